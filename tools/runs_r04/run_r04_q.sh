R=$GRAFT_REPO_ROOT; cd $R
for I in 64 128 256 512 1024; do
GBWT_HIP_SAMPLE_INTERVAL=$I timeout 600 python tools/configs.py high_degree 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('interval $I high_degree kernel_ms', round(j['kernel_ms'],4), 'G/s', round(j['value_kernel']/1e9,1), round(j['value']/1e9,1), 'open', round(j['open_ms'],1))"
done
for I in 256 512; do GBWT_HIP_SAMPLE_INTERVAL=$I GBWT_HIP_DEBUG_DRY_ROWS=1 timeout 600 python - <<'PY'
import os,sys,numpy as np
sys.path.insert(0,'.')
import gbwt_rs_amd as G
from gbwt_rs_amd import synth as S
s = S.Synth.chain(sites=3000, haplotypes=5000, alleles=300, model=S.IID, seed=42)
dev = G.GBWT.from_records(s.data(), s.starts(), s.alphabet_offset, s.alphabet_size, s.sequences, s.size, True)
ids = np.arange(0, s.sequences, 2, dtype=np.uint64)
w=[]
for _ in range(8):
    dev.extract_device(ids); w.append(dev.last_kernel_ms()[0])
print('dry rows interval', os.environ['GBWT_HIP_SAMPLE_INTERVAL'], np.mean(w[3:]), 30e6/np.mean(w[3:])/1e6, 'G/s')
PY
done
