import os, sys
import numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import gbwt_rs_amd as G
from gbwt_rs_amd import synth as S, dist as D
s = S.Synth.chain(333334, 5000, alleles=2, model=S.MOSAIC, seed=42)
dev = G.GBWT.from_records(s.data(), s.starts(), s.alphabet_offset, s.alphabet_size, s.sequences, s.size, True)
ids = np.arange(0, s.sequences, 2, dtype=np.uint64)
out = dev.extract_device(ids)
off, nodes = D.paths_tensors(out, torch.device("cuda", 0))
print("tensor", nodes.shape, nodes.dtype, nodes.device, int(off[-1]))
k = 4321
row = nodes[int(off[k]):int(off[k + 1])].cpu().numpy().astype(np.uint32)
print("row equal:", np.array_equal(row, s.path(k)), " sum of last row on device:", int(nodes[int(off[-2]):].to(torch.int64).sum()), s.path_checksum(4999))
c = nodes[-1000:].clone(); print("clone ok", int(c[-1]))
