#!/usr/bin/env python3
"""What does a large device allocation cost on a box nobody has used yet?  A few fresh PROCESSES in turn, each allocating and filling `gb` GB
(hipMalloc through torch, no caching between them: each is its own process), then freeing; usage: vram_first_touch_probe.py [GB ...]"""
import subprocess, sys, time
CHILD = r'''
import sys, time, torch
gb = int(sys.argv[1])
torch.cuda.init(); torch.zeros(1, device="cuda"); torch.cuda.synchronize()
t0 = time.perf_counter(); x = torch.empty(gb << 30, dtype=torch.uint8, device="cuda"); torch.cuda.synchronize(); t1 = time.perf_counter()
x.zero_(); torch.cuda.synchronize(); t2 = time.perf_counter()
x.zero_(); torch.cuda.synchronize(); t3 = time.perf_counter()
del x; torch.cuda.empty_cache(); torch.cuda.synchronize(); t4 = time.perf_counter()
y = torch.empty(gb << 30, dtype=torch.uint8, device="cuda"); y.zero_(); torch.cuda.synchronize(); t5 = time.perf_counter()
print(f"{gb} GB: alloc {1e3 * (t1 - t0):8.1f} ms, first fill {1e3 * (t2 - t1):8.1f}, second fill {1e3 * (t3 - t2):8.1f}, free {1e3 * (t4 - t3):8.1f}, alloc + fill again in this process {1e3 * (t5 - t4):8.1f}", flush=True)
'''
for gb in [int(a) for a in sys.argv[1:]] or [16, 100, 100, 200, 100]:
    subprocess.run([sys.executable, "-c", CHILD, str(gb)], check=False)
