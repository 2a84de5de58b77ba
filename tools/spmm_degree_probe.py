import os, sys
import numpy as np, torch
sys.path.insert(0, os.getcwd())
import cgc_net_amd
from cgc_net_amd import kernels
K = kernels.get(); dev='cuda:0'
rng = np.random.RandomState(0)
counts = rng.randint(1440, 2161, size=32); n = int(counts.sum()); W = 1140
gptr = torch.tensor(np.cumsum([0]+list(counts)), dtype=torch.int32, device=dev)
x = torch.randn(n, W, device=dev); out = torch.empty_like(x)
big = torch.empty(300*1024*1024//4, device=dev)
for deg in (1, 2, 4, 9):
    cols = []
    off = 0
    for c in counts:
        cols.append(off + rng.randint(0, c, size=(c, deg)))
        off += c
    col = torch.tensor(np.concatenate(cols).reshape(-1), dtype=torch.int32, device=dev)
    rowptr = torch.arange(0, n*deg+1, deg, dtype=torch.int32, device=dev)
    for cold in (True, False):
        ts=[]
        for it in range(6):
            if cold: big.fill_(1.0)
            s,e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record(); K.spmm(rowptr, col, None, None, None, None, x, out, n, W, gptr, 32, int(counts.max())); e.record()
            torch.cuda.synchronize(); ts.append(s.elapsed_time(e))
        ms = float(np.median(ts[1:])); by = 8.0*n*W
        print('deg %d %s: %.1f us  %.0f GB/s algorithmic' % (deg, 'cold' if cold else 'warm', ms*1e3, by/ms/1e6))
# plain copy for reference
for cold in (True, False):
    ts=[]
    for it in range(6):
        if cold: big.fill_(1.0)
        s,e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); out.copy_(x); e.record(); torch.cuda.synchronize(); ts.append(s.elapsed_time(e))
    ms=float(np.median(ts[1:])); print('torch copy %s: %.1f us %.0f GB/s' % ('cold' if cold else 'warm', ms*1e3, 8.0*n*W/ms/1e6))
