"""Where does a workgroup of the dominant GEMM spend its life?  Needs the trace build of the library
(hipcc -DCGC_GEMM_TRACE on csrc/gemm.hip, linked as tools/libcgc_trace.so; run with CGC_LIB=tools/libcgc_trace.so): every 128 x 128
workgroup records wall_clock64 (100 MHz) at entry, at the start of its k loop, at its end and after the epilogue, plus HW_ID /
XCC_ID.  Prints the phase lengths and, per CU, how much of the launch had 2 / 1 / 0 workgroups inside their k loop."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from cgc_net_amd import kernels

K = kernels.get()
dev = 'cuda:0'
n, C, LD = 58761, 1140, 1152
dz = torch.randn(n, LD, device=dev)
W = torch.randn(C, LD, device=dev)
out = torch.empty(n, LD, device=dev)


def product():          # Linear dx: [n, 1140] = dz [n, 1140] @ W [1140, 1140]  (NN, flat: 460 x 9 = 4140 tiles)
    K.gemm(dz, W, out, n, C, C, False, False, LD, LD, LD)


split = '--nosplit' not in sys.argv
K.tail_split = split
for _ in range(5):
    product()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); product(); e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1)
tiles = 460 * 9
nwg = tiles + 1024 if split else tiles
buf = np.zeros((nwg, 6), dtype=np.uint64)
K.lib.cgc_gemm_trace_read.argtypes = [ctypes.c_void_p, ctypes.c_int]
rc = K.lib.cgc_gemm_trace_read(buf.ctypes.data, nwg)
assert rc == 0
t = buf[:, :4].astype(np.float64)
ok = (t[:, 0] > 0) & (t[:, 3] > 0)
t, hw = t[ok], buf[ok, 5]
t0 = t[:, 0].min()
us = (t - t0) / 100.0
print('launch %.1f us (events), first entry .. last exit %.1f us, %d workgroups traced' % (ms * 1e3, us[:, 3].max(), len(us)))
pro, loop, epi = us[:, 1] - us[:, 0], us[:, 2] - us[:, 1], us[:, 3] - us[:, 2]
for name, v in (('prologue (entry -> k loop)', pro), ('k loop', loop), ('epilogue', epi), ('whole', us[:, 3] - us[:, 0])):
    print('  %-28s mean %7.2f  p10 %7.2f  median %7.2f  p90 %7.2f us' % (name, v.mean(), np.percentile(v, 10), np.median(v), np.percentile(v, 90)))
# per CU: (xcc, se, cu) from XCC_ID / HW_ID (gfx9 HW_ID: cu_id bits 8-11, sh 12, se 13-15)
xcc = (hw >> np.uint64(32)) & np.uint64(0xf)
hwid = hw & np.uint64(0xffffffff)
cu = (hwid >> np.uint64(8)) & np.uint64(0xf)
se = (hwid >> np.uint64(13)) & np.uint64(0x7)
sh = (hwid >> np.uint64(12)) & np.uint64(0x1)
key = (xcc * 1000 + se * 100 + sh * 50 + cu).astype(np.int64)
ids = np.unique(key)
print('  distinct (xcc, se, sh, cu): %d' % len(ids))
end = us[:, 3].max()
cover = np.zeros(3)
for k in ids:
    m = key == k
    ev = sorted([(a, 1) for a in us[m, 1]] + [(b, -1) for b in us[m, 2]])
    cur, last = 0, 0.0
    for x, d in ev:
        cover[min(cur, 2)] += x - last
        cur += d
        last = x
    cover[0] += end - last
cover /= cover.sum()
print('  CU time with 0 / 1 / >=2 workgroups inside their k loop: %.3f / %.3f / %.3f' % tuple(cover))
# gaps between consecutive workgroups on one CU slot: exit of one -> k loop start of the next
