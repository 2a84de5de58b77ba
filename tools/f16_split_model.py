#!/usr/bin/env python
"""numpy model of an fp32 product on 16-bit matrix cores, written before csrc/gemm_half.hip existed: error against float64, relative to
sum_k |a||b|, in units of 2^-24, of (1) numpy's fp32 matmul, (2) six bf16 pairs of operands split in three (csrc/gemm_split.hip),
(3) THREE fp16 pairs of operands scaled by a power of two and split in two -- the pair products summed exactly (float64), so that
what is printed for (2) and (3) is the representation + dropped-pair error alone, to be set against the fp32 chain's own accumulation
error (1).  CPU only.  usage: python tools/f16_split_model.py"""
import numpy as np
rng = np.random.default_rng(0)
def bf16(x):
    u = x.astype(np.float32).view(np.uint32).astype(np.uint64)
    r = ((u + 0x7fff + ((u >> 16) & 1)) >> 16) << 16
    return r.astype(np.uint32).view(np.float32)
def split_bf16(x):
    h = bf16(x); r1 = (x - h).astype(np.float32); m = bf16(r1); r2 = (r1 - m).astype(np.float32); l = bf16(r2)
    return h, m, l
def split_f16(x, scale):
    xs = (x * scale).astype(np.float32)
    h = xs.astype(np.float16).astype(np.float32); r = (xs - h).astype(np.float32); l = r.astype(np.float16).astype(np.float32)
    return h, l
def pow2scale(x, target=2.0**14):
    m = np.abs(x).max()
    e = np.floor(np.log2(target / m))
    return np.float32(2.0 ** e)
def run(name, A, B):
    A = A.astype(np.float32); B = B.astype(np.float32)
    ref = A.astype(np.float64) @ B.astype(np.float64)
    den = np.abs(A).astype(np.float64) @ np.abs(B).astype(np.float64)
    ex = (A @ B).astype(np.float64)           # fp32 accumulation (numpy order)
    ah, am, al = split_bf16(A); bh, bm, bl = split_bf16(B)
    d = lambda x, y: x.astype(np.float64) @ y.astype(np.float64)
    s6 = d(al, bh) + d(am, bm) + d(am, bh) + d(ah, bl) + d(ah, bm) + d(ah, bh)
    sa, sb = pow2scale(A), pow2scale(B)
    fh, fl = split_f16(A, sa); gh, gl = split_f16(B, sb)
    s3 = (d(fl, gh) + d(fh, gl) + d(fh, gh)) / (float(sa) * float(sb))
    f = lambda c: (np.abs(c - ref) / den).max() * 2**24
    g = lambda c: np.sqrt((((c - ref) / den) ** 2).mean()) * 2**24
    print('%-28s err / sum|a||b| in units of 2^-24:  fp32 max %.3f rms %.4f | bf16 x6 (exact acc) max %.3f rms %.4f | fp16 x3 (exact acc) max %.3f rms %.4f  scales 2^%d 2^%d' % (
        name, f(ex), g(ex), f(s6), g(s6), f(s3), g(s3), np.log2(sa), np.log2(sb)))
M, N, K = 192, 192, 1140
run('normal', rng.standard_normal((M, K)), rng.standard_normal((K, N)))
A = rng.standard_normal((M, K)) * np.exp2(rng.uniform(-12, 0, (1, K))); B = rng.standard_normal((K, N)) * np.exp2(rng.uniform(-12, 0, (K, 1)))
run('skewk (12 binades along K)', A, B)
run('tiny gradients 1e-7', rng.standard_normal((M, K)) * 1e-7, rng.standard_normal((K, N)))
A = rng.standard_normal((M, K)) * np.exp2(rng.uniform(-24, 0, (M, 1)))
run('rows over 24 binades', A, rng.standard_normal((K, N)))
S = rng.standard_normal((M, K)) * 4; S = np.exp(S - S.max(1, keepdims=True)); S /= S.sum(1, keepdims=True)
run('softmax rows x normal', S, rng.standard_normal((K, N)))
run('softmax^T (K=M) x grads', S.T.copy(), rng.standard_normal((M, N)) * 1e-6 * np.exp2(rng.uniform(-10, 0, (M, 1))))
