#!/usr/bin/env python
"""Which kernel family carries a gradient error?  GPU only.  Runs a golden case on the per-operator path with groups of kernel-table
entries swapped for the torch restatement of the kernel contract (oracle/flat_ref.py, run on the device tensors), and prints the
worst parameter-gradient error against the reference-generated float64 fixture for each swap.  Measurement tool (tests/ territory:
imports the oracle); nothing in the product path uses it.   usage: python tools/parity_bisect.py [case ...]"""
import os
import sys
import types

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests')]
import cgc_net_amd  # noqa: E402,F401
import discrete  # noqa: E402
from cgc_net_amd import kernels, network  # noqa: E402
from oracle.flat_ref import TorchKernels  # noqa: E402
from util import build_model, load_case  # noqa: E402

DEV = 'cuda:0'
GROUPS = {
    'gemm': ['gemm', 'reduce_batch_sum', 'reduce_batched'],
    'spmm': ['spmm'],
    'rowops': ['l2norm_act_stats', 'l2norm_act_bn', 'bn_finalize', 'bn_act_apply', 'bn_bwd_reduce', 'bn_act_l2_bwd', 'colsum'],
    'softmax': ['softmax_fwd', 'softmax_bwd'],
    'segmax': ['segment_max_fwd', 'segment_max_bwd', 'segment_max_bwd_full'],
    'jk': ['jk_supported', 'jk_fwd', 'jk_bwd', 'jk_bwd_params', 'jk_unpack_param_grads'],
    'adj': ['dense_rownorm_fwd', 'dense_rownorm_bwd', 'dense_renorm_fwd', 'dense_renorm_bwd', 'adj_prep_fwd', 'adj_prep_bwd'],
    'sage_fused': ['sage_wide_fwd', 'sage_narrow_fwd', 'sage_narrow_bwd'],
    'edges': ['edge_renorm', 'csr_transpose_vals', 'csr_invdeg'],
}


def run(name, swapped):
    K = kernels.get()
    T = TorchKernels()
    saved = {}
    for g in swapped:
        for m in GROUPS[g]:
            if hasattr(T, m):
                saved[m] = K.__dict__.get(m)
                setattr(K, m, getattr(T, m))
    try:
        fix = discrete.load_reference_fp64(name)
        cfg, batch, sd, _o, _g, _s = load_case(name, DEV)
        model = build_model(network.SoftPoolingGcnEncoder, cfg)
        model.load_state_dict(sd)
        model.to(DEV).train()
        model.native = False
        model.native_head = False
        logits, loss = model(batch)
        loss.backward()
        torch.cuda.synchronize()
        rows = []
        for k, p in model.named_parameters():
            g64 = fix['grad'][k]
            if float(g64.abs().max()) < 1e-12:
                continue
            rows.append((float((p.grad.double().cpu() - g64).abs().max() / g64.abs().max()), k))
        rows.sort(reverse=True)
        return rows
    finally:
        for m, v in saved.items():
            if v is None:
                delattr(K, m)
            else:
                setattr(K, m, v)


for name in (sys.argv[1:] or ['tiny_shipped']):
    print('== %s (per-operator path; worst |grad - ref fp64| / max|ref fp64|)' % name)
    for swapped in [[]] + [[g] for g in GROUPS] + [list(GROUPS)]:
        try:
            rows = run(name, swapped)
            print('  torch for %-40s worst %.1e %-28s 2nd %.1e %s' % ('+'.join(swapped) or '(nothing: all HIP)', rows[0][0], rows[0][1], rows[1][0], rows[1][1]))
        except Exception as e:      # noqa: BLE001
            print('  torch for %-40s FAILED: %s: %s' % ('+'.join(swapped), type(e).__name__, str(e)[:150]))
