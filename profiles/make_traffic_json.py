#!/usr/bin/env python
"""HBM traffic per launch of the bench's two roofline kernels from rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE collected in
SEPARATE runs of `bench.py --steps 3 --warmup 1`).  Correction per MI355X_MICROARCH.md (HBM section): on gfx950 FETCH_SIZE reports
half of the bytes of a wide (16 B/lane) coalesced read -> doubled; WRITE_SIZE taken as is (uncalibrated).  Units: KB -> bytes.
usage: make_traffic_json.py <pmc_FETCH_dir> <pmc_WRITE_dir> > profiles/rNN_traffic.json"""
import collections
import csv
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import source_hash  # noqa: E402  (the kernel sources these counters were measured on: bench.py quotes them only for the same hash)


def per_kernel(d, counter):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(d.rstrip('/') + '/p_counter_collection.csv')):
        if r['Counter_Name'] == counter:
            acc[r['Kernel_Name'].split('(')[0].replace('void ', '')].append(float(r['Counter_Value']))
    return acc


def main():
    f, w = per_kernel(sys.argv[1], 'FETCH_SIZE'), per_kernel(sys.argv[2], 'WRITE_SIZE')
    if len(sys.argv) > 4:          # (round 5: a second pair of passes = the same command with the split-mode GEMM)
        f2, w2 = per_kernel(sys.argv[3], 'FETCH_SIZE'), per_kernel(sys.argv[4], 'WRITE_SIZE')
        f.update({k: v for k, v in f2.items() if k.startswith('k_gemm_split<')})
        w.update({k: v for k, v in w2.items() if k.startswith('k_gemm_split<')})
    if len(sys.argv) > 6:          # (round 6: a third pair = mode CGC_GEMM_SPLIT_F16: the product kernel and its operand-maximum pass)
        f3, w3 = per_kernel(sys.argv[5], 'FETCH_SIZE'), per_kernel(sys.argv[6], 'WRITE_SIZE')
        f.update({k: v for k, v in f3.items() if k.startswith(('k_gemm_half<', 'k_gemm_absmax<'))})
        w.update({k: v for k, v in w3.items() if k.startswith(('k_gemm_half<', 'k_gemm_absmax<'))})
    out = {}
    for tag, match, big_only in (('gemm_128x128', 'k_gemm_f32<2, 2, 2, 2', False), ('spmm_wide', 'k_spmm_wide<', True),
                                 ('gemm_split', 'k_gemm_split<', False), ('gemm_half', 'k_gemm_half<', False),
                                 ('gemm_half_absmax', 'k_gemm_absmax<', False)):
        fs = [v for k in f if k.startswith(match) for v in f[k]]
        ws = [v for k in w if k.startswith(match) for v in w[k]]
        if big_only:      # the wide (cluster-count) launches are the ones moving > 100 MB
            fs, ws = [v for v in fs if v > 5e4], [v for v in ws if v > 5e4]
        if fs and ws:
            out[tag] = {'launches': len(fs), 'fetch_bytes_per_launch': 2.0 * 1e3 * sum(fs) / len(fs),
                        'write_bytes_per_launch': 1e3 * sum(ws) / len(ws)}
            out[tag]['hbm_bytes_per_launch'] = out[tag]['fetch_bytes_per_launch'] + out[tag]['write_bytes_per_launch']
    out['_note'] = ('rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) of `bench.py --steps 3 --warmup 1 --no-cpu-baseline '
                    '--no-kernel-timing`; FETCH_SIZE x2 (gfx950 wide-read correction), KB -> bytes; mean over all launches of the kernel')
    out['source_sha256'] = source_hash()
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main()
