#!/usr/bin/env python
"""Counter-derived matrix-core utilisation of the dominant GEMM from a rocprofv3 --pmc SQ pass (profiles/rNN_bench_c3_pmc_sq.txt,
written by tools/pmc_run.sh around `bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-timing`).

  mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x kernel cycles),  kernel cycles = GRBM_GUI_ACTIVE / 8 XCDs

SQ_VALU_MFMA_BUSY_CYCLES counts cycles in which a SIMD's matrix pipe is executing (64 per v_mfma_f32_32x32x2_f32, summed over
the chip); GRBM_GUI_ACTIVE counts busy clocks per XCD, summed over the 8 XCDs.  The ratio is clock-independent: it says what
fraction of the matrix pipes' cycles did MFMA work, whatever frequency the chip sustained (MI355X_MICROARCH.md, DVFS give-back).
The flops-derived `frac` of bench.py divides by the 2.4 GHz peak instead, so frac = mfma_busy x (sustained clock / 2.4 GHz).
usage: make_counters_json.py profiles/r02_bench_c3_pmc_sq.txt > profiles/r02_counters.json"""
import collections
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import source_hash  # noqa: E402  (the kernel sources these counters were measured on: bench.py quotes them only for the same hash)

vals = collections.defaultdict(dict)
lines = [l for f in sys.argv[1:] for l in open(f)]          # (round 5: a second file = the pass with the split-mode GEMM)
for line in lines:
    parts = line.split()
    if len(parts) < 5 or parts[0] == 'kernel':
        continue
    median = float(parts[-2])
    calls = int(parts[-3])
    counter = parts[-4]
    kernel = ' '.join(parts[:-4])
    vals[kernel].setdefault(counter, (calls, median))
out = {}
tot_busy = tot_cyc = 0.0
for k, c in vals.items():
    if not k.startswith('k_gemm_f32<2, 2, 2, 2') or 'SQ_VALU_MFMA_BUSY_CYCLES' not in c or 'GRBM_GUI_ACTIVE' not in c:
        continue
    calls, busy = c['SQ_VALU_MFMA_BUSY_CYCLES']
    cyc = c['GRBM_GUI_ACTIVE'][1] / 8.0
    wave = c.get('SQ_WAVE_CYCLES', (0, 0.0))[1]
    out[k] = {'launches_profiled': calls, 'mfma_busy': round(busy / (1024.0 * cyc), 4), 'kernel_cycles': round(cyc),
              'wave_wait_any_frac': round(c.get('SQ_WAIT_ANY', (0, 0.0))[1] / wave, 4) if wave else None,
              'wave_wait_inst_frac': round(c.get('SQ_WAIT_INST_ANY', (0, 0.0))[1] / wave, 4) if wave else None,
              'lds_bank_conflict_cycles': c.get('SQ_LDS_BANK_CONFLICT', (0, 0.0))[1]}
    tot_busy += busy * calls
    tot_cyc += 1024.0 * cyc * calls
out['gemm_128x128'] = {'mfma_busy': round(tot_busy / tot_cyc, 4) if tot_cyc else None,
                       'definition': 'SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs * GRBM_GUI_ACTIVE / 8), launch-weighted over the three instantiations'}
# the split-mode kernel (csrc/gemm_split.hip): the same ratio (32 busy cycles per v_mfma_f32_32x32x16_bf16)
sb = sc = 0.0
for k, c in vals.items():
    if not k.startswith('k_gemm_split<') or 'SQ_VALU_MFMA_BUSY_CYCLES' not in c or 'GRBM_GUI_ACTIVE' not in c:
        continue
    calls, busy = c['SQ_VALU_MFMA_BUSY_CYCLES']
    cyc = c['GRBM_GUI_ACTIVE'][1] / 8.0
    wave = c.get('SQ_WAVE_CYCLES', (0, 0.0))[1]
    out[k] = {'launches_profiled': calls, 'mfma_busy': round(busy / (1024.0 * cyc), 4), 'kernel_cycles': round(cyc),
              'wave_wait_any_frac': round(c.get('SQ_WAIT_ANY', (0, 0.0))[1] / wave, 4) if wave else None,
              'wave_wait_inst_frac': round(c.get('SQ_WAIT_INST_ANY', (0, 0.0))[1] / wave, 4) if wave else None,
              'lds_bank_conflict_cycles': c.get('SQ_LDS_BANK_CONFLICT', (0, 0.0))[1]}
    sb += busy * calls
    sc += 1024.0 * cyc * calls
if sc:
    out['gemm_split'] = {'mfma_busy': round(sb / sc, 4),
                         'definition': 'SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs * GRBM_GUI_ACTIVE / 8) over the launches of k_gemm_split<*> (bf16 matrix pipe)'}
# the fp16 three-pair kernel (csrc/gemm_half.hip; a third file = the pass with mode CGC_GEMM_SPLIT_F16): the product kernel alone
# (its operand-maximum pass k_gemm_absmax is a streaming kernel and is listed separately)
hb = hc = 0.0
for k, c in vals.items():
    if not k.startswith('k_gemm_half<') or 'SQ_VALU_MFMA_BUSY_CYCLES' not in c or 'GRBM_GUI_ACTIVE' not in c:
        continue
    calls, busy = c['SQ_VALU_MFMA_BUSY_CYCLES']
    cyc = c['GRBM_GUI_ACTIVE'][1] / 8.0
    wave = c.get('SQ_WAVE_CYCLES', (0, 0.0))[1]
    out[k] = {'launches_profiled': calls, 'mfma_busy': round(busy / (1024.0 * cyc), 4), 'kernel_cycles': round(cyc),
              'wave_wait_any_frac': round(c.get('SQ_WAIT_ANY', (0, 0.0))[1] / wave, 4) if wave else None,
              'wave_wait_inst_frac': round(c.get('SQ_WAIT_INST_ANY', (0, 0.0))[1] / wave, 4) if wave else None,
              'lds_bank_conflict_cycles': c.get('SQ_LDS_BANK_CONFLICT', (0, 0.0))[1]}
    hb += busy * calls
    hc += 1024.0 * cyc * calls
if hc:
    out['gemm_half'] = {'mfma_busy': round(hb / hc, 4),
                        'definition': 'SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs * GRBM_GUI_ACTIVE / 8) over the launches of k_gemm_half<*> (fp16 matrix pipe; the product kernel without its operand-maximum pass)'}
out['source_sha256'] = source_hash()
print(json.dumps(out, indent=1))
