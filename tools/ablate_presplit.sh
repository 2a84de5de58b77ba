#!/bin/bash
# Timing-only ablations of k_gemm_split that bound what PRE-SPLIT operand planes (the producers emit hi | mid | lo bf16 planes, the k loop is
# loads + LDS + MFMA only: verdict r5 item 1a) could gain, built from a patched COPY of gemm_split.hip (the library source is untouched):
#   abl_valu      the split's vector work gone (mid = lo = hi: two v_cvt_pk per group of four values left), loads and LDS writes as they are
#   abl_valu_lds  ... and the LDS writes gone too (what direct-to-LDS loads of ready-made planes would leave: fragment reads + MFMA + loads)
# RESULTS ARE WRONG by construction.  -> cgc-net_amd/csrc/variants/libcgc_abl_*.so ; run: CGC_LIB=... python tools/split_gemm_bench.py 20
set -e
cd "$(dirname "$0")/../cgc-net_amd/csrc"
make -j8 > /dev/null
mkdir -p variants
python3 - <<'PY'
s = open('gemm_split.hip').read()
key = "  GroupState& s = gs[u];\n"
assert s.count(key) == 1
patch = key + """#ifdef S_ABL_PRESPLIT
  if (st >= 1 && st <= 6) {
    if (st == 6) { s.mp[0] = s.hp[0]; s.mp[1] = s.hp[1]; s.lp[0] = s.hp[0]; s.lp[1] = s.hp[1]; }
    return;
  }
#endif
#ifdef S_ABL_NOWRITE
  if (st == 7) return;
#endif
"""
open('variants/gemm_split_abl.hip', 'w').write(s.replace(key, patch))
PY
for spec in "abl_valu:-DS_ABL_PRESPLIT" "abl_valu_lds:-DS_ABL_PRESPLIT -DS_ABL_NOWRITE"; do
  tag="${spec%%:*}"; flags="${spec#*:}"
  (
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -fhip-fp32-correctly-rounded-divide-sqrt -I. $flags -c variants/gemm_split_abl.hip -o variants/${tag}.o
    objs=$(ls *.o | grep -v "^gemm_split.o$" | tr '\n' ' ')
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o variants/libcgc_$tag.so variants/${tag}.o $objs
    rm -f variants/${tag}.o
    echo "built variants/libcgc_$tag.so ($flags)"
  ) &
done
wait
rm -f variants/gemm_split_abl.hip
