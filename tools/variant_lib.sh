#!/bin/bash
# Variant builds of libcgc_hip.so for A/B runs on ONE GPU box (CGC_LIB=<path> selects the library: kernels.lib_path()).
#   tools/variant_lib.sh <tag>:<source.hip>:"<extra flags>" ...     (build container; the objects of the regular build are reused)
# -> cgc-net_amd/csrc/variants/libcgc_<tag>.so   (git-ignored like every *.so; travels to the GPU box with the snapshot)
# Used for: timing ablations of the dominant GEMM's k loop (-DCGC_X_*: they break the RESULT, timing only), -DCGC_JK_PRECISE, ...
set -e
cd "$(dirname "$0")/../cgc-net_amd/csrc"
make -j8 > /dev/null
mkdir -p variants
for spec in "$@"; do
  tag="${spec%%:*}"; rest="${spec#*:}"; src="${rest%%:*}"; flags="${rest#*:}"
  (
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -fhip-fp32-correctly-rounded-divide-sqrt $flags -c $src -o variants/${tag}_${src%.hip}.o
    objs=$(ls *.o | grep -v "^${src%.hip}.o$" | tr '\n' ' ')
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o variants/libcgc_$tag.so variants/${tag}_${src%.hip}.o $objs
    rm -f variants/${tag}_${src%.hip}.o
    echo "built variants/libcgc_$tag.so ($src $flags)"
  ) &
done
wait
