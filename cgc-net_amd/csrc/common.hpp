// Shared device helpers for libcgc_hip.so (gfx950 / CDNA4 only: 64-wide wavefronts are assumed everywhere).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/cgc_hip.h"

#define CGC_WAVE 64
#define CGC_BLOCK 256  // 4 waves: one per SIMD of a CU

#define CGC_RETURN_IF_LAUNCH_FAILED()                   \
  do {                                                  \
    hipError_t e__ = hipGetLastError();                 \
    if (e__ != hipSuccess) return (int)e__;             \
  } while (0)

// measurement hook (timing.hip): index of the record or -1 when no observer is attached
#define CGC_TAG_GEMM_128 1
#define CGC_TAG_SPMM_WIDE 2
int cgc_timing_begin(int tag, int d0, int d1, int d2, int d3, int d4, int d5, int d6, hipStream_t stream);
void cgc_timing_end(int idx, hipStream_t stream);

// A kernel that needs more dynamic LDS than the 64 KB default has the limit raised once per DEVICE (function attributes are per
// device; a process may drive several).  `done` = a zero-initialised static array of CGC_MAX_DEVICES flags owned by the call site.
#define CGC_MAX_DEVICES 64
static inline void cgc_allow_lds(const void* fn, int bytes, bool* done) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= CGC_MAX_DEVICES) dev = 0;
  if (!done[dev]) {
    (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    done[dev] = true;
  }
}

static inline hipStream_t as_stream(cgc_stream_t s) { return reinterpret_cast<hipStream_t>(s); }
static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
static inline int64_t ceil_div64(int64_t a, int64_t b) { return (a + b - 1) / b; }
static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// ---- cross-lane reductions over the `lpr` (power of two, 8..64) consecutive lanes that share a row; every lane of the group
// receives the result.  The steps inside a row of 16 lanes are DPP moves (quad swaps, half mirror, mirror: register to register);
// only the strides 16 and 32 need a cross-row exchange.  __shfl_xor alone compiles to one ds_bpermute_b32 -- an LDS round trip
// -- per step, i.e. 6 dependent round trips per row value for 64-lane rows.
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float group_sum(float v, int lpr) {
  v += dpp_mov<0xB1>(v);                  // quad_perm [1,0,3,2]   (stride 1)
  v += dpp_mov<0x4E>(v);                  // quad_perm [2,3,0,1]   (stride 2)
  if (lpr >= 8) v += dpp_mov<0x141>(v);   // row_half_mirror       (stride 4)
  if (lpr >= 16) v += dpp_mov<0x140>(v);  // row_mirror            (stride 8)
  if (lpr >= 32) v += __shfl_xor(v, 16);
  if (lpr >= 64) v += __shfl_xor(v, 32);
  return v;
}
__device__ __forceinline__ float group_max(float v, int lpr) {
  v = fmaxf(v, dpp_mov<0xB1>(v));
  v = fmaxf(v, dpp_mov<0x4E>(v));
  if (lpr >= 8) v = fmaxf(v, dpp_mov<0x141>(v));
  if (lpr >= 16) v = fmaxf(v, dpp_mov<0x140>(v));
  if (lpr >= 32) v = fmaxf(v, __shfl_xor(v, 16));
  if (lpr >= 64) v = fmaxf(v, __shfl_xor(v, 32));
  return v;
}

// ---- activations (model/network.py:84-91); `act` is wave-uniform
__device__ __forceinline__ float act_fwd(float x, int act) {
  switch (act) {
    case CGC_ACT_RELU: return x > 0.f ? x : 0.f;
    case CGC_ACT_ELU: return x > 0.f ? x : expf(x) - 1.f;
    case CGC_ACT_LEAKYRELU: return x > 0.f ? x : 0.01f * x;
    default: return x;
  }
}
__device__ __forceinline__ float act_bwd(float x, int act) {  // d act / d x evaluated at the pre-activation x
  switch (act) {
    case CGC_ACT_RELU: return x > 0.f ? 1.f : 0.f;
    case CGC_ACT_ELU: return x > 0.f ? 1.f : expf(x);
    case CGC_ACT_LEAKYRELU: return x > 0.f ? 1.f : 0.01f;
    default: return 1.f;
  }
}

// ---- "row group" work distribution used by all row-wise kernels:
// lpr lanes cooperate on one row (lpr = 8/16/32/64, chosen from the row width), a wave carries 64/lpr rows,
// lanes of a group walk the columns with stride lpr*VEC.  Shuffles stay inside a group.
struct RowGroup {
  int sl;      // lane index inside the group
  int sub;     // which of the wave's rows this lane works on
  int rpw;     // rows per wave
  int gwave;   // global wave id
  int nwaves;  // waves in the grid
  __device__ __forceinline__ RowGroup(int lpr) {
    const int lane = threadIdx.x & 63;
    sl = lane & (lpr - 1);
    sub = lane / lpr;
    rpw = 64 / lpr;
    gwave = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    nwaves = gridDim.x * (blockDim.x >> 6);
  }
};

static inline int pick_lpr(int chunks) { return chunks <= 8 ? 8 : chunks <= 16 ? 16 : chunks <= 32 ? 32 : 64; }
static inline int row_blocks(int n, int lpr, int cap = 2048) {
  int rows_per_block = (CGC_BLOCK / 64) * (64 / lpr);
  int b = ceil_div(n > 0 ? n : 1, rows_per_block);
  return b < cap ? b : cap;
}

// vector-of-VEC load/store helpers (VEC = 1 or 4)
template <int VEC> struct Vec;
template <> struct Vec<1> {
  float v[1];
  __device__ __forceinline__ void load(const float* p) { v[0] = p[0]; }
  __device__ __forceinline__ void store(float* p) const { p[0] = v[0]; }
};
template <> struct Vec<4> {
  float v[4];
  __device__ __forceinline__ void load(const float* p) {
    const float4 t = *reinterpret_cast<const float4*>(p);
    v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
  }
  __device__ __forceinline__ void store(float* p) const {
    *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
  }
  // streaming read: the line is not kept in L2 / the Infinity Cache.  A 263 MB tensor that is read once per pass gains nothing from
  // being cached, and the lines it would occupy are the ones the pass's own WRITES want: tools/probes/stream_rows_probe.hip --
  // one read + one write of [57696, 1140] floats at 5.3 TB/s with plain loads (torch's copy_, hipMemcpy: the same), 6.8 TB/s with
  // these.  (Nontemporal STORES cost 25 %.)
  __device__ __forceinline__ void load_stream(const float* p) {
    typedef float vf4 __attribute__((ext_vector_type(4)));
    const vf4 t = __builtin_nontemporal_load(reinterpret_cast<const vf4*>(p));
    v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
  }
};
// rows of more than 1024 floats (16-byte lanes, 5+ chunks per lane: the cluster-count-wide tensors) are read as streams
template <int VEC, int MAXJ>
__device__ __forceinline__ void load_wide(Vec<VEC>& x, const float* p) {
#ifndef CGC_NO_STREAM_LOADS      // (-DCGC_NO_STREAM_LOADS: plain loads, for A/B timing through tools/variant_lib.sh)
  if constexpr (VEC == 4 && MAXJ >= 5) x.load_stream(p);
  else
#endif
    x.load(p);
}
