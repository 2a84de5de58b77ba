// Shared by the kernels that run an fp32 product on the 16-bit matrix cores of gfx950 from operands split in registers
// (gemm_split.hip: three bf16 planes, six pairs; gemm_half.hip: two fp16 planes, three pairs): the 256 x 128 x 16 tile's loaders
// (global -> registers -> [plane][row][16 k] in LDS, the transposition of an operand stored [K, .] done by register naming), the
// operand segments of a k-tile, the 16-byte fragment read.
#pragma once
#include <type_traits>

#include "gemm_common.hpp"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float float2v __attribute__((ext_vector_type(2)));
typedef unsigned uint4v __attribute__((ext_vector_type(4)));

#define SBK 16                          // k-tile
#define SROW 48                         // bytes per LDS row of a plane: 16 bf16 + 16 bytes of padding (16-byte-aligned rows for ds_read_b128)
constexpr int S_BM = 256, S_BN = 128;

__device__ __forceinline__ unsigned pack_bf16(float a, float b) {      // v_cvt_pk_bf16_f32: round to nearest even, a in the low half
  float2v t;
  t[0] = a;
  t[1] = b;
  return __builtin_bit_cast(unsigned, __builtin_convertvector(t, bf16x2));
}
__device__ __forceinline__ float comp(const float4& v, int e) { return e == 0 ? v.x : e == 1 ? v.y : e == 2 ? v.z : v.w; }

// Everything below is written so that, once the per-tile loops are fully unrolled, every array index is a constant (the register
// arrays then live in registers) and no closure survives: free functions with explicit arguments, no lambda inside a lambda (a
// by-reference lambda nested in another kept its closure -- a struct of pointers to locals -- in scratch memory: gemm.hip).

// Operand whose K index is the contiguous one in memory (A stored [M, K]; B stored [N, K]).  ROWS x 16 tile = ROWS * 4 units of
// 16 bytes; thread t: unit q = t & 3 of rows row_of(i), i = 0 .. ROWS / 64 - 1.  A unit is one "group": four consecutive k of one row.
template <int ROWS, int RS = SROW>
struct SplitLoaderK {
  static constexpr int NF = ROWS / 64, NG = NF;
  // row of unit i: 64 i + 16 wave + r16 with r16 = ((t >> 4) & 1) + 8 ((t >> 5) & 1) + 2 ((t >> 2) & 3): the 16 lanes of an LDS write group
  // (8-byte writes) hold rows b, b + 2, b + 4, b + 6 -- on the 48-byte row stride their 32-byte windows start 96 = -32 (mod 128) bytes
  // apart and tile the 128-byte bank window exactly (consecutive rows overlap: a quarter of the LDS cycles of the first version of
  // this kernel were bank conflicts)
  static __device__ __forceinline__ int row_of(int i) {
    const int t = (int)threadIdx.x;
    static_assert(RS == 48, "the row mapping tiles the bank window for 48-byte rows");
    return 64 * i + 16 * (t >> 6) + ((t >> 4) & 1) + 8 * ((t >> 5) & 1) + 2 * ((t >> 2) & 3);
  }
  static __device__ __forceinline__ void offsets(unsigned (&off)[NF], int ld, int row0, int row_last) {
#pragma unroll
    for (int i = 0; i < NF; ++i) off[i] = (unsigned)min(row0 + row_of(i), row_last) * (unsigned)ld * 4u + (threadIdx.x & 3u) * 16u;
  }
  static __device__ __forceinline__ unsigned soffset(int /*ld*/, int k0) { return (unsigned)k0 * 4u; }
  // any tile of any segment: units past the end of K re-read the last valid 16 bytes of their row (the split zeroes them)
  static __device__ __forceinline__ float4 load_any(int i, const float* __restrict__ base, int ld, int row0, int row_last, int k0, int klim) {
    const int row = min(row0 + row_of(i), row_last);
    const int k = min(k0 + (int)(threadIdx.x & 3) * 4, (klim - 1) & ~3);
    return *reinterpret_cast<const float4*>(base + (size_t)row * ld + k);
  }
  static __device__ __forceinline__ void get(const float4 (&reg)[NF], int u, float (&x)[4]) {
    x[0] = reg[u].x; x[1] = reg[u].y; x[2] = reg[u].z; x[3] = reg[u].w;
  }
  static __device__ __forceinline__ int kof(int /*u*/, int e) { return (int)(threadIdx.x & 3) * 4 + e; }      // k of element e inside the tile
  static __device__ __forceinline__ unsigned wbase() { return (unsigned)row_of(0) * RS + (threadIdx.x & 3u) * 8u; }
  static __device__ __forceinline__ void put(unsigned char* st, int u, int plane_off, unsigned w0, unsigned w1) {   // st = stage + region + wbase()
    *reinterpret_cast<uint2*>(st + u * 64 * RS + plane_off) = make_uint2(w0, w1);
  }
};

// Operand whose M / N index is the contiguous one (A stored [K, M]; B stored [K, N]).  16 x COLS tile; a thread owns a KH x 4 block
// (KH = 4 for the 256-wide operand, 2 for the 128-wide one): lane -> (kgrp = t % (16 / KH), g = t / (16 / KH)); float4 j of the block is
// row k = KH * kgrp + j, columns 4 g .. 4 g + 3.  The 16 lanes of an LDS write group then cover 4 column groups x 4 k groups (KH = 4:
// 8-byte writes) or the 32 lanes 4 x 8 (KH = 2: 4-byte writes): at most two lanes per bank on the 48-byte stride (free for 4-byte writes).
// Groups: KH = 4: column c of the block (its four k); KH = 2: columns 2u, 2u + 1 (two k each).  The transposition is a choice of
// register names.
template <int COLS, int RS = SROW>
struct SplitLoaderMN {
  static constexpr int KH = COLS / 64, NF = KH, NG = KH == 4 ? 4 : 2, KG = 16 / KH;
  static __device__ __forceinline__ int kgrp() { return (int)threadIdx.x % KG; }
  static __device__ __forceinline__ int g() { return (int)threadIdx.x / KG; }
  static __device__ __forceinline__ void offsets(unsigned (&off)[NF], int ld, int col0, int col_last4) {
#pragma unroll
    for (int j = 0; j < NF; ++j)
      off[j] = (unsigned)(KH * kgrp() + j) * (unsigned)ld * 4u + (unsigned)min(col0 + 4 * g(), col_last4) * 4u;
  }
  static __device__ __forceinline__ unsigned soffset(int ld, int k0) { return (unsigned)k0 * (unsigned)ld * 4u; }
  // rows (k) past the end re-read row klim - 1 (the split zeroes them)
  static __device__ __forceinline__ float4 load_any(int j, const float* __restrict__ base, int ld, int col0, int col_last4, int k0, int klim) {
    const int k = min(k0 + KH * kgrp() + j, klim - 1);
    return *reinterpret_cast<const float4*>(base + (size_t)k * ld + min(col0 + 4 * g(), col_last4));
  }
  static __device__ __forceinline__ void get(const float4 (&reg)[NF], int u, float (&x)[4]) {
    if constexpr (KH == 4) {
      x[0] = comp(reg[0], u); x[1] = comp(reg[1], u); x[2] = comp(reg[2], u); x[3] = comp(reg[3], u);
    } else {
      x[0] = comp(reg[0], 2 * u); x[1] = comp(reg[1], 2 * u); x[2] = comp(reg[0], 2 * u + 1); x[3] = comp(reg[1], 2 * u + 1);
    }
  }
  static __device__ __forceinline__ int kof(int /*u*/, int e) { return KH == 4 ? 4 * kgrp() + e : 2 * kgrp() + (e & 1); }
  static __device__ __forceinline__ unsigned wbase() { return (unsigned)g() * 4u * RS + (unsigned)kgrp() * (KH == 4 ? 8u : 4u); }
  static __device__ __forceinline__ void put(unsigned char* st, int u, int plane_off, unsigned w0, unsigned w1) {
    if constexpr (KH == 4) {
      *reinterpret_cast<uint2*>(st + u * RS + plane_off) = make_uint2(w0, w1);
    } else {
      *reinterpret_cast<unsigned*>(st + (2 * u) * RS + plane_off) = w0;
      *reinterpret_cast<unsigned*>(st + (2 * u + 1) * RS + plane_off) = w1;
    }
  }
};

// which k-tile (of 16) of which operand segment: the main pair, then the extra K segments (gemm_common.hpp: GemmArgs::nx)
struct SplitSegs {
  const float *A0, *A1, *A2, *B0, *B1, *B2;
  int lda0, lda1, lda2, ldb0, ldb1, ldb2, K0, K1, K2;
  int nk_main, nkx0;
};
struct SplitTile {
  const float* A;
  const float* B;
  int lda, ldb, klim, k0;
};
__device__ __forceinline__ SplitTile split_tile(const SplitSegs t, int kt) {
  const int kx = kt - t.nk_main;
  const bool in_main = kx < 0, in_x0 = kx < t.nkx0;
  SplitTile r;
  r.A = in_main ? t.A0 : in_x0 ? t.A1 : t.A2;
  r.B = in_main ? t.B0 : in_x0 ? t.B1 : t.B2;
  r.lda = in_main ? t.lda0 : in_x0 ? t.lda1 : t.lda2;
  r.ldb = in_main ? t.ldb0 : in_x0 ? t.ldb1 : t.ldb2;
  r.klim = in_main ? t.K0 : in_x0 ? t.K1 : t.K2;
  r.k0 = (in_main ? kt : in_x0 ? kx : kx - t.nkx0) * SBK;
  return r;
}

__device__ __forceinline__ uint4v frag16(const unsigned char* p) { return *reinterpret_cast<const uint4v*>(p); }
