// fp32 products on the bf16 matrix cores of gfx950: C = op(A) op(B) with every fp32 operand element split into three bf16 values
// (hi + mid + lo = x exactly: 8 + 8 + 8 mantissa bits, round-to-nearest at every level, both residuals exact in fp32) and the product
// formed as the SIX most significant bf16 x bf16 pairs -- hh, hm, mh, hl, mm, lh; every partial product is exact in fp32 and the sums
// run in the matrix core's fp32 accumulator.  The three dropped pairs (ml, lm, ll) are below 2^-26 of |a||b| each: the result is
// indistinguishable from the fp32 MFMA chain of gemm.hip (max / rms error against float64 measured per form by
// tests/test_kernels_gpu.py::test_split_gemm_*; profiles/r04_bf16_split_probe.txt has the probe that motivated this).
// v_mfma_f32_32x32x16_bf16 issues in 32 cycles per SIMD where the eight v_mfma_f32_32x32x2_f32 of the same k range take 512: six
// pairs are 192 matrix-pipe cycles per 32 x 32 x 16 block against 512.
//
// This is an opt-in MODE of the same entry points (cgc_gemm_f32_ws / cgc_gemm_f32_cat_ws, mode = CGC_GEMM_SPLIT_BF16; the step
// sequencer: cgc_level_desc.flags bit 1) for the products that take the 128 x 128 route of gemm.hip -- the assignment Linear
// (model/network.py:121-122), S^T (A S), P dA'^T, S dA' of _diff_pool and its backward (:206-207) -- in all their forms: NN / NT / TN,
// ragged M, ragged K, uniform K chunks, extra K segments, beta = 1, tail split.  Everything else stays on the exact kernel.
//
// Domain: finite inputs.  x = hi + mid + lo needs |x| >= 2^-108 or so for lo to be a normal bf16 (below that the low planes lose bits
// gradually and the product degrades towards bf16 x 2 accuracy; zeros are exact); an infinite or > 3.39e38 input gives NaN where the
// exact kernel gives inf (hi = inf, x - hi = NaN).  The network's activations and gradients live in 1e-12 .. 1e3.
//
// Structure (one workgroup per CU: 256 threads = 4 waves, one per SIMD, 512 registers each):
//   tile 256 x 128, wave tile 128 x 64 (4 x 2 accumulators of 32 x 32), k-tiles of 16;
//   LDS: three stages of [3 planes][256 + 128 rows][16 k bf16] (rows of 32 B on a 40 B stride: the two ds_read_b64 of a fragment and
//   the 8-byte writes of both operand orientations are bank-conflict free), 135 KB;
//   global -> registers three k-tiles ahead of the split (buffer loads, no vector address arithmetic in the loop), split in registers
//   -> LDS two k-tiles ahead of the MFMAs, fragments read one k-tile ahead into a second register set: a wave never waits for LDS or
//   memory inside a k-tile, and there is ONE barrier per k-tile;
//   an operand whose k index is the memory row (A stored [K, M], B stored [K, N]) is transposed in registers for free: a thread loads a
//   4 (k) x 4 (m) block -- 2 x 4 for the 128-wide operand -- and packs along k;
//   the ~130 vector instructions of a k-tile's split are dealt out by hand behind its 48 MFMAs (one micro-step of 2-4 instructions per
//   MFMA, a scheduling fence after each): behind a bf16 MFMA up to ~5 plain vector instructions of the SAME wave issue for free
//   (tools/pipe_overlap_probe.hip), another wave's do not.
#include <type_traits>

#include "gemm_common.hpp"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float float2v __attribute__((ext_vector_type(2)));
typedef unsigned uint4v __attribute__((ext_vector_type(4)));

#define SBK 16                          // k-tile
#define SROW 40                         // bytes per LDS row of a plane: 16 bf16 + 8 bytes of padding
constexpr int S_BM = 256, S_BN = 128;
constexpr int S_PLA = S_BM * SROW, S_PLB = S_BN * SROW, S_STAGE = 3 * S_PLA + 3 * S_PLB, S_NSTAGE = 3;
constexpr int S_LDS = S_NSTAGE * S_STAGE;       // 138240 bytes

__device__ __forceinline__ unsigned pack_bf16(float a, float b) {      // v_cvt_pk_bf16_f32: round to nearest even, a in the low half
  float2v t;
  t[0] = a;
  t[1] = b;
  return __builtin_bit_cast(unsigned, __builtin_convertvector(t, bf16x2));
}
template <int E>
__device__ __forceinline__ float comp(const float4& v) { return E == 0 ? v.x : E == 1 ? v.y : E == 2 ? v.z : v.w; }

// Operand whose K index is the contiguous one in memory (A stored [M, K]; B stored [N, K]).  ROWS x 16 tile = ROWS * 4 units of
// 16 bytes; thread t: unit q = t & 3 of rows (t >> 2) + 64 i.  A unit is one "group": four consecutive k of one row.
template <int ROWS>
struct SplitLoaderK {
  static constexpr int NF = ROWS / 64, NG = NF;
  float4 reg[NF];
  __device__ __forceinline__ void offsets(unsigned (&off)[NF], int ld, int row0, int row_last) const {
#pragma unroll
    for (int i = 0; i < NF; ++i)
      off[i] = (unsigned)min(row0 + (int)(threadIdx.x >> 2) + 64 * i, row_last) * (unsigned)ld * 4u + (threadIdx.x & 3u) * 16u;
  }
  static __device__ __forceinline__ unsigned soffset(int /*ld*/, int k0) { return (unsigned)k0 * 4u; }
  __device__ __forceinline__ void load_buf(int i, __amdgpu_buffer_rsrc_t rsrc, const unsigned (&off)[NF], unsigned soff) {
    reg[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, off[i], soff, 0));
  }
  // any tile of any segment: units past the end of K re-read the last valid 16 bytes of their row (the split zeroes them)
  __device__ __forceinline__ void load_any(int i, const float* __restrict__ base, int ld, int row0, int row_last, int k0, int klim) {
    const int row = min(row0 + (int)(threadIdx.x >> 2) + 64 * i, row_last);
    const int k = min(k0 + (int)(threadIdx.x & 3) * 4, (klim - 1) & ~3);
    reg[i] = *reinterpret_cast<const float4*>(base + (size_t)row * ld + k);
  }
  template <int U>
  __device__ __forceinline__ void get(float (&x)[4]) const {
    x[0] = reg[U].x; x[1] = reg[U].y; x[2] = reg[U].z; x[3] = reg[U].w;
  }
  template <int U>
  __device__ __forceinline__ int kof(int e) const { return (int)(threadIdx.x & 3) * 4 + e; }      // k of element e inside the tile
  __device__ __forceinline__ unsigned wbase() const { return (threadIdx.x >> 2) * SROW + (threadIdx.x & 3u) * 8u; }
  template <int U, int PLANE>
  __device__ __forceinline__ void put(unsigned char* st, int p, unsigned w0, unsigned w1) const {   // st = stage base + region + wbase()
    *reinterpret_cast<uint2*>(st + U * 64 * SROW + p * PLANE) = make_uint2(w0, w1);
  }
};

// Operand whose M / N index is the contiguous one (A stored [K, M]; B stored [K, N]).  16 x COLS tile; a thread owns a KH x 4 block
// (KH = 4 for the 256-wide operand, 2 for the 128-wide one): lane -> (kgrp = t % (16 / KH), g = t / (16 / KH)); float4 j of the block is
// row k = KH * kgrp + j, columns 4 g .. 4 g + 3.  The 16 lanes of an LDS write group then cover 4 column groups x 4 k groups (KH = 4:
// 8-byte writes) or the 32 lanes 4 x 8 (KH = 2: 4-byte writes): distinct banks.  Groups: KH = 4: column c of the block (its four k);
// KH = 2: columns 2u, 2u + 1 (two k each).
template <int COLS>
struct SplitLoaderMN {
  static constexpr int KH = COLS / 64, NF = KH, NG = KH == 4 ? 4 : 2, KG = 16 / KH;
  float4 reg[NF];
  __device__ __forceinline__ int kgrp() const { return (int)threadIdx.x % KG; }
  __device__ __forceinline__ int g() const { return (int)threadIdx.x / KG; }
  __device__ __forceinline__ void offsets(unsigned (&off)[NF], int ld, int col0, int col_last4) const {
#pragma unroll
    for (int j = 0; j < NF; ++j)
      off[j] = (unsigned)(KH * kgrp() + j) * (unsigned)ld * 4u + (unsigned)min(col0 + 4 * g(), col_last4) * 4u;
  }
  static __device__ __forceinline__ unsigned soffset(int ld, int k0) { return (unsigned)k0 * (unsigned)ld * 4u; }
  __device__ __forceinline__ void load_buf(int j, __amdgpu_buffer_rsrc_t rsrc, const unsigned (&off)[NF], unsigned soff) {
    reg[j] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, off[j], soff, 0));
  }
  // rows (k) past the end re-read row klim - 1 (the split zeroes them)
  __device__ __forceinline__ void load_any(int j, const float* __restrict__ base, int ld, int col0, int col_last4, int k0, int klim) {
    const int k = min(k0 + KH * kgrp() + j, klim - 1);
    reg[j] = *reinterpret_cast<const float4*>(base + (size_t)k * ld + min(col0 + 4 * g(), col_last4));
  }
  template <int U>
  __device__ __forceinline__ void get(float (&x)[4]) const {
    if constexpr (KH == 4) {
      x[0] = comp<U>(reg[0]); x[1] = comp<U>(reg[1]); x[2] = comp<U>(reg[2]); x[3] = comp<U>(reg[3]);
    } else {
      x[0] = comp<2 * U>(reg[0]); x[1] = comp<2 * U>(reg[1]); x[2] = comp<2 * U + 1>(reg[0]); x[3] = comp<2 * U + 1>(reg[1]);
    }
  }
  template <int U>
  __device__ __forceinline__ int kof(int e) const { return KH == 4 ? 4 * kgrp() + e : 2 * kgrp() + (e & 1); }
  __device__ __forceinline__ unsigned wbase() const { return (unsigned)g() * 4u * SROW + (unsigned)kgrp() * (KH == 4 ? 8u : 4u); }
  template <int U, int PLANE>
  __device__ __forceinline__ void put(unsigned char* st, int p, unsigned w0, unsigned w1) const {
    if constexpr (KH == 4) {
      *reinterpret_cast<uint2*>(st + U * SROW + p * PLANE) = make_uint2(w0, w1);
    } else {
      *reinterpret_cast<unsigned*>(st + (2 * U) * SROW + p * PLANE) = w0;
      *reinterpret_cast<unsigned*>(st + (2 * U + 1) * SROW + p * PLANE) = w1;
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
__device__ __forceinline__ SplitTile split_tile(const SplitSegs& t, int kt) {
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

struct SplitFrags {
  uint4v a[4][3], b[2][3];                 // [sub-tile][plane]: 8 bf16 = the lane's 8 k of its row
};

template <bool TA, bool TB>
__global__ __launch_bounds__(256, 1) void k_gemm_split(const GemmArgs a) {
  constexpr int TM = 4, TN = 2, WGN = 2;
  extern __shared__ __attribute__((aligned(16))) unsigned char slds[];

  int b, tile_id, piece, S;
  unsigned tj;
  {
    TileMap<S_BM> map;
    map.init(a, threadIdx.x & 63);
    if (!map.select(a, blockIdx.x, threadIdx.x & 63, b, tile_id, tj, piece, S)) return;
  }
  b = __builtin_amdgcn_readfirstlane(b);
  tile_id = __builtin_amdgcn_readfirstlane(tile_id);
  piece = __builtin_amdgcn_readfirstlane(piece);
  S = __builtin_amdgcn_readfirstlane(S);
  tj = __builtin_amdgcn_readfirstlane(tj);
  const TileBase tb(a, b);
  const int M = tb.M, K = tb.K, N = a.N;
  const float* A = tb.A;
  const float* B = tb.B;
  float* C = tb.C;
  const int tile_m = tile_id / a.tiles_n, tile_n = tile_id - tile_m * a.tiles_n;
  const int m0 = tile_m * S_BM, n0 = tile_n * S_BN;
  if (m0 >= M) return;

  typedef typename std::conditional<TA, SplitLoaderMN<S_BM>, SplitLoaderK<S_BM>>::type LoaderA;
  typedef typename std::conditional<TB, SplitLoaderK<S_BN>, SplitLoaderMN<S_BN>>::type LoaderB;
  static_assert(LoaderA::NG == 4 && LoaderB::NG == 2, "six groups of four values per thread and k-tile");
  constexpr int NFA = LoaderA::NF, NFB = LoaderB::NF;
  LoaderA la[3];                             // three register sets: tile t lives in set t % 3 from its request until its split
  LoaderB lb[3];

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave / WGN, wn = wave - wm * WGN;
  const int l31 = lane & 31, lhi = lane >> 5;

  floatx16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // k-tiles: the main operand pair, then the extra segments
  const int nk_main = (K + SBK - 1) / SBK, nk_full = K / SBK;
  SplitSegs seg;
  seg.A0 = A; seg.B0 = B; seg.lda0 = a.lda; seg.ldb0 = a.ldb; seg.K0 = K;
  seg.A1 = seg.A2 = A; seg.B1 = seg.B2 = B; seg.lda1 = seg.lda2 = a.lda; seg.ldb1 = seg.ldb2 = a.ldb; seg.K1 = seg.K2 = K;
  seg.nk_main = nk_main;
  seg.nkx0 = 0;
  int nkx1 = 0;
  if (a.nx > 0) {
    const size_t roff = a.ragged == 1 ? (size_t)a.gptr[b] : 0;
    seg.A1 = a.xA[0] + (size_t)b * a.xsA[0] + roff * a.xlda[0];
    seg.B1 = a.xB[0] + (size_t)b * a.xsB[0];
    seg.lda1 = a.xlda[0]; seg.ldb1 = a.xldb[0]; seg.K1 = a.xK[0];
    seg.nkx0 = (a.xK[0] + SBK - 1) / SBK;
    if (a.nx > 1) {
      seg.A2 = a.xA[1] + (size_t)b * a.xsA[1] + roff * a.xlda[1];
      seg.B2 = a.xB[1] + (size_t)b * a.xsB[1];
      seg.lda2 = a.xlda[1]; seg.ldb2 = a.xldb[1]; seg.K2 = a.xK[1];
      nkx1 = (a.xK[1] + SBK - 1) / SBK;
    }
  }
  const int nk = nk_main + seg.nkx0 + nkx1;
  const int kbeg = S > 1 ? (int)(((long long)nk * piece) / S) : 0;
  const int kend = S > 1 ? (int)(((long long)nk * (piece + 1)) / S) : nk;
  const int n = kend - kbeg;                 // this workgroup's k-tiles: local index 0 .. n - 1
  // edges: rows / columns past the extent are clamped to the last valid one (they only reach outputs that are never stored)
  const int a_last = TA ? ((M - 1) & ~3) : M - 1, b_last = TB ? N - 1 : ((N - 1) & ~3);

  unsigned offA[NFA], offB[NFB];
  la[0].offsets(offA, a.lda, m0, a_last);
  lb[0].offsets(offB, a.ldb, n0, b_last);
  const __amdgpu_buffer_rsrc_t rsrcA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(A), 0, 0xffffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrcB = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(B), 0, 0xffffffff, 0x00020000);

  // LDS addresses: per stage, the lane's fragment rows (read) and the thread's units (write)
  const unsigned fa_off = (unsigned)(wm * 128 + l31) * SROW + lhi * 16, fb_off = 3 * S_PLA + (unsigned)(wn * 64 + l31) * SROW + lhi * 16;
  const unsigned wa_off = la[0].wbase(), wb_off = 3 * S_PLA + lb[0].wbase();

  // request local tile `lt` (clamped into the piece: a tile past the end is never used, its addresses must be valid)
  auto request_any = [&](LoaderA& ra, LoaderB& rb, int lt) {
    const SplitTile t = split_tile(seg, kbeg + (lt < n ? lt : n - 1));
#pragma unroll
    for (int i = 0; i < NFA; ++i) ra.load_any(i, t.A, t.lda, m0, a_last, t.k0, t.klim);
#pragma unroll
    for (int i = 0; i < NFB; ++i) rb.load_any(i, t.B, t.ldb, n0, b_last, t.k0, t.klim);
  };

  // ---- the split of one group of four values: eight micro-steps (see the file header)
  struct GroupState {
    float x[4], r1[4], r2[4];
    unsigned hp[2], mp[2], lp[2];
  };
  auto split_step = [&](auto u_c, auto st_c, auto masked_c, GroupState& gs, LoaderA& ra, LoaderB& rb, unsigned char* stage_base, int k0,
                        int klim) {
    constexpr int U = decltype(u_c)::value, ST = decltype(st_c)::value;
    constexpr bool MASKED = decltype(masked_c)::value;
    constexpr bool IS_A = U < 4;
    constexpr int UL = IS_A ? U : U - 4;
    if constexpr (ST == 0) {
      if constexpr (IS_A) ra.template get<UL>(gs.x); else rb.template get<UL>(gs.x);
      if constexpr (MASKED) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int ke = k0 + (IS_A ? ra.template kof<UL>(e) : rb.template kof<UL>(e));
          gs.x[e] = ke < klim ? gs.x[e] : 0.f;
        }
      }
      gs.hp[0] = pack_bf16(gs.x[0], gs.x[1]);
      gs.hp[1] = pack_bf16(gs.x[2], gs.x[3]);
    } else if constexpr (ST == 1 || ST == 2) {
      constexpr int h = ST - 1;
      const float e0 = __builtin_bit_cast(float, gs.hp[h] << 16), e1 = __builtin_bit_cast(float, gs.hp[h] & 0xffff0000u);
      gs.r1[2 * h] = gs.x[2 * h] - e0;
      gs.r1[2 * h + 1] = gs.x[2 * h + 1] - e1;
      asm volatile("" : "+v"(gs.r1[2 * h]), "+v"(gs.r1[2 * h + 1]));
    } else if constexpr (ST == 3) {
      gs.mp[0] = pack_bf16(gs.r1[0], gs.r1[1]);
      gs.mp[1] = pack_bf16(gs.r1[2], gs.r1[3]);
    } else if constexpr (ST == 4 || ST == 5) {
      constexpr int h = ST - 4;
      const float e0 = __builtin_bit_cast(float, gs.mp[h] << 16), e1 = __builtin_bit_cast(float, gs.mp[h] & 0xffff0000u);
      gs.r2[2 * h] = gs.r1[2 * h] - e0;
      gs.r2[2 * h + 1] = gs.r1[2 * h + 1] - e1;
      asm volatile("" : "+v"(gs.r2[2 * h]), "+v"(gs.r2[2 * h + 1]));
    } else if constexpr (ST == 6) {
      gs.lp[0] = pack_bf16(gs.r2[0], gs.r2[1]);
      gs.lp[1] = pack_bf16(gs.r2[2], gs.r2[3]);
    } else {
      if constexpr (IS_A) {
        unsigned char* st = stage_base + wa_off;
        ra.template put<UL, S_PLA>(st, 0, gs.hp[0], gs.hp[1]);
        ra.template put<UL, S_PLA>(st, 1, gs.mp[0], gs.mp[1]);
        ra.template put<UL, S_PLA>(st, 2, gs.lp[0], gs.lp[1]);
      } else {
        unsigned char* st = stage_base + wb_off;
        rb.template put<UL, S_PLB>(st, 0, gs.hp[0], gs.hp[1]);
        rb.template put<UL, S_PLB>(st, 1, gs.mp[0], gs.mp[1]);
        rb.template put<UL, S_PLB>(st, 2, gs.lp[0], gs.lp[1]);
      }
    }
  };
  // micro-step s (0 .. 47) of a tile's split: group s / 8, step s % 8
  auto micro = [&](auto s_c, auto masked_c, GroupState (&gs)[6], LoaderA& ra, LoaderB& rb, unsigned char* stage_base, int k0, int klim) {
    constexpr int SIDX = decltype(s_c)::value;
    split_step(std::integral_constant<int, SIDX / 8>(), std::integral_constant<int, SIDX % 8>(), masked_c, gs[SIDX / 8], ra, rb, stage_base,
               k0, klim);
  };
  // a whole tile's split in one go (prologue)
  auto split_all = [&](LoaderA& ra, LoaderB& rb, unsigned char* stage_base, int lt) {
    const SplitTile t = split_tile(seg, kbeg + (lt < n ? lt : n - 1));
    GroupState gs[6];
    auto run = [&](auto self, auto s_c) {
      constexpr int SIDX = decltype(s_c)::value;
      if constexpr (SIDX < 48) {
        micro(s_c, std::true_type(), gs, ra, rb, stage_base, t.k0, t.klim);
        self(self, std::integral_constant<int, SIDX + 1>());
      }
    };
    run(run, std::integral_constant<int, 0>());
  };
  // fragment q (0 .. 35) of a tile: A sub-tile i, plane p, half h (24 of them), then B
  auto frag_read = [&](auto q_c, SplitFrags& f, const unsigned char* stage_base) {
    constexpr int Q = decltype(q_c)::value;
    if constexpr (Q < 24) {
      constexpr int i = Q / 6, p = (Q % 6) / 2, h = Q % 2;
      const uint2 v = *reinterpret_cast<const uint2*>(stage_base + fa_off + i * 32 * SROW + p * S_PLA + h * 8);
      f.a[i][p][2 * h] = v.x;
      f.a[i][p][2 * h + 1] = v.y;
    } else {
      constexpr int R = Q - 24, j = R / 6, p = (R % 6) / 2, h = R % 2;
      const uint2 v = *reinterpret_cast<const uint2*>(stage_base + fb_off + j * 32 * SROW + p * S_PLB + h * 8);
      f.b[j][p][2 * h] = v.x;
      f.b[j][p][2 * h + 1] = v.y;
    }
  };

  SplitFrags fr[2];
  // ---- prologue: tiles 0, 1 split into stages 0, 1; tiles 2, 3, 4 in flight in the three sets; fragments of tile 0 in fr[0]
  request_any(la[0], lb[0], 0);
  request_any(la[1], lb[1], 1);
  request_any(la[2], lb[2], 2);
  split_all(la[0], lb[0], slds, 0);
  request_any(la[0], lb[0], 3);
  split_all(la[1], lb[1], slds + S_STAGE, 1);
  request_any(la[1], lb[1], 4);
  __syncthreads();
  {
    auto run = [&](auto self, auto q_c) {
      constexpr int Q = decltype(q_c)::value;
      if constexpr (Q < 36) {
        frag_read(q_c, fr[0], slds);
        self(self, std::integral_constant<int, Q + 1>());
      }
    };
    run(run, std::integral_constant<int, 0>());
  }

  // ---- one k-tile.  POS = local tile index mod 6 fixes every buffer: fragments fr[POS & 1] (being multiplied) and fr[~POS & 1] (being
  // read, tile lt + 1, stage (POS + 1) % 3), the set (POS + 2) % 3 being split into stage (POS + 2) % 3 (tile lt + 2) and refilled
  // (tile lt + 5).  FULL: tiles lt + 2 and lt + 5 are whole tiles of the main operand pair -- buffer loads, no masks, no conditional.
  // Otherwise the generic request / masked split.  Past the end of the piece everything still runs, on clamped addresses, into
  // buffers nobody multiplies: no conditional there either.
  constexpr int PA_[6] = {0, 0, 1, 0, 1, 2}, PB_[6] = {0, 1, 0, 2, 1, 0};
  auto tile_step = [&](auto pos_c, auto full_c, int lt) {
    constexpr int POS = decltype(pos_c)::value;
    constexpr bool FULL = decltype(full_c)::value;
    constexpr int FC = POS & 1, FN = FC ^ 1, SR = (POS + 1) % 3, SW = (POS + 2) % 3;
    const unsigned char* rstage = slds + SR * S_STAGE;
    unsigned char* wstage = slds + SW * S_STAGE;
    LoaderA& ra = la[SW];
    LoaderB& rb = lb[SW];
    GroupState gs[6];
    int k0s = 0, klims = 0;
    SplitTile tnext;
    unsigned soffA = 0, soffB = 0;
    if constexpr (!FULL) {
      const SplitTile ts = split_tile(seg, kbeg + (lt + 2 < n ? lt + 2 : n - 1));
      k0s = ts.k0;
      klims = ts.klim;
      tnext = split_tile(seg, kbeg + (lt + 5 < n ? lt + 5 : n - 1));
    } else {
      soffA = LoaderA::soffset(a.lda, (kbeg + lt + 5) * SBK);
      soffB = LoaderB::soffset(a.ldb, (kbeg + lt + 5) * SBK);
    }
    auto run = [&](auto self, auto m_c) {
      constexpr int MI = decltype(m_c)::value;
      if constexpr (MI < 48) {
        constexpr int t = MI / 8, ij = MI % 8, i = ij >> 1, j = ij & 1;
        // operands swapped (B fragment first): the accumulator holds the transposed sub-tile, see gemm_epilogue
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fr[FC].b[j][PB_[t]]),
                                                            __builtin_bit_cast(bf16x8, fr[FC].a[i][PA_[t]]), acc[i][j], 0, 0, 0);
        if constexpr (MI < 36) frag_read(std::integral_constant<int, MI>(), fr[FN], rstage);
        micro(std::integral_constant<int, MI>(), std::integral_constant<bool, !FULL>(), gs, ra, rb, wstage, k0s, klims);
        // the set's registers are free once its groups have been picked up (A: micro-step 24, B: 40): refill
        if constexpr (MI >= 36 && MI < 36 + NFA) {
          if constexpr (FULL) ra.load_buf(MI - 36, rsrcA, offA, soffA);
          else ra.load_any(MI - 36, tnext.A, tnext.lda, m0, a_last, tnext.k0, tnext.klim);
        }
        if constexpr (MI >= 44 && MI < 44 + NFB) {
          if constexpr (FULL) rb.load_buf(MI - 44, rsrcB, offB, soffB);
          else rb.load_any(MI - 44, tnext.B, tnext.ldb, n0, b_last, tnext.k0, tnext.klim);
        }
        __builtin_amdgcn_sched_barrier(0);
        self(self, std::integral_constant<int, MI + 1>());
      }
    };
    run(run, std::integral_constant<int, 0>());
    __syncthreads();
  };
  typedef std::true_type FULL_;
  typedef std::false_type ANY_;
#define SPLIT_POS(P_) std::integral_constant<int, P_>()
  int lt = 0;
  // steps whose tiles lt + 2 and lt + 5 are whole main-pair tiles
  const int full_steps = nk_full - 5 - kbeg;
  for (; lt + 6 <= full_steps; lt += 6) {
    tile_step(SPLIT_POS(0), FULL_(), lt);
    tile_step(SPLIT_POS(1), FULL_(), lt + 1);
    tile_step(SPLIT_POS(2), FULL_(), lt + 2);
    tile_step(SPLIT_POS(3), FULL_(), lt + 3);
    tile_step(SPLIT_POS(4), FULL_(), lt + 4);
    tile_step(SPLIT_POS(5), FULL_(), lt + 5);
  }
  for (; lt < n; ++lt) {
    switch (lt % 6) {
      case 0: tile_step(SPLIT_POS(0), ANY_(), lt); break;
      case 1: tile_step(SPLIT_POS(1), ANY_(), lt); break;
      case 2: tile_step(SPLIT_POS(2), ANY_(), lt); break;
      case 3: tile_step(SPLIT_POS(3), ANY_(), lt); break;
      case 4: tile_step(SPLIT_POS(4), ANY_(), lt); break;
      default: tile_step(SPLIT_POS(5), ANY_(), lt); break;
    }
  }
#undef SPLIT_POS

  float* const lds_f = reinterpret_cast<float*>(slds);
  if (S > 1) {
    // piece of a tail tile: raw accumulators into this piece's slab (k_gemm_fixup<2, 2, 4, 2> adds the slabs; same layout as k_gemm_f32)
    float* slab = a.ws + ((size_t)tj * S + piece) * (size_t)(S_BM * S_BN) + (size_t)wave * (TM * TN * 16 * 64) + lane * 4;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int g = 0; g < 4; ++g)
          *reinterpret_cast<float4*>(slab + ((i * TN + j) * 4 + g) * 256) =
              make_float4(acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]);
    return;
  }
  // (the last step ended with a barrier: the LDS is free for the parking strips of the epilogue)
  gemm_epilogue<TM, TN>(a, C, M, N, m0 + wm * TM * 32, n0 + wn * TN * 32, acc, lds_f + wave * 32 * (TN * 32 + 4), lane);
}

// Workgroups the chip holds at once: one per CU
static const int kSplitResident = 256;

// Launch the split kernel for a product that qualifies (gemm.hip: gemm_dispatch decided: 128 x 128 route, every operand segment
// fit for unguarded 16-byte loads).  Returns CGC_EINVAL when the shape is outside what the kernel indexes (the caller then runs the
// exact kernel).
int gemm_split_launch(const GemmArgs& a0, int transA, int transB, int batch, int m_extent, int k_extent, float* ws, int64_t ws_floats,
                      hipStream_t stream) {
  if (transA && transB) return CGC_EINVAL;
  GemmArgs a = a0;
  a.tiles_n = ceil_div(a.N, S_BN);
  static const int map_mode = getenv("CGC_GEMM_MAP") ? atoi(getenv("CGC_GEMM_MAP")) : 3;
  a.map_mode = map_mode;
  const long long per_batch = (long long)ceil_div(m_extent, S_BM) * a.tiles_n;
  const long long tiles = per_batch * batch;
  if (per_batch <= 0 || tiles > 0x7ffffff0LL) return CGC_EINVAL;
  // k offsets are 32-bit scalar byte offsets (16 k rows of an [K, .] operand at a time): same limits as the exact kernel checked
  a.per_batch = (int)per_batch;
  a.nb = batch;
  a.ws = nullptr;
  a.resident = 0;
  a.s_max = 1;
  int extra = 0;
  static const int split_on = getenv("CGC_GEMM_SPLIT") ? atoi(getenv("CGC_GEMM_SPLIT")) : 1;
  if (ws != nullptr && split_on) {
    long long kt = ceil_div(k_extent, SBK);
    for (int i = 0; i < a.nx; ++i) kt += ceil_div(a.xK[i], SBK);
    const int s_max = (int)(kt / 8 < 12 ? kt / 8 : 12);                 // a piece keeps >= 8 k-tiles: the pipeline is five deep
    const long long max_pieces = kSplitResident + kSplitResident / 2;
    if (s_max >= 2 && max_pieces * S_BM * S_BN <= ws_floats) {
      a.ws = ws;
      a.resident = kSplitResident;
      a.s_max = s_max;
      extra = (int)max_pieces;
    }
  }
  int xk = 0;
  for (int i = 0; i < a.nx; ++i) xk += a.xK[i];
  const int trec = cgc_timing_begin(CGC_TAG_GEMM_128, a.M, a.N, a.K, batch, a.ragged, a.ragged ? (a.ragged == 1 ? m_extent : k_extent) : 0,
                                    xk, stream);
  dim3 grid((unsigned)(tiles + extra)), block(256);
#define SPLIT_LAUNCH(TA_, TB_)                                                                              \
  do {                                                                                                      \
    static bool attr__[CGC_MAX_DEVICES] = {};                                                               \
    cgc_allow_lds(reinterpret_cast<const void*>(&k_gemm_split<TA_, TB_>), S_LDS, attr__);                   \
    hipLaunchKernelGGL((k_gemm_split<TA_, TB_>), grid, block, S_LDS, stream, a);                            \
  } while (0)
  if (!transA && !transB) SPLIT_LAUNCH(false, false);
  else if (!transA) SPLIT_LAUNCH(false, true);
  else SPLIT_LAUNCH(true, false);
#undef SPLIT_LAUNCH
  CGC_RETURN_IF_LAUNCH_FAILED();
  if (a.ws != nullptr) {
    const long long lmax = tiles < kSplitResident ? tiles : kSplitResident - 1;
    hipLaunchKernelGGL((k_gemm_fixup<2, 2, 4, 2>), dim3((unsigned)(lmax * 4 * 4 * 2)), dim3(64), 0, stream, a);
    CGC_RETURN_IF_LAUNCH_FAILED();
  }
  cgc_timing_end(trec, stream);
  return 0;
}
