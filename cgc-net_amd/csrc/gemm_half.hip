// fp32 products on the fp16 matrix cores of gfx950 in THREE passes: every operand element, scaled by a power of two chosen per OUTPUT
// TILE from the largest magnitude of the tile's operand panel, is split into two fp16 values  s x = h + l  (h = fp16(s x), l = fp16(s x - h), both
// round-to-nearest; the residual is exact in fp32), and the product is formed as the three most significant pairs  l h + h l + h h  in
// the matrix core's fp32 accumulator, descaled in registers before the epilogue.  h + l carries 22-23 of the 24 significand bits
// (|s x - h - l| <= 2^-23 |s x| while l is a normal fp16) and the dropped pair l l is below 2^-22 |a||b|: with fp32 accumulation over
// K >= 160 terms the difference to the exact chain of gemm.hip is a fraction of that chain's own rounding (measured per form by
// tests/test_split_gemm_gpu.py; tools/f16_split_model.py is the numpy model that motivated it).  v_mfma_f32_32x32x16_f16 runs at the
// bf16 rate: three pairs are 96 matrix-pipe cycles per 32 x 32 x 16 block against the 192 of gemm_split.hip's six and the 512 of the
// fp32 chain.
//
// What fp16 costs is RANGE, and that is what the scale is for: 5 exponent bits, normal from 2^-14.  A first launch (k_gemm_absmax)
// takes max |x| over the 256 rows of op(A) that an output tile multiplies (all of K, every segment) and over its 128 columns of
// op(B) -- one streaming pass over both operands: the price of the mode, ~50 us per 263 MB operand -- and the product kernel places
// each panel's maximum in [2^14, 2^15).  Elements down to 2^-17 of their panel's maximum then keep both planes normal; below, l and
// finally h go subnormal and the element's ABSOLUTE error stops shrinking at 2^-25 / s = 2^-40 of the panel's maximum (fp32 itself:
// 2^-24 of the element).  Zeros are exact; a panel that is all zero takes s = 1.  (One scale per batch item was the first version:
// the gradient of a 32-graph batch, one item of 58 k rows, had 46 % of its elements below 2^-17 of the tensor's maximum.)
// Domain: finite inputs (an infinite element makes the tiles of its panel NaN; the exact kernel confines it to its row / column).
//
// Third MODE of the same entry points (CGC_GEMM_SPLIT_F16; cgc_level_desc.flags bit 2), for the same products and forms as
// gemm_split.hip, and with its structure: tile 256 x 128, one wave per SIMD with a 128 x 64 wave tile, k-tiles of 16, two LDS stages
// of [2 planes][256 + 128 rows][16 k], global -> registers two k-tiles ahead of the split, split -> LDS two ahead of the MFMAs, a full
// second set of fragment registers (two planes instead of three leave room for it: no retirement order to respect), one barrier per
// k-tile, half tiles.  Per k-tile a wave issues 24 MFMAs (768 pipe cycles) + 12 fragment reads + 12-16 LDS writes + 6 buffer loads +
// ~75 vector instructions.
#include "gemm_split_common.hpp"

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

constexpr int H_PLA = S_BM * SROW, H_PLB = S_BN * SROW, H_STAGE = 2 * H_PLA + 2 * H_PLB, H_LDS = 2 * H_STAGE;   // 73728 bytes
// scale slots at the end of the caller's workspace (floats reserved: cgc_gemm_ws_floats() counts them): max |A| per (batch item, 256-row
// tile), max |B| per (batch item, 128-column tile)
constexpr int H_SCALE_FLOATS = 65536;

__device__ __forceinline__ unsigned pack_f16(float a, float b) {      // round to nearest even, a in the low half
  float2v t;
  t[0] = a;
  t[1] = b;
  return __builtin_bit_cast(unsigned, __builtin_convertvector(t, f16x2));
}

// x - float(h) for the low / high fp16 of a packed pair, one instruction each: v_fma_mix_f32 reads the half in place (the compiler
// converts and subtracts: 8 instead of 4 vector instructions per group of four values).  Exact: the residual of a rounding is a float.
__device__ __forceinline__ float resid_lo(unsigned h, float x) {
  float r;
  asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r) : "v"(h), "v"(x));
  return r;
}
__device__ __forceinline__ float resid_hi(unsigned h, float x) {
  float r;
  asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(h), "v"(x));
  return r;
}

// the scale of an operand from the bits of its largest magnitude: 2^(14 - floor(log2 max)), kept inside the normal floats together
// with its reciprocal; 1 for an operand that is all zero
__device__ __forceinline__ void half_scale(unsigned maxbits, float& s, float& inv) {
  const int e = (int)((maxbits >> 23) & 0xffu);
  int ex = 127 + 14 - (e - 127);
  ex = ex < 1 ? 1 : ex > 253 ? 253 : ex;
  if (maxbits == 0u) ex = 127;
  s = __builtin_bit_cast(float, (unsigned)ex << 23);
  inv = __builtin_bit_cast(float, (unsigned)(254 - ex) << 23);
}

// ---- the split of one group of four values in four micro-steps (sidx = 4 * group + step; groups 0-3: operand A, 4-5: operand B)
struct HalfGroup {
  float x[4], r[4];
  unsigned hp[2], lp[2];
};
template <class LoaderA, class LoaderB, bool MASKED>
__device__ __forceinline__ void half_micro(int sidx, HalfGroup (&gs)[6], const float4 (&ra)[LoaderA::NF], const float4 (&rb)[LoaderB::NF],
                                           unsigned char* wa, unsigned char* wb, int k0, int klim, float sa, float sb) {
  const int u = sidx >> 2, st = sidx & 3;
  HalfGroup& s = gs[u];
  if (st == 0) {
    if (u < 4) LoaderA::get(ra, u, s.x); else LoaderB::get(rb, u - 4, s.x);
    const float sc = u < 4 ? sa : sb;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      s.x[e] *= sc;
      if (MASKED) {
        const int ke = k0 + (u < 4 ? LoaderA::kof(u, e) : LoaderB::kof(u - 4, e));
        s.x[e] = ke < klim ? s.x[e] : 0.f;
      }
    }
    s.hp[0] = pack_f16(s.x[0], s.x[1]);
    s.hp[1] = pack_f16(s.x[2], s.x[3]);
  } else if (st == 1) {
#ifdef H_ABL_NOSPLIT      // (timing-only ablations, tools/variant_lib.sh: the result is wrong)
    return;
#endif
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      s.r[2 * h] = resid_lo(s.hp[h], s.x[2 * h]);
      s.r[2 * h + 1] = resid_hi(s.hp[h], s.x[2 * h + 1]);
    }
    asm volatile("" : "+v"(s.r[0]), "+v"(s.r[1]), "+v"(s.r[2]), "+v"(s.r[3]));
  } else if (st == 2) {
#ifdef H_ABL_NOSPLIT
    s.lp[0] = s.hp[0];
    s.lp[1] = s.hp[1];
    return;
#endif
    s.lp[0] = pack_f16(s.r[0], s.r[1]);
    s.lp[1] = pack_f16(s.r[2], s.r[3]);
  } else {
#ifdef H_ABL_NOWRITE
    return;
#endif
    if (u < 4) {
      LoaderA::put(wa, u, 0, s.hp[0], s.hp[1]);
      LoaderA::put(wa, u, H_PLA, s.lp[0], s.lp[1]);
    } else {
      LoaderB::put(wb, u - 4, 0, s.hp[0], s.hp[1]);
      LoaderB::put(wb, u - 4, H_PLB, s.lp[0], s.lp[1]);
    }
  }
}

// One tile (HALF: a whole tile of <= 128 valid rows: 64 x 64 per wave) or one K piece of a tail tile; see gemm_split.hip's split_body
// for the pipeline, which this follows step by step with two planes and 24 MFMAs per k-tile.
template <bool TA, bool TB, bool HALF>
__device__ __forceinline__ void half_body(const GemmArgs& a, const int b, const int tile_id, const int piece, const int S, const unsigned tj,
                                          unsigned char* const slds) {
  constexpr int TM = 4, TN = 2, WGN = 2, NA = HALF ? 2 : 4;
  const TileBase tb(a, b);
  const int M = tb.M, K = tb.K, N = a.N;
  const float* A = tb.A;
  const float* B = tb.B;
  float* C = tb.C;
  const int tile_m = tile_id / a.tiles_n, tile_n = tile_id - tile_m * a.tiles_n;
  const int m0 = tile_m * S_BM, n0 = tile_n * S_BN;
  if (m0 >= M) return;

  float sa, sb, isa, isb;
  {
    const int ta = a.per_batch / a.tiles_n;                    // row tiles of the largest item: the slot layout of k_gemm_absmax
    const unsigned* sc = a.scale + (size_t)b * (ta + a.tiles_n);
    half_scale(sc[tile_m], sa, isa);
    half_scale(sc[ta + tile_n], sb, isb);
  }

  typedef typename std::conditional<TA, SplitLoaderMN<S_BM, SROW>, SplitLoaderK<S_BM, SROW>>::type LoaderA;
  typedef typename std::conditional<TB, SplitLoaderK<S_BN, SROW>, SplitLoaderMN<S_BN, SROW>>::type LoaderB;
  static_assert(LoaderA::NG == 4 && LoaderB::NG == 2, "six groups of four values per thread and k-tile");
  constexpr int NFA = LoaderA::NF, NFB = LoaderB::NF;
  float4 ra[2][NFA], rb[2][NFB];            // two register sets: tile t lives in set t % 2 from its request until its split

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
  const int n = kend - kbeg;
  const int a_last = TA ? ((M - 1) & ~3) : M - 1, b_last = TB ? N - 1 : ((N - 1) & ~3);

  unsigned offA[NFA], offB[NFB];
  LoaderA::offsets(offA, a.lda, m0, a_last);
  LoaderB::offsets(offB, a.ldb, n0, b_last);
  const __amdgpu_buffer_rsrc_t rsrcA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(A), 0, 0xffffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrcB = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(B), 0, 0xffffffff, 0x00020000);

  const unsigned fa_off = (unsigned)((HALF ? wm * 64 : wm * 128) + l31) * SROW + lhi * 16, fb_off = 2 * H_PLA + (unsigned)(wn * 64 + l31) * SROW + lhi * 16;
  const unsigned wa_off = LoaderA::wbase(), wb_off = 2 * H_PLA + LoaderB::wbase();

  uint4v fa[2][4][2], fb[2][2][2];          // [set = tile parity][sub-tile][plane h, l]
  // ---- prologue: tiles 0, 1 split into stages 0, 1; tiles 2, 3 in flight in the two sets; the fragments of tile 0 in set 0.
  // An empty k range loads nothing and goes straight to the epilogue / its slab with zero accumulators (as gemm_split.hip).
  if (n > 0) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {             // q: 0, 1 request tiles 0, 1; 2: split 0, request 2; 3: split 1, request 3
      const int set = q & 1;
      if (q >= 2) {
        const SplitTile t = split_tile(seg, kbeg + (q - 2 < n ? q - 2 : n - 1));
        HalfGroup gs[6];
#pragma unroll
        for (int sidx = 0; sidx < 24; ++sidx)
          half_micro<LoaderA, LoaderB, true>(sidx, gs, ra[set], rb[set], slds + (q - 2) * H_STAGE + wa_off, slds + (q - 2) * H_STAGE + wb_off,
                                             t.k0, t.klim, sa, sb);
      }
      const SplitTile t = split_tile(seg, kbeg + (q < n ? q : n - 1));
#pragma unroll
      for (int i = 0; i < NFA; ++i) ra[set][i] = LoaderA::load_any(i, t.A, t.lda, m0, a_last, t.k0, t.klim);
#pragma unroll
      for (int i = 0; i < NFB; ++i) rb[set][i] = LoaderB::load_any(i, t.B, t.ldb, n0, b_last, t.k0, t.klim);
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NA; ++i)
#pragma unroll
      for (int p = 0; p < 2; ++p) fa[0][i][p] = frag16(slds + fa_off + i * 32 * SROW + p * H_PLA);
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int p = 0; p < 2; ++p) fb[0][j][p] = frag16(slds + fb_off + j * 32 * SROW + p * H_PLB);
    __syncthreads();                          // (step 0 writes tile 2 into stage 0: everybody has read tile 0 out of it)
  }

  auto tile_step = [&](auto pos_c, auto full_c, int lt) {
    constexpr int POS = decltype(pos_c)::value;          // local tile index mod 2: its stage, its register sets
    constexpr bool FULL = decltype(full_c)::value;
    const unsigned char* rstage = slds + (POS ^ 1) * H_STAGE;      // tile lt + 1
    unsigned char* wa = slds + POS * H_STAGE + wa_off;             // tile lt + 2 goes where tile lt was
    unsigned char* wb = slds + POS * H_STAGE + wb_off;
    HalfGroup gs[6];
    int k0s = 0, klims = 0;
    SplitTile tnext;
    unsigned soffA = 0, soffB = 0;
    if constexpr (!FULL) {
      const SplitTile ts = split_tile(seg, kbeg + (lt + 2 < n ? lt + 2 : n - 1));
      k0s = ts.k0;
      klims = ts.klim;
      tnext = split_tile(seg, kbeg + (lt + 4 < n ? lt + 4 : n - 1));
    } else {
      tnext.A = A; tnext.B = B; tnext.lda = a.lda; tnext.ldb = a.ldb; tnext.klim = K; tnext.k0 = 0;
      const int tl = min(kbeg + lt + 4, nk_full - 1);
      soffA = LoaderA::soffset(a.lda, tl * SBK);
      soffB = LoaderB::soffset(a.ldb, tl * SBK);
    }
    constexpr int PA_[3] = {1, 0, 0}, PB_[3] = {0, 1, 0};          // l h, h l, h h: the small pairs first
#pragma clang loop unroll(full)
    for (int m = 0; m < 24; ++m) {
      const int t = m / 8, ij = m % 8, i = ij >> 1, j = ij & 1;
      if (!HALF || i < 2)
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, fb[POS][j][PB_[t]]), __builtin_bit_cast(f16x8, fa[POS][i][PA_[t]]),
                                                           acc[i][j], 0, 0, 0);
      // the next tile's fragments into the other register set: 12 reads of 16 bytes (8 for a half tile)
#ifndef H_ABL_NOFRAG
      if (m < 8) { if ((m >> 1) < NA) fa[POS ^ 1][m >> 1][m & 1] = frag16(rstage + fa_off + (m >> 1) * 32 * SROW + (m & 1) * H_PLA); }
      else if (m < 12) fb[POS ^ 1][(m - 8) >> 1][m & 1] = frag16(rstage + fb_off + ((m - 8) >> 1) * 32 * SROW + (m & 1) * H_PLB);
#endif
      half_micro<LoaderA, LoaderB, !FULL>(m, gs, ra[POS], rb[POS], wa, wb, k0s, klims, sa, sb);
#ifndef H_ABL_NOLOAD
      if (m >= 13 && m < 13 + NFA) {            // (group 3 of A took its values at m = 12)
        if constexpr (FULL) ra[POS][m - 13] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsrcA, offA[m - 13], soffA, 0));
        else ra[POS][m - 13] = LoaderA::load_any(m - 13, tnext.A, tnext.lda, m0, a_last, tnext.k0, tnext.klim);
      }
      if (m >= 21 && m < 21 + NFB) {            // (group 1 of B took its values at m = 20)
        if constexpr (FULL) rb[POS][m - 21] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsrcB, offB[m - 21], soffB, 0));
        else rb[POS][m - 21] = LoaderB::load_any(m - 21, tnext.B, tnext.ldb, n0, b_last, tnext.k0, tnext.klim);
      }
#endif
      __builtin_amdgcn_sched_barrier(0);
    }
#ifndef H_ABL_NOBAR
    __syncthreads();
#endif
  };
  typedef std::true_type FULL_;
  typedef std::false_type ANY_;
#define HALF_POS(P_) std::integral_constant<int, P_>()
  int lt = 0;
  const int last_special = nk - nk_full;
  const int full_steps = min(last_special > 0 && kend > nk_full ? nk_full - 4 - kbeg : nk_full - 2 - kbeg, n);
  for (; lt + 2 <= full_steps; lt += 2) {
    tile_step(HALF_POS(0), FULL_(), lt);
    tile_step(HALF_POS(1), FULL_(), lt + 1);
  }
  for (; lt < n; ++lt) {
    if ((lt & 1) == 0) tile_step(HALF_POS(0), ANY_(), lt);
    else tile_step(HALF_POS(1), ANY_(), lt);
  }
#undef HALF_POS

  // descale: two multiplications (1 / (sa sb) alone may leave the normal range)
#pragma unroll
  for (int i = 0; i < NA; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = (acc[i][j][r] * isa) * isb;

  float* const lds_f = reinterpret_cast<float*>(slds);
  if (S > 1) {
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
  if constexpr (HALF) {
    floatx16 ah[2][TN];                    // (by value: a reference to a part of acc would put the accumulators in memory)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) ah[i][j] = acc[i][j];
    gemm_epilogue<2, TN>(a, C, M, N, m0 + wm * 64, n0 + wn * TN * 32, ah, lds_f + wave * 32 * (TN * 32 + 4), lane);
  } else {
    gemm_epilogue<TM, TN>(a, C, M, N, m0 + wm * TM * 32, n0 + wn * TN * 32, acc, lds_f + wave * 32 * (TN * 32 + 4), lane);
  }
}

template <bool TA, bool TB>
__global__ __launch_bounds__(256, 1) void k_gemm_half(const GemmArgs a) {
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
  const int rows_left = tb.M - (tile_id / a.tiles_n) * S_BM;
  if (S == 1 && rows_left <= 128) {
    half_body<TA, TB, true>(a, b, tile_id, piece, S, tj, slds);
    return;
  }
  half_body<TA, TB, false>(a, b, tile_id, piece, S, tj, slds);
}

// ---- max |x| per OUTPUT TILE's operand panel: for batch item b, scale[b * (ta + tn) + t] = the bits of max |x| over rows
// 256 t .. 256 t + 255 of op(A) (all of K, every segment) and scale[b * (ta + tn) + ta + t] over columns 128 t .. 128 t + 127 of op(B)
// (bits of a non-negative float: unsigned order is magnitude order; the caller zeroes the slots).  grid (panels x their sub-ranges,
// batch items): a workgroup streams a share of one panel's rows -- stored rows of 16-byte units, one or two per wave-load.
__device__ __forceinline__ float rect_absmax(const float* __restrict__ p, int rows, int cols, int ld, int w, int nw, int lane) {
  float m0 = 0.f, m1 = 0.f;
  const int cv = cols & ~3;
  const int lpr = cols > 128 ? 64 : 32, rpl = 64 / lpr;      // lanes per stored row, stored rows per wave-load
  const int l = lane % lpr, step = nw * rpl;
  int r = w * rpl + lane / lpr;
  for (; r + step < rows; r += 2 * step) {                    // two trips' loads in flight
    const float* q0 = p + (size_t)r * ld;
    const float* q1 = p + (size_t)(r + step) * ld;
#pragma unroll 5
    for (int c = l * 4; c < cv; c += lpr * 4) {
      const float4 u = *reinterpret_cast<const float4*>(q0 + c), v = *reinterpret_cast<const float4*>(q1 + c);
      m0 = fmaxf(m0, fmaxf(fmaxf(fabsf(u.x), fabsf(u.y)), fmaxf(fabsf(u.z), fabsf(u.w))));
      m1 = fmaxf(m1, fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
    }
    if (l < cols - cv) {                                      // (the padding behind the panel's last column is not the operand's)
      m0 = fmaxf(m0, fabsf(q0[cv + l]));
      m1 = fmaxf(m1, fabsf(q1[cv + l]));
    }
  }
  if (r < rows) {
    const float* q0 = p + (size_t)r * ld;
    for (int c = l * 4; c < cv; c += lpr * 4) {
      const float4 u = *reinterpret_cast<const float4*>(q0 + c);
      m0 = fmaxf(m0, fmaxf(fmaxf(fabsf(u.x), fabsf(u.y)), fmaxf(fabsf(u.z), fabsf(u.w))));
    }
    if (l < cols - cv) m0 = fmaxf(m0, fabsf(q0[cv + l]));
  }
  return fmaxf(m0, m1);
}
template <bool TA, bool TB>
__global__ __launch_bounds__(256) void k_gemm_absmax(const GemmArgs a, unsigned* __restrict__ scale, int ta, int sub_a, int sub_b) {
  const int b = blockIdx.y, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const TileBase tb(a, b);
  const int M = tb.M, K = tb.K, N = a.N;
  const bool is_a = (int)blockIdx.x < ta * sub_a;
  const int id = is_a ? (int)blockIdx.x : (int)blockIdx.x - ta * sub_a;
  const int sub = is_a ? sub_a : sub_b, t = id / sub, w = (id - t * sub) * 4 + wave, nw = sub * 4;
  const int lo = t * (is_a ? S_BM : S_BN), ext = is_a ? M : N;
  if (lo >= ext) return;
  const int len = min(is_a ? S_BM : S_BN, ext - lo);
  const size_t roff = a.ragged == 1 ? (size_t)a.gptr[b] : 0;
  float m = 0.f;
  for (int i = -1; i < a.nx; ++i) {            // the main pair, then the extra K segments
    const float* base = is_a ? (i < 0 ? tb.A : a.xA[i] + (size_t)b * a.xsA[i] + roff * a.xlda[i]) : (i < 0 ? tb.B : a.xB[i] + (size_t)b * a.xsB[i]);
    const int ld = is_a ? (i < 0 ? a.lda : a.xlda[i]) : (i < 0 ? a.ldb : a.xldb[i]);
    const int kk = i < 0 ? K : a.xK[i];
    const bool k_is_row = is_a ? TA : !TB;     // the operand is stored [K, .]: the panel is a column window of it
    m = fmaxf(m, k_is_row ? rect_absmax(base + lo, kk, len, ld, w, nw, lane) : rect_absmax(base + (size_t)lo * ld, len, kk, ld, w, nw, lane));
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
  // one atomic per WORKGROUP, and only when it would raise the slot: thousands of atomics on one address execute one after the other
  // at the memory side (the first version -- two per wave, 16 k per launch on two addresses -- spent two thirds of its time there)
  __shared__ float red[4];
  if (lane == 0) red[wave] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned bits = __builtin_bit_cast(unsigned, fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3])));
    unsigned* slot = scale + (size_t)b * (ta + a.tiles_n) + (is_a ? t : ta + t);
    if (bits > __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(slot, bits);
  }
}

// Workgroups the chip holds at once: one per CU (the register budget: 512 per lane)
static const int kHalfResident = 256;

static int64_t g_half_launches = 0;
// tile x k-tile steps below which gemm_half_launch declines (tuning hook, tests: 0 = never decline)
static int64_t g_half_min_work = getenv("CGC_HALF_MIN_WORK") ? atoll(getenv("CGC_HALF_MIN_WORK")) : 28000;
extern "C" int64_t cgc_gemm_half_min_work(int64_t v) {
  const int64_t old = __atomic_load_n(&g_half_min_work, __ATOMIC_RELAXED);
  if (v >= 0) __atomic_store_n(&g_half_min_work, v, __ATOMIC_RELAXED);
  return old;
}
extern "C" int64_t cgc_gemm_half_count(void) { return __atomic_load_n(&g_half_launches, __ATOMIC_RELAXED); }
int64_t gemm_half_scale_floats() { return H_SCALE_FLOATS; }
extern "C" int64_t cgc_gemm_half_ws_floats(void) { return H_SCALE_FLOATS; }

// Launch for a product that qualifies (gemm.hip: gemm_dispatch decided: 128 x 128 route, every operand segment fit for unguarded
// 16-byte loads).  CGC_EINVAL: no workspace for the scales / more panels than slots / a shape outside what the kernel indexes / a
// product too small for the mode to pay -- the caller then tries the bf16 kernel and, failing that, the exact one.
int gemm_half_launch(const GemmArgs& a0, int transA, int transB, int batch, int m_extent, int k_extent, float* ws, int64_t ws_floats,
                     hipStream_t stream) {
  if (transA && transB) return CGC_EINVAL;
  if (ws == nullptr || ws_floats < H_SCALE_FLOATS || batch > 65535) return CGC_EINVAL;
  GemmArgs a = a0;
  a.tiles_n = ceil_div(a.N, S_BN);
  static const int map_mode = getenv("CGC_GEMM_MAP") ? atoi(getenv("CGC_GEMM_MAP")) : 3;
  a.map_mode = map_mode;
  const long long per_batch = (long long)ceil_div(m_extent, S_BM) * a.tiles_n;
  const long long tiles = per_batch * batch;
  if (per_batch <= 0 || tiles > 0x7ffffff0LL) return CGC_EINVAL;
  const int ta = ceil_div(m_extent, S_BM);
  const long long slots = (long long)batch * (ta + a.tiles_n);
  if (slots > H_SCALE_FLOATS) return CGC_EINVAL;
  // The mode pays once the product kernel's saving (~30 % of the bf16 kernel's time) exceeds its own fixed cost (the slot fill, the
  // maximum pass: two launches and one more trip over the operands).  Measured on the step's six products at 32 / 16 / 8 / 4 graphs
  // (profiles/r06_configurations.txt): ahead of the bf16 mode by 18 / 17 / 10 % down to 8 graphs (37-47 k tile x k-tile steps per
  // product), behind it by 6 % at 4 (19-23 k) and at C1 = 180 (5 k).  Below 28 k steps the caller runs the bf16 kernel instead.
  const long long min_work = __atomic_load_n(&g_half_min_work, __ATOMIC_RELAXED);
  {
    long long kt = ceil_div(k_extent, SBK);
    for (int i = 0; i < a.nx; ++i) kt += ceil_div(a.xK[i], SBK);
    if (tiles * kt < min_work) return CGC_EINVAL;
  }
  a.per_batch = (int)per_batch;
  a.nb = batch;
  a.ws = nullptr;
  a.resident = 0;
  a.s_max = 1;
  ws_floats -= H_SCALE_FLOATS;
  unsigned* scale = reinterpret_cast<unsigned*>(ws + ws_floats);
  a.scale = scale;
  int extra = 0;
  static const int split_on = getenv("CGC_GEMM_SPLIT") ? atoi(getenv("CGC_GEMM_SPLIT")) : 1;
  if (split_on) {
    long long kt = ceil_div(k_extent, SBK);
    for (int i = 0; i < a.nx; ++i) kt += ceil_div(a.xK[i], SBK);
    const int s_max = (int)(kt / 8 < 12 ? kt / 8 : 12);                 // a piece keeps >= 8 k-tiles: the pipeline is five deep
    const long long max_pieces = kHalfResident + kHalfResident / 2;
    if (s_max >= 2 && max_pieces * S_BM * S_BN <= ws_floats) {
      a.ws = ws;
      a.resident = kHalfResident;
      a.s_max = s_max;
      extra = (int)max_pieces;
    }
  }
  int xk = 0;
  for (int i = 0; i < a.nx; ++i) xk += a.xK[i];
  const int trec = cgc_timing_begin(CGC_TAG_GEMM_128, a.M, a.N, a.K, batch, a.ragged, a.ragged ? (a.ragged == 1 ? m_extent : k_extent) : 0,
                                    xk, stream);
  {
    const hipError_t e = hipMemsetAsync(scale, 0, sizeof(unsigned) * (size_t)slots, stream);
    if (e != hipSuccess) return (int)e;
  }
  // ~4 workgroups per CU for each operand, a panel's rows shared by up to 64 of them
  auto subs = [&](int panels) { const long long n = (long long)batch * panels; const int v = n >= 1024 ? 1 : ceil_div(1024, (int)n); return v > 64 ? 64 : v; };
  const int sub_a = subs(ta), sub_b = subs(a.tiles_n);
  dim3 grid((unsigned)(tiles + extra)), block(256), mgrid((unsigned)(ta * sub_a + a.tiles_n * sub_b), (unsigned)batch);
#define HALF_LAUNCH(TA_, TB_)                                                                               \
  do {                                                                                                      \
    static bool attr__[CGC_MAX_DEVICES] = {};                                                               \
    cgc_allow_lds(reinterpret_cast<const void*>(&k_gemm_half<TA_, TB_>), H_LDS, attr__);                    \
    hipLaunchKernelGGL((k_gemm_absmax<TA_, TB_>), mgrid, block, 0, stream, a, scale, ta, sub_a, sub_b);                       \
    hipLaunchKernelGGL((k_gemm_half<TA_, TB_>), grid, block, H_LDS, stream, a);                             \
  } while (0)
  if (!transA && !transB) HALF_LAUNCH(false, false);
  else if (!transA) HALF_LAUNCH(false, true);
  else HALF_LAUNCH(true, false);
#undef HALF_LAUNCH
  CGC_RETURN_IF_LAUNCH_FAILED();
  __atomic_fetch_add(&g_half_launches, 1, __ATOMIC_RELAXED);
  if (a.ws != nullptr) {
    const long long lmax = tiles < kHalfResident ? tiles : kHalfResident - 1;
    hipLaunchKernelGGL((k_gemm_fixup<2, 2, 4, 2>), dim3((unsigned)(lmax * 4 * 4 * 2)), dim3(64), 0, stream, a);
    CGC_RETURN_IF_LAUNCH_FAILED();
  }
  cgc_timing_end(trec, stream);
  return 0;
}
