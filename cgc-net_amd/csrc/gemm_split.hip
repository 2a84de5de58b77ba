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
//   LDS: two stages of [3 planes][256 + 128 rows][16 k bf16], rows of 32 B on a 48 B stride (108 KB): a fragment is ONE ds_read_b128
//   (conflict-free: 3 is coprime with the 16 sixteen-byte slots of a bank row), and the 8-byte writes of a k-contiguous operand tile
//   the bank window exactly;
//   global -> registers two k-tiles ahead of the split (two register sets; buffer loads, no vector address arithmetic in the loop),
//   split in registers -> LDS two k-tiles ahead of the MFMAs, fragments read one k-tile ahead, plane by plane into the registers the
//   pair order frees (SplitFrags below): a wave never waits for LDS or memory inside a k-tile, and there is ONE barrier per k-tile;
//   an operand whose k index is the memory row (A stored [K, M], B stored [K, N]) is transposed in registers for free: a thread loads a
//   4 (k) x 4 (m) block -- 2 x 4 for the 128-wide operand -- and packs along k;
//   the ~130 vector instructions of a k-tile's split are dealt out by hand behind its 48 MFMAs (one micro-step of 2-4 instructions per
//   MFMA, a scheduling fence after each): behind a bf16 MFMA up to ~5 plain vector instructions of the SAME wave issue for free
//   (tools/pipe_overlap_probe.hip), another wave's do not.  What costs is every LDS instruction (~6 cycles of issue) and every
//   buffer load (~27): 36 + 6 per k-tile here (the first version of this kernel: 54 + 6 with 8-byte fragment reads on 40-byte rows and
//   three stages -- 2 % slower).
#include "gemm_split_common.hpp"

// ---- the split of one group of four values in eight micro-steps (sidx = 8 * group + step; groups 0-3: operand A, 4-5: operand B)
struct GroupState {
  float x[4], r1[4], r2[4];
  unsigned hp[2], mp[2], lp[2];
};
template <class LoaderA, class LoaderB, bool MASKED, int PLA, int PLB>
__device__ __forceinline__ void split_micro(int sidx, GroupState (&gs)[6], const float4 (&ra)[LoaderA::NF], const float4 (&rb)[LoaderB::NF],
                                            unsigned char* wa, unsigned char* wb, int k0, int klim) {
  const int u = sidx >> 3, st = sidx & 7;
  GroupState& s = gs[u];
  if (st == 0) {
    if (u < 4) LoaderA::get(ra, u, s.x); else LoaderB::get(rb, u - 4, s.x);
    if (MASKED) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int ke = k0 + (u < 4 ? LoaderA::kof(u, e) : LoaderB::kof(u - 4, e));
        s.x[e] = ke < klim ? s.x[e] : 0.f;
      }
    }
    s.hp[0] = pack_bf16(s.x[0], s.x[1]);
    s.hp[1] = pack_bf16(s.x[2], s.x[3]);
  } else if (st == 1 || st == 2) {
    const int h = st - 1;
    const float e0 = __builtin_bit_cast(float, s.hp[h] << 16), e1 = __builtin_bit_cast(float, s.hp[h] & 0xffff0000u);
    s.r1[2 * h] = s.x[2 * h] - e0;
    s.r1[2 * h + 1] = s.x[2 * h + 1] - e1;
    asm volatile("" : "+v"(s.r1[2 * h]), "+v"(s.r1[2 * h + 1]));
  } else if (st == 3) {
    s.mp[0] = pack_bf16(s.r1[0], s.r1[1]);
    s.mp[1] = pack_bf16(s.r1[2], s.r1[3]);
  } else if (st == 4 || st == 5) {
    const int h = st - 4;
    const float e0 = __builtin_bit_cast(float, s.mp[h] << 16), e1 = __builtin_bit_cast(float, s.mp[h] & 0xffff0000u);
    s.r2[2 * h] = s.r1[2 * h] - e0;
    s.r2[2 * h + 1] = s.r1[2 * h + 1] - e1;
    asm volatile("" : "+v"(s.r2[2 * h]), "+v"(s.r2[2 * h + 1]));
  } else if (st == 6) {
    s.lp[0] = pack_bf16(s.r2[0], s.r2[1]);
    s.lp[1] = pack_bf16(s.r2[2], s.r2[3]);
  } else {
    if (u < 4) {
      LoaderA::put(wa, u, 0, s.hp[0], s.hp[1]);
      LoaderA::put(wa, u, PLA, s.mp[0], s.mp[1]);
      LoaderA::put(wa, u, 2 * PLA, s.lp[0], s.lp[1]);
    } else {
      LoaderB::put(wb, u - 4, 0, s.hp[0], s.hp[1]);
      LoaderB::put(wb, u - 4, PLB, s.mp[0], s.mp[1]);
      LoaderB::put(wb, u - 4, 2 * PLB, s.lp[0], s.lp[1]);
    }
  }
}

// Fragment registers of a wave: one set (the k-tile being multiplied) + a second copy of the two hi planes.  The pair order of a
// tile -- lh, mm, mh, hl, hm, hh -- retires the planes one after the other, and the NEXT tile's copy of a plane is read (16 bytes per
// lane and sub-tile: 18 ds_read_b128 per k-tile) as soon as this tile's is dead: A.lo after pair 0, A.mid after pair 2, B.lo after pair
// 3, B.mid after pair 4; A.hi and B.hi live to the end, so the next tile's go into the second copies (24 registers) during pair 0.
// 96 fragment registers instead of the 144 of a full double buffer -- with two register sets of raw operands (48) and the split's
// temporaries everything but the accumulators has to fit 256 architectural registers (a full double buffer spilled 1194).
// No fragment of the tile being multiplied is read from LDS any more, so its stage is free for the tile after next: TWO stages.
constexpr int S_PLA = S_BM * SROW, S_PLB = S_BN * SROW, S_STAGE = 3 * S_PLA + 3 * S_PLB, S_LDS = 2 * S_STAGE;   // 110592 bytes

struct SplitFrags {
  uint4v a[4][3], b[2][3], a0n[4], b0n[2];
};

// One tile (HALF: a tile of <= 128 valid rows, see below) or one K piece of a tail tile.  Two instantiations per kernel, chosen per
// workgroup: as two loop nests inside ONE body the accumulators, fragments and operand registers had to agree at every merge point and
// the allocator spilled 107-124 registers; as two bodies that share nothing but the arguments it spills none.
template <bool TA, bool TB, bool HALF>
__device__ __forceinline__ void split_body(const GemmArgs& a, const int b, const int tile_id, const int piece, const int S, const unsigned tj,
                                           unsigned char* const slds) {
  constexpr int TM = 4, TN = 2, WGN = 2;
  const TileBase tb(a, b);
  const int M = tb.M, K = tb.K, N = a.N;
  const float* A = tb.A;
  const float* B = tb.B;
  float* C = tb.C;
  const int tile_m = tile_id / a.tiles_n, tile_n = tile_id - tile_m * a.tiles_n;
  const int m0 = tile_m * S_BM, n0 = tile_n * S_BN;
  if (m0 >= M) return;

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

  const unsigned fa_off = (unsigned)((HALF ? wm * 64 : wm * 128) + l31) * SROW + lhi * 16, fb_off = 3 * S_PLA + (unsigned)(wn * 64 + l31) * SROW + lhi * 16;
  const unsigned wa_off = LoaderA::wbase(), wb_off = 3 * S_PLA + LoaderB::wbase();

  SplitFrags fr;
  // ---- prologue: tiles 0, 1 split into stages 0, 1; tiles 2, 3 in flight in the two sets; all fragments of tile 0 in registers.
  // An EMPTY k range (an empty graph of a ragged-K batch; a tail-split piece of a graph with fewer k-tiles than pieces) loads
  // nothing -- there is no valid tile to clamp to (kbeg - 1 would be rows of the previous graph, or in front of the operand) --
  // and goes straight to the epilogue / its slab with zero accumulators, as k_gemm_f32 does (gemm.hip: `if (kend > kbeg)`).
  if (n > 0) {
#pragma unroll
  for (int q = 0; q < 4; ++q) {               // q: 0, 1 request tiles 0, 1; 2: split 0, request 2; 3: split 1, request 3
    const int set = q & 1;
    if (q >= 2) {
      const SplitTile t = split_tile(seg, kbeg + (q - 2 < n ? q - 2 : n - 1));
      GroupState gs[6];
#pragma unroll
      for (int sidx = 0; sidx < 48; ++sidx)
        split_micro<LoaderA, LoaderB, true, S_PLA, S_PLB>(sidx, gs, ra[set], rb[set], slds + (q - 2) * S_STAGE + wa_off,
                                                            slds + (q - 2) * S_STAGE + wb_off, t.k0, t.klim);
    }
    const SplitTile t = split_tile(seg, kbeg + (q < n ? q : n - 1));
#pragma unroll
    for (int i = 0; i < NFA; ++i) ra[set][i] = LoaderA::load_any(i, t.A, t.lda, m0, a_last, t.k0, t.klim);
#pragma unroll
    for (int i = 0; i < NFB; ++i) rb[set][i] = LoaderB::load_any(i, t.B, t.ldb, n0, b_last, t.k0, t.klim);
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int p = 0; p < 3; ++p) fr.a[i][p] = frag16(slds + fa_off + i * 32 * SROW + p * S_PLA);
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int p = 0; p < 3; ++p) fr.b[j][p] = frag16(slds + fb_off + j * 32 * SROW + p * S_PLB);
  __syncthreads();                            // (step 0 writes tile 2 into stage 0: everybody has read tile 0 out of it)
  }

  auto tile_step = [&](auto pos_c, auto full_c, int lt) {
    constexpr int POS = decltype(pos_c)::value;          // local tile index mod 2
    constexpr bool FULL = decltype(full_c)::value;       // (HALF: sub-tiles i >= 2 of the wave do not exist)
    const unsigned char* rstage = slds + (POS ^ 1) * S_STAGE;      // tile lt + 1
    unsigned char* wa = slds + POS * S_STAGE + wa_off;             // tile lt + 2 goes where tile lt was
    unsigned char* wb = slds + POS * S_STAGE + wb_off;
    GroupState gs[6];
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
    constexpr int PA_[6] = {2, 1, 1, 0, 0, 0}, PB_[6] = {0, 1, 0, 2, 1, 0};      // lh, mm, mh, hl, hm, hh
#pragma clang loop unroll(full)
    for (int m = 0; m < 48; ++m) {
      const int t = m / 8, ij = m % 8, i = ij >> 1, j = ij & 1;
      if (!HALF || i < 2)
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fr.b[j][PB_[t]]), __builtin_bit_cast(bf16x8, fr.a[i][PA_[t]]),
                                                            acc[i][j], 0, 0, 0);
      // the next tile's fragments, 18 reads of 16 bytes (12 for a half tile), each plane as soon as this tile's copy is dead
      constexpr int NA = HALF ? 2 : 4;
      if (m < 4) { if (m < NA) fr.a0n[m] = frag16(rstage + fa_off + m * 32 * SROW); }                          // A.hi' (second copy)
      else if (m < 6) fr.b0n[m - 4] = frag16(rstage + fb_off + (m - 4) * 32 * SROW);                         // B.hi' (second copy)
      else if (m >= 8 && m < 12) { if (m - 8 < NA) fr.a[m - 8][2] = frag16(rstage + fa_off + (m - 8) * 32 * SROW + 2 * S_PLA); }   // A.lo (pair 0 only)
      else if (m >= 24 && m < 28) { if (m - 24 < NA) fr.a[m - 24][1] = frag16(rstage + fa_off + (m - 24) * 32 * SROW + S_PLA); }   // A.mid (pairs 1, 2)
      else if (m >= 32 && m < 34) fr.b[m - 32][2] = frag16(rstage + fb_off + (m - 32) * 32 * SROW + 2 * S_PLB);   // B.lo (pair 3)
      else if (m >= 40 && m < 42) fr.b[m - 40][1] = frag16(rstage + fb_off + (m - 40) * 32 * SROW + S_PLB);   // B.mid (pairs 1, 4)
      split_micro<LoaderA, LoaderB, !FULL, S_PLA, S_PLB>(m, gs, ra[POS], rb[POS], wa, wb, k0s, klims);
      if (m >= 28 && m < 28 + NFA) {
        if constexpr (FULL) ra[POS][m - 28] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsrcA, offA[m - 28], soffA, 0));
        else ra[POS][m - 28] = LoaderA::load_any(m - 28, tnext.A, tnext.lda, m0, a_last, tnext.k0, tnext.klim);
      }
      if (m >= 44 && m < 44 + NFB) {
        if constexpr (FULL) rb[POS][m - 44] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsrcB, offB[m - 44], soffB, 0));
        else rb[POS][m - 44] = LoaderB::load_any(m - 44, tnext.B, tnext.ldb, n0, b_last, tnext.k0, tnext.klim);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int i = 0; i < (HALF ? 2 : 4); ++i) fr.a[i][0] = fr.a0n[i];
    fr.b[0][0] = fr.b0n[0];
    fr.b[1][0] = fr.b0n[1];
    __syncthreads();
  };
  typedef std::true_type FULL_;
  typedef std::false_type ANY_;
#define SPLIT_POS(P_) std::integral_constant<int, P_>()
  int lt = 0;
  const int last_special = nk - nk_full;
  const int full_steps = min(last_special > 0 && kend > nk_full ? nk_full - 4 - kbeg : nk_full - 2 - kbeg, n);
  for (; lt + 2 <= full_steps; lt += 2) {
    tile_step(SPLIT_POS(0), FULL_(), lt);
    tile_step(SPLIT_POS(1), FULL_(), lt + 1);
  }
  for (; lt < n; ++lt) {
    if ((lt & 1) == 0) tile_step(SPLIT_POS(0), ANY_(), lt);
    else tile_step(SPLIT_POS(1), ANY_(), lt);
  }
#undef SPLIT_POS

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
__global__ __launch_bounds__(256, 1) void k_gemm_split(const GemmArgs a) {
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
#ifndef S_NO_HALF            // (-DS_NO_HALF: A/B timing builds)
  if (S == 1 && rows_left <= 128) {
    split_body<TA, TB, true>(a, b, tile_id, piece, S, tj, slds);
    return;
  }
#endif
  split_body<TA, TB, false>(a, b, tile_id, piece, S, tj, slds);
}

// Workgroups the chip holds at once: one per CU
static const int kSplitResident = 256;

// how many products this process has sent to the split kernel (tests: "did the mode apply to this product?")
static int64_t g_split_launches = 0;
extern "C" int64_t cgc_gemm_split_count(void) { return __atomic_load_n(&g_split_launches, __ATOMIC_RELAXED); }

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
  __atomic_fetch_add(&g_split_launches, 1, __ATOMIC_RELAXED);
  if (a.ws != nullptr) {
    const long long lmax = tiles < kSplitResident ? tiles : kSplitResident - 1;
    hipLaunchKernelGGL((k_gemm_fixup<2, 2, 4, 2>), dim3((unsigned)(lmax * 4 * 4 * 2)), dim3(64), 0, stream, a);
    CGC_RETURN_IF_LAUNCH_FAILED();
  }
  cgc_timing_end(trec, stream);
  return 0;
}
