// fp32 GEMM on the CDNA4 matrix cores: v_mfma_f32_32x32x2_f32 (exact fp32 FMA chain: keeps the 1e-4 parity budget;
// 157.3 TFLOP/s peak on MI355X -- there is no TF32/xf32 on gfx950).
//
// Replaces the dense contractions of the reference's hot path: the assignment Linear
// (model/network.py:122), S^T X and S^T (A S) of _diff_pool (:206-207), adj@x at levels 2-3, and all their backward
// products.  One kernel family covers: NN / NT / TN operand layouts, strided batches, and RAGGED batches whose M or K
// extent is a graph's node count (per-graph row offsets from gptr) -- so the flat [Ntot, *] level-1 tensors are
// contracted per graph without padding.
//
// Tiling (wave = 64 lanes, 4 waves per workgroup = one per SIMD):
//   block tile (WGM*TM*32) x (WGN*TN*32) x 32;  each wave owns TM x TN accumulators of 32x32 (16 VGPRs each);
//   both operand tiles are copied into LDS with 16-byte writes in the orientation they have in memory -- no transposing
//   scalar writes: an mn-contiguous operand (A stored [K,M], B stored [K,N]) lands k-major ([k][mn], row stride mn+4) and
//   is fetched with four ds_read_b32 per 8 k; a k-contiguous operand (A stored [M,K], B stored [N,K]) lands row-major
//   ([mn][k], row stride 36 words: conflict-free for ds_read_b128's 16-lane groups) and is fetched with ONE ds_read_b128
//   per 8 k.  Both fetches hand lane l the k values 8*kb + 4*(l>>5) + {0,1,2,3}; MFMA t of the group consumes element t,
//   i.e. the k-pair {8kb+t, 8kb+4+t} -- a permutation of the k order that A and B share, so the product is unchanged;
//   global->register prefetch of tile t+1 overlaps the 64 MFMA/wave (128x128) on tile t; 2 LDS buffers, 1 barrier/k-tile.
#include "gemm_common.hpp"

// Loader for an operand tile whose K index is the CONTIGUOUS one in memory (A stored [M,K]; B stored [N,K]).
// Logical tile: ROWS (m or n) x BK.  LDS image: [row][k], row stride KC_LD = 36 words.
template <int ROWS>
struct KContigLoader {
  static constexpr int UNITS = ROWS * (BK / 4);       // float4 units in the tile
  static constexpr int PER_T = (UNITS + 255) / 256;   // units per thread
  float4 reg[PER_T];
  __device__ __forceinline__ void load(const float* __restrict__ base, int ld, int row0, int row_lim, int k0, int k_lim, bool vec_ok) {
#pragma unroll
    for (int i = 0; i < PER_T; ++i) {
      const int u = threadIdx.x + i * 256;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (UNITS % 256 == 0 || u < UNITS) {
        const int row = row0 + u / (BK / 4);
        const int k = k0 + (u % (BK / 4)) * 4;
        if (row < row_lim && k < k_lim) {
          const float* p = base + (size_t)row * ld + k;
          if (vec_ok && k + 3 < k_lim) {
            v = *reinterpret_cast<const float4*>(p);
          } else {
            v.x = p[0];
            if (k + 1 < k_lim) v.y = p[1];
            if (k + 2 < k_lim) v.z = p[2];
            if (k + 3 < k_lim) v.w = p[3];
          }
        }
      }
      reg[i] = v;
    }
  }
  // Fast path for interior k-tiles: every unit is one aligned 16-byte load, no predication.  Rows beyond the matrix
  // edge are CLAMPED to the last valid row: they only feed output rows/columns that are never stored.
  __device__ __forceinline__ void load_fast(const float* __restrict__ base, int ld, int row0, int row_last, int k0) {
#pragma unroll
    for (int i = 0; i < PER_T; ++i) {
      const int u = threadIdx.x + i * 256;
      const int row = min(row0 + u / (BK / 4), row_last);
      reg[i] = *reinterpret_cast<const float4*>(base + (size_t)row * ld + k0 + (u % (BK / 4)) * 4);
    }
  }
  // The same fetch as load_fast as a BUFFER load: the operand is described once by a resource descriptor in SGPRs (base, no bounds),
  // a lane contributes a loop-invariant 32-bit byte offset (row clamp included) and the tile's k position is the instruction's
  // scalar offset -- buffer_load_dwordx4 v, v_off, s[rsrc], s_k offen.  The k loop then contains no vector address arithmetic at
  // all: on gfx950 the fp32 MFMA and the vector ALU do not overlap on a SIMD, so every v_lshl_add_u64 of the flat-address form (two
  // per load) is time taken from the matrix pipe.
  __device__ __forceinline__ void offsets(unsigned (&off)[PER_T], int ld, int row0, int row_last) const {
#pragma unroll
    for (int i = 0; i < PER_T; ++i) {
      const int u = threadIdx.x + i * 256;
      off[i] = (unsigned)min(row0 + u / (BK / 4), row_last) * (unsigned)ld * 4u + (unsigned)(u % (BK / 4)) * 16u;
    }
  }
  static __device__ __forceinline__ unsigned tile_soffset(int /*ld*/, int k0) { return (unsigned)k0 * 4u; }   // bytes from the operand base
  template <int AUX = 0>
  __device__ __forceinline__ void load_buf_part(int i, __amdgpu_buffer_rsrc_t rsrc, const unsigned (&off)[PER_T], unsigned soff) {
    reg[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, off[i], soff, AUX));
  }
  // Partial k-tile (k_lim % 4 == 0), still branch-free: units past the end of K re-read the last valid 16 bytes of their
  // row and are zeroed with a select -- the guarded loader above costs ~0.75 of a full tile's time on top of its own.
  // (k_lim need not be a multiple of 4 when the row stride is: the unit that straddles the end is read whole -- the tail of
  // a padded row, or the head of the next one -- and its elements past k_lim are zeroed one by one.)
  __device__ __forceinline__ void load_fast_masked(const float* __restrict__ base, int ld, int row0, int row_last, int k0, int k_lim) {
    const int k_last4 = (k_lim - 1) & ~3;
#pragma unroll
    for (int i = 0; i < PER_T; ++i) {
      const int u = threadIdx.x + i * 256;
      const int row = min(row0 + u / (BK / 4), row_last);
      const int k = k0 + (u % (BK / 4)) * 4;
      float4 v = *reinterpret_cast<const float4*>(base + (size_t)row * ld + min(k, k_last4));
      if (k >= k_lim) v.x = 0.f;
      if (k + 1 >= k_lim) v.y = 0.f;
      if (k + 2 >= k_lim) v.z = 0.f;
      if (k + 3 >= k_lim) v.w = 0.f;
      reg[i] = v;
    }
  }
  // same orientation as memory: one 16-byte write per unit.  store_part(i) writes unit i only, so that the writes of the
  // next tile can be spread between the MFMA groups of the current one.
  __device__ __forceinline__ void store_part(int i, float* __restrict__ lds) const {
    const int u = threadIdx.x + i * 256;
    if (UNITS % 256 == 0 || u < UNITS)
      *reinterpret_cast<float4*>(&lds[(u / (BK / 4)) * KC_LD + (u % (BK / 4)) * 4]) = reg[i];
  }
  __device__ __forceinline__ void store(float* __restrict__ lds) const {
#pragma unroll
    for (int i = 0; i < PER_T; ++i) store_part(i, lds);
  }
};

// Loader for an operand tile whose M/N index is the contiguous one (A stored [K,M]; B stored [K,N]).
// Logical tile: BK x COLS.  LDS image: [k][col], row stride COLS+4 (16-byte aligned rows, 16-byte writes).
template <int COLS>
struct MnContigLoader {
  static constexpr int UNITS = BK * (COLS / 4);
  static constexpr int PER_T = (UNITS + 255) / 256;
  float4 reg[PER_T];
  __device__ __forceinline__ void load(const float* __restrict__ base, int ld, int col0, int col_lim, int k0, int k_lim, bool vec_ok) {
#pragma unroll
    for (int i = 0; i < PER_T; ++i) {
      const int u = threadIdx.x + i * 256;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (UNITS % 256 == 0 || u < UNITS) {
        const int k = k0 + u / (COLS / 4);
        const int c = col0 + (u % (COLS / 4)) * 4;
        if (k < k_lim && c < col_lim) {
          const float* p = base + (size_t)k * ld + c;
          if (vec_ok && c + 3 < col_lim) {
            v = *reinterpret_cast<const float4*>(p);
          } else {
            v.x = p[0];
            if (c + 1 < col_lim) v.y = p[1];
            if (c + 2 < col_lim) v.z = p[2];
            if (c + 3 < col_lim) v.w = p[3];
          }
        }
      }
      reg[i] = v;
    }
  }
  // Fast path (interior k-tile, extent % 4 == 0): columns beyond the edge are clamped to the last valid 16-byte group.
  __device__ __forceinline__ void load_fast(const float* __restrict__ base, int ld, int col0, int col_last4, int k0) {
#pragma unroll
    for (int i = 0; i < PER_T; ++i) {
      const int u = threadIdx.x + i * 256;
      const int c = min(col0 + (u % (COLS / 4)) * 4, col_last4);
      reg[i] = *reinterpret_cast<const float4*>(base + (size_t)(k0 + u / (COLS / 4)) * ld + c);
    }
  }
  // base (uniform) + per-lane 32-bit byte offset: see KContigLoader::offsets
  __device__ __forceinline__ void offsets(unsigned (&off)[PER_T], int ld, int col0, int col_last4) const {
#pragma unroll
    for (int i = 0; i < PER_T; ++i) {
      const int u = threadIdx.x + i * 256;
      off[i] = (unsigned)(u / (COLS / 4)) * (unsigned)ld * 4u + (unsigned)min(col0 + (u % (COLS / 4)) * 4, col_last4) * 4u;
    }
  }
  static __device__ __forceinline__ unsigned tile_soffset(int ld, int k0) { return (unsigned)k0 * (unsigned)ld * 4u; }
  template <int AUX = 0>
  __device__ __forceinline__ void load_buf_part(int i, __amdgpu_buffer_rsrc_t rsrc, const unsigned (&off)[PER_T], unsigned soff) {
    reg[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, off[i], soff, AUX));
  }
  // Partial k-tile: rows (k) past the end re-read row k_lim-1 and are zeroed.
  __device__ __forceinline__ void load_fast_masked(const float* __restrict__ base, int ld, int col0, int col_last4, int k0, int k_lim) {
#pragma unroll
    for (int i = 0; i < PER_T; ++i) {
      const int u = threadIdx.x + i * 256;
      const int c = min(col0 + (u % (COLS / 4)) * 4, col_last4);
      const int k = k0 + u / (COLS / 4);
      float4 v = *reinterpret_cast<const float4*>(base + (size_t)min(k, k_lim - 1) * ld + c);
      if (k >= k_lim) v = make_float4(0.f, 0.f, 0.f, 0.f);
      reg[i] = v;
    }
  }
  __device__ __forceinline__ void store_part(int i, float* __restrict__ lds) const {
    constexpr int LD = COLS + 4;
    const int u = threadIdx.x + i * 256;
    if (UNITS % 256 == 0 || u < UNITS) *reinterpret_cast<float4*>(&lds[(u / (COLS / 4)) * LD + (u % (COLS / 4)) * 4]) = reg[i];
  }
  __device__ __forceinline__ void store(float* __restrict__ lds) const {
#pragma unroll
    for (int i = 0; i < PER_T; ++i) store_part(i, lds);
  }
};

// MFMA operand fragment for the 8 k's of group kb: lane (l31 = l&31, lhi = l>>5) gets k = 8*kb + 4*lhi + {0..3} of row/col l31.
template <bool KMAJOR, int LD>
__device__ __forceinline__ void fetch_frag(const float* __restrict__ tile, int kb, int l31, int lhi, float (&f)[4]) {
  if (KMAJOR) {
#pragma unroll
    for (int t = 0; t < 4; ++t) f[t] = tile[(kb * 8 + lhi * 4 + t) * LD + l31];
  } else {
    const float4 v = *reinterpret_cast<const float4*>(&tile[l31 * LD + kb * 8 + lhi * 4]);
    f[0] = v.x; f[1] = v.y; f[2] = v.z; f[3] = v.w;
  }
}

// FAST: every operand segment qualifies for the unguarded 16-byte loaders (decided on the host, gemm_all_fast): the kernel then
// contains no guarded loader and no conditional inside a k-loop phase.  !FAST: the guarded element-wise loaders throughout.
// cache policy of the pinned operand loads (buffer-load aux bits; 2 = nt: streaming) -- -DCGC_GEMM_A_AUX / _B_AUX for A/B timing
#ifndef CGC_GEMM_A_AUX
#define CGC_GEMM_A_AUX 0
#endif
#ifndef CGC_GEMM_B_AUX
#define CGC_GEMM_B_AUX 0
#endif
template <int WGM, int WGN, int TM, int TN, bool TA, bool TB, bool FAST>
__global__ __launch_bounds__(256, 2) void k_gemm_f32(const GemmArgs a) {   // 2 waves per SIMD = 2 workgroups per CU (the LDS budget)
  GT_MARK(0)
  constexpr int BM = WGM * TM * 32, BN = WGN * TN * 32;
  constexpr int LDA_S = TA ? BM + 4 : KC_LD;   // TA: A stored [K,M] -> k-major tile; else row-major [m][k]
  constexpr int LDB_S = TB ? KC_LD : BN + 4;   // TB: B stored [N,K] -> row-major [n][k]; else k-major
  constexpr int A_SZ = TA ? BK * LDA_S : BM * KC_LD, B_SZ = TB ? BN * KC_LD : BK * LDB_S;   // multiples of 4 words
  __shared__ __attribute__((aligned(16))) float lds[2 * A_SZ + 2 * B_SZ];
  float* const As0 = lds;                 // As[buf] = As0 + buf*A_SZ
  float* const Bs0 = lds + 2 * A_SZ;      // Bs[buf] = Bs0 + buf*B_SZ

  int b, tile_id, piece, S;
  unsigned tj;
  {
    TileMap<BM> map;
    map.init(a, threadIdx.x & 63);
    if (!map.select(a, blockIdx.x, threadIdx.x & 63, b, tile_id, tj, piece, S)) return;
  }
  // wave-uniform by construction; said explicitly so that everything derived from them (k range, tile origin) lives in SGPRs
  b = __builtin_amdgcn_readfirstlane(b);
  tile_id = __builtin_amdgcn_readfirstlane(tile_id);
  piece = __builtin_amdgcn_readfirstlane(piece);
  S = __builtin_amdgcn_readfirstlane(S);
  tj = __builtin_amdgcn_readfirstlane(tj);
  const TileBase tb(a, b);
  const int M = tb.M, K = tb.K;
  const float* A = tb.A;
  const float* B = tb.B;
  float* C = tb.C;
  const int tile_m = tile_id / a.tiles_n, tile_n = tile_id - tile_m * a.tiles_n;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  if (m0 >= M) return;
  const int N = a.N;

#ifdef CGC_GEMM_TRACE
  if (TM == 2 && TN == 2 && threadIdx.x == 0 && blockIdx.x < 65536)
    g_gemm_trace[blockIdx.x][5] = ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32) | __builtin_amdgcn_s_getreg((31 << 11) | 4);
#endif
  const bool vecA = (a.lda % 4 == 0) && ((reinterpret_cast<uintptr_t>(A) & 15u) == 0);
  const bool vecB = (a.ldb % 4 == 0) && ((reinterpret_cast<uintptr_t>(B) & 15u) == 0);

  typedef typename std::conditional<TA, MnContigLoader<BM>, KContigLoader<BM>>::type LoaderA;
  typedef typename std::conditional<TB, KContigLoader<BN>, MnContigLoader<BN>>::type LoaderB;
  LoaderA la0, la1;   // two register stages: one holds tile t+1 (arrived, being written to LDS), the other receives t+2
  LoaderB lb0, lb1;

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

  const int nk_main = (K + BK - 1) / BK, nk_full = K / BK;
  // extra K segments follow the main operand pair in the tile sequence (always through the guarded loader: they are short)
  const float* xA0 = nullptr; const float* xB0 = nullptr; const float* xA1 = nullptr; const float* xB1 = nullptr;
  int nkx0 = 0, nkx1 = 0;
  if (a.nx > 0) {
    const size_t roff = a.ragged == 1 ? (size_t)a.gptr[b] : 0;
    xA0 = a.xA[0] + (size_t)b * a.xsA[0] + roff * a.xlda[0];
    xB0 = a.xB[0] + (size_t)b * a.xsB[0];
    nkx0 = (a.xK[0] + BK - 1) / BK;
    if (a.nx > 1) {
      xA1 = a.xA[1] + (size_t)b * a.xsA[1] + roff * a.xlda[1];
      xB1 = a.xB[1] + (size_t)b * a.xsB[1];
      nkx1 = (a.xK[1] + BK - 1) / BK;
    }
  }
  const int nk = nk_main + nkx0 + nkx1;
  // this workgroup's share of the k-tiles: all of them, or piece `piece` of S (tail split)
  const int kbeg = S > 1 ? (int)(((long long)nk * piece) / S) : 0;
  const int kend = S > 1 ? (int)(((long long)nk * (piece + 1)) / S) : nk;
  // unguarded 16-byte loads for full k-tiles when the layout allows (block-uniform decision)
  // (an extent that is not a multiple of 4 is fine: with a row stride that IS one -- vecA / vecB -- the 16-byte group that
  // straddles the edge stays inside its row's stride: padding, whose values only reach output rows / columns that are never
  // stored, or k positions that are zeroed.  The level-2 cluster count 114 = int(1140 * 0.1) on 116-float rows is the case.)
  const bool fastA = vecA && M >= 1 && (TA ? a.lda >= ((M + 3) & ~3) : true);
  const bool fastB = vecB && N >= 1 && (TB ? true : a.ldb >= ((N + 3) & ~3));
  const int a_last = TA ? ((M - 1) & ~3) : M - 1, b_last = TB ? N - 1 : ((N - 1) & ~3);
  // partial k-tiles can take the masked fast loads: always when k is the row index of the operand (A stored [K,M], B stored
  // [K,N]); for k-contiguous operands when the row stride covers the 16-byte group that holds the last k
  const int K4 = (K + 3) & ~3;
  const bool k4A = K >= 1 && (TA ? true : a.lda >= K4), k4B = K >= 1 && (TB ? a.ldb >= K4 : true);
  auto fetch = [&](LoaderA& la, LoaderB& lb, int kt) {
   if constexpr (!FAST) {                 // (the guarded loaders do not exist in the FAST kernel)
    if (kt < nk_main) {
      if (fastA && kt < nk_full) la.load_fast(A, a.lda, m0, a_last, kt * BK);
      else if (fastA && k4A) la.load_fast_masked(A, a.lda, m0, a_last, kt * BK, K);
      else la.load(A, a.lda, m0, M, kt * BK, K, vecA);
      if (fastB && kt < nk_full) lb.load_fast(B, a.ldb, n0, b_last, kt * BK);
      else if (fastB && k4B) lb.load_fast_masked(B, a.ldb, n0, b_last, kt * BK, K);
      else lb.load(B, a.ldb, n0, N, kt * BK, K, vecB);
    } else {
      int kx = kt - nk_main;
      const bool second = kx >= nkx0;
      if (second) kx -= nkx0;
      const float* Ax = second ? xA1 : xA0;
      const float* Bx = second ? xB1 : xB0;
      const int ldax = a.xlda[second], ldbx = a.xldb[second], Kx = a.xK[second];
      const bool va = (ldax % 4 == 0) && ((reinterpret_cast<uintptr_t>(Ax) & 15u) == 0);
      const bool vb = (ldbx % 4 == 0) && ((reinterpret_cast<uintptr_t>(Bx) & 15u) == 0);
      const bool kx4 = (Kx % 4 == 0) && Kx >= 4;
      // same M / N extents as the main pair: the clamps a_last / b_last apply; short segments are one or two k-tiles
      if (va && (TA ? (M % 4 == 0 && M >= 4) : kx4)) la.load_fast_masked(Ax, ldax, m0, a_last, kx * BK, Kx);
      else la.load(Ax, ldax, m0, M, kx * BK, Kx, va);
      if (vb && (TB ? kx4 : (N % 4 == 0 && N >= 4))) lb.load_fast_masked(Bx, ldbx, n0, b_last, kx * BK, Kx);
      else lb.load(Bx, ldbx, n0, N, kx * BK, Kx, vb);
    }
   }
  };
  // Branch-free fetch of ANY k-tile -- a full or partial tile of the main operand pair or of an extra K segment -- for the
  // prologue and the tail of the k loop: the segment (base pointers, row strides, reduction length) is picked with scalar
  // selects and both operands take the masked unguarded loads.  Valid when every segment qualifies (FAST).
  const SegTable seg = {A, xA0, xA1, B, xB0, xB1, a.lda, a.xlda[0], a.xlda[1], a.ldb, a.xldb[0], a.xldb[1], K, a.xK[0], a.xK[1],
                        nk_main, nkx0};
  // One k-tile: consume LDS buffer `cur` with 4 groups of TM*TN*4 MFMAs.  The fragments of group kb+1 are read from LDS
  // before the MFMAs of group kb are issued; between the groups a quarter of the NEXT tile (already in registers `ls_*`)
  // is written into the other LDS buffer; the tile after that is requested from memory at the top and lands in `ll_*`
  // while all of this runs.  One barrier per k-tile.  The phase comes in compile-time flavours so that NO flavour but the
  // generic one has a conditional inside (the compiler then counts vmcnt / lgkmcnt exactly and interleaves freely):
  //   PH_FULL   interior of the main operand pair: tile kt+2 is a full tile, unguarded 16-byte loads, no conditional at all
  //   PH_MASK   (FAST kernel) the last tiles -- partial tile, extra segments, nothing left to request: masked unguarded loads with
  //             scalar segment selection behind two uniform conditions
  //   PH_ANY    (!FAST kernel) operands that do not qualify for unguarded loads (odd strides, unaligned views): guarded loaders
  // (measured: the FAST kernel is 1-2 % faster on the step's big products -- the guarded tail was NOT what the K = 40 segment of
  // the assignment Linear paid for; that was round quantisation, see the tail split.  A variant with no conditional inside any
  // phase (separate STORE / LAST flavours, parity-alternating epilogue loop) compiled to 256 VGPRs + 500-800 spilled registers.)
#ifndef CGC_GEMM_NOPIN      // (-DCGC_GEMM_NOPIN: the compiler's own order with flat-address loads, for A/B timing)
  unsigned offA[LoaderA::PER_T], offB[LoaderB::PER_T];
  __amdgpu_buffer_rsrc_t rsrcA, rsrcB;
  if constexpr (FAST) {
    la0.offsets(offA, a.lda, m0, a_last);
    lb0.offsets(offB, a.ldb, n0, b_last);
    // raw buffers over the main operand pair (stride 0, no range check to speak of; 0x00020000 = 32-bit data format)
    rsrcA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(A), 0, 0xffffffff, 0x00020000);
    rsrcB = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(B), 0, 0xffffffff, 0x00020000);
  }
#endif
  auto phase = [&](auto mode_c, int kt, auto cur_c, LoaderA& ls_a, LoaderB& ls_b, LoaderA& ll_a, LoaderB& ll_b) {
    constexpr int MODE = decltype(mode_c)::value;
    constexpr int cur = decltype(cur_c)::value;
    bool has_next = true;
    if constexpr (MODE == PH_FULL) {
#ifndef CGC_X_NOLOAD          // (CGC_X_*: timing experiments that break the result -- tools/variant_lib.sh; never defined in the build)
#ifndef CGC_GEMM_NOPIN
      if constexpr (FAST) {     // (issued inside the k groups below)
        static_assert(LoaderA::PER_T <= BK / 8 && LoaderB::PER_T <= BK / 8, "one load per operand and k group");
      } else
#endif
      {
        ll_a.load_fast(A, a.lda, m0, a_last, (kt + 2) * BK);
        ll_b.load_fast(B, a.ldb, n0, b_last, (kt + 2) * BK);
      }
#endif
    } else if constexpr (MODE == PH_MASK) {
      if (kt + 2 < kend) fetch_seg(ll_a, ll_b, seg, kt + 2, m0, a_last, n0, b_last);
      has_next = kt + 1 < kend;
    } else if constexpr (MODE == PH_ANY) {
      if (kt + 2 < kend) fetch(ll_a, ll_b, kt + 2);
      has_next = kt + 1 < kend;
    }
    // k groups (of 8) of THIS tile that hold anything: a partial last tile / a short extra segment is zero past its end, and a group
    // of zeros is 16 MFMAs per wave that add nothing (K = 1140 = 35 x 32 + 20: one group; + 40: three more; + 20: one more)
    int nkb = BK / 8;
    if constexpr (MODE != PH_FULL) {
      const int kx = kt - nk_main;
      const int left = kx < 0 ? K - kt * BK : kx < nkx0 ? a.xK[0] - kx * BK : a.xK[1] - (kx - nkx0) * BK;
      nkb = left >= BK ? BK / 8 : (left + 7) >> 3;
    }
    const float* as = As0 + cur * A_SZ + (TA ? wm * TM * 32 : wm * TM * 32 * KC_LD);
    const float* bs = Bs0 + cur * B_SZ + (TB ? wn * TN * 32 * KC_LD : wn * TN * 32);
    float* an = As0 + (cur ^ 1) * A_SZ;
    float* bn = Bs0 + (cur ^ 1) * B_SZ;
    float av[2][TM][4], bv[2][TN][4];
#pragma unroll
    for (int i = 0; i < TM; ++i) fetch_frag<TA, LDA_S>(as + (TA ? i * 32 : i * 32 * KC_LD), 0, l31, lhi, av[0][i]);
#pragma unroll
    for (int j = 0; j < TN; ++j) fetch_frag<!TB, LDB_S>(bs + (TB ? j * 32 * KC_LD : j * 32), 0, l31, lhi, bv[0][j]);
#pragma unroll
    for (int kb = 0; kb < BK / 8; ++kb) {
#ifdef CGC_X_NOREAD
      if (MODE != PH_FULL && kb + 1 < BK / 8) {
#else
      if (kb + 1 < BK / 8) {
#endif
#pragma unroll
        for (int i = 0; i < TM; ++i) fetch_frag<TA, LDA_S>(as + (TA ? i * 32 : i * 32 * KC_LD), kb + 1, l31, lhi, av[(kb + 1) & 1][i]);
#pragma unroll
        for (int j = 0; j < TN; ++j) fetch_frag<!TB, LDB_S>(bs + (TB ? j * 32 * KC_LD : j * 32), kb + 1, l31, lhi, bv[(kb + 1) & 1][j]);
      }
#ifndef CGC_GEMM_NOPIN
      if constexpr (MODE == PH_FULL && FAST) {
        // pinned interleave (FULL phases): the k group's 16 MFMAs in four runs of TM*TN, and behind each run ONE kind of other work --
        // the next group's fragment reads (issued above, in front of the first run), this group's share of the LDS writes of tile
        // kt+1, this group's share of the global loads of tile kt+2 -- with a scheduling fence after each run, so that no more than
        // a few non-matrix instructions ever sit between two MFMAs (the compiler's own order issues the tile's eight global loads
        // back to back at the end of the phase)
#pragma unroll
        for (int t = 0; t < 4; ++t) {
#pragma unroll
          for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(bv[kb & 1][j][t], av[kb & 1][i][t], acc[i][j], 0, 0, 0);
          if (t == 1) {
            if (kb < LoaderA::PER_T) ls_a.store_part(kb, an);
            if (kb < LoaderB::PER_T) ls_b.store_part(kb, bn);
          }
          if (t == 2) {
            if (kb < LoaderA::PER_T) ll_a.template load_buf_part<CGC_GEMM_A_AUX>(kb, rsrcA, offA, LoaderA::tile_soffset(a.lda, (kt + 2) * BK));
            if (kb < LoaderB::PER_T) ll_b.template load_buf_part<CGC_GEMM_B_AUX>(kb, rsrcB, offB, LoaderB::tile_soffset(a.ldb, (kt + 2) * BK));
          }
          __builtin_amdgcn_sched_barrier(0);
        }
        continue;
      }
#endif
      if (MODE == PH_FULL || kb < nkb) {
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(bv[kb & 1][j][t], av[kb & 1][i][t], acc[i][j], 0, 0, 0);   // C^T tile: see gemm_epilogue
      }
#ifdef CGC_X_NOWRITE
      if (MODE != PH_FULL && has_next) {
#else
      if (has_next) {
#endif
        if (kb < LoaderA::PER_T) ls_a.store_part(kb, an);
        if (kb < LoaderB::PER_T) ls_b.store_part(kb, bn);
      }
    }
#ifdef CGC_X_NOBAR
    if (MODE != PH_FULL) __syncthreads();
#else
    __syncthreads();
#endif
  };
  typedef std::integral_constant<int, PH_FULL> FULL_;
  typedef std::integral_constant<int, PH_MASK> MASK_;
  typedef std::integral_constant<int, PH_ANY> ANY_;
  typedef std::integral_constant<int, 0> B0;
  typedef std::integral_constant<int, 1> B1;

  if constexpr (FAST) {
    if (kend > kbeg) {
      fetch_seg(la0, lb0, seg, kbeg, m0, a_last, n0, b_last);
      la0.store(As0);
      lb0.store(Bs0);
      if (kbeg + 1 < kend) fetch_seg(la0, lb0, seg, kbeg + 1, m0, a_last, n0, b_last);
    }
    __syncthreads();
    GT_MARK(1)
    int kt = kbeg;
    const int fast_end = nk_full < kend ? nk_full : kend;
    for (; kt + 3 < fast_end; kt += 2) {     // both phases prefetch full tiles of the main pair (kt+2, kt+3 < nk_full)
      phase(FULL_(), kt, B0(), la0, lb0, la1, lb1);
      phase(FULL_(), kt + 1, B1(), la1, lb1, la0, lb0);
    }
    // the last tiles (partial tile, extra segments, nothing left to request): two uniform conditions per phase, no guarded loader
    for (; kt < kend; kt += 2) {
      phase(MASK_(), kt, B0(), la0, lb0, la1, lb1);
      if (kt + 1 < kend) phase(MASK_(), kt + 1, B1(), la1, lb1, la0, lb0);
    }
  } else {
    if (kend > kbeg) {
      fetch(la0, lb0, kbeg);
      la0.store(As0);
      lb0.store(Bs0);
      if (kbeg + 1 < kend) fetch(la0, lb0, kbeg + 1);
    }
    __syncthreads();
    int kt = kbeg;
    if (fastA && fastB) {
      const int fast_end = nk_full < kend ? nk_full : kend;
      for (; kt + 3 < fast_end; kt += 2) {
        phase(FULL_(), kt, B0(), la0, lb0, la1, lb1);
        phase(FULL_(), kt + 1, B1(), la1, lb1, la0, lb0);
      }
    }
    for (; kt < kend; kt += 2) {
      phase(ANY_(), kt, B0(), la0, lb0, la1, lb1);
      if (kt + 1 < kend) phase(ANY_(), kt + 1, B1(), la1, lb1, la0, lb0);
    }
  }

  GT_MARK(2)
  if (S > 1) {
    // piece of a tail tile: the raw accumulators go to this piece's slab (k_gemm_fixup adds the S slabs of the tile and applies
    // alpha / beta / bias).  Slab order = register order, 16 bytes per lane: every store instruction writes 1 KiB contiguous.
    float* slab = a.ws + ((size_t)tj * S + piece) * (size_t)(BM * BN) + (size_t)wave * (TM * TN * 16 * 64) + lane * 4;
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
  // epilogue (the last phase ended with a barrier: nobody reads operand tiles any more, the LDS is free for the parking strips)
  static_assert(4 * 32 * (TN * 32 + 4) <= 2 * A_SZ + 2 * B_SZ, "epilogue parking region exceeds the operand LDS");
  gemm_epilogue<TM, TN>(a, C, M, N, m0 + wm * TM * 32, n0 + wn * TN * 32, acc, lds + wave * 32 * (TN * 32 + 4), lane);
  GT_MARK(3)
}

template <int WGM, int WGN, int TM, int TN, bool TA, bool TB>
__global__ __launch_bounds__(256, 3) void k_gemm_f32_shortk(const GemmArgs a) {
  constexpr int BM = WGM * TM * 32, BN = WGN * TN * 32;
  constexpr int LDA_S = TA ? BM + 4 : KC_LD;   // TA: A stored [K,M] -> k-major tile; else row-major [m][k]
  constexpr int LDB_S = TB ? KC_LD : BN + 4;   // TB: B stored [N,K] -> row-major [n][k]; else k-major
  constexpr int A_SZ = TA ? BK * LDA_S : BM * KC_LD, B_SZ = TB ? BN * KC_LD : BK * LDB_S;   // multiples of 4 words
  constexpr int PARK = 4 * 32 * (TN * 32 + 4);                             // epilogue parking strips (gemm_epilogue)
  __shared__ __attribute__((aligned(16))) float lds[A_SZ + B_SZ > PARK ? A_SZ + B_SZ : PARK];   // ONE stage: <= 36.9 KB; 3 workgroups per CU (register-limited)
  float* const As0 = lds;                 // As[buf] = As0 + buf*A_SZ
  float* const Bs0 = lds + A_SZ;

  int b, tile_id;
  {
    TileMap<BM> map;
    int piece, S;
    unsigned tj;
    map.init(a, threadIdx.x & 63);      // (a.ws is never set for this kernel: every tile is computed whole)
    if (!map.select(a, blockIdx.x, threadIdx.x & 63, b, tile_id, tj, piece, S)) return;
  }
  b = __builtin_amdgcn_readfirstlane(b);
  tile_id = __builtin_amdgcn_readfirstlane(tile_id);
  const TileBase tb(a, b);
  const int M = tb.M, K = tb.K;
  const float* A = tb.A;
  const float* B = tb.B;
  float* C = tb.C;
  const int tile_m = tile_id / a.tiles_n, tile_n = tile_id - tile_m * a.tiles_n;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  if (m0 >= M) return;
  const int N = a.N;

  const bool vecA = (a.lda % 4 == 0) && ((reinterpret_cast<uintptr_t>(A) & 15u) == 0);
  const bool vecB = (a.ldb % 4 == 0) && ((reinterpret_cast<uintptr_t>(B) & 15u) == 0);

  typedef typename std::conditional<TA, MnContigLoader<BM>, KContigLoader<BM>>::type LoaderA;
  typedef typename std::conditional<TB, KContigLoader<BN>, MnContigLoader<BN>>::type LoaderB;
  LoaderA la0;
  LoaderB lb0;

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

  const int nk = (K + BK - 1) / BK, nk_full = K / BK;
  // unguarded 16-byte loads for full k-tiles when the layout allows (block-uniform decision)
  const bool fastA = vecA && M >= 1 && (TA ? a.lda >= ((M + 3) & ~3) : true);     // as in k_gemm_f32
  const bool fastB = vecB && N >= 1 && (TB ? true : a.ldb >= ((N + 3) & ~3));
  const int a_last = TA ? ((M - 1) & ~3) : M - 1, b_last = TB ? N - 1 : ((N - 1) & ~3);
  const int K4 = (K + 3) & ~3;
  const bool k4A = K >= 1 && (TA ? true : a.lda >= K4), k4B = K >= 1 && (TB ? a.ldb >= K4 : true);
  auto fetch = [&](LoaderA& la, LoaderB& lb, int kt) {
    if (fastA && kt < nk_full) la.load_fast(A, a.lda, m0, a_last, kt * BK);
    else if (fastA && k4A) la.load_fast_masked(A, a.lda, m0, a_last, kt * BK, K);
    else la.load(A, a.lda, m0, M, kt * BK, K, vecA);
    if (fastB && kt < nk_full) lb.load_fast(B, a.ldb, n0, b_last, kt * BK);
    else if (fastB && k4B) lb.load_fast_masked(B, a.ldb, n0, b_last, kt * BK, K);
    else lb.load(B, a.ldb, n0, N, kt * BK, K, vecB);
  };
  // Short reductions (K <= 96: rank-k updates such as dS += X dX'^T, dA~ = g x^T, h = agg W): the kernel is bound by the
  // C tile it writes (and reads when beta != 0), not by the MFMAs.  One LDS stage keeps the footprint at 36.9 KB so that three
  // workgroups share a CU and overlap each other's load / compute / store phases.
  for (int kt = 0; kt < nk; ++kt) {
    fetch(la0, lb0, kt);
    if (kt > 0) __syncthreads();          // everybody finished reading the previous tile
    la0.store(As0);
    lb0.store(Bs0);
    __syncthreads();
    const float* as = As0 + (TA ? wm * TM * 32 : wm * TM * 32 * KC_LD);
    const float* bs = Bs0 + (TB ? wn * TN * 32 * KC_LD : wn * TN * 32);
    const int left = K - kt * BK, nkb = left >= BK ? BK / 8 : (left + 7) >> 3;     // (k groups of 8 past the end of K hold zeros: skipped)
#pragma unroll
    for (int kb = 0; kb < BK / 8; ++kb) {
      if (kb < nkb) {                     // (uniform; a `break` here keeps the loop from being unrolled)
        float av[TM][4], bv[TN][4];
#pragma unroll
        for (int i = 0; i < TM; ++i) fetch_frag<TA, LDA_S>(as + (TA ? i * 32 : i * 32 * KC_LD), kb, l31, lhi, av[i]);
#pragma unroll
        for (int j = 0; j < TN; ++j) fetch_frag<!TB, LDB_S>(bs + (TB ? j * 32 * KC_LD : j * 32), kb, l31, lhi, bv[j]);
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(bv[j][t], av[i][t], acc[i][j], 0, 0, 0);
      }
    }
  }

  __syncthreads();                          // everybody finished reading the last operand tile: the LDS becomes the parking area
  gemm_epilogue<TM, TN>(a, C, M, N, m0 + wm * TM * 32, n0 + wn * TN * 32, acc, lds + wave * 32 * (TN * 32 + 4), lane);
}

// Whether every operand segment of the product qualifies for the unguarded 16-byte loaders of k_gemm_f32<.., FAST = true>:
// 16-byte-aligned bases (also after the per-batch stride and the per-graph row offset), row strides that are multiples of 4
// floats, and rows long enough to hold the 16-byte group that straddles the end of an extent that is not a multiple of 4 (its
// surplus elements only reach output rows / columns that are never stored, or k positions that the loader zeroes).  The ragged
// extent (M of ragged 1, K of ragged 2) is always the ROW index of the operands it applies to, never the contiguous one.
static bool gemm_all_fast(const GemmArgs& a, int transA, int transB, int batch, long long max_rows, long long max_k) {
  auto up4 = [](int v) { return (v + 3) & ~3; };
  auto seg = [&](const float* A, const float* B, int lda, int ldb, long long sA, long long sB, int K) {
    if (lda % 4 != 0 || ldb % 4 != 0 || !aligned16(A) || !aligned16(B)) return false;
    // per-lane byte offsets inside one batch item's operand are 32-bit (KContigLoader::offsets): rows x row stride must stay below 4 GiB
    const long long rowsA = transA ? 32 : (a.ragged == 1 ? max_rows : a.M), rowsB = transB ? a.N : 32;
    if (rowsA * lda * 4 >= (1LL << 32) || rowsB * ldb * 4 >= (1LL << 32)) return false;
    // ... and so is the scalar k offset of a tile (k rows x row stride when k is the operand's row index)
    const long long kmax = a.ragged >= 2 ? max_k : K;
    if ((transA ? kmax * lda : kmax) * 4 >= (1LL << 31) || (transB ? kmax : kmax * ldb) * 4 >= (1LL << 31)) return false;
    if (batch > 1 && (sA % 4 != 0 || sB % 4 != 0)) return false;
    if (transA ? lda < up4(a.M) : lda < up4(K)) return false;      // (ragged 1: transA = 0; ragged 2: transA = 1, K is the row index)
    if (transB ? ldb < up4(K) : ldb < up4(a.N)) return false;
    return true;
  };
  if (!seg(a.A, a.B, a.lda, a.ldb, a.strideA, a.strideB, a.ragged >= 2 ? 0 : a.K)) return false;
  for (int i = 0; i < a.nx; ++i)
    if (!seg(a.xA[i], a.xB[i], a.xlda[i], a.xldb[i], a.xsA[i], a.xsB[i], a.xK[i])) return false;
  return true;
}

// Workgroups of k_gemm_f32 the chip holds at once: 2 per CU (LDS: 73.7 KB for the 128 x 128 tile; __launch_bounds__(256, 2))
static const int kResident = 512;
// slab workspace the tail split may use (floats): up to 1.5 * kResident pieces of one 128 x 128 tile
// + the scale slots of mode CGC_GEMM_SPLIT_F16 (gemm_half.hip), which sit at the workspace's end
int64_t gemm_half_scale_floats();      // gemm_half.hip
extern "C" int64_t cgc_gemm_ws_floats(void) { return (int64_t)(kResident + kResident / 2) * 128 * 128 + gemm_half_scale_floats(); }

template <int WGM, int WGN, int TM, int TN>
static int launch_cfg(const GemmArgs& a0, int transA, int transB, int batch, int m_extent, int k_extent, bool short_k, float* ws,
                      int64_t ws_floats, hipStream_t stream) {
  constexpr int BM = WGM * TM * 32, BN = WGN * TN * 32;
  GemmArgs a = a0;
  a.tiles_n = ceil_div(a.N, BN);
  static const int map_mode = getenv("CGC_GEMM_MAP") ? atoi(getenv("CGC_GEMM_MAP")) : 3;
  a.map_mode = map_mode;
  const long long per_batch = (long long)ceil_div(m_extent, BM) * a.tiles_n;
  const long long tiles = per_batch * batch;
  if (per_batch <= 0 || tiles > 0x7ffffff0LL) return CGC_EINVAL;
  a.per_batch = (int)per_batch;
  a.nb = batch;
  a.ws = nullptr;
  a.resident = 0;
  a.s_max = 1;
  if (transA && transB) return CGC_EINVAL;
  // tail split: pipelined kernel only, long reductions only (a piece keeps >= 3 k-tiles), slabs must fit the workspace
  static const int split_on = getenv("CGC_GEMM_SPLIT") ? atoi(getenv("CGC_GEMM_SPLIT")) : 1;
  int extra = 0;
  // the smaller tile shapes hold more workgroups per CU than the 512 the split is sized for: they only split when the launch is
  // clearly under-filled (the thin level-2 products of a 4-graph shard: 72 workgroups walking 36 k-tiles each = 36 us of pure
  // latency; as 504 pieces of 5 k-tiles + fix-up: 14 us)
  static const int small_max = getenv("CGC_GEMM_SMALL_SPLIT") ? atoi(getenv("CGC_GEMM_SMALL_SPLIT")) : 384;
  const bool big = BM == 128 && BN == 128;
  if (!short_k && ws != nullptr && split_on && (big || split_on >= 2 || tiles <= small_max)) {
    long long kt = ceil_div(k_extent, BK);
    for (int i = 0; i < a.nx; ++i) kt += ceil_div(a.xK[i], BK);
    const int s_max = (int)(kt / 3 < 12 ? kt / 3 : 12);
    const long long max_pieces = kResident + kResident / 2;            // L * S <= 1.5 R by construction (TileMap::init)
    if (s_max >= 2 && max_pieces * BM * BN <= ws_floats) {
      a.ws = ws;
      a.resident = kResident;
      a.s_max = s_max;
      extra = (int)max_pieces;
    }
  }
  dim3 grid((unsigned)(tiles + extra)), block(256);
  int xk = 0;
  for (int i = 0; i < a.nx; ++i) xk += a.xK[i];
  const int trec = (BM == 128 && BN == 128 && !short_k)
                       ? cgc_timing_begin(CGC_TAG_GEMM_128, a.M, a.N, a.K, batch, a.ragged, a.ragged ? (a.ragged == 1 ? m_extent : k_extent) : 0, xk, stream)
                       : -1;
  static const int fast_on = getenv("CGC_GEMM_FAST") ? atoi(getenv("CGC_GEMM_FAST")) : 1;   // CGC_GEMM_FAST=0: A-B timing against the guarded kernel
  const bool fast = fast_on && gemm_all_fast(a, transA, transB, batch, m_extent, k_extent);
  if (short_k) {
    if (!transA && !transB) hipLaunchKernelGGL((k_gemm_f32_shortk<WGM, WGN, TM, TN, false, false>), grid, block, 0, stream, a);
    else if (!transA) hipLaunchKernelGGL((k_gemm_f32_shortk<WGM, WGN, TM, TN, false, true>), grid, block, 0, stream, a);
    else hipLaunchKernelGGL((k_gemm_f32_shortk<WGM, WGN, TM, TN, true, false>), grid, block, 0, stream, a);
  } else if (fast) {
    if (!transA && !transB) hipLaunchKernelGGL((k_gemm_f32<WGM, WGN, TM, TN, false, false, true>), grid, block, 0, stream, a);
    else if (!transA) hipLaunchKernelGGL((k_gemm_f32<WGM, WGN, TM, TN, false, true, true>), grid, block, 0, stream, a);
    else hipLaunchKernelGGL((k_gemm_f32<WGM, WGN, TM, TN, true, false, true>), grid, block, 0, stream, a);
  } else {
    if (!transA && !transB) hipLaunchKernelGGL((k_gemm_f32<WGM, WGN, TM, TN, false, false, false>), grid, block, 0, stream, a);
    else if (!transA) hipLaunchKernelGGL((k_gemm_f32<WGM, WGN, TM, TN, false, true, false>), grid, block, 0, stream, a);
    else hipLaunchKernelGGL((k_gemm_f32<WGM, WGN, TM, TN, true, false, false>), grid, block, 0, stream, a);
  }
  CGC_RETURN_IF_LAUNCH_FAILED();
  if (a.ws != nullptr) {
    // at most min(T, R - 1) tail tiles; which ones (if any) is decided on the device exactly as in the kernel above
    const long long lmax = tiles < kResident ? tiles : kResident - 1;
    hipLaunchKernelGGL((k_gemm_fixup<WGM, WGN, TM, TN>), dim3((unsigned)(lmax * 4 * TM * TN)), dim3(64), 0, stream, a);
    CGC_RETURN_IF_LAUNCH_FAILED();
  }
  cgc_timing_end(trec, stream);
  return 0;
}

// tuning hook (tools/gemm_cfg_sweep.py): 0 = automatic tile selection (default)
static int g_force_cfg = getenv("CGC_GEMM_CFG") ? atoi(getenv("CGC_GEMM_CFG")) : 0;
extern "C" int cgc_gemm_tuning(int cfg) {
  const int old = g_force_cfg;
  g_force_cfg = cfg;
  return old;
}

int gemm_split_launch(const GemmArgs& a0, int transA, int transB, int batch, int m_extent, int k_extent, float* ws, int64_t ws_floats,
                      hipStream_t stream);      // gemm_split.hip
int gemm_half_launch(const GemmArgs& a0, int transA, int transB, int batch, int m_extent, int k_extent, float* ws, int64_t ws_floats,
                     hipStream_t stream);       // gemm_half.hip

static int gemm_dispatch(GemmArgs& a, int transA, int transB, int batch, int max_ragged, float* ws, int64_t ws_floats, int mode,
                         hipStream_t stream) {
  const int M = a.M, N = a.N, K = a.K, ragged = a.ragged;
  if (batch <= 0 || N <= 0) return 0;
#ifdef CGC_GEMM_ONLY_128   // compile-time experiments on the dominant kernel alone (not part of the build)
  return launch_cfg<2, 2, 2, 2>(a, transA, transB, batch, ragged == 1 ? max_ragged : M, ragged >= 2 ? max_ragged : K, false, ws, ws_floats, stream);
#endif
  // the tail split is tuned for the 128 x 128 tile (512 resident workgroups); the other pipelined tile shapes use it for under-filled
  // launches only (launch_cfg; CGC_GEMM_SPLIT=2: always)
  float* const ws_any = ws;
  if (ragged == 1 && (transA || a.gptr == nullptr)) return CGC_EINVAL;
  if (ragged == 2 && (!transA || transB || a.gptr == nullptr || a.nx > 0)) return CGC_EINVAL;
  if (ragged == 3 && (!transA || transB || a.nx > 0 || max_ragged <= 0 || K <= 0 || batch % ceil_div(K, max_ragged) != 0)) return CGC_EINVAL;
  if (ragged < 0 || ragged > 3) return CGC_EINVAL;
  a.chunk = ragged == 3 ? max_ragged : 0;
  const int m_extent = ragged == 1 ? max_ragged : M;
  if (m_extent <= 0) return 0;
  // tile shape by output aspect: the hot contractions are (>=1140) x (>=1140); the skinny ones are K- or output-bound.
  // `fill` = workgroups a 128-row tiling would launch; below ~448 (256 CUs x 2 resident) the tile is halved in M.
  const int k_extent = ragged >= 2 ? max_ragged : K;
  static const int shortk_max = getenv("CGC_GEMM_SHORTK") ? atoi(getenv("CGC_GEMM_SHORTK")) : 160;
  static const int fill_min = getenv("CGC_GEMM_FILL") ? atoi(getenv("CGC_GEMM_FILL")) : 448;
  const bool sk = k_extent <= shortk_max && a.nx == 0;
  const long long fill = (long long)ceil_div(m_extent, 128) * batch;
  // the 128 x 128 pipelined route; mode CGC_GEMM_SPLIT_BF16: as six bf16 MFMA pairs on 256 x 128 tiles (gemm_split.hip) when every
  // operand segment is fit for unguarded 16-byte loads; otherwise -- and for every other route -- the exact kernel
  static const int force_mode = getenv("CGC_GEMM_MODE") ? atoi(getenv("CGC_GEMM_MODE")) : -1;      // experiments: overrides the argument
  // mode CGC_GEMM_SPLIT_F16: as three fp16 MFMA pairs of operands scaled per batch item (gemm_half.hip), same tiles, same condition
  const int eff_mode = force_mode >= 0 ? force_mode : mode;
  const bool want_split = eff_mode == CGC_GEMM_SPLIT_BF16, want_half = eff_mode == CGC_GEMM_SPLIT_F16;
  auto big_route = [&](bool shortk, float* w) -> int {
    if ((want_split || want_half) && !shortk && !(transA && transB) && gemm_all_fast(a, transA, transB, batch, m_extent, k_extent)) {
      int rc = want_half ? gemm_half_launch(a, transA, transB, batch, m_extent, k_extent, w, ws_floats, stream) : CGC_EINVAL;
      if (rc == CGC_EINVAL) rc = gemm_split_launch(a, transA, transB, batch, m_extent, k_extent, w, ws_floats, stream);   // (mode F16: products too small for its maximum pass to pay)
      if (rc != CGC_EINVAL) return rc;
    }
    return launch_cfg<2, 2, 2, 2>(a, transA, transB, batch, m_extent, k_extent, shortk, w, ws_floats, stream);
  };
  // experiment hook: CGC_GEMM_CFG = 1..6 forces a tile shape (128x128, 128x64, 64x128, 64x64, 128x32, 32x128), +10 forces the
  // pipelined kernel, +20 the short-K kernel
  const int force = g_force_cfg;
  if (force > 0) {
    const bool fsk = force >= 20 ? true : force >= 10 ? false : sk;
    switch (force % 10) {
      case 1: return big_route(fsk, ws);
      case 2: return launch_cfg<4, 1, 1, 2>(a, transA, transB, batch, m_extent, k_extent, fsk, ws_any, ws_floats, stream);
      case 3: return launch_cfg<1, 4, 2, 1>(a, transA, transB, batch, m_extent, k_extent, fsk, ws_any, ws_floats, stream);
      case 4: return launch_cfg<2, 2, 1, 1>(a, transA, transB, batch, m_extent, k_extent, fsk, ws_any, ws_floats, stream);
      case 5: return launch_cfg<4, 1, 1, 1>(a, transA, transB, batch, m_extent, k_extent, fsk, ws_any, ws_floats, stream);
      case 6: return launch_cfg<1, 4, 1, 1>(a, transA, transB, batch, m_extent, k_extent, fsk, ws_any, ws_floats, stream);
      default: break;
    }
  }
  if (N <= 32) return launch_cfg<4, 1, 1, 1>(a, transA, transB, batch, m_extent, k_extent, sk, ws_any, ws_floats, stream);          // 128 x 32
  if (N <= 64) {
    if (m_extent > 64 && fill < fill_min) return launch_cfg<2, 2, 1, 1>(a, transA, transB, batch, m_extent, k_extent, sk, ws_any, ws_floats, stream);   // 64 x 64
    return launch_cfg<4, 1, 1, 2>(a, transA, transB, batch, m_extent, k_extent, sk, ws_any, ws_floats, stream);                                     // 128 x 64
  }
  if (m_extent <= 32) return launch_cfg<1, 4, 1, 1>(a, transA, transB, batch, m_extent, k_extent, sk, ws_any, ws_floats, stream);   // 32 x 128
  if (m_extent <= 64) return launch_cfg<1, 4, 2, 1>(a, transA, transB, batch, m_extent, k_extent, sk, ws_any, ws_floats, stream);   // 64 x 128
  if (fill * ceil_div(N, 128) < fill_min) {   // too few 128x128 tiles to fill the chip (tools/gemm_cfg_sweep.py over the step's shapes)
    if (N <= 128) {                           // one column tile: long reductions stream A through 128x32 tiles (4 column tiles share the
      if (k_extent > 256 && m_extent > 128)   // A panel in L2; [32 x 1140 x 1140] x [1140 x 114]: 200 -> 155 us), short ones take 64x64
        return launch_cfg<4, 1, 1, 1>(a, transA, transB, batch, m_extent, k_extent, sk, ws_any, ws_floats, stream);
      return launch_cfg<2, 2, 1, 1>(a, transA, transB, batch, m_extent, k_extent, sk, ws_any, ws_floats, stream);
    }
    if (transA) {
      // long reductions (S^T P of a 4-graph shard: 324 tiles of 57 k-tiles): whole 128 x 128 tiles, cut in two along K, beat
      // twice as many 128 x 64 tiles (235 -> 208 us).  (More pieces per tile -- 3 = 1.9 rounds of a third -- were measured
      // too: 204 us, and a cost model that picked the piece count by rounds x length made the small tails slower.)
      if (!sk && ws != nullptr && k_extent >= 24 * BK) return big_route(sk, ws);
      return launch_cfg<4, 1, 1, 2>(a, transA, transB, batch, m_extent, k_extent, sk, ws_any, ws_floats, stream);
    }
    return launch_cfg<1, 4, 2, 1>(a, transA, transB, batch, m_extent, k_extent, sk, ws_any, ws_floats, stream);
  }
  return big_route(sk, ws);                        // 128 x 128
}

static void gemm_fill(GemmArgs& a, int M, int N, int K, float alpha, const float* A, int lda, const float* B, int ldb, float beta,
                      float* C, int ldc, const float* bias, int64_t strideA, int64_t strideB, int64_t strideC, const int* gptr,
                      int ragged) {
  a.A = A; a.B = B; a.C = C; a.bias = bias; a.gptr = gptr;
  a.M = M; a.N = N; a.K = K; a.lda = lda; a.ldb = ldb; a.ldc = ldc;
  a.strideA = strideA; a.strideB = strideB; a.strideC = strideC;
  a.alpha = alpha; a.beta = beta; a.ragged = ragged; a.tiles_n = 0;
  a.per_batch = a.nb = 0; a.ws = nullptr; a.resident = 0; a.s_max = 1; a.map_mode = 0; a.chunk = 0; a.scale = nullptr;
  a.nx = 0;
  for (int i = 0; i < 2; ++i) { a.xA[i] = a.xB[i] = nullptr; a.xlda[i] = a.xldb[i] = a.xK[i] = 0; a.xsA[i] = a.xsB[i] = 0; }
}

extern "C" int cgc_gemm_f32_ws(int transA, int transB, int M, int N, int K, float alpha, const float* A, int lda, const float* B,
                               int ldb, float beta, float* C, int ldc, const float* bias, int batch, int64_t strideA, int64_t strideB,
                               int64_t strideC, const int* gptr, int ragged, int max_ragged, float* ws, int64_t ws_floats, int mode,
                               cgc_stream_t stream_) {
  if (mode != CGC_GEMM_EXACT && mode != CGC_GEMM_SPLIT_BF16 && mode != CGC_GEMM_SPLIT_F16) return CGC_EINVAL;
  GemmArgs a;
  gemm_fill(a, M, N, K, alpha, A, lda, B, ldb, beta, C, ldc, bias, strideA, strideB, strideC, gptr, ragged);
  return gemm_dispatch(a, transA, transB, batch, max_ragged, ws, ws_floats, mode, as_stream(stream_));
}

extern "C" int cgc_gemm_f32(int transA, int transB, int M, int N, int K, float alpha, const float* A, int lda, const float* B, int ldb,
                            float beta, float* C, int ldc, const float* bias, int batch, int64_t strideA, int64_t strideB,
                            int64_t strideC, const int* gptr, int ragged, int max_ragged, cgc_stream_t stream_) {
  return cgc_gemm_f32_ws(transA, transB, M, N, K, alpha, A, lda, B, ldb, beta, C, ldc, bias, batch, strideA, strideB, strideC, gptr,
                         ragged, max_ragged, nullptr, 0, CGC_GEMM_EXACT, stream_);
}

extern "C" int cgc_gemm_f32_cat_ws(int transA, int transB, int M, int N, int K, float alpha, const float* A, int lda, const float* B,
                                   int ldb, float beta, float* C, int ldc, const float* bias, int batch, int64_t strideA,
                                   int64_t strideB, int64_t strideC, const int* gptr, int ragged, int max_ragged, int nx,
                                   const float* const* xA, const int* xlda, const int64_t* xstrideA, const float* const* xB,
                                   const int* xldb, const int64_t* xstrideB, const int* xK, float* ws, int64_t ws_floats, int mode,
                                   cgc_stream_t stream_) {
  if (nx < 0 || nx > 2) return CGC_EINVAL;
  if (mode != CGC_GEMM_EXACT && mode != CGC_GEMM_SPLIT_BF16 && mode != CGC_GEMM_SPLIT_F16) return CGC_EINVAL;
  GemmArgs a;
  gemm_fill(a, M, N, K, alpha, A, lda, B, ldb, beta, C, ldc, bias, strideA, strideB, strideC, gptr, ragged);
  int kept = 0;
  for (int i = 0; i < nx; ++i) {
    if (xK[i] <= 0) continue;
    a.xA[kept] = xA[i]; a.xB[kept] = xB[i]; a.xlda[kept] = xlda[i]; a.xldb[kept] = xldb[i]; a.xK[kept] = xK[i];
    a.xsA[kept] = xstrideA[i]; a.xsB[kept] = xstrideB[i];
    ++kept;
  }
  a.nx = kept;
  return gemm_dispatch(a, transA, transB, batch, max_ragged, ws, ws_floats, mode, as_stream(stream_));
}

extern "C" int cgc_gemm_f32_cat(int transA, int transB, int M, int N, int K, float alpha, const float* A, int lda, const float* B,
                                int ldb, float beta, float* C, int ldc, const float* bias, int batch, int64_t strideA,
                                int64_t strideB, int64_t strideC, const int* gptr, int ragged, int max_ragged, int nx,
                                const float* const* xA, const int* xlda, const int64_t* xstrideA, const float* const* xB,
                                const int* xldb, const int64_t* xstrideB, const int* xK, cgc_stream_t stream_) {
  return cgc_gemm_f32_cat_ws(transA, transB, M, N, K, alpha, A, lda, B, ldb, beta, C, ldc, bias, batch, strideA, strideB, strideC,
                             gptr, ragged, max_ragged, nx, xA, xlda, xstrideA, xB, xldb, xstrideB, xK, nullptr, 0, CGC_GEMM_EXACT, stream_);
}

// ---- deterministic split-K combine
// (width, ldo): the result is a [numel / width, width] matrix written with row stride ldo -- a weight-gradient piece goes straight into its
// columns of the flat gradient buffer instead of through a temporary + a copy launch; width = ldo = numel: a flat vector as before
__global__ void k_reduce_batch_sum(const float* __restrict__ ws, float* __restrict__ out, int parts, long long numel, float beta, int width,
                                   int ldo) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < numel; i += (long long)gridDim.x * blockDim.x) {
    float s = 0.f;
    for (int p = 0; p < parts; ++p) s += ws[(size_t)p * numel + i];
    const long long o = (i / width) * ldo + i % width;
    out[o] = beta != 0.f ? s + beta * out[o] : s;
  }
}

// many slices of a small tensor (the weight gradients of the narrow layers: 50-110 slices of <= 64 x 64): 8 slice groups x
// 32 elements per workgroup, 8 loads in flight per thread, fixed summation order
__global__ __launch_bounds__(256) void k_reduce_many_parts(const float* __restrict__ ws, float* __restrict__ out, int parts, int numel,
                                                           float beta, int width, int ldo) {
  __shared__ float part[8][32];
  const int cl = threadIdx.x & 31, grp = threadIdx.x >> 5;
  const int c = blockIdx.x * 32 + cl;
  ws += (size_t)blockIdx.y * parts * numel;     // blockIdx.y = outer batch: ws is [outer][parts][numel], out is [outer][numel]
  out += (size_t)blockIdx.y * numel;            // (outer > 1 only with width = ldo = numel)
  float s = 0.f;
  if (c < numel) {
    float a[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) a[u] = 0.f;
    int k = grp;
    for (; k + 56 < parts; k += 64) {
#pragma unroll
      for (int u = 0; u < 8; ++u) a[u] += ws[(size_t)(k + 8 * u) * numel + c];
    }
    for (; k < parts; k += 8) a[0] += ws[(size_t)k * numel + c];
    s = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
  }
  part[grp][cl] = s;
  __syncthreads();
  if (grp == 0 && c < numel) {
    float t = part[0][cl];
#pragma unroll
    for (int g = 1; g < 8; ++g) t += part[g][cl];
    const int o = (c / width) * ldo + c % width;
    out[o] = beta != 0.f ? t + beta * out[o] : t;
  }
}

extern "C" int cgc_reduce_batched(const float* ws, float* out, int outer, int parts, int numel, float beta, cgc_stream_t stream) {
  if (numel <= 0 || outer <= 0) return 0;
  if (outer > 65535) return CGC_EINVAL;
  hipLaunchKernelGGL(k_reduce_many_parts, dim3((unsigned)ceil_div(numel, 32), (unsigned)outer), dim3(256), 0, as_stream(stream), ws, out,
                     parts, numel, beta, numel, numel);
  CGC_RETURN_IF_LAUNCH_FAILED();
  return 0;
}

// out[r * ldo + c] (r < rows, c < width) = sum over parts of ws[p][r * width + c]: cgc_reduce_batch_sum with a row stride on the result
// (the same kernels, the same summation order: the same bits)
int reduce_batch_sum_rows(const float* ws, float* out, int parts, int rows, int width, int ldo, float beta, hipStream_t stream) {
  const int64_t numel = (int64_t)rows * width;
  if (numel <= 0) return 0;
  if (width <= 0 || ldo < width || numel > 0x7fffffff) return CGC_EINVAL;
  if (parts >= 16 && numel <= (1 << 20)) {
    hipLaunchKernelGGL(k_reduce_many_parts, dim3((unsigned)ceil_div((int)numel, 32), 1u), dim3(256), 0, stream, ws, out, parts, (int)numel, beta,
                       width, ldo);
  } else {
    const int64_t blocks = ceil_div64(numel, 256);
    hipLaunchKernelGGL(k_reduce_batch_sum, dim3((unsigned)(blocks < 4096 ? blocks : 4096)), dim3(256), 0, stream, ws, out, parts,
                       (long long)numel, beta, width, ldo);
  }
  CGC_RETURN_IF_LAUNCH_FAILED();
  return 0;
}

extern "C" int cgc_reduce_batch_sum(const float* ws, float* out, int parts, int64_t numel, float beta, cgc_stream_t stream) {
  if (numel <= 0) return 0;
  if (parts >= 16 && numel <= (1 << 20)) return cgc_reduce_batched(ws, out, 1, parts, (int)numel, beta, stream);
  const int64_t blocks = ceil_div64(numel, 256);
  const int w = numel <= 0x7fffffff ? (int)numel : 0x7fffffff;       // (one row: the stride is never applied)
  hipLaunchKernelGGL(k_reduce_batch_sum, dim3((unsigned)(blocks < 4096 ? blocks : 4096)), dim3(256), 0, as_stream(stream), ws, out,
                     parts, (long long)numel, beta, w, w);
  CGC_RETURN_IF_LAUNCH_FAILED();
  return 0;
}
