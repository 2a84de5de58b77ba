// Shared pieces of the matrix-core GEMM kernels (gemm.hip: exact fp32 MFMA; gemm_split.hip: fp32 products as six bf16 MFMA pairs):
// argument block, tile map with tail split, per-item operand bases, the epilogue through LDS, the fix-up kernel of the tail split.
#pragma once
#include <stdlib.h>

#include <type_traits>

#include "common.hpp"

typedef float floatx16 __attribute__((ext_vector_type(16)));

struct GemmArgs {
  const float* A;
  const float* B;
  float* C;
  const float* bias;
  const int* gptr;
  int M, N, K, lda, ldb, ldc;
  long long strideA, strideB, strideC;
  float alpha, beta;
  int ragged;
  int tiles_n;
  int map_mode;   // bit 0: compact tile list for ragged M, bit 1: K-balanced dealing for ragged K (TileMap)
  int per_batch;  // tiles of one batch item at the largest extent; the 1-D grid holds per_batch * nb ids (+ tail pieces)
  int nb;         // batch items
  // tail split (see TileMap): slabs of raw accumulators for the pieces of the tail tiles; nullptr = every tile is computed whole
  float* ws;
  int resident;   // workgroups the chip holds at once = the quantum of a "round"
  int chunk;      // ragged 3: rows per part
  int s_max;      // most pieces a tail tile is cut into
  const unsigned* scale;   // gemm_half.hip only: bits of max |A|, max |B| per batch item (2 per item); nullptr elsewhere
  // optional extra K segments: C += alpha * A_x[s] * B_x[s] (same op() orientation, M, N as the main pair), i.e. the
  // product of the column-concatenated [A | A_x0 | A_x1] with the row-concatenated [B ; B_x0 ; B_x1] without ever
  // materialising the concatenation (Linear over cat[x1,x2,x3]; dS = P dA'^T + X dX'^T)
  int nx;
  const float* xA[2];
  const float* xB[2];
  int xlda[2], xldb[2], xK[2];
  long long xsA[2], xsB[2];
};

#define BK 32
#ifdef CGC_GEMM_TRACE      // experiment build (tools/gemm_wg_timeline.py): per-workgroup timestamps of the 128 x 128 kernel's phases
__device__ unsigned long long g_gemm_trace[65536][6];
#define GT_MARK(slot_) \
  if (TM == 2 && TN == 2 && threadIdx.x == 0 && blockIdx.x < 65536) g_gemm_trace[blockIdx.x][slot_] = wall_clock64();
extern "C" int cgc_gemm_trace_read(unsigned long long* host, int n) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_gemm_trace), sizeof(unsigned long long) * 6 * (size_t)n);
}
#else
#define GT_MARK(slot_)
#endif
enum { PH_FULL = 0, PH_MASK = 1, PH_ANY = 4 };   // flavours of a k-loop phase (k_gemm_f32)
#define KC_LD 36   // LDS row stride (words) of a row-major [mn][k] tile: 16-byte aligned rows, conflict-free ds_read_b128

__device__ __forceinline__ const float* sgpr_ptr(const float* p) {     // a wave-uniform pointer, pinned to scalar registers
  const uintptr_t v = reinterpret_cast<uintptr_t>(p);
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
  return reinterpret_cast<const float*>(((uintptr_t)hi << 32) | lo);
}

// Branch-free fetch of ANY k-tile -- a full or partial tile of the main operand pair or of an extra K segment -- for the
// prologue and the tail of the k loop of k_gemm_f32<FAST>: the segment (base pointers, row strides, reduction length) is picked
// with scalar selects and both operands take the masked unguarded loads.  A free function over a table of VALUES: as a lambda
// capturing by reference inside the phase lambda its closure (a struct of pointers to locals) survived into the generated code
// and put the locals it referred to in scratch memory.
struct SegTable {
  const float *A0, *A1, *A2, *B0, *B1, *B2;
  int lda0, lda1, lda2, ldb0, ldb1, ldb2, K0, K1, K2;
  int nk_main, nkx0;
};
template <class LoaderA, class LoaderB>
__device__ __forceinline__ void fetch_seg(LoaderA& la, LoaderB& lb, const SegTable t, int kt, int m0, int a_last, int n0, int b_last) {
  const int kx = kt - t.nk_main;
  const bool in_main = kx < 0, in_x0 = kx < t.nkx0;
  const float* Ap = in_main ? t.A0 : in_x0 ? t.A1 : t.A2;
  const float* Bp = in_main ? t.B0 : in_x0 ? t.B1 : t.B2;
  const int lda = in_main ? t.lda0 : in_x0 ? t.lda1 : t.lda2;
  const int ldb = in_main ? t.ldb0 : in_x0 ? t.ldb1 : t.ldb2;
  const int Ks = in_main ? t.K0 : in_x0 ? t.K1 : t.K2;
  const int k0 = (in_main ? kt : in_x0 ? kx : kx - t.nkx0) * BK;
  la.load_fast_masked(Ap, lda, m0, a_last, k0, Ks);
  lb.load_fast_masked(Bp, ldb, n0, b_last, k0, Ks);
}

// Which (batch, tile) a workgroup computes.  Speed only -- every tile is computed exactly once whatever the hardware's dispatch
// order is.  Workgroups are dealt round-robin to the 8 XCDs, each with a private 4 MiB L2: linear id lin -> XCD lin & 7, slot
// lin >> 3.  Every XCD gets a CONTIGUOUS run of (batch, tile_m, tile_n) ids, tile_n fastest: the ~64 workgroups resident on an
// XCD then form a (few tile_m) x (all tile_n) super-tile that streams each A panel and each B panel through that L2 once
// (measured before this remap: 31-50 % L2 hit rate and ~9x the algorithmic bytes fetched from the fabric).
//   * uniform batches: the runs are cut from the launched grid.
//   * ragged M (per-graph row counts): the grid is sized for the LARGEST graph, so cutting it into equal runs hands an XCD
//     whose graphs are small mostly empty tiles (C3: per-XCD work spread +-8 %).  The runs are cut from the COMPACT list of
//     real tiles instead: each wave derives the per-graph tile counts from gptr with a wave scan (batch <= 64).
//   * ragged K (per-graph reduction length): every graph has the same tiles but a different duration.  Graphs are ranked by
//     K and dealt to the XCDs in serpentine order (longest first), so that the sums of K per XCD agree within ~1 %
//     (batch a multiple of 8, <= 64; otherwise the plain cut).
//
// TAIL SPLIT (round 3).  T tiles on R resident workgroups run in ceil(T / R) rounds, and the last round is as long as the others
// however few tiles it holds: 4140 tiles (the step's big products at 32 graphs) on 512 slots are 8.09 -> 9 rounds, 522 tiles
// (4 graphs per GPU, the strong-scaling shard) 1.02 -> 2.  With a slab workspace the L = T mod R tiles of the last round (all T
// when T < R) are cut along K into S ~ R / L pieces each; a piece parks its raw accumulators in its slab and k_gemm_fixup adds a
// tile's S slabs in a fixed order and applies alpha / beta / bias -- the launch-boundary reduce: deterministic, no flags, no
// spinning.  The L * S pieces fill one round of 1/S the length.  Tail tiles are taken evenly from the END of every XCD's run
// (the whole-tile part stays a multiple of R, i.e. of 8).  Which tiles are split is a function of (T, R, s_max) only, computed
// the same way by both kernels.
template <int BM>
struct TileMap {
  int mode;                 // 0 plain cut, 1 compact list (ragged M), 2 serpentine by K (ragged K)
  unsigned T, Tdp, L, S;    // real tiles; tiles computed whole; tail tiles; pieces per tail tile
  unsigned q8, r8;          // every XCD owns q8 (+1 for the first r8) consecutive tile ids
  int incl, t, rank;        // per-lane state of modes 1 / 2
  __device__ __forceinline__ void init(const GemmArgs& a, int lane) {
    const unsigned nb = a.nb;
    mode = 0;
    incl = t = rank = 0;
    T = (unsigned)a.per_batch * nb;
    if (a.ragged == 1 && nb <= 64 && (a.map_mode & 1)) {
      mode = 1;
      const int ext = lane < (int)nb ? a.gptr[lane + 1] - a.gptr[lane] : 0;
      t = (ext + BM - 1) / BM;                     // m-tile rows of graph `lane`
      incl = t;
      for (int o = 1; o < 64; o <<= 1) {
        const int up = __shfl_up(incl, o);
        if (lane >= o) incl += up;
      }
      T = (unsigned)__shfl(incl, 63) * a.tiles_n;  // real tiles of the launch
    } else if (a.ragged == 2 && nb <= 64 && (nb & 7u) == 0 && (a.map_mode & 2)) {
      mode = 2;
      const int ext = lane < (int)nb ? a.gptr[lane + 1] - a.gptr[lane] : -1;
      for (int j = 0; j < (int)nb; ++j) {
        const int ej = __shfl(ext, j);
        rank += (ej > ext || (ej == ext && j < lane)) ? 1 : 0;
      }
    }
    q8 = T >> 3;
    r8 = T & 7u;
    Tdp = T;
    L = 0;
    S = 1;
    if (a.ws != nullptr && a.resident > 0 && T > 0) {
      const unsigned R = a.resident;
      const unsigned l = T < R ? T : T % R;
      if (l > 0) {
        unsigned s = (R + l / 2) / l;
        if (s > (unsigned)a.s_max) s = a.s_max;
        if (s >= 2) {
          L = l;
          S = s;
          Tdp = T - l;
        }
      }
    }
  }
  // tail tile j (0 <= j < L) -> its place (XCD, slot) in the runs: the slots behind the whole-tile part, XCD fastest
  __device__ __forceinline__ void tail_slot(unsigned j, unsigned& xcd, unsigned& slot) const {
    const unsigned qd = Tdp >> 3, common = (q8 - qd) * 8u;
    if (j < common) {
      xcd = j & 7u;
      slot = qd + (j >> 3);
    } else {
      xcd = j - common;
      slot = q8;
    }
  }
  __device__ __forceinline__ void locate(const GemmArgs& a, unsigned xcd, unsigned slot, int lane, int& b, int& tile_id) const {
    if (mode == 1) {
      const unsigned cid = xcd * q8 + (xcd < r8 ? xcd : r8) + slot;
      const int row = cid / a.tiles_n;
      b = __popcll(__ballot(incl <= row));                    // graphs that end at or before this row
      const int first = __shfl(incl - t, b);
      tile_id = (row - first) * a.tiles_n + (cid - row * a.tiles_n);
    } else if (mode == 2) {
      const unsigned per_batch = a.per_batch;
      const unsigned p = slot / per_batch;
      const unsigned q = p * 8 + ((p & 1u) ? 7u - xcd : xcd);
      b = __ffsll((unsigned long long)__ballot(lane < a.nb && rank == (int)q)) - 1;
      tile_id = slot - p * per_batch;
    } else {
      const unsigned vb = xcd * q8 + (xcd < r8 ? xcd : r8) + slot;
      b = vb / a.per_batch;
      tile_id = vb - b * a.per_batch;
    }
  }
  // workgroup `lin` of the main kernel: a whole tile (S_out = 1) or piece `piece` of tail tile `tj`
  __device__ __forceinline__ bool select(const GemmArgs& a, unsigned lin, int lane, int& b, int& tile_id, unsigned& tj, int& piece,
                                         int& S_out) const {
    unsigned xcd, slot;
    tj = 0;
    piece = 0;
    S_out = 1;
    if (lin < Tdp) {
      xcd = lin & 7u;
      slot = lin >> 3;
    } else {
      const unsigned j = lin - Tdp;
      if (j >= L * S) return false;
      tj = j / S;
      piece = j - tj * S;
      S_out = S;
      tail_slot(tj, xcd, slot);
    }
    locate(a, xcd, slot, lane, b, tile_id);
    return true;
  }
};

// Operand / output bases and extents of batch item b (ragged: per-graph row offsets and extents from gptr)
struct TileBase {
  const float* A;
  const float* B;
  float* C;
  int M, K;
  __device__ __forceinline__ TileBase(const GemmArgs& a, int b) {
    M = a.M;
    K = a.K;
    A = a.A + (size_t)b * a.strideA;
    B = a.B + (size_t)b * a.strideB;
    C = a.C + (size_t)b * a.strideC;
    if (a.ragged == 1) {
      const int g0 = a.gptr[b];
      M = a.gptr[b + 1] - g0;
      A += (size_t)g0 * a.lda;
      C += (size_t)g0 * a.ldc;
    } else if (a.ragged == 2) {
      const int g0 = a.gptr[b];
      K = a.gptr[b + 1] - g0;
      A += (size_t)g0 * a.lda;
      B += (size_t)g0 * a.ldb;
    } else if (a.ragged == 3) {
      // uniform row chunks (split-K without an offset array): item b = (outer, part); every outer item reduces over a.K rows, cut
      // into parts of a.chunk rows; A (stored [K,M]) and B (stored [K,N]) are the flattened [outer * K, .] row blocks
      const int parts = (a.K + a.chunk - 1) / a.chunk;
      const int outer = b / parts, part = b - outer * parts;
      const long long g0 = (long long)outer * a.K + (long long)part * a.chunk;
      const int left = a.K - part * a.chunk;
      K = left < a.chunk ? left : a.chunk;
      A = a.A + (size_t)g0 * a.lda;
      B = a.B + (size_t)g0 * a.ldb;
    }
  }
};


// Epilogue shared by the kernels below.  The MFMAs are issued with the operands SWAPPED (B fragment first), i.e. every
// 32x32 accumulator holds the TRANSPOSED sub-tile: lane (l31, lhi) owns output ROW l31 and its 16 registers the columns
// (r&3) + 8*(r>>2) + 4*lhi -- four runs of four consecutive columns.  (a*b is commutative and the k order is unchanged, so
// the values are bitwise those of the unswapped product.)  A wave first parks its 32 x (TN*32) strip in a wave-private LDS
// region with 16-byte writes (row stride TN*32+4 words: conflict-free for the 8-lane groups of ds_write_b128), then reads it
// back with lanes running ALONG the rows and stores 16 bytes per lane: every store instruction covers whole 128-byte lines
// (4 rows x 256 B for TN = 2) instead of 64 four-byte pieces of two rows, and the accumulate mode (beta != 0) and the bias read
// with the same pattern.  The former row-per-register layout needed 16 store instructions per accumulator and left the
// output-bound short-K products at 1.6 TB/s.  LDS operations of one wave execute in order, so no barrier is needed between
// the parking writes and the read-back; the caller guarantees that no wave still reads operand tiles from this LDS.
template <int TM, int TN>
__device__ __forceinline__ void gemm_epilogue(const GemmArgs& a, float* __restrict__ C, int M, int N, int m_base, int n_base,
                                              floatx16 (&acc)[TM][TN], float* __restrict__ st, int lane) {
  constexpr int COLS = TN * 32, SLD = COLS + 4, LPR = COLS / 4, RPI = 64 / LPR;   // lanes per row, rows per store instruction
  const int l31 = lane & 31, lhi = lane >> 5;
  const float alpha = a.alpha, beta = a.beta;
  const int cu = lane % LPR, rsub = lane / LPR;
  const int gcol = n_base + cu * 4;
  const bool vec = (a.ldc % 4 == 0) && ((reinterpret_cast<uintptr_t>(C) & 15u) == 0) && (gcol + 3 < N);
  float4 bia = make_float4(0.f, 0.f, 0.f, 0.f);
  if (a.bias != nullptr) {
    if (gcol < N) bia.x = a.bias[gcol];
    if (gcol + 1 < N) bia.y = a.bias[gcol + 1];
    if (gcol + 2 < N) bia.z = a.bias[gcol + 2];
    if (gcol + 3 < N) bia.w = a.bias[gcol + 3];
  }
#pragma unroll
  for (int i = 0; i < TM; ++i) {
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int g = 0; g < 4; ++g)
        *reinterpret_cast<float4*>(&st[l31 * SLD + j * 32 + 8 * g + 4 * lhi]) =
            make_float4(acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]);
    __builtin_amdgcn_wave_barrier();
    float4 v[32 / RPI];
#pragma unroll
    for (int it = 0; it < 32 / RPI; ++it) v[it] = *reinterpret_cast<const float4*>(&st[(it * RPI + rsub) * SLD + cu * 4]);
    __builtin_amdgcn_wave_barrier();
    const int row0 = m_base + i * 32 + rsub;
    if (vec) {
      if (beta != 0.f) {                     // accumulate mode: all reads of the strip in flight before the first write
        float4 cold[32 / RPI];
#pragma unroll
        for (int it = 0; it < 32 / RPI; ++it) {
          const int row = row0 + it * RPI;
          cold[it] = row < M ? *reinterpret_cast<const float4*>(&C[(size_t)row * a.ldc + gcol]) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int it = 0; it < 32 / RPI; ++it) {
          const int row = row0 + it * RPI;
          if (row < M)
            *reinterpret_cast<float4*>(&C[(size_t)row * a.ldc + gcol]) =
                make_float4(fmaf(beta, cold[it].x, alpha * v[it].x + bia.x), fmaf(beta, cold[it].y, alpha * v[it].y + bia.y),
                            fmaf(beta, cold[it].z, alpha * v[it].z + bia.z), fmaf(beta, cold[it].w, alpha * v[it].w + bia.w));
        }
      } else {
#pragma unroll
        for (int it = 0; it < 32 / RPI; ++it) {
          const int row = row0 + it * RPI;
          if (row < M)
            *reinterpret_cast<float4*>(&C[(size_t)row * a.ldc + gcol]) =
                make_float4(alpha * v[it].x + bia.x, alpha * v[it].y + bia.y, alpha * v[it].z + bia.z, alpha * v[it].w + bia.w);
        }
      }
    } else {                                  // unaligned C / ragged right edge: element-wise with the same arithmetic
#pragma unroll
      for (int it = 0; it < 32 / RPI; ++it) {
        const int row = row0 + it * RPI;
        if (row >= M) continue;
        const float vv[4] = {v[it].x, v[it].y, v[it].z, v[it].w};
        const float bb[4] = {bia.x, bia.y, bia.z, bia.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          if (gcol + e >= N) break;
          float* p = &C[(size_t)row * a.ldc + gcol + e];
          *p = beta != 0.f ? fmaf(beta, *p, alpha * vv[e] + bb[e]) : alpha * vv[e] + bb[e];
        }
      }
    }
  }
}

// Second half of the tail split: the S slabs of a tail tile are added in piece order (fixed: the result does not depend on which
// piece finished first) and go through the same epilogue.  One single-wave workgroup per 32 x 32 accumulator (16 per 128 x 128
// tile) with every slab load of a pass in flight at once: with one 4-wave workgroup per tile walking its 16 accumulators and S
// slabs one load after the other the fix-up of 10 tail tiles took 60 us (of a 210 us product).
template <int WGM, int WGN, int TM, int TN>
__global__ __launch_bounds__(64) void k_gemm_fixup(const GemmArgs a) {
  constexpr int BM = WGM * TM * 32, BN = WGN * TN * 32, SUB = 4 * TM * TN;
  __shared__ __attribute__((aligned(16))) float park[32 * 36];
  const int lane = threadIdx.x;
  TileMap<BM> map;
  map.init(a, lane);
  const unsigned tj = blockIdx.x / SUB;
  const int sub = blockIdx.x - tj * SUB;                 // (wave, i, j) of the main kernel's thread geometry
  if (tj >= map.L) return;
  unsigned xcd, slot;
  map.tail_slot(tj, xcd, slot);
  int b, tile_id;
  map.locate(a, xcd, slot, lane, b, tile_id);
  const TileBase tb(a, b);
  const int tile_m = tile_id / a.tiles_n, tile_n = tile_id - tile_m * a.tiles_n;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  if (m0 >= tb.M) return;
  const int wave = sub / (TM * TN), ij = sub - wave * (TM * TN), i = ij / TN, j = ij - i * TN;
  const int wm = wave / WGN, wn = wave - wm * WGN;
  const int S = map.S;
  const float* slab = a.ws + (size_t)tj * S * (size_t)(BM * BN) + (size_t)wave * (TM * TN * 16 * 64) + (size_t)ij * (4 * 256) + lane * 4;
  float4 v[4];
#pragma unroll
  for (int g = 0; g < 4; ++g) v[g] = *reinterpret_cast<const float4*>(slab + g * 256);
  int p = 1;
  for (; p + 3 < S; p += 4) {                            // four slabs per pass: 16 loads in flight, added in piece order
    float4 w[4][4];
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int g = 0; g < 4; ++g) w[q][g] = *reinterpret_cast<const float4*>(slab + (size_t)(p + q) * (BM * BN) + g * 256);
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int g = 0; g < 4; ++g) { v[g].x += w[q][g].x; v[g].y += w[q][g].y; v[g].z += w[q][g].z; v[g].w += w[q][g].w; }
  }
  for (; p < S; ++p) {
    float4 w[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) w[g] = *reinterpret_cast<const float4*>(slab + (size_t)p * (BM * BN) + g * 256);
#pragma unroll
    for (int g = 0; g < 4; ++g) { v[g].x += w[g].x; v[g].y += w[g].y; v[g].z += w[g].z; v[g].w += w[g].w; }
  }
  floatx16 acc[1][1];
#pragma unroll
  for (int g = 0; g < 4; ++g) { acc[0][0][4 * g] = v[g].x; acc[0][0][4 * g + 1] = v[g].y; acc[0][0][4 * g + 2] = v[g].z; acc[0][0][4 * g + 3] = v[g].w; }
  gemm_epilogue<1, 1>(a, tb.C, tb.M, a.N, m0 + (wm * TM + i) * 32, n0 + (wn * TN + j) * 32, acc, park, lane);
}

