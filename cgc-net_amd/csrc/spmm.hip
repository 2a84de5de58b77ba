// Neighbour aggregation on the CSR:  out[i,:] = post[i] * sum_{k in row i} w_k * pre[col[k]] * x[col[k],:]
//
// Replaces torch.matmul(adj, x) of DenseSAGEConv at level 1 (model/network.py:114-116; narrow widths 16..64) and the
// inner product A*S of (S^T A) S (model/network.py:207; width = cluster count, "K4").  HBM-bound gather kernel:
//   * a row group of `lpr` lanes owns one row x one column tile; every lane holds VEC (=4 -> 16-byte) consecutive columns,
//     so each neighbour row is fetched as one contiguous, fully coalesced segment (1 KiB per wave for wide rows);
//   * the row's (col, weight) pairs are read ONCE by the group's lanes (coalesced) and staged in registers; they reach the
//     other lanes through wavefront shuffles -- no per-edge scalar re-reads, no atomics (the row owner writes its result);
//   * gathers are issued four at a time before the first FMA so that ~4 KiB per wave is in flight;
//   * wide rows are cut into column tiles and the block index is remapped so that one XCD (private 4 MiB L2) works on a
//     contiguous range of rows (= a few whole graphs): the ~9x re-read of neighbour rows is then served by that L2.
#include <stdlib.h>

#include "common.hpp"

#ifndef GATHER_U
#define GATHER_U 9   // neighbour rows requested before the first FMA: one batch covers a k-NN(8)+self row
#endif

template <int VEC>
__global__ __launch_bounds__(256) void k_spmm(const int* __restrict__ rowptr, const int* __restrict__ col, const int* __restrict__ perm,
                                              const float* __restrict__ val, const float* __restrict__ pre,
                                              const float* __restrict__ post, const float* __restrict__ x, float* __restrict__ out,
                                              int n, int W, int ld, int lpr, int n_ctiles, int rows_per_block, int blocks_per_ct,
                                              int n_chunks, int nt_store, const int* __restrict__ gptr) {
  // XCD-contiguous virtual block id (blocks are dealt round-robin to the 8 XCDs; speed only, never correctness)
  const int nb = gridDim.x, b = blockIdx.x;
  const int vb = (nb % 8 == 0) ? (b % 8) * (nb / 8) + b / 8 : b;
  const int per_chunk = blocks_per_ct * n_ctiles;
  const int chunk = vb / per_chunk;
  if (chunk >= n_chunks) return;
  const int rem = vb - chunk * per_chunk;
  const int ct = rem / blocks_per_ct, rb = rem - ct * blocks_per_ct;
  // chunk = rows whose neighbour rows one XCD should keep in its L2 while it sweeps the column tiles: a whole graph
  // when the caller told us the graph boundaries, a fixed row range otherwise
  int row0, row_end;
  if (gptr != nullptr) {
    const int g0 = gptr[chunk], g1 = gptr[chunk + 1];
    row0 = g0 + rb * rows_per_block;
    if (row0 >= g1) return;
    row_end = min(row0 + rows_per_block, g1);
  } else {
    row0 = (chunk * blocks_per_ct + rb) * rows_per_block;
    row_end = min(row0 + rows_per_block, n);
  }

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int sl = lane & (lpr - 1), sub = lane / lpr, rpw = 64 / lpr;
  const int c0 = (ct * lpr + sl) * VEC;
  const bool colok = c0 < W;

  for (int rbase = row0 + wave * rpw; rbase < row_end; rbase += 4 * rpw) {
    const int r = rbase + sub;
    const bool valid = r < row_end;
    int s = 0, e = 0;
    if (valid) { s = rowptr[r]; e = rowptr[r + 1]; }
    float acc[VEC];
#pragma unroll
    for (int v = 0; v < VEC; ++v) acc[v] = 0.f;
    for (int k0 = s; k0 < e; k0 += lpr) {
      // stage up to lpr (col, weight) pairs of this row in the group's lanes
      const int kk = k0 + sl;
      int myc = 0;
      float myw = 0.f;
      if (kk < e) {
        myc = col[kk];
        float w = val != nullptr ? val[perm != nullptr ? perm[kk] : kk] : 1.f;
        if (pre != nullptr) w *= pre[myc];
        myw = w;
      }
      const int cnt = min(lpr, e - k0);
      for (int t = 0; t < cnt; t += GATHER_U) {
        Vec<VEC> xv[GATHER_U];
        float ww[GATHER_U];
#pragma unroll
        for (int u = 0; u < GATHER_U; ++u) {
          const int tt = t + u;
          const int cc = __shfl(myc, tt & (lpr - 1), lpr);
          ww[u] = __shfl(myw, tt & (lpr - 1), lpr);
          if (tt < cnt && colok) {
            xv[u].load(x + (size_t)cc * ld + c0);
          } else {
            ww[u] = 0.f;
#pragma unroll
            for (int v = 0; v < VEC; ++v) xv[u].v[v] = 0.f;
          }
        }
#pragma unroll
        for (int u = 0; u < GATHER_U; ++u)
#pragma unroll
          for (int v = 0; v < VEC; ++v) acc[v] = fmaf(ww[u], xv[u].v[v], acc[v]);
      }
    }
    if (valid && colok) {
      const float ps = post != nullptr ? post[r] : 1.f;
      float* op = out + (size_t)r * ld + c0;
      if (nt_store && VEC == 4) {   // streaming result: keep it from evicting the re-read neighbour rows out of L2
        typedef float f4v __attribute__((ext_vector_type(4)));
        f4v o4 = {acc[0] * ps, acc[VEC > 1 ? 1 : 0] * ps, acc[VEC > 2 ? 2 : 0] * ps, acc[VEC > 3 ? 3 : 0] * ps};
        __builtin_nontemporal_store(o4, reinterpret_cast<f4v*>(op));
      } else {
        Vec<VEC> o;
#pragma unroll
        for (int v = 0; v < VEC; ++v) o.v[v] = acc[v] * ps;
        o.store(op);
      }
    }
  }
}

// Wide rows (> 32 sixteen-byte chunks, "K4"): one WAVE owns (row, 1 KiB column tile), so everything about the sparsity
// pattern is wave-uniform.  Row extents, neighbour ids and edge weights are fetched with SCALAR loads into SGPRs (no lane
// staging, no ds_bpermute shuffles; each gather is one global_load_dwordx4 off a scalar row base), U gathers are in
// flight before the first FMA.  Workgroups are short (4 waves x RUN rows) and dispatched in (graph, column tile, row)
// order on XCD-contiguous ids, so the rows in flight on one XCD stay inside ONE (graph, column tile) slab
// (1800 rows x 1 KiB = 1.8 MB) and the ~9x re-read of neighbour rows is served by that XCD's 4 MiB L2.  (A persistent,
// strided schedule with index prefetch was measured 1.5-1.9x slower: waves drift apart and several slabs compete for L2.)
typedef float wide_f4 __attribute__((ext_vector_type(4)));

// CNT neighbour rows of one (row, column tile): ids and weights through the scalar unit, CNT gathers in flight, then FMAs.
template <int CNT, bool VAL, bool PERM, bool PRE>
__device__ __forceinline__ void wide_batch(const int* __restrict__ col, const int* __restrict__ perm,
                                           const float* __restrict__ val, const float* __restrict__ pre,
                                           const float* __restrict__ xl, int W, int k, wide_f4& acc) {
  int cc[CNT];
  wide_f4 xv[CNT];
  float ww[CNT];
#pragma unroll
  for (int u = 0; u < CNT; ++u) cc[u] = col[k + u];
#pragma unroll
  for (int u = 0; u < CNT; ++u) xv[u] = *reinterpret_cast<const wide_f4*>(xl + (size_t)cc[u] * W);   // (nt loads: 2.1x slower, they skip L2)
#pragma unroll
  for (int u = 0; u < CNT; ++u) ww[u] = 1.f;
  if (VAL) {
    int pi[CNT];
#pragma unroll
    for (int u = 0; u < CNT; ++u) pi[u] = PERM ? perm[k + u] : k + u;
#pragma unroll
    for (int u = 0; u < CNT; ++u) ww[u] = val[pi[u]];
  }
  if (PRE) {
    float pr[CNT];
#pragma unroll
    for (int u = 0; u < CNT; ++u) pr[u] = pre[cc[u]];
#pragma unroll
    for (int u = 0; u < CNT; ++u) ww[u] *= pr[u];
  }
#pragma unroll
  for (int u = 0; u < CNT; ++u) acc += ww[u] * xv[u];
}

// row tail (wave-uniform count 0..N): dispatch to the batch of exactly that size
template <int N, bool VAL, bool PERM, bool PRE>
__device__ __forceinline__ void wide_tail(int cnt, const int* __restrict__ col, const int* __restrict__ perm,
                                          const float* __restrict__ val, const float* __restrict__ pre,
                                          const float* __restrict__ xl, int W, int k, wide_f4& acc) {
  if constexpr (N > 0) {
    if (cnt == N) wide_batch<N, VAL, PERM, PRE>(col, perm, val, pre, xl, W, k, acc);
    else wide_tail<N - 1, VAL, PERM, PRE>(cnt, col, perm, val, pre, xl, W, k, acc);
  }
}

template <int U, bool VAL, bool PERM, bool PRE>
__global__ __launch_bounds__(256) void k_spmm_wide(const int* __restrict__ rowptr, const int* __restrict__ col,
                                                   const int* __restrict__ perm, const float* __restrict__ val,
                                                   const float* __restrict__ pre, const float* __restrict__ post,
                                                   const float* __restrict__ x, float* __restrict__ out, int n, int W, int ld,
                                                   int n_ctiles, int run, int blocks_per_ct, int n_chunks,
                                                   const int* __restrict__ gptr, int order, const int* __restrict__ gorder) {
  typedef float f4v __attribute__((ext_vector_type(4)));
  const int nb = gridDim.x, b = blockIdx.x;
  const int vb = (nb % 8 == 0) ? (b % 8) * (nb / 8) + b / 8 : b;
  const int per_chunk = blocks_per_ct * n_ctiles;
  const int chunk = vb / per_chunk;
  if (chunk >= n_chunks) return;
  const int rem = vb - chunk * per_chunk;
  const int ct = rem / blocks_per_ct, rb = rem - ct * blocks_per_ct;
  const int rows_per_block = 4 * run;
  int row0, row_end;
  if (gptr != nullptr) {
    // Graph visited by this dispatch slot (a hint about what the 256 MB Infinity Cache still holds of x, which was just
    // written by the previous kernel).  1: x was written in ascending row order (row kernels) -- the 8 XCDs start together
    // on the last 8 graphs and walk down.  2: x was written by a batched GEMM whose XCD-contiguous tile order left the tail
    // of every eighth of the batch in cache -- every XCD walks its own eighth backwards.  0: ascending.
    int gi = chunk;
    if (gorder != nullptr) {              // caller-supplied visiting sequence (size-balanced over the XCDs: large graphs)
      gi = gorder[chunk];
    } else if (order == 1) {
      const int per_x = n_chunks >> 3;
      gi = ((n_chunks & 7) == 0 && (nb & 7) == 0) ? (per_x - 1 - chunk % per_x) * 8 + chunk / per_x : n_chunks - 1 - chunk;
    } else if (order == 2) {
      const int per_x = n_chunks >> 3;
      gi = ((n_chunks & 7) == 0 && (nb & 7) == 0) ? (chunk / per_x) * per_x + (per_x - 1 - chunk % per_x) : n_chunks - 1 - chunk;
    }
    const int g0 = gptr[gi], g1 = gptr[gi + 1];
    row0 = g0 + rb * rows_per_block;
    row_end = min(row0 + rows_per_block, g1);
  } else {
    row0 = (chunk * blocks_per_ct + rb) * rows_per_block;
    row_end = min(row0 + rows_per_block, n);
  }
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int c0 = (ct * 64 + lane) * 4;
  if (c0 >= W) return;                       // lanes past the row end (last column tile) retire; indices stay scalar
  const float* __restrict__ xl = x + c0;
  int r = row0 + wave * run;
  const int r_end = min(r + run, row_end);
  if (r >= r_end) return;
  int s = rowptr[r];
  for (; r < r_end; ++r) {
    const int e = rowptr[r + 1];
    f4v acc = {0.f, 0.f, 0.f, 0.f};
    int k = s;
    for (; k + U <= e; k += U) wide_batch<U, VAL, PERM, PRE>(col, perm, val, pre, xl, ld, k, acc);
    wide_tail<U - 1, VAL, PERM, PRE>(e - k, col, perm, val, pre, xl, ld, k, acc);   // exactly as many gathers as entries left
    if (post != nullptr) acc *= post[r];
    __builtin_nontemporal_store(acc, reinterpret_cast<f4v*>(out + (size_t)r * ld + c0));
    s = e;
  }
}

// ------------------------------------------------------------------------------------------------
// Wide rows, SPATIALLY ORDERED nodes (round 5): the neighbour union of a block of consecutive rows staged in LDS.
//
// k_spmm_wide re-reads every neighbour row ~9x from L2 (2.9 GB of L2 -> CU traffic per launch at C3 for 531 MB of algorithmic
// bytes: it runs at the rate of its gathers, 0.42 of the HBM peak).  When the nuclei of a graph are listed grid cell by grid cell
// (data.spatial_order: cell = the k-NN radius), the neighbours of RB = 32 CONSECUTIVE rows -- a strip ~6 cells long -- lie in the 3 x 8
// cells around it: ~130 distinct rows instead of 32 x 9 gathers.  One workgroup owns such a block for ALL column tiles:
//   once:      the block's edges -> the window [min col, max col] -> a presence table in LDS -> prefix sum = slot of every union row,
//              every edge's (slot, weight) parked in LDS;
//   per tile:  the U union rows' 512-byte pieces copied into LDS ONCE (coalesced 16-byte loads, 32 lanes per row), then the 32 rows
//              of the block gather from LDS (ds_read_b128, slot / weight by broadcast reads) and stream their 512 bytes out.
// L2 -> CU traffic drops to U / RB ~ 4x the rows instead of 9x, and two workgroups share a CU (80 KB of LDS each): one stages while the
// other gathers.  A block whose union, edge count or id window exceeds the LDS budget (nodes NOT in spatial order, hubs) falls back to
// direct gathers for that block -- correct for any input, fast only for ordered ones: the caller says so (visit bit 2).
#define PATCH_RB 32         // rows per block
#define PATCH_T 128         // floats per column tile (512 bytes: 32 lanes x 16 bytes)
#define PATCH_CAP 144       // union rows the stage holds
#define PATCH_ECAP 512      // edges of a block
#define PATCH_TW 2048       // id window (max col - min col + 1) the presence table covers
struct PatchLds {
  float stage[PATCH_CAP * PATCH_T];          // 73728 B
  unsigned short table[PATCH_TW];            // presence, then slot (0xffff: absent)
  unsigned short eslot[PATCH_ECAP];
  float ew[PATCH_ECAP];
  unsigned short list[PATCH_CAP];            // slot -> id - lo
  int scan[8];
  int lo, hi, U, ok;
};

template <bool VAL, bool PERM, bool PRE>
__global__ __launch_bounds__(256, 2) void k_spmm_patch(const int* __restrict__ rowptr, const int* __restrict__ col, const int* __restrict__ perm,
                                                       const float* __restrict__ val, const float* __restrict__ pre,
                                                       const float* __restrict__ post, const float* __restrict__ x, float* __restrict__ out,
                                                       int W, int ld, const int* __restrict__ gptr, int B, int blocks_per_graph,
                                                       const int* __restrict__ gorder) {
  typedef float f4v __attribute__((ext_vector_type(4)));
  extern __shared__ __attribute__((aligned(16))) unsigned char patch_raw[];
  PatchLds& L = *reinterpret_cast<PatchLds*>(patch_raw);
  const int nb = gridDim.x, b = blockIdx.x;
  const int vb = (nb % 8 == 0) ? (b % 8) * (nb / 8) + b / 8 : b;       // XCD-contiguous: neighbouring blocks share an L2
  const int gi = vb / blocks_per_graph;
  if (gi >= B) return;
  const int g = gorder != nullptr ? gorder[gi] : gi;
  const int g0 = gptr[g], g1 = gptr[g + 1];
  const int r0 = g0 + (vb - gi * blocks_per_graph) * PATCH_RB;
  if (r0 >= g1) return;
  const int r1 = min(r0 + PATCH_RB, g1);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l32 = lane & 31, half = lane >> 5;
  const int s0 = rowptr[r0], e1 = rowptr[r1], E = e1 - s0;
  const int n_ct = (W + PATCH_T - 1) / PATCH_T;

  // ---- once per block: id window, presence table, slots, per-edge (slot, weight)
  if (tid == 0) { L.lo = 0x7fffffff; L.hi = -1; L.ok = 1; }
  __syncthreads();
  {
    int mn = 0x7fffffff, mx = -1;
    for (int k = s0 + tid; k < e1; k += 256) {
      const int c = col[k];
      mn = min(mn, c);
      mx = max(mx, c);
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
      mn = min(mn, __shfl_xor(mn, o));
      mx = max(mx, __shfl_xor(mx, o));
    }
    if (lane == 0 && mx >= 0) {
      atomicMin(&L.lo, mn);
      atomicMax(&L.hi, mx);
    }
  }
  __syncthreads();
  const int lo = L.lo, span = L.hi - lo + 1;            // (E == 0: span <= 0, nothing to stage)
  bool direct = E > PATCH_ECAP || span > PATCH_TW;
  if (!direct && E > 0) {
    for (int i = tid; i < span; i += 256) L.table[i] = 0;
    __syncthreads();
    for (int k = s0 + tid; k < e1; k += 256) L.table[col[k] - lo] = 1;
    __syncthreads();
    // exclusive prefix sum of the presence flags: 8 consecutive entries per thread (span <= 2048), wave scan, 4 wave totals
    const int base = tid * 8;
    int cnt = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) cnt += (base + j < span && L.table[base + j]) ? 1 : 0;
    int incl = cnt;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int up = __shfl_up(incl, o);
      if (lane >= o) incl += up;
    }
    if (lane == 63) L.scan[wave] = incl;
    __syncthreads();
    int off = incl - cnt;
    for (int w = 0; w < wave; ++w) off += L.scan[w];
    const int U = L.scan[0] + L.scan[1] + L.scan[2] + L.scan[3];
    __syncthreads();                                     // (everybody has read the flags' wave totals; the table is rewritten below)
    if (U <= PATCH_CAP) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        if (base + j < span) {
          const bool present = L.table[base + j] != 0;
          L.table[base + j] = present ? (unsigned short)off : (unsigned short)0xffff;
          if (present) {
            L.list[off] = (unsigned short)(base + j);
            ++off;
          }
        }
      }
    }
    if (tid == 0) L.U = U;
    __syncthreads();
    if (U > PATCH_CAP) {
      direct = true;
    } else {
      for (int k = s0 + tid; k < e1; k += 256) {
        const int c = col[k];
        float w = VAL ? val[PERM ? perm[k] : k] : 1.f;
        if (PRE) w *= pre[c];
        L.eslot[k - s0] = L.table[c - lo];
        L.ew[k - s0] = w;
      }
    }
  }
  __syncthreads();
  const int U = direct ? 0 : (E > 0 ? L.U : 0);

  for (int ct = 0; ct < n_ct; ++ct) {
    const int c0 = ct * PATCH_T + l32 * 4;
    const bool colok = c0 < W;
    if (!direct) {
      // stage: half a wave per union row, 8 rows per trip and workgroup
      for (int sl = wave * 2 + half; sl < U; sl += 32) {          // four rows per lane in flight
        f4v v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int q = min(sl + 8 * j, U - 1);
          v[j] = colok ? *reinterpret_cast<const f4v*>(x + (size_t)(lo + L.list[q]) * ld + c0) : f4v{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (sl + 8 * j < U) *reinterpret_cast<f4v*>(&L.stage[(sl + 8 * j) * PATCH_T + l32 * 4]) = v[j];
      }
      __syncthreads();
    }
    // gather: half a wave per output row
    for (int r = r0 + wave * 2 + half; r < r1; r += 8) {
      const int s = rowptr[r], e = rowptr[r + 1];
      f4v acc = {0.f, 0.f, 0.f, 0.f};
      if (!direct) {
        // eight edges at a time: their slots and weights first (broadcast reads), then the eight 16-byte gathers, then the FMAs in
        // edge order -- one after the other each edge is two dependent LDS round trips
        const int ke = e - s0;
        for (int k = s - s0; k < ke; k += 8) {
          int sl[8];
          float w[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            const int kk = min(k + u, ke - 1);
            sl[u] = (int)L.eslot[kk];
            w[u] = k + u < ke ? L.ew[kk] : 0.f;
          }
          f4v v[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const f4v*>(&L.stage[sl[u] * PATCH_T + l32 * 4]);
#pragma unroll
          for (int u = 0; u < 8; ++u) acc += w[u] * v[u];
        }
      } else if (colok) {
        for (int k = s; k < e; ++k) {
          const int c = col[k];
          float w = VAL ? val[PERM ? perm[k] : k] : 1.f;
          if (PRE) w *= pre[c];
          acc += w * *reinterpret_cast<const f4v*>(x + (size_t)c * ld + c0);
        }
      }
      if (colok) {
        if (post != nullptr) acc *= post[r];
        __builtin_nontemporal_store(acc, reinterpret_cast<f4v*>(out + (size_t)r * ld + c0));
      }
    }
    if (!direct) __syncthreads();                        // (the stage is overwritten by the next column tile)
  }
}

// tuning knobs (read once from the environment; defaults are the measured best for ~1800-node cell graphs)
static int knob(const char* name, int dflt) {
  const char* v = getenv(name);
  return v != nullptr ? atoi(v) : dflt;
}

static int launch_gather(const int* rowptr, const int* col, const int* perm, const float* val, const float* pre, const float* post,
                         const float* x, float* out, int n, int width, int ld, const int* gptr, int B, int nmax, int visit,
                         hipStream_t stream, const int* gorder = nullptr) {
  static const int k_passes = knob("CGC_SPMM_PASSES", 2), k_chunk = knob("CGC_SPMM_CHUNK", 2048);
  static const int k_nt = knob("CGC_SPMM_NT", 1), k_lds = knob("CGC_SPMM_LDS", 0), k_lpr = knob("CGC_SPMM_LPR", 64);
  const bool vec = (width % 4 == 0) && (ld % 4 == 0) && aligned16(x) && aligned16(out);
  const int chunks = vec ? width / 4 : width;        // per-lane column chunks in a row
  int lpr = pick_lpr(chunks);
  if (lpr > k_lpr) lpr = k_lpr;     // narrower column tiles for wide rows (experiment)
  const int n_ctiles = ceil_div(chunks, lpr);         // > 1 only for rows wider than 64 chunks (lpr == 64)
  const int rpw = 64 / lpr;
  const int rows_per_block = 4 * rpw * k_passes;
  int blocks_per_ct, n_chunks;
  if (gptr != nullptr && n_ctiles > 1) {              // graph-aligned chunks (only matters when rows are tiled)
    blocks_per_ct = ceil_div(nmax, rows_per_block);
    n_chunks = B;
  } else {
    gptr = nullptr;
    const int chunk_rows = ceil_div(k_chunk, rows_per_block) * rows_per_block;
    blocks_per_ct = chunk_rows / rows_per_block;
    n_chunks = ceil_div(n, chunk_rows);
  }
  static const int k_wide = knob("CGC_SPMM_WIDE", 1), k_run = knob("CGC_SPMM_RUN", 2);
  static const int k_order = knob("CGC_SPMM_ORDER", -1);      // >= 0 overrides the caller's visiting-order hint
  const int order = k_order >= 0 ? k_order : visit;
  if (k_wide && vec && lpr == 64) {                   // wave-per-row: scalar index path
    const int rpb = 4 * k_run;
    if (gptr != nullptr) {
      blocks_per_ct = ceil_div(nmax, rpb);
    } else {
      const int chunk_rows = ceil_div(k_chunk, rpb) * rpb;
      blocks_per_ct = chunk_rows / rpb;
      n_chunks = ceil_div(n, chunk_rows);
    }
    const int nbw = ceil_div(n_chunks * blocks_per_ct * n_ctiles, 8) * 8;
#define WIDE_LAUNCH(V, P, Q)                                                                                              \
  hipLaunchKernelGGL((k_spmm_wide<GATHER_U, V, P, Q>), dim3(nbw), dim3(CGC_BLOCK), 0, stream, rowptr, col, perm, val, pre, \
                     post, x, out, n, width, ld, n_ctiles, k_run, blocks_per_ct, n_chunks, gptr, order, gorder)
    const bool hv = val != nullptr, hp = hv && perm != nullptr, hq = pre != nullptr;
    if (!hv && !hq) WIDE_LAUNCH(false, false, false);
    else if (!hv) WIDE_LAUNCH(false, false, true);
    else if (!hp && !hq) WIDE_LAUNCH(true, false, false);
    else if (!hp) WIDE_LAUNCH(true, false, true);
    else if (!hq) WIDE_LAUNCH(true, true, false);
    else WIDE_LAUNCH(true, true, true);
#undef WIDE_LAUNCH
    CGC_RETURN_IF_LAUNCH_FAILED();
    return 0;
  }
  int nb = n_chunks * blocks_per_ct * n_ctiles;
  nb = ceil_div(nb, 8) * 8;
  dim3 grid(nb), block(CGC_BLOCK);
  const int nt = (n_ctiles > 1) ? k_nt : 0;
  if (vec)
    hipLaunchKernelGGL(k_spmm<4>, grid, block, k_lds, stream, rowptr, col, perm, val, pre, post, x, out, n, width, ld, lpr,
                       n_ctiles, rows_per_block, blocks_per_ct, n_chunks, nt, gptr);
  else
    hipLaunchKernelGGL(k_spmm<1>, grid, block, 0, stream, rowptr, col, perm, val, pre, post, x, out, n, width, ld, lpr,
                       n_ctiles, rows_per_block, blocks_per_ct, n_chunks, nt, gptr);
  CGC_RETURN_IF_LAUNCH_FAILED();
  return 0;
}

extern "C" int cgc_spmm(const int* rowptr, const int* col, const int* perm, const float* val, const float* pre, const float* post,
                        const float* x, float* out, int n, int width, cgc_stream_t stream) {
  if (n <= 0 || width <= 0) return 0;
  return launch_gather(rowptr, col, perm, val, pre, post, x, out, n, width, width, nullptr, 0, 0, 0, as_stream(stream));
}

// ------------------------------------------------------------------------------------------------
// Wide rows ("K4": A*S with width = cluster count): graph-slab kernel.
// The gather kernel above re-reads every neighbour row ~9x through L2/Infinity Cache (measured 12 TB/s of cache traffic
// for 2.5 TB/s of algorithmic bandwidth).  Cell graphs are block diagonal and small (~1800 nodes), so instead one
// workgroup takes (graph g, T-column tile): it copies the graph's N_g x T slab of X into LDS ONCE with coalesced 16-byte
// loads (pre-scaled by `pre`), then every output row of the graph gathers its ~9 neighbours from LDS (ds_read_b128) and
// streams its T results out with non-temporal stores.  HBM traffic = algorithmic traffic; the 9x reuse happens in LDS.
// Workgroups of one graph get consecutive (XCD-contiguous) ids so that the two column tiles sharing a 128-byte line run
// on the same XCD back to back.
template <int T, int NTHR>
__global__ __launch_bounds__(NTHR) void k_spmm_slab(const int* __restrict__ rowptr, const int* __restrict__ col,
                                                    const int* __restrict__ perm, const float* __restrict__ val,
                                                    const float* __restrict__ pre, const float* __restrict__ post,
                                                    const float* __restrict__ x, float* __restrict__ out,
                                                    const int* __restrict__ gptr, int B, int W, int n_ctiles) {
  extern __shared__ __attribute__((aligned(16))) float slab[];   // [N_g][T]
  constexpr int Q = T / 4;                                        // 16-byte units per slab row
  const int nb = gridDim.x, b = blockIdx.x;
  const int vb = (nb % 8 == 0) ? (b % 8) * (nb / 8) + b / 8 : b;
  const int g = vb / n_ctiles;
  if (g >= B) return;
  const int ct = vb - g * n_ctiles;
  const int g0 = gptr[g], ng = gptr[g + 1] - g0;
  const int tid = threadIdx.x;

  // phase 1: slab <- pre * X[g0 : g0+ng, ct*T : ct*T+T]
  for (int u = tid; u < ng * Q; u += NTHR) {
    const int r = u / Q, q = u - r * Q;
    const int c = ct * T + q * 4;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (c < W) {
      v = *reinterpret_cast<const float4*>(x + (size_t)(g0 + r) * W + c);
      if (pre != nullptr) {
        const float p = pre[g0 + r];
        v.x *= p; v.y *= p; v.z *= p; v.w *= p;
      }
    }
    *reinterpret_cast<float4*>(&slab[r * T + q * 4]) = v;
  }
  __syncthreads();

  // phase 2: Q lanes per output row
  const int q = tid % Q;
  const int c = ct * T + q * 4;
  for (int r = tid / Q; r < ng; r += NTHR / Q) {
    const int i = g0 + r;
    const int s = rowptr[i], e = rowptr[i + 1];
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int k0 = s; k0 < e; k0 += 8) {
      int cc[8];
      float ww[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {      // indices and weights first (independent loads), LDS gathers after
        const int k = k0 + u;
        cc[u] = -1;
        ww[u] = 0.f;
        if (k < e) {
          cc[u] = col[k] - g0;
          ww[u] = val != nullptr ? val[perm != nullptr ? perm[k] : k] : 1.f;
        }
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        if (cc[u] >= 0) {
          const float4 v = *reinterpret_cast<const float4*>(&slab[cc[u] * T + q * 4]);
          acc.x = fmaf(ww[u], v.x, acc.x);
          acc.y = fmaf(ww[u], v.y, acc.y);
          acc.z = fmaf(ww[u], v.z, acc.z);
          acc.w = fmaf(ww[u], v.w, acc.w);
        }
      }
    }
    if (c < W) {
      const float ps = post != nullptr ? post[i] : 1.f;
      float* op = out + (size_t)i * W + c;
      typedef float f4v __attribute__((ext_vector_type(4)));
      f4v o4 = {acc.x * ps, acc.y * ps, acc.z * ps, acc.w * ps};
      __builtin_nontemporal_store(o4, reinterpret_cast<f4v*>(op));
    }
  }
}

template <int T, int NTHR>
static int launch_slab(const int* rowptr, const int* col, const int* perm, const float* val, const float* pre, const float* post,
                       const float* x, float* out, const int* gptr, int B, int nmax, int W, hipStream_t stream) {
  const int n_ctiles = ceil_div(W, T);
  const size_t lds = sizeof(float) * (size_t)nmax * T;
  static bool attr_set[CGC_MAX_DEVICES] = {};      // per instantiation and device
  cgc_allow_lds(reinterpret_cast<const void*>(&k_spmm_slab<T, NTHR>), 160 * 1024, attr_set);
  const int nb = ceil_div(B * n_ctiles, 8) * 8;
  hipLaunchKernelGGL((k_spmm_slab<T, NTHR>), dim3(nb), dim3(NTHR), lds, stream, rowptr, col, perm, val, pre, post, x, out, gptr, B, W, n_ctiles);
  CGC_RETURN_IF_LAUNCH_FAILED();
  return 0;
}

// Graph-aware entry point: gptr[B+1] = first row of each graph (every row's neighbours lie inside its own graph),
// nmax = largest graph.  Wide, 16-byte-aligned rows of graphs that fit LDS take the slab kernel; everything else the
// gather kernel.
extern "C" int cgc_spmm_graphs_ordered(const int* rowptr, const int* col, const int* perm, const float* val, const float* pre,
                                       const float* post, const float* x, float* out, int n, int width, int ld, const int* gptr, int B,
                                       int nmax, int visit, const int* gorder, cgc_stream_t stream);

extern "C" int cgc_spmm_graphs(const int* rowptr, const int* col, const int* perm, const float* val, const float* pre,
                               const float* post, const float* x, float* out, int n, int width, int ld, const int* gptr, int B,
                               int nmax, int visit, cgc_stream_t stream) {
  return cgc_spmm_graphs_ordered(rowptr, col, perm, val, pre, post, x, out, n, width, ld, gptr, B, nmax, visit, nullptr, stream);
}

// gorder (optional, B ints, a permutation of the graphs): the sequence in which the graphs are visited; the eight XCDs take
// consecutive eighths of it.  Scheduling only -- the result does not depend on it.
extern "C" int cgc_spmm_graphs_ordered(const int* rowptr, const int* col, const int* perm, const float* val, const float* pre,
                                       const float* post, const float* x, float* out, int n, int width, int ld, const int* gptr, int B,
                                       int nmax, int visit, const int* gorder, cgc_stream_t stream) {
  if (n <= 0 || width <= 0) return 0;
  if (ld < width) return CGC_EINVAL;
  static const int k_slab = knob("CGC_SPMM_SLAB", 0);
  const size_t budget = 150 * 1024;
  if (k_slab && ld == width && gptr != nullptr && B > 0 && width > 64 && width % 4 == 0 && aligned16(x) && aligned16(out) && nmax > 0) {
    hipStream_t st = as_stream(stream);
    static const int k_t = knob("CGC_SPMM_T", 16), k_thr = knob("CGC_SPMM_THR", 1024);
#define SLAB_ARGS rowptr, col, perm, val, pre, post, x, out, gptr, B, nmax, width, st
    if (k_t >= 16 && (size_t)nmax * 16 * 4 <= budget) return k_thr >= 1024 ? launch_slab<16, 1024>(SLAB_ARGS) : launch_slab<16, 512>(SLAB_ARGS);
    if (k_t >= 8 && (size_t)nmax * 8 * 4 <= budget) return k_thr >= 1024 ? launch_slab<8, 1024>(SLAB_ARGS) : launch_slab<8, 512>(SLAB_ARGS);
    if ((size_t)nmax * 4 * 4 <= budget) return k_thr >= 1024 ? launch_slab<4, 1024>(SLAB_ARGS) : launch_slab<4, 512>(SLAB_ARGS);
#undef SLAB_ARGS
  }
  const int trec = width > 64 ? cgc_timing_begin(CGC_TAG_SPMM_WIDE, n, width, ld, val != nullptr, 0, 0, 0, as_stream(stream)) : -1;
  // visit bit 2: the caller lists every graph's nodes grid cell by grid cell.  The LDS-staged patch kernel that could exploit it is an
  // EXPERIMENT (CGC_SPMM_PATCH=1): measured 277-312 us against 150 us for k_spmm_wide on the same ordered graphs at C3 (profiles/
  // r05_k4_patch_kernel.txt) -- its staging and gather phases are latency chains of a few waves, the gather kernel
  // keeps 32 waves x 9 loads in flight per CU.  Off by default; the hint itself is free and already helps the gather kernel (+3 %)
  static const int k_patch = knob("CGC_SPMM_PATCH", 0);
  int rc;
  if (((visit & 8) || (k_patch && (visit & 4))) && gptr != nullptr && B > 0 && nmax > 0 && width > 256 && width % 4 == 0 && ld % 4 == 0 &&
      aligned16(x) && aligned16(out)) {
    hipStream_t st = as_stream(stream);
    const int bpg = ceil_div(nmax, PATCH_RB);
    const int nbp = ceil_div(B * bpg, 8) * 8;
    const bool hv = val != nullptr, hp = hv && perm != nullptr, hq = pre != nullptr;
#define PATCH_LAUNCH(V, P, Q)                                                                                                         \
  do {                                                                                                                                \
    static bool attr__[CGC_MAX_DEVICES] = {};                                                                                         \
    cgc_allow_lds(reinterpret_cast<const void*>(&k_spmm_patch<V, P, Q>), (int)sizeof(PatchLds), attr__);                              \
    hipLaunchKernelGGL((k_spmm_patch<V, P, Q>), dim3(nbp), dim3(256), sizeof(PatchLds), st, rowptr, col, perm, val, pre, post, x, out, \
                       width, ld, gptr, B, bpg, gorder);                                                                              \
  } while (0)
    if (!hv && !hq) PATCH_LAUNCH(false, false, false);
    else if (!hv) PATCH_LAUNCH(false, false, true);
    else if (!hp && !hq) PATCH_LAUNCH(true, false, false);
    else if (!hp) PATCH_LAUNCH(true, false, true);
    else if (!hq) PATCH_LAUNCH(true, true, false);
    else PATCH_LAUNCH(true, true, true);
#undef PATCH_LAUNCH
    hipError_t e__ = hipGetLastError();
    rc = e__ == hipSuccess ? 0 : (int)e__;
    // (the experiment needs 2 x 81 KB of dynamic LDS per CU: a launch the device refuses falls back to the gather kernel)
    if (rc != 0) rc = launch_gather(rowptr, col, perm, val, pre, post, x, out, n, width, ld, gptr, B, nmax, visit & 3, as_stream(stream), gorder);
  } else {
    rc = launch_gather(rowptr, col, perm, val, pre, post, x, out, n, width, ld, gptr, B, nmax, visit & 3, as_stream(stream), gorder);
  }
  cgc_timing_end(trec, as_stream(stream));
  return rc;
}
