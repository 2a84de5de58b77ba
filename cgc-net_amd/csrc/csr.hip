// Graph structure on the device: COO edge list -> column-sorted, de-duplicated CSR + its transpose.
// Replaces the reference's densification (to_dense_adj, model/utils.py:3-36): instead of a [B,Nmax,Nmax] float
// tensor (592 MB at batch 32 x ~1800 nodes) the batch keeps ~9 int32 per node.  Integer work, HBM/latency-bound;
// counting uses int atomics (order-independent), the per-row sort makes the result deterministic.
// No host synchronisation: nnz stays on the device (rowptr[n]); arrays have capacity E (+n).
#include "common.hpp"

#define RENORM_EPS 1e-15f

__global__ void k_hist_rows(const int64_t* __restrict__ ei, int64_t E, int n, int add_diag, int* __restrict__ cnt) {
  const int64_t total = E + (add_diag ? n : 0);
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    if (i < E) {
      // ids outside [0, n) (a bad batch offset, a foreign Batch) must not reach the atomics: such edges are dropped and
      // COUNTED in cnt[n] (reported through bad_edges) -- the reference's dense indexing raises an IndexError instead
      const int64_t r = ei[i], c = ei[E + i];
      if (r < 0 || r >= n || c < 0 || c >= n) { atomicAdd(&cnt[n], 1); continue; }
      atomicAdd(&cnt[(int)r], 1);
    } else {
      atomicAdd(&cnt[(int)(i - E)], 1);
    }
  }
}

__global__ __launch_bounds__(256) void k_zero_ints(int* __restrict__ p, int count) {
  const int i = (blockIdx.x * 256 + threadIdx.x) * 4;
#pragma unroll
  for (int u = 0; u < 4; ++u)
    if (i + u < count) p[i + u] = 0;
}

// exclusive scan of in[0..n) into out[0..n], out[n] = total.  One workgroup of 1024 threads, each owning a contiguous slice.
__global__ __launch_bounds__(1024) void k_exclusive_scan(const int* __restrict__ in, int* __restrict__ out, int n) {
  __shared__ int wave_tot[16];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int per = (n + 1023) / 1024;
  const int lo = min(t * per, n), hi = min(lo + per, n);
  int s = 0;
#pragma unroll 8
  for (int i = lo; i < hi; ++i) s += in[i];     // independent loads: unrolled so that 8 are in flight
  // inclusive scan of the 1024 slice totals: wave scan, then scan of wave totals
  int incl = s;
  for (int o = 1; o < 64; o <<= 1) {
    const int up = __shfl_up(incl, o);
    if (lane >= o) incl += up;
  }
  if (lane == 63) wave_tot[wave] = incl;
  __syncthreads();
  int wbase = 0;
  for (int w = 0; w < wave; ++w) wbase += wave_tot[w];
  int run = wbase + incl - s;   // exclusive prefix of this thread's slice
#pragma unroll 8
  for (int i = lo; i < hi; ++i) {
    const int v = in[i];
    out[i] = run;
    run += v;
  }
  if (t == 1023) out[n] = wbase + incl;
}

// The same scan cut over many workgroups (n = 57.7 k rows took 28 us in the single-workgroup form, three times per batch):
// pass 1 leaves each 2048-element block's total, pass 2 lets every block add the totals of the blocks before it to its own
// exclusive scan.  Two ~4 us launches instead of one 28 us launch.
#define SCAN_CH 2048
__global__ __launch_bounds__(256) void k_scan_block_sums(const int* __restrict__ in, int n, int* __restrict__ bsum) {
  __shared__ int wsum[4];
  const int base = blockIdx.x * SCAN_CH + threadIdx.x * 8;
  int s = 0;
#pragma unroll
  for (int u = 0; u < 8; ++u) s += (base + u < n) ? in[base + u] : 0;
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) bsum[blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

__global__ __launch_bounds__(256) void k_scan_apply(const int* __restrict__ in, int n, const int* __restrict__ bsum, int nblk,
                                                    int* __restrict__ out) {
  __shared__ int wtot[4];
  __shared__ int s_prefix;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  int pre = 0;                                           // totals of the blocks before this one (and of all, for out[n])
  for (int j = t; j < (int)blockIdx.x; j += 256) pre += bsum[j];
  for (int o = 32; o > 0; o >>= 1) pre += __shfl_xor(pre, o);
  if (lane == 0) wtot[wave] = pre;
  __syncthreads();
  if (t == 0) s_prefix = wtot[0] + wtot[1] + wtot[2] + wtot[3];
  __syncthreads();
  const int block_prefix = s_prefix;
  __syncthreads();
  const int base = blockIdx.x * SCAN_CH + t * 8;
  int v[8], s = 0;
#pragma unroll
  for (int u = 0; u < 8; ++u) { v[u] = (base + u < n) ? in[base + u] : 0; s += v[u]; }
  int incl = s;
  for (int o = 1; o < 64; o <<= 1) {
    const int up = __shfl_up(incl, o);
    if (lane >= o) incl += up;
  }
  if (lane == 63) wtot[wave] = incl;
  __syncthreads();
  int run = block_prefix + incl - s;
  for (int w = 0; w < wave; ++w) run += wtot[w];
#pragma unroll
  for (int u = 0; u < 8; ++u) {
    if (base + u < n) out[base + u] = run;
    run += v[u];
  }
  if ((int)blockIdx.x == nblk - 1 && t == 255) out[n] = run;     // the last thread of the last block ends on the grand total
}

// exclusive scan of in[0..n) into out[0..n] (out[n] = total); `scratch` holds `scratch_ints` ints (may be too small: then the
// single-workgroup kernel runs)
static void exclusive_scan(const int* in, int* out, int n, int* scratch, int64_t scratch_ints, hipStream_t stream) {
  const int nblk = ceil_div(n, SCAN_CH);
  if (nblk < 4 || nblk > scratch_ints) {
    hipLaunchKernelGGL(k_exclusive_scan, dim3(1), dim3(1024), 0, stream, in, out, n);
    return;
  }
  hipLaunchKernelGGL(k_scan_block_sums, dim3(nblk), dim3(256), 0, stream, in, n, scratch);
  hipLaunchKernelGGL(k_scan_apply, dim3(nblk), dim3(256), 0, stream, in, n, scratch, nblk, out);
}

__global__ void k_fill_rows(const int64_t* __restrict__ ei, int64_t E, int n, int add_diag, const int* __restrict__ start,
                            int* __restrict__ cursor, int* __restrict__ colraw) {
  const int64_t total = E + (add_diag ? n : 0);
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    int r, c;
    if (i < E) {
      const int64_t r64 = ei[i], c64 = ei[E + i];
      if (r64 < 0 || r64 >= n || c64 < 0 || c64 >= n) continue;       // dropped and counted by k_hist_rows
      r = (int)r64; c = (int)c64;
    } else { r = c = (int)(i - E); }
    const int pos = start[r] + atomicAdd(&cursor[r], 1);
    colraw[pos] = c;
  }
}

// one thread per row: insertion-sort the row's columns (degree ~9 for k-NN cell graphs), drop duplicates, count uniques
__global__ void k_sort_dedup_rows(const int* __restrict__ start, int* __restrict__ colraw, int n, int* __restrict__ ucnt) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n) return;
  const int s = start[r], e = start[r + 1];
  for (int i = s + 1; i < e; ++i) {
    const int v = colraw[i];
    int j = i - 1;
    while (j >= s && colraw[j] > v) { colraw[j + 1] = colraw[j]; --j; }
    colraw[j + 1] = v;
  }
  int u = 0;
  for (int i = s; i < e; ++i)
    if (i == s || colraw[i] != colraw[i - 1]) colraw[s + u++] = colraw[i];
  ucnt[r] = u;
}

__global__ void k_compact_rows(const int* __restrict__ start, const int* __restrict__ colraw, const int* __restrict__ rowptr, int n,
                               int* __restrict__ col, int* __restrict__ rowidx, const int* __restrict__ bad_cnt, int* __restrict__ bad_out) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r == 0) *bad_out = *bad_cnt;      // number of out-of-range edges that were dropped (0 for a well-formed batch)
  if (r >= n) return;
  const int s = start[r], d = rowptr[r], u = rowptr[r + 1] - d;
  for (int k = 0; k < u; ++k) { col[d + k] = colraw[s + k]; rowidx[d + k] = r; }
}

// ---- transpose (bucket by column; keys are unique after the de-duplication above)
__global__ void k_hist_cols(const int* __restrict__ rowptr, int n, const int* __restrict__ col, int cap, int* __restrict__ cnt) {
  const int nnz = rowptr[n];
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < cap; i += gridDim.x * blockDim.x)
    if (i < nnz) atomicAdd(&cnt[col[i]], 1);
}

__global__ void k_fill_cols(const int* __restrict__ rowptr, int n, const int* __restrict__ col, const int* __restrict__ rowidx, int cap,
                            const int* __restrict__ t_rowptr, int* __restrict__ cursor, int* __restrict__ t_col, int* __restrict__ t_perm) {
  const int nnz = rowptr[n];
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < cap; i += gridDim.x * blockDim.x)
    if (i < nnz) {
      const int c = col[i];
      const int pos = t_rowptr[c] + atomicAdd(&cursor[c], 1);
      t_col[pos] = rowidx[i];
      t_perm[pos] = i;
    }
}

__global__ void k_sort_pairs_rows(const int* __restrict__ start, int* __restrict__ key, int* __restrict__ payload, int n) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n) return;
  const int s = start[r], e = start[r + 1];
  for (int i = s + 1; i < e; ++i) {
    const int v = key[i], p = payload[i];
    int j = i - 1;
    while (j >= s && key[j] > v) { key[j + 1] = key[j]; payload[j + 1] = payload[j]; --j; }
    key[j + 1] = v;
    payload[j + 1] = p;
  }
}

// ws[cgc_csr_bad_edges_offset()] receives the number of edges whose ids were outside [0, n) (they are dropped)
extern "C" int64_t cgc_csr_bad_edges_offset(int64_t E, int n, int add_diag) {
  const int64_t cap = E + (add_diag ? n : 0);
  return 3 * ((int64_t)n + 1) + (cap > 1 ? cap : 1);
}

extern "C" int cgc_csr_build(const int64_t* edge_index, int64_t E, int n, int add_diag, int* rowptr, int* col, int* rowidx,
                             int* t_rowptr, int* t_col, int* t_perm, int* ws, cgc_stream_t stream_) {
  hipStream_t stream = as_stream(stream_);
  if (n < 0 || E < 0) return CGC_EINVAL;
  if (n == 0) {
    (void)hipMemsetAsync(rowptr, 0, sizeof(int), stream);
    (void)hipMemsetAsync(t_rowptr, 0, sizeof(int), stream);
    (void)hipMemsetAsync(ws + cgc_csr_bad_edges_offset(E, n, add_diag), 0, sizeof(int), stream);
    return 0;
  }
  const int64_t cap64 = E + (add_diag ? n : 0);
  if (cap64 > 0x7fffffff) return CGC_EINVAL;
  const int cap = (int)cap64;
  int* cnt = ws;
  int* start = ws + (n + 1);
  int* cursor = ws + 2 * (n + 1);
  int* colraw = ws + 3 * (n + 1);
  int* scan_ws = ws + cgc_csr_bad_edges_offset(E, n, add_diag) + 1;          // the rest of the second `cap` region
  const int64_t scan_ints = (cap64 > 1 ? cap64 : 1) - 1;
  const int tb = 256;
  const int g_edges = (int)(ceil_div64(cap64 > 0 ? cap64 : 1, tb) < 4096 ? ceil_div64(cap64 > 0 ? cap64 : 1, tb) : 4096);
  const int g_rows = ceil_div(n, tb);

  // (one launch each: hipMemsetAsync of a size that is not a multiple of its fill width is TWO runtime kernels -- six fill launches
  // per build, 28 us of a step, for the three memsets that stood here)
  hipLaunchKernelGGL(k_zero_ints, dim3(ceil_div(3 * (n + 1), 1024)), dim3(256), 0, stream, cnt, 3 * (n + 1));   // cnt, start, cursor
  hipLaunchKernelGGL(k_hist_rows, dim3(g_edges), dim3(tb), 0, stream, edge_index, E, n, add_diag, cnt);
  exclusive_scan(cnt, start, n, scan_ws, scan_ints, stream);
  hipLaunchKernelGGL(k_fill_rows, dim3(g_edges), dim3(tb), 0, stream, edge_index, E, n, add_diag, start, cursor, colraw);
  hipLaunchKernelGGL(k_sort_dedup_rows, dim3(g_rows), dim3(tb), 0, stream, start, colraw, n, cnt);   // cnt := unique count
  exclusive_scan(cnt, rowptr, n, scan_ws, scan_ints, stream);
  hipLaunchKernelGGL(k_compact_rows, dim3(g_rows), dim3(tb), 0, stream, start, colraw, rowptr, n, col, rowidx, cnt + n,
                     ws + cgc_csr_bad_edges_offset(E, n, add_diag));
  CGC_RETURN_IF_LAUNCH_FAILED();

  hipLaunchKernelGGL(k_zero_ints, dim3(ceil_div(3 * (n + 1), 1024)), dim3(256), 0, stream, cnt, 3 * (n + 1));   // cnt, (start: dead by now), cursor
  hipLaunchKernelGGL(k_hist_cols, dim3(g_edges), dim3(tb), 0, stream, rowptr, n, col, cap, cnt);
  exclusive_scan(cnt, t_rowptr, n, scan_ws, scan_ints, stream);
  hipLaunchKernelGGL(k_fill_cols, dim3(g_edges), dim3(tb), 0, stream, rowptr, n, col, rowidx, cap, t_rowptr, cursor, t_col, t_perm);
  hipLaunchKernelGGL(k_sort_pairs_rows, dim3(g_rows), dim3(tb), 0, stream, t_rowptr, t_col, t_perm, n);
  CGC_RETURN_IF_LAUNCH_FAILED();
  return 0;
}

// ---- level-1 _re_norm_adj on the CSR (A6) and the clamped mean divisor (A4)
__global__ void k_edge_renorm(const int* __restrict__ rowptr, const int* __restrict__ col, int n, float p, float* __restrict__ val) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n) return;
  const int s = rowptr[r], e = rowptr[r + 1];
  int off = 0;
  for (int k = s; k < e; ++k) off += (col[k] != r);
  const float w = (1.f / ((float)off + RENORM_EPS)) * (1.f - p);
  for (int k = s; k < e; ++k) val[k] = (col[k] == r) ? p : w;
}

// weights in transposed slot order: t_val[k] = val[t_perm[k]] for the nnz live slots (the backward aggregations then read
// their weights contiguously instead of through a dependent index load per edge)
__global__ void k_transpose_vals(const int* __restrict__ t_rowptr, const int* __restrict__ t_perm, const float* __restrict__ val,
                                 int n, float* __restrict__ t_val) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n) return;
  for (int k = t_rowptr[r]; k < t_rowptr[r + 1]; ++k) t_val[k] = val[t_perm[k]];
}

extern "C" int cgc_csr_transpose_vals(const int* t_rowptr, const int* t_perm, const float* val, int n, float* t_val,
                                      cgc_stream_t stream) {
  if (n <= 0) return 0;
  hipLaunchKernelGGL(k_transpose_vals, dim3(ceil_div(n, 256)), dim3(256), 0, as_stream(stream), t_rowptr, t_perm, val, n, t_val);
  CGC_RETURN_IF_LAUNCH_FAILED();
  return 0;
}

__global__ void k_csr_invdeg(const int* __restrict__ rowptr, const float* __restrict__ val, int n, float* __restrict__ out) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n) return;
  const int s = rowptr[r], e = rowptr[r + 1];
  float sum;
  if (val == nullptr) {
    sum = (float)(e - s);
  } else {
    sum = 0.f;
    for (int k = s; k < e; ++k) sum += val[k];
  }
  out[r] = 1.f / fmaxf(sum, 1.f);
}

extern "C" int cgc_edge_renorm(const int* rowptr, const int* col, int n, float p, float* val, cgc_stream_t stream) {
  if (n <= 0) return 0;
  hipLaunchKernelGGL(k_edge_renorm, dim3(ceil_div(n, 256)), dim3(256), 0, as_stream(stream), rowptr, col, n, p, val);
  CGC_RETURN_IF_LAUNCH_FAILED();
  return 0;
}

extern "C" int cgc_csr_invdeg(const int* rowptr, const float* val, int n, float* out, cgc_stream_t stream) {
  if (n <= 0) return 0;
  hipLaunchKernelGGL(k_csr_invdeg, dim3(ceil_div(n, 256)), dim3(256), 0, as_stream(stream), rowptr, val, n, out);
  CGC_RETURN_IF_LAUNCH_FAILED();
  return 0;
}

// Everything graph.BatchGraph holds, behind one call: cgc_csr_build, then (renorm_p >= 0) cgc_edge_renorm + cgc_csr_transpose_vals,
// then cgc_csr_invdeg.  val / t_val may be NULL when renorm_p < 0.  Same kernels in the same order: one host call instead of four.
extern "C" int cgc_graph_build(const int64_t* edge_index, int64_t E, int n, float renorm_p, int* rowptr, int* col, int* rowidx, int* t_rowptr,
                               int* t_col, int* t_perm, float* val, float* t_val, float* inv_d, int* ws, cgc_stream_t stream) {
  const bool renorm = renorm_p >= 0.f;
  int rc = cgc_csr_build(edge_index, E, n, renorm ? 1 : 0, rowptr, col, rowidx, t_rowptr, t_col, t_perm, ws, stream);
  if (rc != 0) return rc;
  if (renorm) {
    if (val == nullptr || t_val == nullptr) return CGC_EINVAL;
    rc = cgc_edge_renorm(rowptr, col, n, renorm_p, val, stream);
    if (rc != 0) return rc;
    rc = cgc_csr_transpose_vals(t_rowptr, t_perm, val, n, t_val, stream);
    if (rc != 0) return rc;
  }
  return cgc_csr_invdeg(rowptr, renorm ? val : nullptr, n, inv_d, stream);
}
