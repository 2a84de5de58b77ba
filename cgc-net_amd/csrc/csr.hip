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

// ------------------------------------------------------------------------------------------------
// The same structure for a batch whose edge list is GROUPED BY GRAPH (round 6): what Batch.from_data_list emits -- the edges of
// graph g are edge_index[:, eptr[g] .. eptr[g+1]) and stay inside its node range gptr[g] .. gptr[g+1].  Then nothing crosses a
// graph: ONE workgroup builds one graph's rows with everything it indexes at random in LDS (histogram, scan, fill, per-row sort +
// de-duplication, unique counts), and a second launch -- it needs every graph's unique count for its place in the compacted arrays --
// compacts, transposes (again histogram / scan / fill / per-column sort in LDS) and forms the edge weights of _re_norm_adj, their
// transposed copy and the mean divisor.  TWO launches instead of the 19-23 of cgc_graph_build.  Every array comes out bit for bit as
// cgc_graph_build writes it (sorted rows: the order in which the atomics hand out slots does not survive the sort).
// What the first version of these kernels taught (57 + 132 us per batch with 32 workgroups -- no faster than the 20 launches): a lone
// workgroup per graph is a chain of memory round trips unless every pass over global memory is SLOT-parallel -- lane k touches element
// k: coalesced, independent, several in flight per thread -- and everything indexed by row or column lives in LDS.  A thread that
// walks "its" row in global memory pays a round trip per element, and a wave storing 64 ten-element runs touches 64 cache lines per
// instruction.
// Differences by design: (1) an edge whose end points are not both inside its own graph's node range is dropped and counted as a bad
// edge (the general build keeps an edge between two graphs of the batch; the reference's dense indexing cannot represent one);
// (2) the envelope: nmax <= 4095 nodes, emax + nmax <= 32767 entries per graph, and both kernels' LDS (20 (nmax + 1) + 4 (emax + nmax)
// bytes + 4 KB) within the 160 KB of a CU -- the entry point refuses anything else and the caller takes the general build.
#define GL_MAXN 4095
#define GL_MAXE 32767
#define GL_T 1024
#define GL_SORT 16          // rows / columns up to this length are sorted by rank in registers

struct GlArgs {
  const int64_t* ei;
  int64_t E;
  int n, B, add_diag, n1, ec;  // n1 = nmax + 1 rounded up to 4 ints, ec = emax + nmax rounded up to 64 entries
  float p;                     // < 0: no edge weights
  const int* gptr;
  const int* eptr;
  int *rowptr, *col, *rowidx, *t_rowptr, *t_col, *t_perm;
  float *val, *t_val, *inv_d;
  int *gnnz, *gbad, *dlg, *colraw, *bad_out;
};

// exclusive scan of v[0 .. m) into out[0 .. m], out[m] = total; GL_T threads, every thread a contiguous slice.  `tot`: 16 ints of LDS.
// Ends with a barrier (and needs one in front of it if v was just written).
__device__ __forceinline__ int block_scan_lds(const int* __restrict__ v, int* __restrict__ out, int m, int* __restrict__ tot) {
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int per = (m + GL_T - 1) / GL_T;
  const int lo = min(t * per, m), hi = min(lo + per, m);
  int s = 0;
  for (int i = lo; i < hi; ++i) s += v[i];
  int incl = s;
  for (int o = 1; o < 64; o <<= 1) {
    const int up = __shfl_up(incl, o);
    if (lane >= o) incl += up;
  }
  if (lane == 63) tot[wave] = incl;
  __syncthreads();
  int wbase = 0, total = 0;
  for (int w = 0; w < GL_T / 64; ++w) {
    if (w < wave) wbase += tot[w];
    total += tot[w];
  }
  int run = wbase + incl - s;
  for (int i = lo; i < hi; ++i) {
    const int x = v[i];
    out[i] = run;
    run += x;
  }
  if (t == 0) out[m] = total;
  __syncthreads();
  return total;
}

// blk[b] = the segment (row of ptr[0 .. m]) that holds element 64 b, for b = 0 .. ceil(total / 64); then owner(k) walks a few
// segments from blk[k >> 6] instead of bisecting all of ptr for every element.
__device__ __forceinline__ void build_block_index(const int* __restrict__ ptr, int m, int total, unsigned short* __restrict__ blk) {
  for (int b = threadIdx.x; b * 64 < total; b += GL_T) {
    const int k = b * 64;
    int lo = 0, hi = m;                              // largest i with ptr[i] <= k
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if (ptr[mid] <= k) lo = mid; else hi = mid;
    }
    blk[b] = (unsigned short)lo;
  }
}
// Lanes of a wave whose key equals their predecessor lane's form a RUN (an edge list sorted by centre: ~9 consecutive edges share a
// row).  One LDS atomic per run instead of one per lane: same-address atomics of a wave are executed one after the other, and with
// every lane in a 9-way collision the two atomic passes over 16 k edges WERE the kernel (50 of its 52 us).  Returns the lane's position
// inside its run, the lane that heads it, and -- for the head -- the run's length.  Invalid lanes belong to no run.
__device__ __forceinline__ void wave_runs(int key, bool valid, int lane, bool& head, int& head_lane, int& pos, int& len) {
  const int prev = __shfl_up(key, 1);
  const int pv = __shfl_up((int)valid, 1);
  head = valid && (lane == 0 || !pv || prev != key);
  const unsigned long long hm = __ballot(head), vm = __ballot(valid);
  const unsigned long long below = hm & (lane == 63 ? ~0ull : ((2ull << lane) - 1ull));
  head_lane = below ? 63 - __clzll((long long)below) : lane;
  pos = lane - head_lane;
  const unsigned long long above = lane == 63 ? 0ull : ((hm | ~vm) >> (lane + 1));
  len = above ? __ffsll((unsigned long long)above) : 64 - lane;
}

__device__ __forceinline__ int owner_of(const int* __restrict__ ptr, const unsigned short* __restrict__ blk, int k) {
  int i = blk[k >> 6];
  while (ptr[i + 1] <= k) ++i;
  return i;
}

// EPT = elements per thread: every thread's share of the graph's edges (rows kernel) / compacted slots (finish kernel) is requested
// in ONE batch of independent loads and stays in registers for all passes -- a load per loop iteration next to the loop's stores and
// atomics is a memory round trip per iteration (the second version of these kernels: 50 + 50 us per batch; the first: 57 + 132).
template <int EPT>
__global__ __launch_bounds__(GL_T) void k_graph_local_rows(const GlArgs a) {
  extern __shared__ __attribute__((aligned(16))) int gl_lds[];
  int* const cnt = gl_lds;
  int* const start = cnt + a.n1;
  int* const cur = start + a.n1;
  unsigned short* const craw = reinterpret_cast<unsigned short*>(cur + a.n1);      // [ec] local column ids, row by row
  unsigned short* const blk = craw + a.ec;                                          // [ec / 64 + 1]
  __shared__ int tot[16];
  const int g = blockIdx.x, t = threadIdx.x;
  const int g0 = a.gptr[g], ng = a.gptr[g + 1] - g0;
  const int e0 = a.eptr[g], e1 = a.eptr[g + 1];
  // A caller whose nmax / emax understate this graph (the LDS areas and EPT were sized from them): the graph is built EMPTY and all its
  // edges are reported as bad ones -- never an access past the areas.  (Batch.from_data_list's own numbers cannot disagree.)
  if (ng + 1 > a.n1 || (e1 - e0) + ng > a.ec || e1 - e0 > EPT * GL_T || ng < 0 || e1 < e0) {      // (workgroup-uniform)
    for (int i = t; i < ng; i += GL_T) {
      a.rowptr[g0 + i] = 0;
      a.dlg[g0 + i] = 0;
    }
    if (t == 0) {
      a.gnnz[g] = 0;
      a.gbad[g] = e1 > e0 ? e1 - e0 : 0;
    }
    return;
  }
  const int base = e0 + (a.add_diag ? g0 : 0);            // this graph's segment of colraw (capacity order = the general build's)
  unsigned pk[EPT];                                       // (local row << 16) | local column; 0xffffffff: no edge / a bad edge
  int rk[EPT];                                            // the edge's rank inside its row (handed out by the histogram's atomic)
  {
    int64_t r[EPT], c[EPT];
#pragma unroll
    for (int u = 0; u < EPT; ++u) {
      const int idx = min(e0 + t + u * GL_T, e1 - 1);
      r[u] = e1 > e0 ? a.ei[idx] : -1;
      c[u] = e1 > e0 ? a.ei[a.E + idx] : -1;
    }
    for (int i = t; i <= ng; i += GL_T) cnt[i] = 0;
    __syncthreads();
    int bad = 0;
    const int lane = t & 63;
#pragma unroll
    for (int u = 0; u < EPT; ++u) {
      pk[u] = 0xffffffffu;
      rk[u] = 0;
      bool valid = false;
      int rl = -1;
      if (e0 + t + u * GL_T < e1) {
        const int64_t rl64 = r[u] - g0, cl64 = c[u] - g0;
        if (rl64 < 0 || rl64 >= ng || cl64 < 0 || cl64 >= ng) ++bad;
        else {
          valid = true;
          rl = (int)rl64;
          pk[u] = ((unsigned)rl << 16) | (unsigned)cl64;
        }
      }
      bool head;
      int head_lane, pos, len;
      wave_runs(rl, valid, lane, head, head_lane, pos, len);
      int old = 0;
      if (head) old = atomicAdd(&cnt[rl], len);
      old = __shfl(old, head_lane);
      rk[u] = old + pos;
    }
    for (int o = 32; o > 0; o >>= 1) bad += __shfl_xor(bad, o);
    if ((t & 63) == 0) tot[t >> 6] = bad;
    __syncthreads();
    if (t == 0) {
      int bsum = 0;
      for (int w = 0; w < GL_T / 64; ++w) bsum += tot[w];
      a.gbad[g] = bsum;
    }
    if (a.add_diag)
      for (int i = t; i < ng; i += GL_T) cnt[i] += 1;     // the diagonal entry: the row's last slot before the sort
    __syncthreads();
  }
  block_scan_lds(cnt, start, ng, tot);
#pragma unroll
  for (int u = 0; u < EPT; ++u)
    if (pk[u] != 0xffffffffu) craw[start[pk[u] >> 16] + rk[u]] = (unsigned short)(pk[u] & 0xffffu);
  if (a.add_diag)
    for (int i = t; i < ng; i += GL_T) craw[start[i + 1] - 1] = (unsigned short)i;
  __syncthreads();
  // sort every row by column and drop duplicates, in place in LDS; the unique count goes where the histogram was
  for (int i = t; i < ng; i += GL_T) {
    unsigned short* row = craw + start[i];
    const int len = start[i + 1] - start[i];
    if (len <= GL_SORT) {                                  // by RANK, out of registers: no chain of dependent LDS round trips
      unsigned v[GL_SORT];
#pragma unroll
      for (int k = 0; k < GL_SORT; ++k) v[k] = k < len ? (unsigned)row[k] : 0xffffffffu;
#pragma unroll
      for (int k = 0; k < GL_SORT; ++k) {
        int rank = 0;
#pragma unroll
        for (int q = 0; q < GL_SORT; ++q) rank += (v[q] < v[k]) || (v[q] == v[k] && q < k);
        if (k < len) row[rank] = (unsigned short)v[k];
      }
    } else {
      for (int k = 1; k < len; ++k) {
        const unsigned short v = row[k];
        int j = k - 1;
        while (j >= 0 && row[j] > v) { row[j + 1] = row[j]; --j; }
        row[j + 1] = v;
      }
    }
    int u = 0, less = 0, diag = 0;
    for (int k = 0; k < len; ++k)
      if (k == 0 || row[k] != row[k - 1]) {
        less += row[k] < i;
        diag |= row[k] == i;
        row[u++] = row[k];
      }
    cnt[i] = u;
    a.dlg[g0 + i] = diag | (less << 16);                  // where the row's diagonal entry sits (second launch: edge weights, mean divisor)
  }
  __syncthreads();
  const int ug = block_scan_lds(cnt, cur, ng, tot);       // cur := the graph's LOCAL row pointers (its offset in the batch: second launch)
  build_block_index(cur, ng, ug, blk);
  for (int i = t; i < ng; i += GL_T) a.rowptr[g0 + i] = cur[i];
  if (t == 0) a.gnnz[g] = ug;
  __syncthreads();
  for (int k = t; k < ug; k += GL_T) {                    // the compacted rows, element k by lane k (stores only: nothing waits for them):
    const int i = owner_of(cur, blk, k);                  // (local row << 16) | local column -- the second launch never searches for a row
    a.colraw[base + k] = (i << 16) | (int)craw[start[i] + (k - cur[i])];
  }
}

template <int EPT>
__global__ __launch_bounds__(GL_T) void k_graph_local_finish(const GlArgs a) {
  extern __shared__ __attribute__((aligned(16))) int gl_lds[];
  int* const lrp = gl_lds;                                // local row pointers
  int* const tc = lrp + a.n1;                             // column histogram, then the fill cursors
  int* const ts = tc + a.n1;                              // local transposed row pointers
  int* const dl = ts + a.n1;                              // per row (from the first launch): has a diagonal entry (bit 0), entries in front of it (<< 16)
  float* const w = reinterpret_cast<float*>(dl + a.n1);   // per row: the off-diagonal weight of _re_norm_adj
  unsigned* const tpair = reinterpret_cast<unsigned*>(w + a.n1);                    // [ec] (source row << 16) | (forward slot << 1) | diagonal, column by column
  __shared__ int tot[16];
  __shared__ int s_G, s_bad;
  const int g = blockIdx.x, t = threadIdx.x;
  const int g0 = a.gptr[g], ng_all = a.gptr[g + 1] - g0;
  const int base = a.eptr[g] + (a.add_diag ? g0 : 0);
  const int ug = a.gnnz[g];
  // (a graph the rows kernel refused -- see there -- has no entries; its rows still get their pointers and mean divisors below as
  // long as they fit the LDS areas, otherwise they are written directly)
  const bool refused = ng_all + 1 > a.n1 || ng_all < 0;
  const int ng = refused ? 0 : ng_all;
  int cl[EPT];                                             // this thread's slots t, t + 1024, ...: (local row << 16) | local column, one batch of loads
#pragma unroll
  for (int u = 0; u < EPT; ++u) cl[u] = a.colraw[base + min(t + u * GL_T, max(ug - 1, 0))];
  {                                                        // place of this graph in the compacted arrays: unique counts of the graphs before it
    int s = 0, b = 0;
    for (int j = t; j < a.B; j += GL_T) {
      if (j < g) s += a.gnnz[j];
      b += a.gbad[j];
    }
    for (int o = 32; o > 0; o >>= 1) { s += __shfl_xor(s, o); b += __shfl_xor(b, o); }
    __shared__ int ps[16], pb[16];
    if ((t & 63) == 0) { ps[t >> 6] = s; pb[t >> 6] = b; }
    __syncthreads();
    if (t == 0) {
      int S = 0, Bd = 0;
      for (int q = 0; q < GL_T / 64; ++q) { S += ps[q]; Bd += pb[q]; }
      s_G = S;
      s_bad = Bd;
    }
    __syncthreads();
  }
  const int G = s_G;
  if (refused)
    for (int i = t; i < ng_all; i += GL_T) {
      a.rowptr[g0 + i] = G;
      a.t_rowptr[g0 + i] = G;
      a.inv_d[g0 + i] = 1.f;
    }
  for (int i = t; i <= ng; i += GL_T) {
    lrp[i] = i < ng ? a.rowptr[g0 + i] : ug;
    tc[i] = 0;
    dl[i] = i < ng ? a.dlg[g0 + i] : 0;
  }
  __syncthreads();
  if (g == a.B - 1 && t == 0) {
    a.rowptr[a.n] = G + ug;
    a.t_rowptr[a.n] = G + ug;
  }
  if (g == 0 && t == 0) *a.bad_out = s_bad + (int)(a.E - ((int64_t)a.eptr[a.B] - a.eptr[0]));    // + edges outside every graph's range of the list
  const bool weights = a.p >= 0.f;
#pragma unroll
  for (int u = 0; u < EPT; ++u)                            // pass 1 over the slots: column histogram (the slots of one row have distinct columns:
    if (t + u * GL_T < ug) atomicAdd(&tc[cl[u] & 0xffff], 1);                                              // these rarely collide)
  __syncthreads();
  for (int i = t; i < ng; i += GL_T) {                    // per row: weight, mean divisor (the sum in slot order, as cgc_csr_invdeg forms it)
    const int u = lrp[i + 1] - lrp[i], diag = dl[i] & 0xffff, less = dl[i] >> 16;
    float sum = (float)u, wv = 0.f;
    if (weights) {
      wv = (1.f / ((float)(u - diag) + RENORM_EPS)) * (1.f - a.p);
      sum = 0.f;
      for (int k = 0; k < u; ++k) sum += (diag && k == less) ? a.p : wv;
    }
    w[i] = wv;
    a.inv_d[g0 + i] = 1.f / fmaxf(sum, 1.f);
    a.rowptr[g0 + i] = G + lrp[i];
  }
  block_scan_lds(tc, ts, ng, tot);                         // (starts with its own reads of tc: the barrier above covers them)
  for (int i = t; i <= ng; i += GL_T) tc[i] = 0;
  for (int i = t; i < ng; i += GL_T) a.t_rowptr[g0 + i] = G + ts[i];
  __syncthreads();
#pragma unroll
  for (int u = 0; u < EPT; ++u) {                          // pass 2 over the slots: the forward arrays out, the transposed fill into LDS
    const int k = t + u * GL_T;
    if (k < ug) {
      const int c = cl[u] & 0xffff, i = cl[u] >> 16;
      a.col[G + k] = g0 + c;
      a.rowidx[G + k] = g0 + i;
      if (weights) a.val[G + k] = (c == i) ? a.p : w[i];
      tpair[ts[c] + atomicAdd(&tc[c], 1)] = ((unsigned)i << 16) | ((unsigned)k << 1) | (unsigned)(c == i);
    }
  }
  __syncthreads();
  for (int j = t; j < ng; j += GL_T) {                    // every column's sources ascending (the key's high bits; unique per column)
    unsigned* seg = tpair + ts[j];
    const int len = ts[j + 1] - ts[j];
    if (len <= GL_SORT) {
      unsigned v[GL_SORT];
#pragma unroll
      for (int k = 0; k < GL_SORT; ++k) v[k] = k < len ? seg[k] : 0xffffffffu;
#pragma unroll
      for (int k = 0; k < GL_SORT; ++k) {
        int rank = 0;
#pragma unroll
        for (int q = 0; q < GL_SORT; ++q) rank += v[q] < v[k];
        if (k < len) seg[rank] = v[k];
      }
    } else {
      for (int k = 1; k < len; ++k) {
        const unsigned v = seg[k];
        int q = k - 1;
        while (q >= 0 && seg[q] > v) { seg[q + 1] = seg[q]; --q; }
        seg[q + 1] = v;
      }
    }
  }
  __syncthreads();
  for (int k = t; k < ug; k += GL_T) {                    // the transposed arrays out, element k by lane k
    const unsigned key = tpair[k];
    const int i = (int)(key >> 16), slot = (int)((key >> 1) & 32767u);
    a.t_col[G + k] = g0 + i;
    a.t_perm[G + k] = G + slot;
    if (weights) a.t_val[G + k] = (key & 1u) ? a.p : w[i];
  }
}

extern "C" int cgc_graph_local_max_nodes(void) { return GL_MAXN; }

// Same outputs and workspace as cgc_graph_build (ws: 3 * (n + 1) + 2 * max(cap, 1) ints; the bad-edge count at the same offset).
// gptr [B + 1], eptr [B + 1] (device, int32); nmax = the largest graph, emax = the most edges of one graph (host).  CGC_EINVAL --
// nothing launched, take cgc_graph_build -- outside the envelope described above (or B > 65535, 2 B > n + 1: the per-graph counts
// live in the workspace's node part).
extern "C" int cgc_graph_build_local(const int64_t* edge_index, int64_t E, int n, const int* gptr, const int* eptr, int B, int nmax, int emax,
                                     float renorm_p, int* rowptr, int* col, int* rowidx, int* t_rowptr, int* t_col, int* t_perm, float* val,
                                     float* t_val, float* inv_d, int* ws, cgc_stream_t stream_) {
  hipStream_t stream = as_stream(stream_);
  const bool renorm = renorm_p >= 0.f;
  if (n <= 0 || E < 0 || B <= 0 || nmax <= 0 || emax < 0 || gptr == nullptr || eptr == nullptr) return CGC_EINVAL;
  if (nmax > GL_MAXN || (int64_t)emax + nmax > GL_MAXE || B > 65535 || 2 * (int64_t)B > (int64_t)n + 1) return CGC_EINVAL;
  if (renorm && (val == nullptr || t_val == nullptr)) return CGC_EINVAL;
  const int64_t cap64 = E + (renorm ? n : 0);
  if (cap64 > 0x7fffffff) return CGC_EINVAL;
  GlArgs a;
  a.ei = edge_index; a.E = E; a.n = n; a.B = B; a.add_diag = renorm ? 1 : 0; a.p = renorm_p;
  a.n1 = (nmax + 1 + 3) & ~3;
  a.ec = (emax + nmax + 63) & ~63;
  const size_t lds_rows = sizeof(int) * 3 * (size_t)a.n1 + sizeof(unsigned short) * ((size_t)a.ec + a.ec / 64 + 8);
  const size_t lds_fin = sizeof(int) * 5 * (size_t)a.n1 + sizeof(unsigned) * (size_t)a.ec + sizeof(unsigned short) * ((size_t)a.ec / 64 + 8);
  if (lds_fin > 156 * 1024 || lds_rows > 156 * 1024) return CGC_EINVAL;
  a.gptr = gptr; a.eptr = eptr;
  a.rowptr = rowptr; a.col = col; a.rowidx = rowidx; a.t_rowptr = t_rowptr; a.t_col = t_col; a.t_perm = t_perm;
  a.val = val; a.t_val = t_val; a.inv_d = inv_d;
  a.gnnz = ws;                                     // [B]      unique entries per graph
  a.gbad = ws + B;                                 // [B]      dropped edges per graph
  a.dlg = ws + (n + 1);                            // [n]      per row: diagonal present | entries in front of it << 16
  a.colraw = ws + 3 * (n + 1);                     // [cap]    as in cgc_csr_build: here every graph's compacted rows at its capacity offset
  a.bad_out = ws + cgc_csr_bad_edges_offset(E, n, a.add_diag);
  const int ept_e = ceil_div(emax > 0 ? emax : 1, GL_T), ept_s = ceil_div(emax + nmax, GL_T);     // both <= 32 by the envelope
#define GL_LAUNCH(KERNEL, EPT_, LDS_)                                                               \
  do {                                                                                              \
    static bool attr__[CGC_MAX_DEVICES] = {};                                                       \
    cgc_allow_lds(reinterpret_cast<const void*>(&KERNEL<EPT_>), 156 * 1024, attr__);                \
    hipLaunchKernelGGL((KERNEL<EPT_>), dim3(B), dim3(GL_T), LDS_, stream, a);                       \
  } while (0)
  if (ept_e <= 8) GL_LAUNCH(k_graph_local_rows, 8, lds_rows);
  else if (ept_e <= 16) GL_LAUNCH(k_graph_local_rows, 16, lds_rows);
  else if (ept_e <= 24) GL_LAUNCH(k_graph_local_rows, 24, lds_rows);
  else GL_LAUNCH(k_graph_local_rows, 32, lds_rows);
  if (ept_s <= 8) GL_LAUNCH(k_graph_local_finish, 8, lds_fin);
  else if (ept_s <= 16) GL_LAUNCH(k_graph_local_finish, 16, lds_fin);
  else if (ept_s <= 24) GL_LAUNCH(k_graph_local_finish, 24, lds_fin);
  else GL_LAUNCH(k_graph_local_finish, 32, lds_fin);
#undef GL_LAUNCH
  CGC_RETURN_IF_LAUNCH_FAILED();
  return 0;
}
