// Cell-graph construction on the GPU ("F2"): for every nucleus its <= k nearest others within radius r, plus itself.
// Replaces torch_cluster.radius_graph(pos, r, None, loop, max_num_neighbors) -- cKDTree.query(k+1, distance_upper_bound =
// r + 1e-8) per graph on the host (dataflow/data.py:246,255,297,348; dataflow/prepare_cv_dataset.py:102).
//
// Uniform-grid bucket search, a whole batch of graphs per call (block-diagonal: a node only sees its own graph):
//   k_knn_bbox   one workgroup per graph: bounding box, cell size = max(r, sqrt(w h / n_g), max(w,h) / n_g)  (>= r so that
//                the 3x3 neighbourhood is exhaustive; the other two terms bound the cell count by 3 n_g + 1 whatever the
//                coordinates are), grid extent, and -- last graph done -- the per-graph cell offsets
//   k_knn_hist / k_knn_scan / k_knn_fill   counting sort of the nodes by cell
//   k_knn_query  one thread per node: scans its 3x3 cells, squared distances in fp64 (the host tree works in double too),
//                keeps the k+1 best (distance, index) pairs in registers (static insertion network) -> ELL rows sorted by
//                distance, ties: the node itself first, then by index (the tree's tie order is unspecified)
//   k_knn_emit   ELL -> COO edge_index [2, nnz] int64 with GLOBAL node ids, rows ascending (what Batch/from_data_list and
//                cgc_csr_build consume)
// HBM-light, latency-bound work (58 k nodes: ~50 candidates each); no atomics on floating point, output deterministic.
#include <stdint.h>

#include "common.hpp"

struct KnnGrid {          // per graph
  float minx, miny, inv_cell;
  int gx, gy, cell0;      // grid extent, first cell id
};

__global__ __launch_bounds__(256) void k_knn_bbox(const float* __restrict__ pos, const int* __restrict__ gptr, int B, float r,
                                                  KnnGrid* __restrict__ grid, int* __restrict__ done) {
  __shared__ float red[4][4];
  const int g = blockIdx.x;
  const int lo = gptr[g], hi = gptr[g + 1];
  float mnx = 3.0e38f, mny = 3.0e38f, mxx = -3.0e38f, mxy = -3.0e38f;
  for (int i = lo + threadIdx.x; i < hi; i += 256) {
    const float x = pos[2 * (size_t)i], y = pos[2 * (size_t)i + 1];
    mnx = fminf(mnx, x); mxx = fmaxf(mxx, x);
    mny = fminf(mny, y); mxy = fmaxf(mxy, y);
  }
  for (int o = 32; o > 0; o >>= 1) {
    mnx = fminf(mnx, __shfl_xor(mnx, o)); mxx = fmaxf(mxx, __shfl_xor(mxx, o));
    mny = fminf(mny, __shfl_xor(mny, o)); mxy = fmaxf(mxy, __shfl_xor(mxy, o));
  }
  const int wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { red[wave][0] = mnx; red[wave][1] = mny; red[wave][2] = mxx; red[wave][3] = mxy; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < 4; ++w) {
      mnx = fminf(mnx, red[w][0]); mny = fminf(mny, red[w][1]);
      mxx = fmaxf(mxx, red[w][2]); mxy = fmaxf(mxy, red[w][3]);
    }
    KnnGrid k;
    const int ng = hi - lo;
    if (ng <= 0) {
      k.minx = k.miny = 0.f; k.inv_cell = 0.f; k.gx = k.gy = 0;
    } else {
      const float w = mxx - mnx, h = mxy - mny;
      float cell = fmaxf(r, sqrtf(w * h / (float)ng));
      cell = fmaxf(cell, fmaxf(w, h) / (float)ng);
      if (!(cell > 0.f)) cell = 1.f;                           // r = 0 and all points coincident
      k.minx = mnx; k.miny = mny; k.inv_cell = 1.f / cell;
      k.gx = (int)(w * k.inv_cell) + 1;
      k.gy = (int)(h * k.inv_cell) + 1;
    }
    k.cell0 = 0;
    grid[g] = k;
    __threadfence();
    if (atomicAdd(done, 1) == B - 1) {                         // last graph: cell offsets (B is small), reset the ticket
      int c = 0;
      for (int q = 0; q < B; ++q) {
        KnnGrid t = grid[q];
        t.cell0 = c;
        c += t.gx * t.gy;
        grid[q].cell0 = t.cell0;
      }
      *done = 0;
    }
  }
}

__device__ __forceinline__ int knn_graph_of(const int* __restrict__ gptr, int B, int i) {
  int lo = 0, hi = B;                                          // largest g with gptr[g] <= i
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (gptr[mid] <= i) lo = mid; else hi = mid;
  }
  return lo;
}

__device__ __forceinline__ void knn_cell_of(const KnnGrid& k, float x, float y, int& cx, int& cy) {
  cx = min(max((int)((x - k.minx) * k.inv_cell), 0), k.gx - 1);
  cy = min(max((int)((y - k.miny) * k.inv_cell), 0), k.gy - 1);
}

__global__ void k_knn_hist(const float* __restrict__ pos, const int* __restrict__ gptr, int B, int n, const KnnGrid* __restrict__ grid,
                           int* __restrict__ key, int* __restrict__ gid, int* __restrict__ cnt) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int g = knn_graph_of(gptr, B, i);
  const KnnGrid k = grid[g];
  int cx, cy;
  knn_cell_of(k, pos[2 * (size_t)i], pos[2 * (size_t)i + 1], cx, cy);
  const int c = k.cell0 + cy * k.gx + cx;
  key[i] = c;
  gid[i] = g;
  atomicAdd(&cnt[c], 1);
}

// exclusive scan of in[0..n) into out[0..n], out[n] = total; one workgroup (the array has at most ~3 n entries)
__global__ __launch_bounds__(1024) void k_knn_scan(const int* __restrict__ in, int* __restrict__ out, int n) {
  __shared__ int wave_tot[16];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int per = (n + 1023) / 1024;
  const int lo = min(t * per, n), hi = min(lo + per, n);
  int s = 0;
#pragma unroll 8
  for (int i = lo; i < hi; ++i) s += in[i];
  int incl = s;
  for (int o = 1; o < 64; o <<= 1) {
    const int up = __shfl_up(incl, o);
    if (lane >= o) incl += up;
  }
  if (lane == 63) wave_tot[wave] = incl;
  __syncthreads();
  int wbase = 0;
  for (int w = 0; w < wave; ++w) wbase += wave_tot[w];
  int run = wbase + incl - s;
#pragma unroll 8
  for (int i = lo; i < hi; ++i) {
    const int v = in[i];
    out[i] = run;
    run += v;
  }
  if (t == 1023) out[n] = wbase + incl;
}

__global__ void k_knn_fill(const int* __restrict__ key, int n, const int* __restrict__ start, int* __restrict__ cursor,
                           int* __restrict__ sorted) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int c = key[i];
  sorted[start[c] + atomicAdd(&cursor[c], 1)] = i;
}

template <int KP>    // KP = k + 1 slots (the node itself is always among its own nearest)
__global__ __launch_bounds__(128) void k_knn_query(const float* __restrict__ pos, int n, const KnnGrid* __restrict__ grid,
                                                   const int* __restrict__ key, const int* __restrict__ gid,
                                                   const int* __restrict__ start, const int* __restrict__ sorted, double r2, int kp,
                                                   int loop, int* __restrict__ nbr, int* __restrict__ cnt) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const KnnGrid k = grid[gid[i]];
  const double xi = (double)pos[2 * (size_t)i], yi = (double)pos[2 * (size_t)i + 1];
  const int local = key[i] - k.cell0, cy = local / k.gx, cx = local - cy * k.gx;
  double bd[KP];
  int bi[KP];
#pragma unroll
  for (int u = 0; u < KP; ++u) { bd[u] = 1.0e300; bi[u] = 0x7fffffff; }
  for (int dy = -1; dy <= 1; ++dy) {
    const int yy = cy + dy;
    if (yy < 0 || yy >= k.gy) continue;
    const int x0 = max(cx - 1, 0), x1 = min(cx + 1, k.gx - 1);
    const int c0 = k.cell0 + yy * k.gx;
    const int s = start[c0 + x0], e = start[c0 + x1 + 1];     // the three cells of a grid row are contiguous
    for (int q = s; q < e; ++q) {
      const int j = sorted[q];
      const double dx = (double)pos[2 * (size_t)j] - xi, dyy = (double)pos[2 * (size_t)j + 1] - yi;
      double dj = dx * dx + dyy * dyy;
      int jj = (j == i) ? -1 : j;            // coincident points: the node itself sorts first among equal distances
      if (dj > r2 || !(dj < bd[KP - 1] || (dj == bd[KP - 1] && jj < bi[KP - 1]))) continue;
      // insertion into the sorted list: one bubble pass from the top (static indices only)
#pragma unroll
      for (int u = 0; u < KP; ++u) {
        const bool before = dj < bd[u] || (dj == bd[u] && jj < bi[u]);
        const double td = bd[u];
        const int ti = bi[u];
        if (before) { bd[u] = dj; bi[u] = jj; dj = td; jj = ti; }
      }
    }
  }
  // cKDTree semantics: the k+1 nearest INCLUDING the node itself, then drop the node unless loop
  int c = 0;
#pragma unroll
  for (int u = 0; u < KP; ++u) {
    if (u < kp && bi[u] != 0x7fffffff && (loop || bi[u] != -1)) nbr[(size_t)i * kp + c++] = bi[u] == -1 ? i : bi[u];
  }
  cnt[i] = c;
}

__global__ void k_knn_emit(const int* __restrict__ nbr, const int* __restrict__ rowptr, int n, int kp, int64_t nnz,
                           int64_t* __restrict__ edge_index) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int s = rowptr[i], e = rowptr[i + 1];
  for (int q = s; q < e; ++q) {
    edge_index[q] = i;
    edge_index[nnz + q] = nbr[(size_t)i * kp + (q - s)];
  }
}

// Workspace (ints): grid B*6 (+2 pad) | ticket 1 | key n | gid n | cnt ncell+1 | start ncell+1 | cursor ncell+1 | sorted n, ncell = 3n + B
extern "C" int64_t cgc_radius_knn_ws_ints(int n, int B) {
  const int64_t ncell = 3 * (int64_t)n + B + 1;
  return 6 * (int64_t)B + 8 + 3 * (int64_t)n + 3 * (ncell + 1);
}

extern "C" int cgc_radius_knn(const float* pos, const int* gptr, int B, int n, float r, int k, int loop, int* nbr, int* cnt,
                              int* rowptr, int* ws, cgc_stream_t stream) {
  if (n <= 0 || B <= 0) return 0;
  if (k < 0 || k > 32 || !(r >= 0.f)) return CGC_EINVAL;
  hipStream_t st = as_stream(stream);
  const int64_t ncell = 3 * (int64_t)n + B + 1;
  if (ncell > 0x7ffffff0) return CGC_EINVAL;
  KnnGrid* grid = reinterpret_cast<KnnGrid*>(ws);
  int* ticket = ws + 6 * (size_t)B + 2;
  int* key = ws + 6 * (size_t)B + 8;
  int* gid = key + n;
  int* ccnt = gid + n;
  int* start = ccnt + (ncell + 1);
  int* cursor = start + (ncell + 1);
  int* sorted = cursor + (ncell + 1);
  (void)hipMemsetAsync(ticket, 0, sizeof(int), st);
  (void)hipMemsetAsync(ccnt, 0, sizeof(int) * (size_t)(ncell + 1), st);
  (void)hipMemsetAsync(cursor, 0, sizeof(int) * (size_t)(ncell + 1), st);
  const int tb = 256, gb = ceil_div(n, tb);
  hipLaunchKernelGGL(k_knn_bbox, dim3(B), dim3(256), 0, st, pos, gptr, B, r, grid, ticket);
  hipLaunchKernelGGL(k_knn_hist, dim3(gb), dim3(tb), 0, st, pos, gptr, B, n, grid, key, gid, ccnt);
  hipLaunchKernelGGL(k_knn_scan, dim3(1), dim3(1024), 0, st, ccnt, start, (int)ncell);
  hipLaunchKernelGGL(k_knn_fill, dim3(gb), dim3(tb), 0, st, key, n, start, cursor, sorted);
  const double rr = (double)r + 1e-8, r2 = rr * rr;     // the host tree's distance_upper_bound
  const int kp = k + 1, gq = ceil_div(n, 128);
#define KNN_Q(KP) hipLaunchKernelGGL(k_knn_query<KP>, dim3(gq), dim3(128), 0, st, pos, n, grid, key, gid, start, sorted, r2, kp, loop, nbr, cnt)
  if (kp <= 9) KNN_Q(9);
  else if (kp <= 17) KNN_Q(17);
  else KNN_Q(33);
#undef KNN_Q
  hipLaunchKernelGGL(k_knn_scan, dim3(1), dim3(1024), 0, st, cnt, rowptr, n);
  CGC_RETURN_IF_LAUNCH_FAILED();
  return 0;
}

extern "C" int cgc_knn_emit_edges(const int* nbr, const int* rowptr, int n, int k, int64_t nnz, int64_t* edge_index, cgc_stream_t stream) {
  if (n <= 0) return 0;
  hipLaunchKernelGGL(k_knn_emit, dim3(ceil_div(n, 256)), dim3(256), 0, as_stream(stream), nbr, rowptr, n, k + 1, nnz, edge_index);
  CGC_RETURN_IF_LAUNCH_FAILED();
  return 0;
}
