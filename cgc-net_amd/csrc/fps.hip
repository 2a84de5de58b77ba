// Farthest-point sampling on nucleus coordinates, a whole batch of graphs per call ("F3").
// Replaces common/utils.py:187-197 (FarthestSampler: k sequential argmax / minimum passes over rows of a precomputed
// n x n int16 distance table, dataflow/construct_feature_graph.py:17-24) inside the 'farthest' / 'fuse' node samplers
// (dataflow/data.py:195-225) -- the table (260 MB per 11 k-node image) is not needed: distances come from the coordinates.
//
// One workgroup of 1024 threads per graph; thread t owns nodes t, t+1024, ... of its graph and keeps their running
// "distance to the chosen set" in REGISTERS (<= FPS_PER nodes per thread), so one iteration is: broadcast the last chosen
// point, fmin-update the owned distances, local argmax, wavefront + LDS argmax across the workgroup.  The recurrence is
// inherently sequential in k (each pick depends on the previous one); ~1.5 us per pick, all graphs of the batch in parallel.
// Two distance definitions (template parameter TABLE16):
//   false  squared distances in fp64 without contraction (numpy float64 on the coordinates): the geometric sampler;
//   true   the REFERENCE's table entries, re-derived on the fly: int16(sqrt(dx*dx + dy*dy)) evaluated in fp32 exactly as
//          euc_dist does on the float32 coordinate file (dataflow/construct_feature_graph.py:17-24: float32 subtract, square,
//          add, sqrt, then astype(int16) = truncation), so that the picks are index-for-index those of FarthestSampler on
//          the stored table (the truncation creates many ties, which numpy.argmax breaks towards the lowest index).
// Ties: lowest index (numpy.argmax).
#include <stdint.h>

#include "common.hpp"

#define FPS_THREADS 1024
#define FPS_PER 16            // nodes per thread: graphs up to 16384 nodes (the register budget of 16 waves per CU)

template <bool TABLE16>
__global__ __launch_bounds__(FPS_THREADS) void k_fps(const float* __restrict__ pos, const int* __restrict__ gptr,
                                                     const int* __restrict__ start, const int* __restrict__ optr,
                                                     int* __restrict__ out) {
#pragma clang fp contract(off)
  __shared__ double s_val[16];
  __shared__ int s_idx[16];
  __shared__ int s_cur;
  const int g = blockIdx.x, lo = gptr[g], ng = gptr[g + 1] - lo;
  const int o0 = optr[g], k = optr[g + 1] - o0;
  if (ng <= 0 || k <= 0) return;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  float px[FPS_PER], py[FPS_PER];                    // coordinates stay fp32 (exact), distances are formed in fp64
  double dist[FPS_PER];
#pragma unroll
  for (int u = 0; u < FPS_PER; ++u) {
    const int i = t + u * FPS_THREADS;
    px[u] = i < ng ? pos[2 * (size_t)(lo + i)] : 0.f;
    py[u] = i < ng ? pos[2 * (size_t)(lo + i) + 1] : 0.f;
    dist[u] = i < ng ? 1.0e300 : -1.0;               // infinity for real nodes, never the argmax for padding
  }
  int cur = min(max(start[g], 0), ng - 1);
  for (int it = 0; it < k; ++it) {
    if (t == 0) out[o0 + it] = lo + cur;
    const float cxf = pos[2 * (size_t)(lo + cur)], cyf = pos[2 * (size_t)(lo + cur) + 1];
    const double cx = (double)cxf, cy = (double)cyf;
    double best = -2.0;
    int bi = 0x7fffffff;
#pragma unroll
    for (int u = 0; u < FPS_PER; ++u) {
      if (u * FPS_THREADS < ng) {                    // uniform across the workgroup: skips unused register slots
        double d;
        if (TABLE16) {
          const float fx = px[u] - cxf, fy = py[u] - cyf;
          const float sx = fx * fx, sy = fy * fy;                        // each rounded to fp32 (no contraction), like numpy
          d = (double)(int)(short)(int)sqrtf(sx + sy);             // astype(int16): truncate, wrap to 16 bits
        } else {
          const double dx = (double)px[u] - cx, dy = (double)py[u] - cy;
          d = dx * dx + dy * dy;
        }
        dist[u] = fmin(dist[u], d);
        if (dist[u] > best) { best = dist[u]; bi = t + u * FPS_THREADS; }   // ascending index within the thread: first max wins
      }
    }
    for (int o = 32; o > 0; o >>= 1) {
      const double ov = __shfl_xor(best, o);
      const int oi = __shfl_xor(bi, o);
      if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
    }
    if (lane == 0) { s_val[wave] = best; s_idx[wave] = bi; }
    __syncthreads();
    if (t == 0) {
      double b = s_val[0];
      int i0 = s_idx[0];
      for (int w = 1; w < FPS_THREADS / 64; ++w)
        if (s_val[w] > b || (s_val[w] == b && s_idx[w] < i0)) { b = s_val[w]; i0 = s_idx[w]; }
      s_cur = i0;
    }
    __syncthreads();
    cur = s_cur;
  }
}

// pos [n,2] f32; gptr [B+1] first node of each graph; start [B] first pick (local index, as the reference draws it with
// np.random.randint); optr [B+1] offsets of each graph's picks in out (k_g = optr[g+1] - optr[g] <= n_g); out: GLOBAL ids.
extern "C" int cgc_farthest_point_sample(const float* pos, const int* gptr, int B, int max_nodes, const int* start, const int* optr,
                                         int* out, cgc_stream_t stream) {
  if (B <= 0) return 0;
  if (max_nodes > FPS_THREADS * FPS_PER) return CGC_EINVAL;
  hipLaunchKernelGGL(k_fps<false>, dim3(B), dim3(FPS_THREADS), 0, as_stream(stream), pos, gptr, start, optr, out);
  CGC_RETURN_IF_LAUNCH_FAILED();
  return 0;
}

// Same picks as the reference's FarthestSampler on its stored int16 distance table (common/utils.py:187-197 reading the file
// written at dataflow/construct_feature_graph.py:17-24), the table entries being recomputed from the float32 coordinates.
extern "C" int cgc_farthest_point_sample_table16(const float* pos, const int* gptr, int B, int max_nodes, const int* start,
                                                 const int* optr, int* out, cgc_stream_t stream) {
  if (B <= 0) return 0;
  if (max_nodes > FPS_THREADS * FPS_PER) return CGC_EINVAL;
  hipLaunchKernelGGL(k_fps<true>, dim3(B), dim3(FPS_THREADS), 0, as_stream(stream), pos, gptr, start, optr, out);
  CGC_RETURN_IF_LAUNCH_FAILED();
  return 0;
}
