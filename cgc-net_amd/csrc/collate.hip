// Device-side finish of Batch.from_data_list ("F1", the loader front-end: dataflow/data.py:330-354 builds the items,
// model/network.py:239-243 consumes the batch).  The host packs the raw per-graph arrays of a batch into ONE pinned buffer
// and issues ONE host-to-device copy; this kernel then does, in one launch, what the reference does item by item on the
// host: feature z-scoring  x = (x - mean) / std  (dataflow/data.py:353), the sorted ``batch`` vector (graph id per node) and
// the cumulative node offsets on edge_index (torch_geometric Batch.from_data_list).
#include <stdint.h>

#include "common.hpp"

__device__ __forceinline__ int seg_of(const int* __restrict__ ptr, int B, int64_t i) {     // largest g with ptr[g] <= i
  int lo = 0, hi = B;
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if ((int64_t)ptr[mid] <= i) lo = mid; else hi = mid;
  }
  return lo;
}

__global__ void k_collate(float* __restrict__ x, int n, int F, const float* __restrict__ mean, const float* __restrict__ stdv,
                          const int* __restrict__ gptr, int B, int64_t* __restrict__ batch, int64_t* __restrict__ ei, int64_t E,
                          const int* __restrict__ eptr) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x, t0 = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (mean != nullptr) {
    const int64_t tot = (int64_t)n * F;
    for (int64_t i = t0; i < tot; i += stride) {
      const int f = (int)(i % F);
      x[i] = (x[i] - mean[f]) / stdv[f];          // IEEE division: bit-identical to the host expression
    }
  }
  if (batch != nullptr)
    for (int64_t i = t0; i < n; i += stride) batch[i] = seg_of(gptr, B, i);
  if (ei != nullptr)
    for (int64_t e = t0; e < E; e += stride) {
      const int64_t off = gptr[seg_of(eptr, B, e)];
      ei[e] += off;
      ei[E + e] += off;
    }
}

extern "C" int cgc_collate(float* x, int n, int F, const float* mean, const float* stdv, const int* gptr, int B, int64_t* batch,
                           int64_t* edge_index, int64_t E, const int* eptr, cgc_stream_t stream) {
  if (n <= 0 || B <= 0) return 0;
  if ((mean == nullptr) != (stdv == nullptr) || (edge_index != nullptr && eptr == nullptr)) return CGC_EINVAL;
  int64_t work = (int64_t)n * (mean != nullptr ? F : 1);
  if (edge_index != nullptr && E > work) work = E;
  int blocks = (int)((work + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(k_collate, dim3(blocks), dim3(256), 0, as_stream(stream), x, n, F, mean, stdv, gptr, B, batch, edge_index, E, eptr);
  CGC_RETURN_IF_LAUNCH_FAILED();
  return 0;
}
