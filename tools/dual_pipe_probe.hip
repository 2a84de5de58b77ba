// Can a CU run fp32 MFMA and packed fp32 VALU FMAs at the same time?  Both pipes are rated 157 TFLOP/s on MI355X; the dominant
// kernel of the step is bound by the matrix pipe alone (0.78 of its peak).  Roles per wave inside one 512-thread workgroup (two
// waves per SIMD): mode 0 = both MFMA, 1 = both VALU, 2 = one of each on every SIMD.  Register-resident operands only: this is the
// ceiling for a hybrid GEMM, not a GEMM.    hipcc --offload-arch=gfx950 -O3 tools/dual_pipe_probe.hip -o /tmp/dpp && /tmp/dpp
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float float2_ __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(512) void k_probe(float* out, int iters, int mode) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const bool mfma = mode == 0 || (mode == 2 && wave < 4);
  float r = 0.f;
  if (mfma) {
    floatx16 a0, a1, a2, a3;
    for (int i = 0; i < 16; ++i) { a0[i] = a1[i] = a2[i] = a3[i] = 0.f; }
    const float x = 1.f + lane * 1e-3f, y = 1.f - lane * 1e-3f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, x, a1, 0, 0, 0);
        a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, x, a2, 0, 0, 0);
        a3 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, y, a3, 0, 0, 0);
      }
    }
    r = a0[0] + a1[1] + a2[2] + a3[3];
  } else {
    // 32 independent packed accumulators (64 floats), operands in registers: 16 MFMAs = 65536 flops per wave <-> 256 pk_fma
    float2_ acc[32];
    for (int i = 0; i < 32; ++i) acc[i] = (float2_){0.f, 0.f};
    float2_ x = {1.f + lane * 1e-3f, 1.f - lane * 1e-3f}, y = {0.999f, 1.001f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 8; ++u) {
#pragma unroll
        for (int i = 0; i < 32; ++i) acc[i] = __builtin_elementwise_fma(x, y, acc[i]);
        x = x * y;                          // (keeps the compiler from hoisting; 1 extra instruction per 32)
      }
    }
    for (int i = 0; i < 32; ++i) r += acc[i][0] + acc[i][1];
  }
  if (r == 12345.678f) out[blockIdx.x * 512 + threadIdx.x] = r;
}

int main() {
  float* out;
  (void)hipMalloc(&out, 1 << 20);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int iters = 20000, grid = 256;      // ~20 ms per launch, after a warm-up launch of the same length (the clock ramps over ms)
  for (int mode = 0; mode < 3; ++mode) {
    hipLaunchKernelGGL(k_probe, dim3(grid), dim3(512), 0, 0, out, iters, mode);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k_probe, dim3(grid), dim3(512), 0, 0, out, iters, mode);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    // flops: an MFMA wave does iters*16 MFMAs * 4096; a VALU wave iters*8*32 pk_fma * 64 lanes * 4 flops
    const double fm = (double)iters * 16 * 4096, fv = (double)iters * 8 * 32 * 64 * 4;
    const int nm = mode == 0 ? 8 : mode == 1 ? 0 : 4, nv = 8 - nm;
    const double tot = grid * (nm * fm + nv * fv);
    printf("mode %d (%d MFMA + %d VALU waves per workgroup): %.3f ms  %.1f TFLOP/s  (MFMA part %.1f, VALU part %.1f)\n", mode, nm, nv, ms,
           tot / ms / 1e9, grid * nm * fm / ms / 1e9, grid * nv * fv / ms / 1e9);
  }
  return 0;
}
