// Sustained rate of back-to-back v_mfma_f32_32x32x2_f32 (register operands) by waves per CU and independent accumulators per wave;
// every configuration is run once to warm up (the clock ramps over milliseconds) and timed on the second launch.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float floatx16 __attribute__((ext_vector_type(16)));
template <int NACC>
__global__ void k(float* out, int iters) {
  const int lane = threadIdx.x & 63;
  floatx16 a[NACC];
  for (int j = 0; j < NACC; ++j) for (int i = 0; i < 16; ++i) a[j][i] = 0.f;
  const float x = 1.f + lane * 1e-3f, y = 1.f - lane * 1e-3f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 16 / NACC; ++u)
#pragma unroll
      for (int j = 0; j < NACC; ++j) a[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a[j], 0, 0, 0);
  }
  float r = 0.f;
  for (int j = 0; j < NACC; ++j) r += a[j][j];
  if (r == 12345.678f) out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
template <int NACC> void run(float* out, int threads, int grid) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int iters = 20000;
  hipLaunchKernelGGL(k<NACC>, dim3(grid), dim3(threads), 0, 0, out, iters); (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0); hipLaunchKernelGGL(k<NACC>, dim3(grid), dim3(threads), 0, 0, out, iters); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  const double fl = (double)grid * (threads / 64) * iters * 16 * 4096;
  printf("acc %d threads %4d grid %4d: %.3f ms %.1f TF\n", NACC, threads, grid, ms, fl / ms / 1e9);
}
int main() {
  float* out; (void)hipMalloc(&out, 1 << 22);
  run<4>(out, 256, 256); run<4>(out, 512, 256); run<4>(out, 1024, 256); run<4>(out, 256, 512);
  run<2>(out, 512, 256); run<1>(out, 1024, 256);
  return 0;
}
