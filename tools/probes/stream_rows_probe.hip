// What a "wide row pass" (one read + one write of a [57696, 1140] fp32 tensor with per-column constants) can reach on MI355X:
// the shape of k_bn_act_apply<4,5> (one wave per row, 5 float4 per lane) and variants of it.  torch's copy_ moves the same bytes at
// 5.3 TB/s, the library kernel at 4.45.   build: hipcc --offload-arch=gfx950 -O3 -o tools/probes/stream_rows_probe tools/probes/stream_rows_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef float vf4 __attribute__((ext_vector_type(4)));

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int ROWS, bool NTL, bool NTS>
__global__ __launch_bounds__(256) void k_rows(const float* __restrict__ x, const float* __restrict__ mean, const float* __restrict__ scale,
                                              const float* __restrict__ shift, float* __restrict__ y, int n, int F) {
  const int lane = threadIdx.x & 63, gw = blockIdx.x * 4 + (threadIdx.x >> 6), nw = gridDim.x * 4;
  float4 mu[5], sc[5], sh[5];
#pragma unroll
  for (int j = 0; j < 5; ++j) {
    const int c = (lane + 64 * j) * 4;
    mu[j] = sc[j] = sh[j] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (c < F) {
      mu[j] = *reinterpret_cast<const float4*>(mean + c);
      sc[j] = *reinterpret_cast<const float4*>(scale + c);
      sh[j] = *reinterpret_cast<const float4*>(shift + c);
    }
  }
  for (int base = gw * ROWS; base < n; base += nw * ROWS) {
    float4 v[ROWS][5];
#pragma unroll
    for (int r = 0; r < ROWS; ++r)
#pragma unroll
      for (int j = 0; j < 5; ++j) {
        const int c = (lane + 64 * j) * 4;
        if (c < F && base + r < n) {
          const float4* p = reinterpret_cast<const float4*>(x + (size_t)(base + r) * F + c);
          if (NTL) { const vf4 t = __builtin_nontemporal_load(reinterpret_cast<const vf4*>(p)); v[r][j] = make_float4(t.x, t.y, t.z, t.w); } else v[r][j] = *p;
        }
      }
#pragma unroll
    for (int r = 0; r < ROWS; ++r)
#pragma unroll
      for (int j = 0; j < 5; ++j) {
        const int c = (lane + 64 * j) * 4;
        if (c < F && base + r < n) {
          float4 o;
          o.x = fmaf(fmaxf(v[r][j].x, 0.f) - mu[j].x, sc[j].x, sh[j].x);
          o.y = fmaf(fmaxf(v[r][j].y, 0.f) - mu[j].y, sc[j].y, sh[j].y);
          o.z = fmaf(fmaxf(v[r][j].z, 0.f) - mu[j].z, sc[j].z, sh[j].z);
          o.w = fmaf(fmaxf(v[r][j].w, 0.f) - mu[j].w, sc[j].w, sh[j].w);
          float4* q = reinterpret_cast<float4*>(y + (size_t)(base + r) * F + c);
          if (NTS) { vf4 t = {o.x, o.y, o.z, o.w}; __builtin_nontemporal_store(t, reinterpret_cast<vf4*>(q)); } else *q = o;
        }
      }
  }
}

// flat stream: thread i handles float4 i, i + T, ...; the column constants come from LDS
template <int UNROLL>
__global__ __launch_bounds__(256) void k_flat(const float* __restrict__ x, const float* __restrict__ mean, const float* __restrict__ scale,
                                              const float* __restrict__ shift, float* __restrict__ y, int n, int F) {
  extern __shared__ float4 cst[];       // [3][F/4]
  const int F4 = F >> 2;
  for (int i = threadIdx.x; i < F4; i += 256) {
    cst[i] = reinterpret_cast<const float4*>(mean)[i];
    cst[F4 + i] = reinterpret_cast<const float4*>(scale)[i];
    cst[2 * F4 + i] = reinterpret_cast<const float4*>(shift)[i];
  }
  __syncthreads();
  const long long total = (long long)n * F4, T = (long long)gridDim.x * 256;
  for (long long i0 = (long long)blockIdx.x * 256 * UNROLL + threadIdx.x; i0 < total; i0 += T * UNROLL) {
    float4 v[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      const long long i = i0 + (long long)u * 256;
      if (i < total) v[u] = reinterpret_cast<const float4*>(x)[i];
    }
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      const long long i = i0 + (long long)u * 256;
      if (i < total) {
        const int c = (int)(i % F4);
        const float4 mu = cst[c], sc = cst[F4 + c], sh = cst[2 * F4 + c];
        float4 o;
        o.x = fmaf(fmaxf(v[u].x, 0.f) - mu.x, sc.x, sh.x);
        o.y = fmaf(fmaxf(v[u].y, 0.f) - mu.y, sc.y, sh.y);
        o.z = fmaf(fmaxf(v[u].z, 0.f) - mu.z, sc.z, sh.z);
        o.w = fmaf(fmaxf(v[u].w, 0.f) - mu.w, sc.w, sh.w);
        reinterpret_cast<float4*>(y)[i] = o;
      }
    }
  }
}

// the library kernel's form: unconditional loads from clamped addresses (a chunk past F re-reads the row's last chunk), stores predicated
template <int ROWS, bool NTL>
__global__ __launch_bounds__(256) void k_rows_clamped(const float* __restrict__ x, const float* __restrict__ mean, const float* __restrict__ scale,
                                                      const float* __restrict__ shift, float* __restrict__ y, int n, int F, int ldy) {
  const int lane = threadIdx.x & 63, gw = blockIdx.x * 4 + (threadIdx.x >> 6), nw = gridDim.x * 4;
  float mu[5][4], sc[5][4], sh[5][4];
#pragma unroll
  for (int j = 0; j < 5; ++j)
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      const int c = min((lane + 64 * j) * 4 + v, F - 1);
      mu[j][v] = mean[c]; sc[j][v] = scale[c]; sh[j][v] = shift[c];
    }
  const int last = F - 4;
  for (int base = gw * ROWS; base < n; base += nw * ROWS) {
    float4 v[ROWS][5];
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
      const int row = min(base + r, n - 1);
#pragma unroll
      for (int j = 0; j < 5; ++j) {
        const float4* p = reinterpret_cast<const float4*>(x + (size_t)row * F + min((lane + 64 * j) * 4, last));
        if (NTL) { const vf4 t = __builtin_nontemporal_load(reinterpret_cast<const vf4*>(p)); v[r][j] = make_float4(t.x, t.y, t.z, t.w); } else v[r][j] = *p;
      }
    }
#pragma unroll
    for (int r = 0; r < ROWS; ++r)
#pragma unroll
      for (int j = 0; j < 5; ++j) {
        const int c = (lane + 64 * j) * 4;
        float4 o;
        o.x = fmaf(fmaxf(v[r][j].x, 0.f) - mu[j][0], sc[j][0], sh[j][0]);
        o.y = fmaf(fmaxf(v[r][j].y, 0.f) - mu[j][1], sc[j][1], sh[j][1]);
        o.z = fmaf(fmaxf(v[r][j].z, 0.f) - mu[j][2], sc[j][2], sh[j][2]);
        o.w = fmaf(fmaxf(v[r][j].w, 0.f) - mu[j][3], sc[j][3], sh[j][3]);
        if (c < F && base + r < n) *reinterpret_cast<float4*>(y + (size_t)(base + r) * ldy + c) = o;
      }
  }
}

template <typename L>
static void timeit(const char* name, int blocks, L launch, double mb) {
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) launch(blocks);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  const int reps = 30;
  for (int i = 0; i < reps; ++i) launch(blocks);
  CHECK(hipEventRecord(e1));
  CHECK(hipEventSynchronize(e1));
  float ms;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  printf("%-44s blocks %5d  %7.1f us  %5.2f TB/s\n", name, blocks, ms * 1e3 / reps, 2 * mb / (ms * 1e3 / reps));
}

int main() {
  const int n = 57696, F = 1140;
  float *x, *y, *c;
  CHECK(hipMalloc(&x, (size_t)n * F * 4)); CHECK(hipMalloc(&y, (size_t)n * F * 4)); CHECK(hipMalloc(&c, 3 * F * 4));
  std::vector<float> h((size_t)n * F);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u) % 2001) / 1000.f - 1.f;
  CHECK(hipMemcpy(x, h.data(), h.size() * 4, hipMemcpyHostToDevice));
  CHECK(hipMemset(c, 0, 3 * F * 4));
  const double mb = (double)n * F * 4 / 1e6;
  const float *mean = c, *scale = c + F, *shift = c + 2 * F;
  for (int blocks : {512, 1024, 2048, 4096}) {
    timeit("rows<1> (the library kernel's shape)", blocks, [&](int b) { hipLaunchKernelGGL((k_rows<1, false, false>), dim3(b), dim3(256), 0, 0, x, mean, scale, shift, y, n, F); }, mb);
    timeit("rows<2>", blocks, [&](int b) { hipLaunchKernelGGL((k_rows<2, false, false>), dim3(b), dim3(256), 0, 0, x, mean, scale, shift, y, n, F); }, mb);
    timeit("rows<4>", blocks, [&](int b) { hipLaunchKernelGGL((k_rows<4, false, false>), dim3(b), dim3(256), 0, 0, x, mean, scale, shift, y, n, F); }, mb);
    timeit("rows<2> nontemporal loads", blocks, [&](int b) { hipLaunchKernelGGL((k_rows<2, true, false>), dim3(b), dim3(256), 0, 0, x, mean, scale, shift, y, n, F); }, mb);
    timeit("rows<2> nontemporal loads + stores", blocks, [&](int b) { hipLaunchKernelGGL((k_rows<2, true, true>), dim3(b), dim3(256), 0, 0, x, mean, scale, shift, y, n, F); }, mb);
    timeit("flat<1> constants in LDS", blocks, [&](int b) { hipLaunchKernelGGL((k_flat<1>), dim3(b), dim3(256), 3 * F * 4, 0, x, mean, scale, shift, y, n, F); }, mb);
    timeit("flat<4> constants in LDS", blocks, [&](int b) { hipLaunchKernelGGL((k_flat<4>), dim3(b), dim3(256), 3 * F * 4, 0, x, mean, scale, shift, y, n, F); }, mb);
    timeit("flat<8> constants in LDS", blocks, [&](int b) { hipLaunchKernelGGL((k_flat<8>), dim3(b), dim3(256), 3 * F * 4, 0, x, mean, scale, shift, y, n, F); }, mb);
  }
  for (int blocks : {1024, 2048}) {
    timeit("rows<1> nontemporal loads", blocks, [&](int b) { hipLaunchKernelGGL((k_rows<1, true, false>), dim3(b), dim3(256), 0, 0, x, mean, scale, shift, y, n, F); }, mb);
    timeit("rows<4> nontemporal loads", blocks, [&](int b) { hipLaunchKernelGGL((k_rows<4, true, false>), dim3(b), dim3(256), 0, 0, x, mean, scale, shift, y, n, F); }, mb);
    timeit("clamped rows<1> plain loads", blocks, [&](int b) { hipLaunchKernelGGL((k_rows_clamped<1, false>), dim3(b), dim3(256), 0, 0, x, mean, scale, shift, y, n, F, F); }, mb);
    timeit("clamped rows<1> nontemporal loads", blocks, [&](int b) { hipLaunchKernelGGL((k_rows_clamped<1, true>), dim3(b), dim3(256), 0, 0, x, mean, scale, shift, y, n, F, F); }, mb);
    timeit("clamped rows<2> nontemporal loads", blocks, [&](int b) { hipLaunchKernelGGL((k_rows_clamped<2, true>), dim3(b), dim3(256), 0, 0, x, mean, scale, shift, y, n, F, F); }, mb);
  }
  timeit("flat<4> constants in LDS", 16384, [&](int b) { hipLaunchKernelGGL((k_flat<4>), dim3(b), dim3(256), 3 * F * 4, 0, x, mean, scale, shift, y, n, F); }, mb);
  CHECK(hipMemcpyAsync(y, x, (size_t)n * F * 4, hipMemcpyDeviceToDevice, 0));
  timeit("hipMemcpyAsync d2d", 0, [&](int) { (void)hipMemcpyAsync(y, x, (size_t)n * F * 4, hipMemcpyDeviceToDevice, 0); }, mb);
  return 0;
}
