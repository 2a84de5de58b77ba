// Which vector work hides behind which MFMA on gfx950?  (follow-up of dual_pipe_probe.hip, for the bf16-pair product of bf16x9_probe.hip)
// One 512-thread workgroup per CU = two waves per SIMD.  Every wave does a fixed amount of work of ONE kind; a launch runs either the
// same kind in all eight waves (time = 2 x the kind's per-wave time) or kind A in waves 0-3 and kind B in waves 4-7 (one of each on
// every SIMD).  No overlap: t(A|B) = (t(A|A) + t(B|B)) / 2.  Full overlap: t(A|B) = max(t(A|A), t(B|B)) / 2.
//   kinds: 0 fp32 MFMA 32x32x2   1 bf16 MFMA 32x32x16   2 v_pk_fma_f32   3 v_fma_f32   4 the fp32 -> hi/mid/lo bf16 split   5 ds_read_b128
//   hipcc --offload-arch=gfx950 -O3 tools/pipe_overlap_probe.hip -o tools/pipe_overlap_probe.bin && tools/pipe_overlap_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float float2_ __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

template <int KIND>
__device__ __forceinline__ float work(int iters, int lane, const float* lds) {
  float r = 0.f;
  if constexpr (KIND == 0) {
    floatx16 a0 = {}, a1 = {}, a2 = {}, a3 = {};
    const float x = 1.f + lane * 1e-3f, y = 1.f - lane * 1e-3f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 2; ++u) {                   // 8 x 64 cycles per iteration
        a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, x, a1, 0, 0, 0);
        a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, x, a2, 0, 0, 0);
        a3 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, y, a3, 0, 0, 0);
      }
    }
    r = a0[0] + a1[1] + a2[2] + a3[3];
  } else if constexpr (KIND == 1) {
    floatx16 a0 = {}, a1 = {}, a2 = {}, a3 = {};
    bf16x8 x, y;
    for (int i = 0; i < 8; ++i) { x[i] = (__bf16)(1.f + lane * 1e-3f + i); y[i] = (__bf16)(1.f - lane * 1e-3f); }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {                   // 16 x 32 cycles per iteration
        a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(y, x, a1, 0, 0, 0);
        a2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, x, a2, 0, 0, 0);
        a3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(y, y, a3, 0, 0, 0);
      }
    }
    r = a0[0] + a1[1] + a2[2] + a3[3];
  } else if constexpr (KIND == 2) {
    float2_ acc[32];
    for (int i = 0; i < 32; ++i) acc[i] = (float2_){0.f, 0.f};
    float2_ x = {1.f + lane * 1e-3f, 1.f - lane * 1e-3f}, y = {0.999f, 1.001f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 2; ++u) {                   // 64 packed FMAs per iteration
#pragma unroll
        for (int i = 0; i < 32; ++i) acc[i] = __builtin_elementwise_fma(x, y, acc[i]);
        x = x * y;
      }
    }
    for (int i = 0; i < 32; ++i) r += acc[i][0] + acc[i][1];
  } else if constexpr (KIND == 3) {
    float acc[32];
    for (int i = 0; i < 32; ++i) acc[i] = 0.f;
    float x = 1.f + lane * 1e-3f, y = 0.999f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {                   // 128 scalar FMAs per iteration
#pragma unroll
        for (int i = 0; i < 32; ++i) {
          acc[i] = __builtin_fmaf(x, y, acc[i]);
          asm volatile("" : "+v"(acc[i]));            // (keeps the compiler from pairing them into v_pk_fma_f32)
        }
        x = x * y;
      }
    }
    for (int i = 0; i < 32; ++i) r += acc[i];
  } else if constexpr (KIND == 4) {
    // 16 float2 values per iteration -> hi / mid / lo: 3 v_cvt_pk_bf16_f32, 2 x (2 expands + 1 packed or 2 scalar subtractions) each
    float2_ v[16];
    for (int i = 0; i < 16; ++i) v[i] = (float2_){1.f + lane * 1e-3f + i, 1.f - lane * 1e-3f - i};
    unsigned acc = 0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const bf16x2 h = __builtin_convertvector(v[i], bf16x2);
        const float2_ r1 = v[i] - __builtin_convertvector(h, float2_);
        const bf16x2 m = __builtin_convertvector(r1, bf16x2);
        const float2_ r2 = r1 - __builtin_convertvector(m, float2_);
        const bf16x2 l = __builtin_convertvector(r2, bf16x2);
        acc ^= __builtin_bit_cast(unsigned, h) + __builtin_bit_cast(unsigned, m) + __builtin_bit_cast(unsigned, l);
        v[i] = v[i] * 1.0001f;
      }
    }
    r = (float)acc;
  } else {
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    const float4* p = reinterpret_cast<const float4*>(lds) + lane;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 16; ++u) {                  // 16 ds_read_b128 per iteration
        const float4 t = p[(u * 64 + it) & 1023];
        s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w;
      }
    }
    r = s.x + s.y + s.z + s.w;
  }
  return r;
}

// The same question INSIDE one wave: NF independent vector instructions behind every MFMA (one wave per SIMD).  FK: 0 = v_fma_f32,
// 1 = v_cvt_pk_bf16_f32 + expand + subtract (the split's mix), 2 = v_pk_fma_f32.  BF = bf16 MFMA (32 cycles) or fp32 MFMA (64).
template <bool BF, int NF, int FK>
__global__ __launch_bounds__(256) void k_inwave(float* out, int iters) {
  const int lane = threadIdx.x & 63;
  floatx16 a0 = {}, a1 = {}, a2 = {}, a3 = {};
  bf16x8 xb, yb;
  for (int i = 0; i < 8; ++i) { xb[i] = (__bf16)(1.f + lane * 1e-3f + i); yb[i] = (__bf16)(1.f - lane * 1e-3f); }
  const float xf = 1.f + lane * 1e-3f, yf = 1.f - lane * 1e-3f;
  float f[16];
  float2_ g[16];
  for (int i = 0; i < 16; ++i) { f[i] = i + lane; g[i] = (float2_){(float)i, (float)lane}; }
  unsigned acc = 0;
  auto filler = [&](int slot) {
#pragma unroll
    for (int q = 0; q < NF; ++q) {
      const int i = (slot * NF + q) & 15;
      if constexpr (FK == 0) {
        f[i] = __builtin_fmaf(f[i], 1.0001f, 0.5f);
        asm volatile("" : "+v"(f[i]));
      } else if constexpr (FK == 2) {
        g[i] = __builtin_elementwise_fma(g[i], (float2_){1.0001f, 0.9999f}, (float2_){0.5f, 0.25f});
      } else {
        if ((q & 1) == 0) {
          const bf16x2 h = __builtin_convertvector(g[i], bf16x2);
          acc ^= __builtin_bit_cast(unsigned, h);
          asm volatile("" : "+v"(acc));
        } else {
          g[i] = g[i] - (float2_){__builtin_bit_cast(float, acc << 16), __builtin_bit_cast(float, acc & 0xffff0000u)};
        }
      }
    }
  };
  auto mm = [&](floatx16& a, int which) {
    if constexpr (BF) a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(which & 1 ? yb : xb, which & 2 ? yb : xb, a, 0, 0, 0);
    else a = __builtin_amdgcn_mfma_f32_32x32x2f32(which & 1 ? yf : xf, which & 2 ? yf : xf, a, 0, 0, 0);
  };
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < (BF ? 4 : 2); ++u) {
      mm(a0, 0); filler(4 * u); __builtin_amdgcn_sched_barrier(0);
      mm(a1, 1); filler(4 * u + 1); __builtin_amdgcn_sched_barrier(0);
      mm(a2, 2); filler(4 * u + 2); __builtin_amdgcn_sched_barrier(0);
      mm(a3, 3); filler(4 * u + 3); __builtin_amdgcn_sched_barrier(0);
    }
  }
  float r = a0[0] + a1[1] + a2[2] + a3[3] + (float)acc;
  for (int i = 0; i < 16; ++i) r += f[i] + g[i][0] + g[i][1];
  if (r == 12345.678f) out[blockIdx.x * 256 + threadIdx.x] = r;
}

template <bool BF, int NF, int FK>
static float run_inwave(float* out, int iters) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  hipLaunchKernelGGL((k_inwave<BF, NF, FK>), dim3(256), dim3(256), 0, 0, out, iters);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL((k_inwave<BF, NF, FK>), dim3(256), dim3(256), 0, 0, out, iters);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms;
  (void)hipEventElapsedTime(&ms, e0, e1);
  return ms;
}

template <bool BF, int FK>
static void inwave_row(float* out, int iters, const char* fname) {
  const float t0 = run_inwave<BF, 0, FK>(out, iters);
  const float t[6] = {run_inwave<BF, 1, FK>(out, iters), run_inwave<BF, 2, FK>(out, iters), run_inwave<BF, 3, FK>(out, iters),
                      run_inwave<BF, 4, FK>(out, iters), run_inwave<BF, 6, FK>(out, iters), run_inwave<BF, 8, FK>(out, iters)};
  const double cyc = (BF ? 32.0 : 64.0) / t0;      // cycles per ms of this launch, from the MFMA-only time
  printf("%s + %-22s per MFMA: 0 fillers %6.3f ms (= %d cycles);  1: %5.1f  2: %5.1f  3: %5.1f  4: %5.1f  6: %5.1f  8: %5.1f cycles per MFMA\n",
         BF ? "bf16 MFMA" : "fp32 MFMA", fname, t0, BF ? 32 : 64, t[0] * cyc, t[1] * cyc, t[2] * cyc, t[3] * cyc, t[4] * cyc, t[5] * cyc);
}

template <int KA, int KB>
__global__ __launch_bounds__(512) void k_probe(float* out, int iters) {
  __shared__ float lds[4096 + 64 * 4];
  for (int i = threadIdx.x; i < 4096 + 256; i += 512) lds[i] = (float)i;
  __syncthreads();
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  float r;
  if (wave < 4) r = work<KA>(iters, lane, lds);
  else r = work<KB>(iters, lane, lds);
  if (r == 12345.678f) out[blockIdx.x * 512 + threadIdx.x] = r;
}

template <int KA, int KB>
static float run(float* out, int iters) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  hipLaunchKernelGGL((k_probe<KA, KB>), dim3(256), dim3(512), 0, 0, out, iters);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL((k_probe<KA, KB>), dim3(256), dim3(512), 0, 0, out, iters);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms;
  (void)hipEventElapsedTime(&ms, e0, e1);
  return ms;
}

static const char* NAMES[6] = {"fp32 MFMA 32x32x2", "bf16 MFMA 32x32x16", "v_pk_fma_f32", "v_fma_f32", "fp32 -> 3 bf16 split", "ds_read_b128"};

template <int KA, int KB>
static void pair(float* out, int iters) {
  const float aa = run<KA, KA>(out, iters), bb = run<KB, KB>(out, iters), ab = run<KA, KB>(out, iters);
  const float none = 0.5f * (aa + bb), full = 0.5f * (aa > bb ? aa : bb);
  printf("%-20s | %-20s : alone %7.3f / %7.3f ms   together %7.3f ms   (no overlap %7.3f, full overlap %7.3f)  -> %3.0f %% hidden\n", NAMES[KA],
         NAMES[KB], aa, bb, ab, none, full, none > full ? 100.f * (none - ab) / (none - full) : 0.f);
}

int main() {
  float* out;
  (void)hipMalloc(&out, 1 << 20);
  const int iters = 20000;
  pair<0, 2>(out, iters);
  pair<0, 3>(out, iters);
  pair<0, 4>(out, iters);
  pair<0, 5>(out, iters);
  pair<1, 2>(out, iters);
  pair<1, 3>(out, iters);
  pair<1, 4>(out, iters);
  pair<1, 5>(out, iters);
  pair<0, 1>(out, iters);
  inwave_row<true, 0>(out, iters, "v_fma_f32");
  inwave_row<true, 1>(out, iters, "cvt_pk / subtract mix");
  inwave_row<true, 2>(out, iters, "v_pk_fma_f32");
  inwave_row<false, 0>(out, iters, "v_fma_f32");
  inwave_row<false, 1>(out, iters, "cvt_pk / subtract mix");
  inwave_row<false, 2>(out, iters, "v_pk_fma_f32");
  return 0;
}
