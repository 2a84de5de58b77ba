// Row-wise kernels of the conv epilogue, BatchNorm, assignment softmax, max readout and the dense
// adjacency transforms.  All are HBM-bandwidth-bound streaming kernels:
//   * lanes of a "row group" (8..64 lanes, common.hpp) read a row with 16-byte loads when the layout allows,
//   * row reductions are wavefront shuffles, column reductions are per-lane register accumulators combined
//     deterministically (wave -> LDS -> per-block slot -> second-stage kernel): no float atomics anywhere.
// Reference arithmetic: see the citations in include/cgc_hip.h.
#include <type_traits>

#include "common.hpp"
#include "groups.hpp"

#define L2_EPS 1e-12f
#define RENORM_EPS 1e-15f
#define MAX_SLOTS 1024

// ------------------------------------------------------------------------------------------------
// column accumulators: block-level combine + second stage
// ------------------------------------------------------------------------------------------------
template <int VEC, int MAXJ, int NQ>
__device__ __forceinline__ void col_reduce_store(float (&acc)[NQ][MAXJ][VEC], int F, int lpr, float* smem, float* slot) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int q = 0; q < NQ; ++q)
#pragma unroll
    for (int j = 0; j < MAXJ; ++j)
#pragma unroll
      for (int v = 0; v < VEC; ++v)
        for (int o = 32; o >= lpr; o >>= 1) acc[q][j][v] += __shfl_xor(acc[q][j][v], o);
  if (wave > 0 && lane < lpr) {
#pragma unroll
    for (int q = 0; q < NQ; ++q)
#pragma unroll
      for (int j = 0; j < MAXJ; ++j)
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
          const int c = (lane + lpr * j) * VEC + v;
          if (c < F) smem[((wave - 1) * NQ + q) * F + c] = acc[q][j][v];
        }
  }
  __syncthreads();
  if (wave == 0 && lane < lpr) {
#pragma unroll
    for (int q = 0; q < NQ; ++q)
#pragma unroll
      for (int j = 0; j < MAXJ; ++j)
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
          const int c = (lane + lpr * j) * VEC + v;
          if (c < F) {
            float t = acc[q][j][v];
            for (int w = 0; w < 3; ++w) t += smem[(w * NQ + q) * F + c];
            slot[q * F + c] = t;
          }
        }
  }
}

// The forward BatchNorm statistics (sum of o, sum of o^2 per column) are accumulated in DOUBLE from the first addition on: the
// variance is formed as E[o^2] - E[o]^2, and on the coarsened levels the rows of a batch are nearly identical (std << |mean|), so
// partial sums carried in fp32 -- 1e-7 relative each -- left the variance with 1e-7 * mean^2 / var of relative error: 3e-4 on the
// gradients of a level-2 block of the reference-generated `tiny_shipped` fixture (tools/parity_bisect.py), 20x what a one-ulp
// perturbation of the parameters produces.  o and o^2 are exact in double; slots hold doubles (4F floats of workspace per slot).
// One quantity at a time through LDS (3 F doubles).
template <int VEC, int MAXJ>
__device__ __forceinline__ void col_reduce_store_f64(double (&acc)[2][MAXJ][VEC], int F, int lpr, double* smem, double* slot) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int q = 0; q < 2; ++q) {
#pragma unroll
    for (int j = 0; j < MAXJ; ++j)
#pragma unroll
      for (int v = 0; v < VEC; ++v)
        for (int o = 32; o >= lpr; o >>= 1) acc[q][j][v] += __shfl_xor(acc[q][j][v], o);
    if (q > 0) __syncthreads();
    if (wave > 0 && lane < lpr) {
#pragma unroll
      for (int j = 0; j < MAXJ; ++j)
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
          const int c = (lane + lpr * j) * VEC + v;
          if (c < F) smem[(wave - 1) * F + c] = acc[q][j][v];
        }
    }
    __syncthreads();
    if (wave == 0 && lane < lpr) {
#pragma unroll
      for (int j = 0; j < MAXJ; ++j)
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
          const int c = (lane + lpr * j) * VEC + v;
          if (c < F) {
            double t = acc[q][j][v];
            for (int w = 0; w < 3; ++w) t += smem[w * F + c];
            slot[q * F + c] = t;
          }
        }
    }
  }
}

// out[c] = sum_s ws[s*width + c]   (fixed summation order: deterministic; accumulated in fp64 so that the
// BatchNorm variance E[x^2] - E[x]^2 formed from these sums keeps ~1e-7 accuracy even when |mean| >> std).
// 1024 threads = 32 slot groups x 32 columns, 4 independent loads in flight per thread: the kernel is a chain of dependent
// load rounds (slots / 128 of them; it was slots / 64 with 8 groups) and little else -- 6 us -> 4 us for the ~450-slot
// reductions of the narrow layers, of which a step has ~45.
#define RS_GROUPS 32
template <typename OUT, typename IN = float>
__global__ __launch_bounds__(32 * RS_GROUPS) void k_reduce_slots(const IN* __restrict__ ws, int slots, int width, OUT* __restrict__ out) {
  __shared__ double part[RS_GROUPS][32];
  const int cl = threadIdx.x & 31, grp = threadIdx.x >> 5;
  const int c = blockIdx.x * 32 + cl;
  double s = 0.0;
  if (c < width) {
    double a[4] = {0.0, 0.0, 0.0, 0.0};
    int k = grp;
    for (; k + 3 * RS_GROUPS < slots; k += 4 * RS_GROUPS) {
#pragma unroll
      for (int u = 0; u < 4; ++u) a[u] += (double)ws[(size_t)(k + RS_GROUPS * u) * width + c];
    }
    for (; k < slots; k += RS_GROUPS) a[0] += (double)ws[(size_t)k * width + c];
    s = (a[0] + a[1]) + (a[2] + a[3]);
  }
  part[grp][cl] = s;
  __syncthreads();
  if (grp == 0 && c < width) {
    double t[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int g = 0; g < RS_GROUPS; g += 4) { t[0] += part[g][cl]; t[1] += part[g + 1][cl]; t[2] += part[g + 2][cl]; t[3] += part[g + 3][cl]; }
    out[c] = (OUT)((t[0] + t[1]) + (t[2] + t[3]));
  }
}
#define REDUCE_SLOTS_GRID(width) dim3(ceil_div((width), 32))

// the same for two slot areas in one launch (blockIdx.y picks the pair)
__global__ __launch_bounds__(32 * RS_GROUPS) void k_reduce_slots_pair(const float* __restrict__ ws0, float* __restrict__ out0,
                                                                     const float* __restrict__ ws1, float* __restrict__ out1, int slots, int width) {
  const float* __restrict__ ws = blockIdx.y ? ws1 : ws0;
  float* __restrict__ out = blockIdx.y ? out1 : out0;
  __shared__ double part[RS_GROUPS][32];
  const int cl = threadIdx.x & 31, grp = threadIdx.x >> 5;
  const int c = blockIdx.x * 32 + cl;
  double s = 0.0;
  if (c < width) {
    double a[4] = {0.0, 0.0, 0.0, 0.0};
    int k = grp;
    for (; k + 3 * RS_GROUPS < slots; k += 4 * RS_GROUPS) {
#pragma unroll
      for (int u = 0; u < 4; ++u) a[u] += (double)ws[(size_t)(k + RS_GROUPS * u) * width + c];
    }
    for (; k < slots; k += RS_GROUPS) a[0] += (double)ws[(size_t)k * width + c];
    s = (a[0] + a[1]) + (a[2] + a[3]);
  }
  part[grp][cl] = s;
  __syncthreads();
  if (grp == 0 && c < width) {
    double t[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int g = 0; g < RS_GROUPS; g += 4) { t[0] += part[g][cl]; t[1] += part[g + 1][cl]; t[2] += part[g + 2][cl]; t[3] += part[g + 3][cl]; }
    out[c] = (float)((t[0] + t[1]) + (t[2] + t[3]));
  }
}
int launch_reduce_slots_f32_pair(const float* ws0, float* out0, const float* ws1, float* out1, int ng, int slots, int width, hipStream_t stream) {
  hipLaunchKernelGGL(k_reduce_slots_pair, dim3(ceil_div(width, 32), ng), dim3(32 * RS_GROUPS), 0, stream, ws0, out0, ws1, out1, slots, width);
  CGC_RETURN_IF_LAUNCH_FAILED();
  return 0;
}

int launch_reduce_slots_f32(const float* ws, int slots, int width, float* out, hipStream_t stream) {
  hipLaunchKernelGGL(k_reduce_slots<float>, REDUCE_SLOTS_GRID(width), dim3(32 * RS_GROUPS), 0, stream, ws, slots, width, out);
  CGC_RETURN_IF_LAUNCH_FAILED();
  return 0;
}

struct ColCfg {
  int vec, maxj, lpr, blocks;
  bool ok;
};
static ColCfg col_cfg(int n, int F, bool vec_ok) {
  ColCfg c;
  c.vec = vec_ok ? 4 : 1;
  c.ok = true;
  int chunks = F / c.vec;
  if (chunks <= 64) {
    c.maxj = 1;
    c.lpr = pick_lpr(chunks);
  } else {
    c.lpr = 64;
    int need = ceil_div(chunks, 64);           // per-lane chunks; instantiated: 16-byte lanes up to 8, scalar lanes up to 32
    if (c.vec == 4 && need > 8) {              // rows wider than 2048 floats: scalar lanes
      c.vec = 1;
      need = ceil_div(F, 64);
    }
    c.ok = need <= 32;                         // F <= 2048 (covers the reference's cluster counts 1140 / 1600)
    c.maxj = need <= 2 ? 2 : need <= 4 ? 4 : (need == 5 && c.vec == 4) ? 5 : need <= 8 ? 8 : need <= 16 ? 16 : 32;   // 5: F = 1140
  }
  c.blocks = row_blocks(n, c.lpr, MAX_SLOTS);
  return c;
}

#define DISPATCH_COL(KERNEL, cfg, smem_bytes, stream, ...)                                                   \
  do {                                                                                                       \
    dim3 g__((cfg).blocks), b__(CGC_BLOCK);                                                                  \
    if ((cfg).vec == 4) {                                                                                    \
      switch ((cfg).maxj) {                                                                                  \
        case 1: hipLaunchKernelGGL((KERNEL<4, 1>), g__, b__, smem_bytes, stream, __VA_ARGS__); break;        \
        case 2: hipLaunchKernelGGL((KERNEL<4, 2>), g__, b__, smem_bytes, stream, __VA_ARGS__); break;        \
        case 4: hipLaunchKernelGGL((KERNEL<4, 4>), g__, b__, smem_bytes, stream, __VA_ARGS__); break;        \
        case 5: hipLaunchKernelGGL((KERNEL<4, 5>), g__, b__, smem_bytes, stream, __VA_ARGS__); break;        \
        default: hipLaunchKernelGGL((KERNEL<4, 8>), g__, b__, smem_bytes, stream, __VA_ARGS__); break;       \
      }                                                                                                      \
    } else {                                                                                                 \
      switch ((cfg).maxj) {                                                                                  \
        case 1: hipLaunchKernelGGL((KERNEL<1, 1>), g__, b__, smem_bytes, stream, __VA_ARGS__); break;        \
        case 2: hipLaunchKernelGGL((KERNEL<1, 2>), g__, b__, smem_bytes, stream, __VA_ARGS__); break;        \
        case 4: hipLaunchKernelGGL((KERNEL<1, 4>), g__, b__, smem_bytes, stream, __VA_ARGS__); break;        \
        case 8: hipLaunchKernelGGL((KERNEL<1, 8>), g__, b__, smem_bytes, stream, __VA_ARGS__); break;        \
        case 16: hipLaunchKernelGGL((KERNEL<1, 16>), g__, b__, smem_bytes, stream, __VA_ARGS__); break;      \
        default: hipLaunchKernelGGL((KERNEL<1, 32>), g__, b__, smem_bytes, stream, __VA_ARGS__); break;      \
      }                                                                                                      \
    }                                                                                                        \
  } while (0)

#define DISPATCH_COL_Y(KERNEL, cfg, ny, smem_bytes, stream, ...)                                                   \
  do {                                                                                                       \
    dim3 g__((cfg).blocks, (ny)), b__(CGC_BLOCK);                                                                    \
    if ((cfg).vec == 4) {                                                                                    \
      switch ((cfg).maxj) {                                                                                  \
        case 1: hipLaunchKernelGGL((KERNEL<4, 1>), g__, b__, smem_bytes, stream, __VA_ARGS__); break;        \
        case 2: hipLaunchKernelGGL((KERNEL<4, 2>), g__, b__, smem_bytes, stream, __VA_ARGS__); break;        \
        case 4: hipLaunchKernelGGL((KERNEL<4, 4>), g__, b__, smem_bytes, stream, __VA_ARGS__); break;        \
        case 5: hipLaunchKernelGGL((KERNEL<4, 5>), g__, b__, smem_bytes, stream, __VA_ARGS__); break;        \
        default: hipLaunchKernelGGL((KERNEL<4, 8>), g__, b__, smem_bytes, stream, __VA_ARGS__); break;       \
      }                                                                                                      \
    } else {                                                                                                 \
      switch ((cfg).maxj) {                                                                                  \
        case 1: hipLaunchKernelGGL((KERNEL<1, 1>), g__, b__, smem_bytes, stream, __VA_ARGS__); break;        \
        case 2: hipLaunchKernelGGL((KERNEL<1, 2>), g__, b__, smem_bytes, stream, __VA_ARGS__); break;        \
        case 4: hipLaunchKernelGGL((KERNEL<1, 4>), g__, b__, smem_bytes, stream, __VA_ARGS__); break;        \
        case 8: hipLaunchKernelGGL((KERNEL<1, 8>), g__, b__, smem_bytes, stream, __VA_ARGS__); break;        \
        case 16: hipLaunchKernelGGL((KERNEL<1, 16>), g__, b__, smem_bytes, stream, __VA_ARGS__); break;      \
        default: hipLaunchKernelGGL((KERNEL<1, 32>), g__, b__, smem_bytes, stream, __VA_ARGS__); break;      \
      }                                                                                                      \
    }                                                                                                        \
  } while (0)


extern "C" int cgc_stats_blocks(int n, int F) {
  (void)F;
  int b = ceil_div(n > 0 ? n : 1, 4);
  return b < MAX_SLOTS ? b : MAX_SLOTS;
}
// floats of the workspace behind the forward statistics (cgc_l2norm_act_stats / cgc_l2norm_act_bn / cgc_sage_wide_fwd /
// cgc_sage_narrow_fwd): one slot of 2F DOUBLES per partial sum (+ 4F + 2 spare floats)
extern "C" int64_t cgc_stats_ws_floats(int n, int F) {
  const int b = cgc_stats_blocks(n, F);
  return (int64_t)(b > 1 ? b : 1) * 4 * F + 4 * (int64_t)F + 2;
}

// ------------------------------------------------------------------------------------------------
// l2norm (+ activation statistics for BatchNorm)
// ------------------------------------------------------------------------------------------------
template <int VEC, int MAXJ>
__global__ __launch_bounds__(256) void k_l2norm_act_stats(const float* __restrict__ h, int n, int F, int lpr, int normalize,
                                                          int act, float* __restrict__ hn, float* __restrict__ rinv,
                                                          float* __restrict__ ws) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const RowGroup rg(lpr);
  double acc[2][MAXJ][VEC];          // see col_reduce_store_f64
#pragma unroll
  for (int j = 0; j < MAXJ; ++j)
#pragma unroll
    for (int v = 0; v < VEC; ++v) acc[0][j][v] = acc[1][j][v] = 0.0;

  for (int base = rg.gwave * rg.rpw; base < n; base += rg.nwaves * rg.rpw) {
    const int row = base + rg.sub;
    const bool valid = row < n;
    Vec<VEC> x[MAXJ];
    float ss = 0.f;
#pragma unroll
    for (int j = 0; j < MAXJ; ++j) {
      const int c = (rg.sl + lpr * j) * VEC;
      if (valid && c < F) {
        load_wide<VEC, MAXJ>(x[j], h + (size_t)row * F + c);
#pragma unroll
        for (int v = 0; v < VEC; ++v) ss += x[j].v[v] * x[j].v[v];
      } else {
#pragma unroll
        for (int v = 0; v < VEC; ++v) x[j].v[v] = 0.f;
      }
    }
    ss = group_sum(ss, lpr);
    const float r = normalize ? 1.f / fmaxf(sqrtf(ss), L2_EPS) : 1.f;
#pragma unroll
    for (int j = 0; j < MAXJ; ++j) {
      const int c = (rg.sl + lpr * j) * VEC;
      if (valid && c < F) {
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
          x[j].v[v] *= r;
          const double o = (double)act_fwd(x[j].v[v], act);
          acc[0][j][v] += o;
          acc[1][j][v] = fma(o, o, acc[1][j][v]);
        }
        x[j].store(hn + (size_t)row * F + c);
      }
    }
    if (valid && rg.sl == 0) rinv[row] = r;
  }
  if (ws != nullptr)
    col_reduce_store_f64<VEC, MAXJ>(acc, F, lpr, reinterpret_cast<double*>(smem), reinterpret_cast<double*>(ws) + (size_t)blockIdx.x * 2 * F);
}

extern "C" int cgc_l2norm_act_stats(const float* h, int n, int F, int normalize, int act, float* hn, float* rinv,
                                    double* stats, float* ws, cgc_stream_t stream) {
  if (n <= 0 || F <= 0) {
    if (stats && F > 0) (void)hipMemsetAsync(stats, 0, sizeof(double) * 2 * F, as_stream(stream));
    return 0;
  }
  const bool vec_ok = (F % 4 == 0) && aligned16(h) && aligned16(hn);
  ColCfg cfg = col_cfg(n, F, vec_ok);
  if (!cfg.ok) return CGC_EINVAL;
  float* wsp = stats ? ws : nullptr;
  if (stats && !ws) return CGC_EINVAL;
  const size_t smem = sizeof(float) * 3 * 2 * F;
  DISPATCH_COL(k_l2norm_act_stats, cfg, smem, as_stream(stream), h, n, F, cfg.lpr, normalize, act, hn, rinv, wsp);
  CGC_RETURN_IF_LAUNCH_FAILED();
  if (stats) {
    hipLaunchKernelGGL((k_reduce_slots<double, double>), REDUCE_SLOTS_GRID(2 * F), dim3(32 * RS_GROUPS), 0, as_stream(stream),
                       reinterpret_cast<const double*>(ws), cfg.blocks, 2 * F, stats);
    CGC_RETURN_IF_LAUNCH_FAILED();
  }
  return 0;
}

// ------------------------------------------------------------------------------------------------
// BatchNorm statistics -> mean / inverse std (+ running statistics)
// ------------------------------------------------------------------------------------------------
__global__ void k_bn_finalize(const double* __restrict__ stats, int F, double count, float eps, float momentum,
                              float* running_mean, float* running_var, float* __restrict__ mean, float* __restrict__ istd,
                              long long* nbt) {
  const int f = blockIdx.x * blockDim.x + threadIdx.x;
  if (f == 0 && nbt != nullptr) *nbt += 1;       // nn.BatchNorm1d's num_batches_tracked
  if (f >= F) return;
  const double m = stats[f] / count;
  double var = stats[F + f] / count - m * m;   // biased; the padded zero rows are part of `count`
  if (var < 0.0) var = 0.0;
  mean[f] = (float)m;
  istd[f] = (float)(1.0 / sqrt(var + (double)eps));
  if (running_mean != nullptr) {
    const double unbiased = var * count / (count > 1.0 ? count - 1.0 : 1.0);
    running_mean[f] = (1.f - momentum) * running_mean[f] + momentum * (float)m;
    running_var[f] = (1.f - momentum) * running_var[f] + momentum * (float)unbiased;
  }
}

extern "C" int cgc_bn_finalize(const double* stats, int F, double count, float eps, float momentum, float* running_mean,
                               float* running_var, float* mean, float* istd, cgc_stream_t stream) {
  if (F <= 0) return 0;
  hipLaunchKernelGGL(k_bn_finalize, dim3(ceil_div(F, 256)), dim3(256), 0, as_stream(stream), stats, F, count, eps, momentum,
                     running_mean, running_var, mean, istd, (long long*)nullptr);
  CGC_RETURN_IF_LAUNCH_FAILED();
  return 0;
}

__global__ void k_bn_running_stats(const float* __restrict__ rm, const float* __restrict__ rv, int F, float eps, float* __restrict__ mean,
                                   float* __restrict__ istd) {
  const int f = blockIdx.x * blockDim.x + threadIdx.x;
  if (f >= F) return;
  mean[f] = rm[f];
  istd[f] = 1.f / sqrtf(rv[f] + eps);
}
extern "C" int cgc_bn_running_stats(const float* running_mean, const float* running_var, int F, float eps, float* mean, float* istd,
                                    cgc_stream_t stream) {
  if (F <= 0) return 0;
  if (running_mean == nullptr || running_var == nullptr || mean == nullptr || istd == nullptr) return CGC_EINVAL;
  hipLaunchKernelGGL(k_bn_running_stats, dim3(ceil_div(F, 256)), dim3(256), 0, as_stream(stream), running_mean, running_var, F, eps, mean, istd);
  CGC_RETURN_IF_LAUNCH_FAILED();
  return 0;
}

// second stage of the statistics + finalize in one kernel: the column sums of the slots (same grouping and order as
// k_reduce_slots<double>, so the same bits) and mean / istd / running statistics / num_batches_tracked of k_bn_finalize
__global__ __launch_bounds__(32 * RS_GROUPS) void k_stats_finalize(const StatsFinPtrs p0, const StatsFinPtrs p1, int slots, int F, double count) {
  const StatsFinPtrs& p = blockIdx.y ? p1 : p0;
  const double* __restrict__ ws = reinterpret_cast<const double*>(p.ws);      // slots of doubles: [sum o | sum o^2] per slot
  const float eps = p.eps, momentum = p.momentum;
  float* running_mean = p.running_mean;
  float* running_var = p.running_var;
  float* __restrict__ mean = p.mean;
  float* __restrict__ istd = p.istd;
  long long* nbt = p.nbt;
  __shared__ double part[2][RS_GROUPS][32];
  const int cl = threadIdx.x & 31, grp = threadIdx.x >> 5;
  const int f = blockIdx.x * 32 + cl;
  if (blockIdx.x == 0 && threadIdx.x == 0 && nbt != nullptr) *nbt += 1;       // nn.BatchNorm1d's num_batches_tracked
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    double s = 0.0;
    if (f < F) {
      const double* col = ws + (size_t)q * F + f;
      const size_t width = 2 * (size_t)F;
      double a[4] = {0.0, 0.0, 0.0, 0.0};
      int k = grp;
      for (; k + 3 * RS_GROUPS < slots; k += 4 * RS_GROUPS) {
#pragma unroll
        for (int u = 0; u < 4; ++u) a[u] += col[(size_t)(k + RS_GROUPS * u) * width];
      }
      for (; k < slots; k += RS_GROUPS) a[0] += col[(size_t)k * width];
      s = (a[0] + a[1]) + (a[2] + a[3]);
    }
    part[q][grp][cl] = s;
  }
  __syncthreads();
  if (grp != 0 || f >= F) return;
  double t0[4] = {0.0, 0.0, 0.0, 0.0}, t1[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int g = 0; g < RS_GROUPS; g += 4)
#pragma unroll
    for (int u = 0; u < 4; ++u) { t0[u] += part[0][g + u][cl]; t1[u] += part[1][g + u][cl]; }
  const double s0 = (t0[0] + t0[1]) + (t0[2] + t0[3]), s1 = (t1[0] + t1[1]) + (t1[2] + t1[3]);
  const double m = s0 / count;
  double var = s1 / count - m * m;             // biased; the padded zero rows are part of `count`
  if (var < 0.0) var = 0.0;
  mean[f] = (float)m;
  istd[f] = (float)(1.0 / sqrt(var + (double)eps));
  if (running_mean != nullptr) {
    const double unbiased = var * count / (count > 1.0 ? count - 1.0 : 1.0);
    running_mean[f] = (1.f - momentum) * running_mean[f] + momentum * (float)m;
    running_var[f] = (1.f - momentum) * running_var[f] + momentum * (float)unbiased;
  }
}

int launch_stats_finalize_groups(const StatsFinPtrs* g, int ng, int slots, int F, double count, hipStream_t stream) {
  hipLaunchKernelGGL(k_stats_finalize, dim3(ceil_div(F, 32), ng), dim3(32 * RS_GROUPS), 0, stream, g[0], g[ng - 1], slots, F, count);
  CGC_RETURN_IF_LAUNCH_FAILED();
  return 0;
}
int launch_stats_finalize(const float* ws, int slots, int F, double count, float eps, float momentum, float* running_mean,
                          float* running_var, float* mean, float* istd, int64_t* nbt, hipStream_t stream) {
  const StatsFinPtrs g{ws, running_mean, running_var, mean, istd, reinterpret_cast<long long*>(nbt), eps, momentum};
  return launch_stats_finalize_groups(&g, 1, slots, F, count, stream);
}

// The training forward's statistics as ONE call: l2norm + activation sums, then second stage + finalize + running statistics +
// num_batches_tracked += 1 in one kernel (two launches; one host call instead of three plus torch's counter increment).
// ws: cgc_stats_ws_floats(n, F) floats (slots of doubles), 8-byte aligned.
extern "C" int cgc_l2norm_act_bn(const float* h, int n, int F, int normalize, int act, float* hn, float* rinv, float* ws,
                                 double count, float eps, float momentum, float* running_mean, float* running_var,
                                 int64_t* num_batches_tracked, float* mean, float* istd, cgc_stream_t stream) {
  if (F <= 0) return 0;
  if (ws == nullptr || mean == nullptr || istd == nullptr) return CGC_EINVAL;
  int slots = 0;
  if (n > 0) {
    const bool vec_ok = (F % 4 == 0) && aligned16(h) && aligned16(hn);
    ColCfg cfg = col_cfg(n, F, vec_ok);
    if (!cfg.ok) return CGC_EINVAL;
    const size_t smem = sizeof(float) * 3 * 2 * F;
    DISPATCH_COL(k_l2norm_act_stats, cfg, smem, as_stream(stream), h, n, F, cfg.lpr, normalize, act, hn, rinv, ws);
    CGC_RETURN_IF_LAUNCH_FAILED();
    slots = cfg.blocks;
  }
  return launch_stats_finalize(ws, slots, F, count, eps, momentum, running_mean, running_var, mean, istd, num_batches_tracked, as_stream(stream));
}

// ------------------------------------------------------------------------------------------------
// y = BN(act(hn))   (elementwise; y may be a column slice of a wider buffer: ldy)
// ------------------------------------------------------------------------------------------------
#ifndef CGC_BNAPPLY_UR
#define CGC_BNAPPLY_UR 1
#endif
template <int VEC, int MAXJ>
__global__ __launch_bounds__(256) void k_bn_act_apply(const BnApplyPtrs p0, const BnApplyPtrs p1, int n, int F, int lpr, int act, int ldy) {
  const BnApplyPtrs& p = blockIdx.y ? p1 : p0;
  const float* __restrict__ hn = p.hn;
  const float* __restrict__ mean = p.mean;
  const float* __restrict__ istd = p.istd;
  const float* __restrict__ gamma = p.gamma;
  const float* __restrict__ beta = p.beta;
  float* __restrict__ y = p.y;
  float* __restrict__ y2 = p.y2;
  const int ldy2 = p.ldy2;
  const RowGroup rg(lpr);
  // y = (act(hn) - mu) * sc + sh with sc = istd*gamma: the lane's columns never change, so the constants live in
  // registers.  (o - mu) is formed first, as nn.BatchNorm1d does: folding mu into the shift would cancel catastrophically
  // when |mean| >> std, e.g. after the un-normalised GIN convolution.)
  float mu[MAXJ][VEC], sc[MAXJ][VEC], sh[MAXJ][VEC];
#pragma unroll
  for (int j = 0; j < MAXJ; ++j)
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
      mu[j][v] = 0.f;
      sc[j][v] = 1.f;
      sh[j][v] = 0.f;
    }
  if (mean != nullptr) {
    // branch-free: a clamped index instead of a predicate per constant (20 conditional blocks, each with its own wait for four
    // dependent-latency loads, were 15-20 us at the start of every wave); the values of lanes past F are never used
#pragma unroll
    for (int j = 0; j < MAXJ; ++j)
#pragma unroll
      for (int v = 0; v < VEC; ++v) {
        const int c = min((rg.sl + lpr * j) * VEC + v, F - 1);
        mu[j][v] = mean[c];
        sc[j][v] = istd[c] * gamma[c];
        sh[j][v] = beta[c];
      }
  }
  // The activation code is a run-time argument; as a switch inside the element loop it splits every chunk into its own basic blocks
  // (load -> wait -> branch -> compute -> store, one chunk at a time: 118 us on [57.7k, 1140]).  Here the switch is outside the row
  // loop, the loads of a row are unconditional (a chunk past F re-reads the row's last chunk) and all requested before the first
  // value is used; only the stores are predicated.
  auto rows = [&](auto act_c) {
    constexpr int ACT = decltype(act_c)::value;
    const int last = F - VEC;                            // (F >= VEC: F % VEC == 0 and F > 0)
    constexpr int UR = (VEC == 4 && MAXJ == 5) ? CGC_BNAPPLY_UR : 1;      // adjacent rows per pass (-DCGC_BNAPPLY_UR; 1: 96 us, 2: 103, 4: 91 on [57.7k, 1140])
    for (int base = rg.gwave * rg.rpw * UR; base < n; base += rg.nwaves * rg.rpw * UR) {
      Vec<VEC> x[UR][MAXJ];
#pragma unroll
      for (int u = 0; u < UR; ++u) {
        const int row = min(base + u * rg.rpw + rg.sub, n - 1);
#pragma unroll
        for (int j = 0; j < MAXJ; ++j) load_wide<VEC, MAXJ>(x[u][j], hn + (size_t)row * F + min((rg.sl + lpr * j) * VEC, last));
      }
#pragma unroll
      for (int u = 0; u < UR; ++u) {
        const int row = base + u * rg.rpw + rg.sub;
        const bool rowok = row < n;
#pragma unroll
        for (int j = 0; j < MAXJ; ++j) {
          const int c = (rg.sl + lpr * j) * VEC;
#pragma unroll
          for (int v = 0; v < VEC; ++v) x[u][j].v[v] = fmaf(act_fwd(x[u][j].v[v], ACT) - mu[j][v], sc[j][v], sh[j][v]);
          if (rowok && c < F) {
            x[u][j].store(y + (size_t)row * ldy + c);
            if (y2 != nullptr) x[u][j].store(y2 + (size_t)row * ldy2 + c);
          }
        }
      }
    }
  };
  switch (act) {
    case CGC_ACT_RELU: rows(std::integral_constant<int, CGC_ACT_RELU>()); break;
    case CGC_ACT_ELU: rows(std::integral_constant<int, CGC_ACT_ELU>()); break;
    case CGC_ACT_LEAKYRELU: rows(std::integral_constant<int, CGC_ACT_LEAKYRELU>()); break;
    default: rows(std::integral_constant<int, CGC_ACT_IDENTITY>()); break;
  }
}

extern "C" int cgc_bn_act_apply(const float* hn, int n, int F, int act, const float* mean, const float* istd,
                                const float* gamma, const float* beta, float* y, int ldy, cgc_stream_t stream) {
  return cgc_bn_act_apply2(hn, n, F, act, mean, istd, gamma, beta, y, ldy, nullptr, 0, stream);
}

// the same result written to a second destination as well (y2 [n, F], row stride ldy2; NULL: none): a layer's output goes into the
// buffer the next aggregation reads AND into its slot of the block's concatenation (model/network.py:118)
int bn_act_apply_groups(const BnApplyPtrs* g, int ng, int n, int F, int act, int ldy, hipStream_t stream) {
  if (n <= 0 || F <= 0) return 0;
  if (ng < 1 || ng > 2) return CGC_EINVAL;
  bool vec = (F % 4 == 0) && (ldy % 4 == 0);
  for (int i = 0; i < ng; ++i)
    vec = vec && aligned16(g[i].hn) && aligned16(g[i].y) && (g[i].y2 == nullptr || (g[i].ldy2 % 4 == 0 && aligned16(g[i].y2)));
  ColCfg cfg = col_cfg(n, F, vec);
  if (!cfg.ok) return CGC_EINVAL;
  cfg.blocks = row_blocks(n, cfg.lpr, 1024);      // every wave first loads its columns' constants: 1024 longer-lived workgroups
                                                  // beat 2048 (119 vs 147 us on [57.7k, 1140])
  DISPATCH_COL_Y(k_bn_act_apply, cfg, ng, 0, stream, g[0], g[ng - 1], n, F, cfg.lpr, act, ldy);
  CGC_RETURN_IF_LAUNCH_FAILED();
  return 0;
}
extern "C" int cgc_bn_act_apply2(const float* hn, int n, int F, int act, const float* mean, const float* istd, const float* gamma,
                                 const float* beta, float* y, int ldy, float* y2, int ldy2, cgc_stream_t stream) {
  const BnApplyPtrs g{hn, mean, istd, gamma, beta, y, y2, ldy2};
  return bn_act_apply_groups(&g, 1, n, F, act, ldy, as_stream(stream));
}

// ------------------------------------------------------------------------------------------------
// BatchNorm backward, stage 1: sums[0] = sum dy, sums[1] = sum dy * xhat
// ------------------------------------------------------------------------------------------------
template <int VEC, int MAXJ>
__global__ __launch_bounds__(256) void k_bn_bwd_reduce(const BnRedPtrs p0, const BnRedPtrs p1, int ldy, int n, int F, int lpr, int act) {
  const BnRedPtrs& p = blockIdx.y ? p1 : p0;
  const float* __restrict__ dy = p.dy;
  const float* __restrict__ hn = p.hn;
  const float* __restrict__ mean = p.mean;
  const float* __restrict__ istd = p.istd;
  float* __restrict__ ws = p.ws;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const RowGroup rg(lpr);
  float acc[2][MAXJ][VEC];
  float mu[MAXJ][VEC], is[MAXJ][VEC];
#pragma unroll
  for (int j = 0; j < MAXJ; ++j)
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
      acc[0][j][v] = acc[1][j][v] = 0.f;
      const int c = min((rg.sl + lpr * j) * VEC + v, F - 1);      // (clamped, not predicated: see k_bn_act_apply; unused past F)
      mu[j][v] = mean[c];
      is[j][v] = istd[c];
    }
  // (the form of k_bn_act_apply -- activation switch outside the row loop, the row's ten loads requested up front -- was measured
  // here: 90 -> 95 us; 50 more registers take a workgroup per CU away and this pass only reads: 526 MB at 5.8 TB/s)
  for (int base = rg.gwave * rg.rpw; base < n; base += rg.nwaves * rg.rpw) {
    const int row = base + rg.sub;
    if (row < n) {
#pragma unroll
      for (int j = 0; j < MAXJ; ++j) {
        const int c = (rg.sl + lpr * j) * VEC;
        if (c < F) {
          Vec<VEC> d, x;
          load_wide<VEC, MAXJ>(d, dy + (size_t)row * ldy + c);
          load_wide<VEC, MAXJ>(x, hn + (size_t)row * F + c);
#pragma unroll
          for (int v = 0; v < VEC; ++v) {
            const float xhat = (act_fwd(x.v[v], act) - mu[j][v]) * is[j][v];
            acc[0][j][v] += d.v[v];
            acc[1][j][v] += d.v[v] * xhat;
          }
        }
      }
    }
  }
  col_reduce_store<VEC, MAXJ, 2>(acc, F, lpr, smem, ws + (size_t)blockIdx.x * 2 * F);
}

int bn_bwd_reduce_groups(const BnRedPtrs* g, float* const* sums, int ng, int ldy, int n, int F, int act, hipStream_t stream) {
  if (F <= 0) return 0;
  if (ng < 1 || ng > 2) return CGC_EINVAL;
  if (n <= 0) {
    for (int i = 0; i < ng; ++i) (void)hipMemsetAsync(sums[i], 0, sizeof(float) * 2 * F, stream);
    return 0;
  }
  bool vec_ok = (F % 4 == 0) && (ldy % 4 == 0);
  for (int i = 0; i < ng; ++i) vec_ok = vec_ok && aligned16(g[i].dy) && aligned16(g[i].hn);
  ColCfg cfg = col_cfg(n, F, vec_ok);
  if (!cfg.ok) return CGC_EINVAL;
  const size_t smem = sizeof(float) * 3 * 2 * F;
  DISPATCH_COL_Y(k_bn_bwd_reduce, cfg, ng, smem, stream, g[0], g[ng - 1], ldy, n, F, cfg.lpr, act);
  CGC_RETURN_IF_LAUNCH_FAILED();
  return launch_reduce_slots_f32_pair(g[0].ws, sums[0], g[ng - 1].ws, sums[ng - 1], ng, cfg.blocks, 2 * F, stream);
}
extern "C" int cgc_bn_bwd_reduce(const float* dy, int ldy, const float* hn, int n, int F, int act, const float* mean,
                                 const float* istd, float* sums, float* ws, cgc_stream_t stream) {
  const BnRedPtrs g{dy, hn, mean, istd, ws};
  float* const out[1] = {sums};
  return bn_bwd_reduce_groups(&g, out, 1, ldy, n, F, act, as_stream(stream));
}

// ------------------------------------------------------------------------------------------------
// BatchNorm backward stage 2 fused with activation and l2norm backward:  dy -> dh
// ------------------------------------------------------------------------------------------------
template <int VEC, int MAXJ>
__global__ __launch_bounds__(256) void k_bn_act_l2_bwd(const float* __restrict__ dy, int ldy, const float* __restrict__ hn,
                                                       const float* __restrict__ rinv, int n, int F, int lpr, int act,
                                                       int normalize, int mode, const float* __restrict__ mean,
                                                       const float* __restrict__ istd, const float* __restrict__ gamma,
                                                       const float* __restrict__ sums, float inv_count, float* __restrict__ dh,
                                                       float* __restrict__ ws /* per-block column sums of dh, or null */) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const RowGroup rg(lpr);
  float csum[1][MAXJ][VEC];
#pragma unroll
  for (int j = 0; j < MAXJ; ++j)
#pragma unroll
    for (int v = 0; v < VEC; ++v) csum[0][j][v] = 0.f;
  // per-column constants: do = ca*dy - cb - xhat*cc  with  ca = gamma*istd, cb = ca*s0/count, cc = ca*s1/count.
  // Narrow rows keep them in registers (the lane's columns never change); wide rows (MAXJ*VEC > 8: 5 x 32 registers would
  // drop the kernel to one wave per SIMD) re-derive them from the L1-resident parameter vectors at every use.
  constexpr bool REGC = MAXJ * VEC <= 8;
  constexpr int RJ = REGC ? MAXJ : 1, RV = REGC ? VEC : 1;
  float ca[RJ][RV], cb[RJ][RV], cc[RJ][RV], mu[RJ][RV], is[RJ][RV];
  if (REGC) {
#pragma unroll
    for (int j = 0; j < RJ; ++j)
#pragma unroll
      for (int v = 0; v < RV; ++v) {
        const int c = (rg.sl + lpr * j) * VEC + v;
        ca[j][v] = 1.f; cb[j][v] = cc[j][v] = mu[j][v] = is[j][v] = 0.f;
        if (mode != 0 && c < F) {
          ca[j][v] = gamma[c] * istd[c];
          if (mode == 2) {
            cb[j][v] = ca[j][v] * sums[c] * inv_count;
            cc[j][v] = ca[j][v] * sums[F + c] * inv_count;
            mu[j][v] = mean[c];
            is[j][v] = istd[c];
          }
        }
      }
  }
  // wide rows: the five per-column constants live in LDS (filled once per workgroup; 5 F floats behind the 3 F floats of the
  // column-sum exchange) -- as loads from the parameter vectors they were 25 extra 16-byte loads per row and lane
  float* const lc = smem + 3 * (size_t)F;
  if (!REGC && mode != 0) {
    for (int c = threadIdx.x; c < F; c += blockDim.x) {
      const float a_ = gamma[c] * istd[c];
      lc[c] = a_;
      lc[F + c] = mode == 2 ? a_ * sums[c] * inv_count : 0.f;
      lc[2 * F + c] = mode == 2 ? a_ * sums[F + c] * inv_count : 0.f;
      lc[3 * F + c] = mode == 2 ? mean[c] : 0.f;
      lc[4 * F + c] = mode == 2 ? istd[c] : 0.f;
    }
    __syncthreads();
  }
  // (the activation switch outside the row loop: see k_bn_act_apply; the training configuration -- batch statistics + l2norm -- is a
  // compile-time case of its own, every other combination keeps the run-time flags)
  auto rows = [&](auto act_c, auto hot_c) {
    constexpr int ACT = decltype(act_c)::value;
    constexpr bool HOT = decltype(hot_c)::value;
    const int mode_rt = mode, normalize_rt = normalize;
    const int mode = HOT ? 2 : mode_rt;
    const int normalize = HOT ? 1 : normalize_rt;
    for (int base = rg.gwave * rg.rpw; base < n; base += rg.nwaves * rg.rpw) {
      const int row = min(base + rg.sub, n - 1);
      const bool valid = base + rg.sub < n;
      Vec<VEC> g[MAXJ], x[MAXJ];
      float dot = 0.f;
  #pragma unroll
      for (int j = 0; j < MAXJ; ++j) {           // unconditional (a chunk past F re-reads the row's last chunk; unused)
        const int c = min((rg.sl + lpr * j) * VEC, F - VEC);
        load_wide<VEC, MAXJ>(g[j], dy + (size_t)row * ldy + c);
        load_wide<VEC, MAXJ>(x[j], hn + (size_t)row * F + c);
      }
  #pragma unroll
      for (int j = 0; j < MAXJ; ++j) {
        const int c = (rg.sl + lpr * j) * VEC;
        if (valid && c < F) {
          Vec<VEC> pa, pb, pc, pm, pi;      // wide rows: constants of these columns
          if (!REGC && mode != 0) {
            pa.load(lc + c);
            if (mode == 2) { pb.load(lc + F + c); pc.load(lc + 2 * F + c); pm.load(lc + 3 * F + c); pi.load(lc + 4 * F + c); }
          }
  #pragma unroll
          for (int v = 0; v < VEC; ++v) {
            const float xv = x[j].v[v];
            float a_, b_ = 0.f, c_ = 0.f, m_ = 0.f, i_ = 0.f;
            if (REGC) {
              a_ = ca[REGC ? j : 0][REGC ? v : 0]; b_ = cb[REGC ? j : 0][REGC ? v : 0]; c_ = cc[REGC ? j : 0][REGC ? v : 0];
              m_ = mu[REGC ? j : 0][REGC ? v : 0]; i_ = is[REGC ? j : 0][REGC ? v : 0];
            } else if (mode != 0) {
              a_ = pa.v[v];
              if (mode == 2) { b_ = pb.v[v]; c_ = pc.v[v]; m_ = pm.v[v]; i_ = pi.v[v]; }
            } else {
              a_ = 1.f;
            }
            float go = a_ * g[j].v[v];
            if (mode == 2) go = go - b_ - (act_fwd(xv, ACT) - m_) * i_ * c_;
            go *= act_bwd(xv, ACT);
            g[j].v[v] = go;                 // d(hn)
            dot += xv * go;
          }
        } else {
  #pragma unroll
          for (int v = 0; v < VEC; ++v) g[j].v[v] = x[j].v[v] = 0.f;
        }
      }
      if (normalize) dot = group_sum(dot, lpr);
      if (!valid) continue;
      const float r = normalize ? rinv[row] : 1.f;
      const bool clamped = normalize && !(r < 1.f / L2_EPS);   // ||h|| <= eps: F.normalize divided by the constant eps
  #pragma unroll
      for (int j = 0; j < MAXJ; ++j) {
        const int c = (rg.sl + lpr * j) * VEC;
        if (c < F) {
          Vec<VEC> o;
  #pragma unroll
          for (int v = 0; v < VEC; ++v) {
            const float gv = g[j].v[v];
            o.v[v] = !normalize ? gv : (clamped ? gv * (1.f / L2_EPS) : r * (gv - x[j].v[v] * dot));
            csum[0][j][v] += o.v[v];
          }
          o.store(dh + (size_t)row * F + c);
        }
      }
    }
  };
  if (mode == 2 && normalize) {
    switch (act) {
      case CGC_ACT_RELU: rows(std::integral_constant<int, CGC_ACT_RELU>(), std::true_type()); break;
      case CGC_ACT_ELU: rows(std::integral_constant<int, CGC_ACT_ELU>(), std::true_type()); break;
      case CGC_ACT_LEAKYRELU: rows(std::integral_constant<int, CGC_ACT_LEAKYRELU>(), std::true_type()); break;
      default: rows(std::integral_constant<int, CGC_ACT_IDENTITY>(), std::true_type()); break;
    }
  } else {
    switch (act) {
      case CGC_ACT_RELU: rows(std::integral_constant<int, CGC_ACT_RELU>(), std::false_type()); break;
      case CGC_ACT_ELU: rows(std::integral_constant<int, CGC_ACT_ELU>(), std::false_type()); break;
      case CGC_ACT_LEAKYRELU: rows(std::integral_constant<int, CGC_ACT_LEAKYRELU>(), std::false_type()); break;
      default: rows(std::integral_constant<int, CGC_ACT_IDENTITY>(), std::false_type()); break;
    }
  }
  if (ws != nullptr) col_reduce_store<VEC, MAXJ, 1>(csum, F, lpr, smem, ws + (size_t)blockIdx.x * F);
}

extern "C" int cgc_bn_act_l2_bwd(const float* dy, int ldy, const float* hn, const float* rinv, int n, int F, int act,
                                 int normalize, int mode, const float* mean, const float* istd, const float* gamma,
                                 const float* sums, double count, float* dh, float* dh_colsum, float* ws, cgc_stream_t stream) {
  if (F <= 0) return 0;
  if (n <= 0) {
    if (dh_colsum) (void)hipMemsetAsync(dh_colsum, 0, sizeof(float) * F, as_stream(stream));
    return 0;
  }
  if (dh_colsum != nullptr && ws == nullptr) return CGC_EINVAL;
  const bool vec = (F % 4 == 0) && (ldy % 4 == 0) && aligned16(dy) && aligned16(hn) && aligned16(dh);
  ColCfg cfg = col_cfg(n, F, vec);
  if (!cfg.ok) return CGC_EINVAL;
  if (dh_colsum == nullptr) cfg.blocks = row_blocks(n, cfg.lpr);   // no reduction slots needed: use the full grid
  const float inv_count = (float)(1.0 / count);
  const size_t smem = sizeof(float) * (3 + 5) * F;      // column-sum exchange + the per-column constants of wide rows
  DISPATCH_COL(k_bn_act_l2_bwd, cfg, smem, as_stream(stream), dy, ldy, hn, rinv, n, F, cfg.lpr, act, normalize, mode, mean, istd,
               gamma, sums, inv_count, dh, dh_colsum ? ws : (float*)nullptr);
  CGC_RETURN_IF_LAUNCH_FAILED();
  if (dh_colsum) {
    hipLaunchKernelGGL(k_reduce_slots<float>, REDUCE_SLOTS_GRID(F), dim3(32 * RS_GROUPS), 0, as_stream(stream), ws, cfg.blocks, F, dh_colsum);
    CGC_RETURN_IF_LAUNCH_FAILED();
  }
  return 0;
}

// ------------------------------------------------------------------------------------------------
// column sums (bias gradients)
// ------------------------------------------------------------------------------------------------
template <int VEC, int MAXJ>
__global__ __launch_bounds__(256) void k_colsum(const float* __restrict__ x, int ld, int n, int F, int lpr, float* __restrict__ ws) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const RowGroup rg(lpr);
  float acc[1][MAXJ][VEC];
#pragma unroll
  for (int j = 0; j < MAXJ; ++j)
#pragma unroll
    for (int v = 0; v < VEC; ++v) acc[0][j][v] = 0.f;
  for (int base = rg.gwave * rg.rpw; base < n; base += rg.nwaves * rg.rpw) {
    const int row = base + rg.sub;
    if (row < n) {
#pragma unroll
      for (int j = 0; j < MAXJ; ++j) {
        const int c = (rg.sl + lpr * j) * VEC;
        if (c < F) {
          Vec<VEC> d;
          d.load(x + (size_t)row * ld + c);
#pragma unroll
          for (int v = 0; v < VEC; ++v) acc[0][j][v] += d.v[v];
        }
      }
    }
  }
  col_reduce_store<VEC, MAXJ, 1>(acc, F, lpr, smem, ws + (size_t)blockIdx.x * F);
}

extern "C" int cgc_colsum(const float* x, int ld, int n, int F, float* out, float* ws, cgc_stream_t stream) {
  if (F <= 0) return 0;
  if (n <= 0) {
    (void)hipMemsetAsync(out, 0, sizeof(float) * F, as_stream(stream));
    return 0;
  }
  const bool vec_ok = (F % 4 == 0) && (ld % 4 == 0) && aligned16(x);
  ColCfg cfg = col_cfg(n, F, vec_ok);
  if (!cfg.ok) return CGC_EINVAL;
  const size_t smem = sizeof(float) * 3 * F;
  DISPATCH_COL(k_colsum, cfg, smem, as_stream(stream), x, ld, n, F, cfg.lpr, ws);
  CGC_RETURN_IF_LAUNCH_FAILED();
  hipLaunchKernelGGL(k_reduce_slots<float>, REDUCE_SLOTS_GRID(F), dim3(32 * RS_GROUPS), 0, as_stream(stream), ws, cfg.blocks, F, out);
  CGC_RETURN_IF_LAUNCH_FAILED();
  return 0;
}

// ------------------------------------------------------------------------------------------------
// row softmax (assignment matrix)
// ------------------------------------------------------------------------------------------------
template <int VEC>
__global__ __launch_bounds__(256) void k_softmax_fwd(const float* __restrict__ x, int n, int C, int ld, int lpr, float* __restrict__ out) {
  const RowGroup rg(lpr);
  for (int base = rg.gwave * rg.rpw; base < n; base += rg.nwaves * rg.rpw) {
    const int row = base + rg.sub;
    const bool valid = row < n;
    const float* xr = x + (size_t)row * ld;
    float m = -INFINITY;
    if (valid)
      for (int c = rg.sl * VEC; c < C; c += lpr * VEC) {
        Vec<VEC> t;
        t.load(xr + c);
#pragma unroll
        for (int v = 0; v < VEC; ++v) m = fmaxf(m, t.v[v]);
      }
    m = group_max(m, lpr);
    float s = 0.f;
    if (valid)
      for (int c = rg.sl * VEC; c < C; c += lpr * VEC) {   // second and third pass hit L1/L2: the row was just read
        Vec<VEC> t;
        t.load(xr + c);
#pragma unroll
        for (int v = 0; v < VEC; ++v) s += expf(t.v[v] - m);
      }
    s = group_sum(s, lpr);
    if (!valid) continue;
    const float inv = 1.f / s;
    for (int c = rg.sl * VEC; c < C; c += lpr * VEC) {
      Vec<VEC> t;
      t.load(xr + c);
#pragma unroll
      for (int v = 0; v < VEC; ++v) t.v[v] = expf(t.v[v] - m) * inv;
      t.store(out + (size_t)row * ld + c);
    }
  }
}

// rows that fit the lanes' registers (C <= 64 * MAXJ * VEC): ONE read of the row, one expf per element (same operations in
// the same order as the three-pass kernel above: same bits)
template <int VEC, int MAXJ>
__global__ __launch_bounds__(256) void k_softmax_fwd_reg(const float* __restrict__ x, int n, int C, int ld, int lpr,
                                                         float* __restrict__ out) {
  const RowGroup rg(lpr);
  for (int base = rg.gwave * rg.rpw; base < n; base += rg.nwaves * rg.rpw) {
    const int row = base + rg.sub;
    const bool valid = row < n;
    const float* xr = x + (size_t)row * ld;
    Vec<VEC> t[MAXJ];
    float m = -INFINITY;
#pragma unroll
    for (int j = 0; j < MAXJ; ++j) {
      const int c = (rg.sl + lpr * j) * VEC;
      if (valid && c < C) {
        t[j].load(xr + c);          // (plain: streaming loads gain nothing on the in-place pass and cost the aggregation that reads S next 6 %)
#pragma unroll
        for (int v = 0; v < VEC; ++v) m = fmaxf(m, t[j].v[v]);
      }
    }
    m = group_max(m, lpr);
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < MAXJ; ++j) {
      const int c = (rg.sl + lpr * j) * VEC;
      if (valid && c < C) {
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
          t[j].v[v] = expf(t[j].v[v] - m);
          s += t[j].v[v];
        }
      }
    }
    s = group_sum(s, lpr);
    if (!valid) continue;
    const float inv = 1.f / s;
#pragma unroll
    for (int j = 0; j < MAXJ; ++j) {
      const int c = (rg.sl + lpr * j) * VEC;
      if (c < C) {
#pragma unroll
        for (int v = 0; v < VEC; ++v) t[j].v[v] *= inv;
        t[j].store(out + (size_t)row * ld + c);
      }
    }
  }
}

template <int VEC, int MAXJ>
__global__ __launch_bounds__(256) void k_softmax_bwd(const float* __restrict__ S, const float* __restrict__ dS, int n, int C,
                                                     int ld, int lpr, float* __restrict__ dx, float* __restrict__ ws) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const RowGroup rg(lpr);
  float csum[1][MAXJ][VEC];
#pragma unroll
  for (int j = 0; j < MAXJ; ++j)
#pragma unroll
    for (int v = 0; v < VEC; ++v) csum[0][j][v] = 0.f;
  for (int base = rg.gwave * rg.rpw; base < n; base += rg.nwaves * rg.rpw) {
    const int row = min(base + rg.sub, n - 1);
    const bool valid = base + rg.sub < n;
    Vec<VEC> s[MAXJ], d[MAXJ];
    float dot = 0.f;
    // unconditional loads, all requested before the first use (a chunk past C re-reads the row's last chunk and is zeroed below)
#pragma unroll
    for (int j = 0; j < MAXJ; ++j) {
      const int c = min((rg.sl + lpr * j) * VEC, C - VEC);
      load_wide<VEC, MAXJ>(s[j], S + (size_t)row * ld + c);
      load_wide<VEC, MAXJ>(d[j], dS + (size_t)row * ld + c);
    }
#pragma unroll
    for (int j = 0; j < MAXJ; ++j) {
      const bool ok = valid && (rg.sl + lpr * j) * VEC < C;
#pragma unroll
      for (int v = 0; v < VEC; ++v) {
        s[j].v[v] = ok ? s[j].v[v] : 0.f;
        d[j].v[v] = ok ? d[j].v[v] : 0.f;
        dot += s[j].v[v] * d[j].v[v];
      }
    }
    dot = group_sum(dot, lpr);
    if (!valid) continue;
#pragma unroll
    for (int j = 0; j < MAXJ; ++j) {
      const int c = (rg.sl + lpr * j) * VEC;
      if (c < C) {
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
          d[j].v[v] = s[j].v[v] * (d[j].v[v] - dot);
          csum[0][j][v] += d[j].v[v];
        }
        d[j].store(dx + (size_t)row * ld + c);
      }
    }
  }
  if (ws != nullptr) col_reduce_store<VEC, MAXJ, 1>(csum, C, lpr, smem, ws + (size_t)blockIdx.x * C);
}

extern "C" int cgc_softmax_fwd(const float* x, int n, int C, int ld, float* out, cgc_stream_t stream) {
  if (n <= 0 || C <= 0) return 0;
  if (ld < C) return CGC_EINVAL;
  const bool vec = (C % 4 == 0) && (ld % 4 == 0) && aligned16(x) && aligned16(out);
  ColCfg cfg = col_cfg(n, C, vec);
  if (cfg.ok && cfg.maxj <= 8) {               // the row fits the lanes' registers
    cfg.blocks = row_blocks(n, cfg.lpr);
    DISPATCH_COL(k_softmax_fwd_reg, cfg, 0, as_stream(stream), x, n, C, ld, cfg.lpr, out);
    CGC_RETURN_IF_LAUNCH_FAILED();
    return 0;
  }
  const int lpr = pick_lpr(vec ? C / 4 : C);
  dim3 grid(row_blocks(n, lpr)), block(CGC_BLOCK);
  if (vec)
    hipLaunchKernelGGL(k_softmax_fwd<4>, grid, block, 0, as_stream(stream), x, n, C, ld, lpr, out);
  else
    hipLaunchKernelGGL(k_softmax_fwd<1>, grid, block, 0, as_stream(stream), x, n, C, ld, lpr, out);
  CGC_RETURN_IF_LAUNCH_FAILED();
  return 0;
}

extern "C" int cgc_softmax_bwd(const float* S, const float* dS, int n, int C, int ld, float* dx, float* dx_colsum, float* ws,
                               cgc_stream_t stream) {
  if (C <= 0) return 0;
  if (ld < C) return CGC_EINVAL;
  if (n <= 0) {
    if (dx_colsum) (void)hipMemsetAsync(dx_colsum, 0, sizeof(float) * C, as_stream(stream));
    return 0;
  }
  if (dx_colsum != nullptr && ws == nullptr) return CGC_EINVAL;
  const bool vec = (C % 4 == 0) && (ld % 4 == 0) && aligned16(S) && aligned16(dS) && aligned16(dx);
  ColCfg cfg = col_cfg(n, C, vec);
  if (!cfg.ok) return CGC_EINVAL;
  if (dx_colsum == nullptr) cfg.blocks = row_blocks(n, cfg.lpr);
  const size_t smem = dx_colsum ? sizeof(float) * 3 * C : 0;
  DISPATCH_COL(k_softmax_bwd, cfg, smem, as_stream(stream), S, dS, n, C, ld, cfg.lpr, dx, dx_colsum ? ws : (float*)nullptr);
  CGC_RETURN_IF_LAUNCH_FAILED();
  if (dx_colsum) {
    hipLaunchKernelGGL(k_reduce_slots<float>, REDUCE_SLOTS_GRID(C), dim3(32 * RS_GROUPS), 0, as_stream(stream), ws, cfg.blocks, C, dx_colsum);
    CGC_RETURN_IF_LAUNCH_FAILED();
  }
  return 0;
}

// ------------------------------------------------------------------------------------------------
// max readout per graph (with the implicit zero padding rows of the dense layout)
// ------------------------------------------------------------------------------------------------
#define SEGMAX_WAVES 16
// one workgroup per (graph, 64-column group); with D <= 32 columns a wave carries 64 / D' rows at a time (D' = D rounded up to a
// power of two) instead of idling 44 of its 64 lanes on the 20-column readouts of this network
__global__ __launch_bounds__(64 * SEGMAX_WAVES) void k_segment_max_fwd(const float* __restrict__ x, const int* __restrict__ gptr,
                                                                       int D, int nmax, float* __restrict__ out,
                                                                       int* __restrict__ arg) {
  __shared__ float bv[SEGMAX_WAVES][64];
  __shared__ int bi[SEGMAX_WAVES][64];
  const int b = blockIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int dp = 64;                                         // lanes per row
  if (D <= 32) dp = D <= 2 ? 2 : D <= 4 ? 4 : D <= 8 ? 8 : D <= 16 ? 16 : 32;
  const int rpw = 64 / dp, sub = lane / dp, dl = lane - sub * dp;
  const int d = blockIdx.y * 64 + dl;
  const int lo = gptr[b], hi = gptr[b + 1];
  float best = -INFINITY;
  int idx = -1;
  if (d < D) {
    const int stride = SEGMAX_WAVES * rpw;
    int r = lo + wave * rpw + sub;
    for (; r + 3 * stride < hi; r += 4 * stride) {       // four independent loads in flight (the scan is pure latency: ~60 dependent
      float v[4];                                        // iterations per wave on an 1800-node graph); compared in row order
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = x[(size_t)(r + u * stride) * D + d];
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (v[u] > best) { best = v[u]; idx = r + u * stride; }
    }
    for (; r < hi; r += stride) {                        // increasing rows + strict '>' keeps the FIRST maximum
      const float v = x[(size_t)r * D + d];
      if (v > best) { best = v; idx = r; }
    }
  }
  bv[wave][lane] = best;
  bi[wave][lane] = idx;
  __syncthreads();
  if (wave == 0 && sub == 0 && d < D) {
    for (int w = 0; w < SEGMAX_WAVES; ++w)
      for (int s2 = (w == 0 ? 1 : 0); s2 < rpw; ++s2) {
        const float v = bv[w][s2 * dp + dl];
        const int i = bi[w][s2 * dp + dl];
        if (i >= 0 && (v > best || (v == best && i < idx) || idx < 0)) { best = v; idx = i; }
      }
    if (idx < 0) { best = 0.f; }                                   // empty graph: only padding rows
    else if (hi - lo < nmax && best < 0.f) { best = 0.f; idx = -1; }  // a zero padding row wins (ties go to the real row)
    out[(size_t)b * D + d] = best;
    arg[(size_t)b * D + d] = idx;
  }
}

__global__ void k_segment_max_bwd(const float* __restrict__ dout, const int* __restrict__ arg, int total, int D, float* __restrict__ dx) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int a = arg[i];
  if (a >= 0) dx[(size_t)a * D + (i % D)] = dout[i];
}

extern "C" int cgc_segment_max_fwd(const float* x, const int* gptr, int B, int D, int nmax, float* out, int* arg, cgc_stream_t stream) {
  if (B <= 0 || D <= 0) return 0;
  hipLaunchKernelGGL(k_segment_max_fwd, dim3(B, ceil_div(D, 64)), dim3(64 * SEGMAX_WAVES), 0, as_stream(stream), x, gptr, D, nmax, out, arg);
  CGC_RETURN_IF_LAUNCH_FAILED();
  return 0;
}

// the same scatter writing EVERY element of dx (zeros included): one kernel instead of a fill + a scatter
__global__ void k_segment_max_bwd_full(const float* __restrict__ dout, const int* __restrict__ arg, const int* __restrict__ gptr, int D,
                                       float* __restrict__ dx) {
  const int b = blockIdx.y;
  const int g0 = gptr[b], cnt = (gptr[b + 1] - g0) * D;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < cnt; i += gridDim.x * blockDim.x) {
    const int r = i / D, d = i - r * D;
    dx[(size_t)g0 * D + i] = (arg[b * D + d] == g0 + r) ? dout[b * D + d] : 0.f;
  }
}

extern "C" int cgc_segment_max_bwd_full(const float* dout, const int* arg, const int* gptr, int B, int D, int nmax, float* dx,
                                        cgc_stream_t stream) {
  if (B <= 0 || D <= 0 || nmax <= 0) return 0;
  if (B > 65535) return CGC_EINVAL;
  int bx = ceil_div(nmax * D, 256 * 4);
  if (bx < 1) bx = 1;
  hipLaunchKernelGGL(k_segment_max_bwd_full, dim3(bx, B), dim3(256), 0, as_stream(stream), dout, arg, gptr, D, dx);
  CGC_RETURN_IF_LAUNCH_FAILED();
  return 0;
}

extern "C" int cgc_segment_max_bwd(const float* dout, const int* arg, int B, int D, float* dx_zeroed, cgc_stream_t stream) {
  if (B <= 0 || D <= 0) return 0;
  const int total = B * D;
  hipLaunchKernelGGL(k_segment_max_bwd, dim3(ceil_div(total, 256)), dim3(256), 0, as_stream(stream), dout, arg, total, D, dx_zeroed);
  CGC_RETURN_IF_LAUNCH_FAILED();
  return 0;
}

// ------------------------------------------------------------------------------------------------
// dense adjacency (levels 2-3): A / clamp(rowsum,1)   and   _re_norm_adj, each with its backward
// ------------------------------------------------------------------------------------------------
template <int VEC>
__global__ __launch_bounds__(256) void k_dense_rownorm_fwd(const float* __restrict__ A, int R, int C, int lpr, float* __restrict__ out,
                                                           float* __restrict__ invd, float* __restrict__ ge1) {
  const RowGroup rg(lpr);
  for (int base = rg.gwave * rg.rpw; base < R; base += rg.nwaves * rg.rpw) {
    const int row = base + rg.sub;
    const bool valid = row < R;
    float s = 0.f;
    if (valid)
      for (int c = rg.sl * VEC; c < C; c += lpr * VEC) {
        Vec<VEC> t;
        t.load(A + (size_t)row * C + c);
#pragma unroll
        for (int v = 0; v < VEC; ++v) s += t.v[v];
      }
    s = group_sum(s, lpr);
    if (!valid) continue;
    const float inv = 1.f / fmaxf(s, 1.f);
    for (int c = rg.sl * VEC; c < C; c += lpr * VEC) {
      Vec<VEC> t;
      t.load(A + (size_t)row * C + c);
#pragma unroll
      for (int v = 0; v < VEC; ++v) t.v[v] *= inv;
      t.store(out + (size_t)row * C + c);
    }
    if (rg.sl == 0) {
      invd[row] = inv;
      ge1[row] = s >= 1.f ? 1.f : 0.f;
    }
  }
}

template <int VEC>
__global__ __launch_bounds__(256) void k_dense_rownorm_bwd(const float* __restrict__ dOut, const float* __restrict__ An,
                                                           const float* __restrict__ invd, const float* __restrict__ ge1, int R, int C,
                                                           int lpr, float* __restrict__ dA) {
  const RowGroup rg(lpr);
  for (int base = rg.gwave * rg.rpw; base < R; base += rg.nwaves * rg.rpw) {
    const int row = base + rg.sub;
    const bool valid = row < R;
    float t = 0.f;
    if (valid)
      for (int c = rg.sl * VEC; c < C; c += lpr * VEC) {
        Vec<VEC> g, a;
        g.load(dOut + (size_t)row * C + c);
        a.load(An + (size_t)row * C + c);
#pragma unroll
        for (int v = 0; v < VEC; ++v) t += g.v[v] * a.v[v];
      }
    t = group_sum(t, lpr);
    if (!valid) continue;
    const float inv = invd[row], sub = ge1[row] * t;
    for (int c = rg.sl * VEC; c < C; c += lpr * VEC) {
      Vec<VEC> g;
      g.load(dOut + (size_t)row * C + c);
#pragma unroll
      for (int v = 0; v < VEC; ++v) g.v[v] = inv * (g.v[v] - sub);
      g.store(dA + (size_t)row * C + c);
    }
  }
}

template <int VEC>
__global__ __launch_bounds__(256) void k_dense_renorm_fwd(const float* __restrict__ A, int R, int C, int lpr, float p, float* __restrict__ out) {
  const RowGroup rg(lpr);
  const float omp = 1.f - p;
  for (int base = rg.gwave * rg.rpw; base < R; base += rg.nwaves * rg.rpw) {
    const int row = base + rg.sub;
    const bool valid = row < R;
    const int diag = valid ? row % C : -1;
    float s = 0.f;
    if (valid)
      for (int c = rg.sl * VEC; c < C; c += lpr * VEC) {
        Vec<VEC> t;
        t.load(A + (size_t)row * C + c);
#pragma unroll
        for (int v = 0; v < VEC; ++v) s += (c + v == diag) ? 0.f : t.v[v];
      }
    s = group_sum(s, lpr);
    if (!valid) continue;
    const float den = s + RENORM_EPS;
    for (int c = rg.sl * VEC; c < C; c += lpr * VEC) {
      Vec<VEC> t;
      t.load(A + (size_t)row * C + c);
#pragma unroll
      for (int v = 0; v < VEC; ++v) t.v[v] = (c + v == diag) ? p : (t.v[v] / den) * omp;
      t.store(out + (size_t)row * C + c);
    }
  }
}

template <int VEC>
__global__ __launch_bounds__(256) void k_dense_renorm_bwd(const float* __restrict__ A, const float* __restrict__ dOut, int R, int C,
                                                          int lpr, float p, float* __restrict__ dA) {
  const RowGroup rg(lpr);
  const float omp = 1.f - p;
  for (int base = rg.gwave * rg.rpw; base < R; base += rg.nwaves * rg.rpw) {
    const int row = base + rg.sub;
    const bool valid = row < R;
    const int diag = valid ? row % C : -1;
    float s = 0.f, t = 0.f;
    if (valid)
      for (int c = rg.sl * VEC; c < C; c += lpr * VEC) {
        Vec<VEC> a, g;
        a.load(A + (size_t)row * C + c);
        g.load(dOut + (size_t)row * C + c);
#pragma unroll
        for (int v = 0; v < VEC; ++v)
          if (c + v != diag) { s += a.v[v]; t += a.v[v] * g.v[v]; }
      }
    s = group_sum(s, lpr);
    t = group_sum(t, lpr);
    if (!valid) continue;
    const float q = 1.f / (s + RENORM_EPS);
    for (int c = rg.sl * VEC; c < C; c += lpr * VEC) {
      Vec<VEC> g;
      g.load(dOut + (size_t)row * C + c);
#pragma unroll
      for (int v = 0; v < VEC; ++v) g.v[v] = (c + v == diag) ? 0.f : omp * q * (g.v[v] - q * t);
      g.store(dA + (size_t)row * C + c);
    }
  }
}

// ---- fused adjacency preparation of a dense level (levels 2-3): optional _re_norm_adj (model/network.py:183-191) followed by
// the clamp(min=1) row normalisation of DenseSAGEConv, in ONE pass over the [B,C,C] adjacency each way.  Unfused the forward
// was two read+write passes and the backward three (row-norm backward, the autograd sum of the two gradient streams into the
// re-normalised adjacency, re-norm backward): 9 x 166 MB at C3 -> 5 x 166 MB here.  The later passes over a row re-read it
// from L2 (a row is 4.5 KB).  p < 0: no re-normalisation (At is not written; pass At = NULL).
template <int VEC>
__global__ __launch_bounds__(256) void k_adj_prep_fwd(const float* __restrict__ A, int R, int C, int lpr, float p, float* __restrict__ At,
                                                      float* __restrict__ An, float* __restrict__ invd, float* __restrict__ ge1) {
  const RowGroup rg(lpr);
  const bool renorm = p >= 0.f;
  const float omp = 1.f - p;
  for (int base = rg.gwave * rg.rpw; base < R; base += rg.nwaves * rg.rpw) {
    const int row = base + rg.sub;
    const bool valid = row < R;
    const int diag = (valid && renorm) ? row % C : -1;
    float s = 0.f;
    if (valid)
      for (int c = rg.sl * VEC; c < C; c += lpr * VEC) {
        Vec<VEC> t;
        t.load(A + (size_t)row * C + c);
#pragma unroll
        for (int v = 0; v < VEC; ++v) s += (c + v == diag) ? 0.f : t.v[v];
      }
    s = group_sum(s, lpr);
    float st = s;                                        // row sum of the (re-normalised) adjacency, from the STORED values
    const float den = s + RENORM_EPS;
    if (renorm) {
      st = 0.f;
      if (valid)
        for (int c = rg.sl * VEC; c < C; c += lpr * VEC) {
          Vec<VEC> t;
          t.load(A + (size_t)row * C + c);
#pragma unroll
          for (int v = 0; v < VEC; ++v) { t.v[v] = (c + v == diag) ? p : (t.v[v] / den) * omp; st += t.v[v]; }
          t.store(At + (size_t)row * C + c);
        }
      st = group_sum(st, lpr);
    }
    if (!valid) continue;
    const float inv = 1.f / fmaxf(st, 1.f);
    for (int c = rg.sl * VEC; c < C; c += lpr * VEC) {
      Vec<VEC> t;
      t.load(A + (size_t)row * C + c);
#pragma unroll
      for (int v = 0; v < VEC; ++v) {
        const float at = renorm ? ((c + v == diag) ? p : (t.v[v] / den) * omp) : t.v[v];
        t.v[v] = at * inv;
      }
      t.store(An + (size_t)row * C + c);
    }
    if (rg.sl == 0) {
      invd[row] = inv;
      ge1[row] = st >= 1.f ? 1.f : 0.f;
    }
  }
}

// dAt = invd*(gAn - ge1*<gAn,An>) + gAt  (row-norm backward + the gradient that reaches the re-normalised adjacency directly,
// e.g. from A~ S of _diff_pool; gAt may be NULL), then, if p >= 0, the re-norm backward dA = (1-p) q (dAt - q <A,dAt>_offdiag)
// off the diagonal and 0 on it, q = 1/(rowsum_offdiag(A)+1e-15).  <A,dAt> is assembled from four row reductions of the inputs.
template <int VEC>
__global__ __launch_bounds__(256) void k_adj_prep_bwd(const float* __restrict__ A, const float* __restrict__ An, const float* __restrict__ invd,
                                                      const float* __restrict__ ge1, const float* __restrict__ gAn,
                                                      const float* __restrict__ gAt, int R, int C, int lpr, float p, float* __restrict__ dA) {
  const RowGroup rg(lpr);
  const bool renorm = p >= 0.f;
  const float omp = 1.f - p;
  for (int base = rg.gwave * rg.rpw; base < R; base += rg.nwaves * rg.rpw) {
    const int row = base + rg.sub;
    const bool valid = row < R;
    const int diag = (valid && renorm) ? row % C : -1;
    float t1 = 0.f, s = 0.f, u = 0.f, w = 0.f;
    if (valid)
      for (int c = rg.sl * VEC; c < C; c += lpr * VEC) {
        Vec<VEC> g, an, a, h;
        g.load(gAn + (size_t)row * C + c);
        an.load(An + (size_t)row * C + c);
#pragma unroll
        for (int v = 0; v < VEC; ++v) t1 += g.v[v] * an.v[v];
        if (renorm) {
          a.load(A + (size_t)row * C + c);
          if (gAt != nullptr) h.load(gAt + (size_t)row * C + c);
#pragma unroll
          for (int v = 0; v < VEC; ++v)
            if (c + v != diag) { s += a.v[v]; u += a.v[v] * g.v[v]; if (gAt != nullptr) w += a.v[v] * h.v[v]; }
        }
      }
    t1 = group_sum(t1, lpr);
    if (renorm) { s = group_sum(s, lpr); u = group_sum(u, lpr); w = group_sum(w, lpr); }
    if (!valid) continue;
    const float inv = invd[row], sub = ge1[row] * t1;
    const float q = 1.f / (s + RENORM_EPS);
    const float t = inv * (u - sub * s) + w;             // <A, dAt> over the off-diagonal entries
    for (int c = rg.sl * VEC; c < C; c += lpr * VEC) {
      Vec<VEC> g, h;
      g.load(gAn + (size_t)row * C + c);
      if (gAt != nullptr) h.load(gAt + (size_t)row * C + c);
#pragma unroll
      for (int v = 0; v < VEC; ++v) {
        float d = inv * (g.v[v] - sub);
        if (gAt != nullptr) d += h.v[v];
        g.v[v] = renorm ? ((c + v == diag) ? 0.f : omp * q * (d - q * t)) : d;
      }
      g.store(dA + (size_t)row * C + c);
    }
  }
}

// register-resident forms of the two kernels above for rows that fit the lanes' registers (C <= 64 * MAXJ * VEC: the 1140 / 1600
// clusters of level 2): every input is read ONCE, the divisions are done once; same operations in the same order (same bits).
template <int VEC, int MAXJ>
__global__ __launch_bounds__(256) void k_adj_prep_fwd_reg(const float* __restrict__ A, int R, int C, int lpr, float p, float* __restrict__ At,
                                                          float* __restrict__ An, float* __restrict__ invd, float* __restrict__ ge1) {
  const RowGroup rg(lpr);
  const bool renorm = p >= 0.f;
  const float omp = 1.f - p;
  for (int base = rg.gwave * rg.rpw; base < R; base += rg.nwaves * rg.rpw) {
    const int row = base + rg.sub;
    const bool valid = row < R;
    const int diag = (valid && renorm) ? row % C : -1;
    Vec<VEC> t[MAXJ];
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < MAXJ; ++j) {
      const int c = (rg.sl + lpr * j) * VEC;
      if (valid && c < C) {
        t[j].load(A + (size_t)row * C + c);      // (plain: no gain from a streaming load here)
#pragma unroll
        for (int v = 0; v < VEC; ++v) s += (c + v == diag) ? 0.f : t[j].v[v];
      }
    }
    s = group_sum(s, lpr);
    float st = s;
    const float den = s + RENORM_EPS;
    if (renorm) {
      st = 0.f;
#pragma unroll
      for (int j = 0; j < MAXJ; ++j) {
        const int c = (rg.sl + lpr * j) * VEC;
        if (valid && c < C) {
#pragma unroll
          for (int v = 0; v < VEC; ++v) { t[j].v[v] = (c + v == diag) ? p : (t[j].v[v] / den) * omp; st += t[j].v[v]; }
          t[j].store(At + (size_t)row * C + c);
        }
      }
      st = group_sum(st, lpr);
    }
    if (!valid) continue;
    const float inv = 1.f / fmaxf(st, 1.f);
#pragma unroll
    for (int j = 0; j < MAXJ; ++j) {
      const int c = (rg.sl + lpr * j) * VEC;
      if (c < C) {
#pragma unroll
        for (int v = 0; v < VEC; ++v) t[j].v[v] *= inv;
        t[j].store(An + (size_t)row * C + c);
      }
    }
    if (rg.sl == 0) {
      invd[row] = inv;
      ge1[row] = st >= 1.f ? 1.f : 0.f;
    }
  }
}

template <int VEC, int MAXJ>
__global__ __launch_bounds__(256) void k_adj_prep_bwd_reg(const float* __restrict__ A, const float* __restrict__ An,
                                                          const float* __restrict__ invd, const float* __restrict__ ge1,
                                                          const float* __restrict__ gAn, const float* __restrict__ gAt, int R, int C,
                                                          int lpr, float p, float* __restrict__ dA) {
  const RowGroup rg(lpr);
  const bool renorm = p >= 0.f;
  const float omp = 1.f - p;
  for (int base = rg.gwave * rg.rpw; base < R; base += rg.nwaves * rg.rpw) {
    const int row = base + rg.sub;
    const bool valid = row < R;
    const int diag = (valid && renorm) ? row % C : -1;
    float t1 = 0.f, s = 0.f, u = 0.f, w = 0.f;
    Vec<VEC> g[MAXJ], h[MAXJ];
    if (renorm) {
      // An is NOT read here: it is re-formed from A with the arithmetic of k_adj_prep_fwd_reg -- the same row sum in the same order,
      // the same division, the same two multiplications -- and the pass reads 166 MB less (of 830 at the C3 level 2: 143 -> 130 us).
      // (dA moves by 6e-8 to 1.4e-7 of its largest entry against the version that read An: the compiler fuses the products of the
      // row terms into FMAs differently in the two code shapes.)
      Vec<VEC> a[MAXJ];
#pragma unroll
      for (int j = 0; j < MAXJ; ++j) {
        const int c = (rg.sl + lpr * j) * VEC;
        if (valid && c < C) {
          load_wide<VEC, MAXJ>(g[j], gAn + (size_t)row * C + c);
          load_wide<VEC, MAXJ>(a[j], A + (size_t)row * C + c);
          if (gAt != nullptr) load_wide<VEC, MAXJ>(h[j], gAt + (size_t)row * C + c);
#pragma unroll
          for (int v = 0; v < VEC; ++v) s += (c + v == diag) ? 0.f : a[j].v[v];
        }
      }
      s = group_sum(s, lpr);
      const float den = s + RENORM_EPS, inv_f = valid ? invd[row] : 0.f;
#pragma unroll
      for (int j = 0; j < MAXJ; ++j) {
        const int c = (rg.sl + lpr * j) * VEC;
        if (valid && c < C) {
#pragma unroll
          for (int v = 0; v < VEC; ++v) {
            const bool dg = c + v == diag;
            const float at = dg ? p : (a[j].v[v] / den) * omp;
            t1 += g[j].v[v] * (at * inv_f);
            if (!dg) { u += a[j].v[v] * g[j].v[v]; if (gAt != nullptr) w += a[j].v[v] * h[j].v[v]; }
          }
        }
      }
    } else {
#pragma unroll
      for (int j = 0; j < MAXJ; ++j) {
        const int c = (rg.sl + lpr * j) * VEC;
        if (valid && c < C) {
          Vec<VEC> an;
          load_wide<VEC, MAXJ>(g[j], gAn + (size_t)row * C + c);
          load_wide<VEC, MAXJ>(an, An + (size_t)row * C + c);
#pragma unroll
          for (int v = 0; v < VEC; ++v) t1 += g[j].v[v] * an.v[v];
          if (gAt != nullptr) load_wide<VEC, MAXJ>(h[j], gAt + (size_t)row * C + c);
        }
      }
    }
    t1 = group_sum(t1, lpr);
    if (renorm) { u = group_sum(u, lpr); w = group_sum(w, lpr); }
    if (!valid) continue;
    const float inv = invd[row], sub = ge1[row] * t1;
    const float q = 1.f / (s + RENORM_EPS);
    const float t = inv * (u - sub * s) + w;             // <A, dAt> over the off-diagonal entries
#pragma unroll
    for (int j = 0; j < MAXJ; ++j) {
      const int c = (rg.sl + lpr * j) * VEC;
      if (c < C) {
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
          float d = inv * (g[j].v[v] - sub);
          if (gAt != nullptr) d += h[j].v[v];
          g[j].v[v] = renorm ? ((c + v == diag) ? 0.f : omp * q * (d - q * t)) : d;
        }
        g[j].store(dA + (size_t)row * C + c);
      }
    }
  }
}

#define LAUNCH_ROW(KERNEL, vec, lpr, R, stream, ...)                                                          \
  do {                                                                                                        \
    dim3 g__(row_blocks(R, lpr)), b__(CGC_BLOCK);                                                             \
    if (vec) hipLaunchKernelGGL(KERNEL<4>, g__, b__, 0, stream, __VA_ARGS__);                                 \
    else hipLaunchKernelGGL(KERNEL<1>, g__, b__, 0, stream, __VA_ARGS__);                                     \
  } while (0)

extern "C" int cgc_dense_rownorm_fwd(const float* A, int R, int C, float* out, float* invd, float* ge1, cgc_stream_t stream) {
  if (R <= 0 || C <= 0) return 0;
  const bool vec = (C % 4 == 0) && aligned16(A) && aligned16(out);
  const int lpr = pick_lpr(vec ? C / 4 : C);
  LAUNCH_ROW(k_dense_rownorm_fwd, vec, lpr, R, as_stream(stream), A, R, C, lpr, out, invd, ge1);
  CGC_RETURN_IF_LAUNCH_FAILED();
  return 0;
}
extern "C" int cgc_dense_rownorm_bwd(const float* dOut, const float* Anorm, const float* invd, const float* ge1, int R, int C,
                                     float* dA, cgc_stream_t stream) {
  if (R <= 0 || C <= 0) return 0;
  const bool vec = (C % 4 == 0) && aligned16(dOut) && aligned16(Anorm) && aligned16(dA);
  const int lpr = pick_lpr(vec ? C / 4 : C);
  LAUNCH_ROW(k_dense_rownorm_bwd, vec, lpr, R, as_stream(stream), dOut, Anorm, invd, ge1, R, C, lpr, dA);
  CGC_RETURN_IF_LAUNCH_FAILED();
  return 0;
}
extern "C" int cgc_dense_renorm_fwd(const float* A, int R, int C, float p, float* out, cgc_stream_t stream) {
  if (R <= 0 || C <= 0) return 0;
  const bool vec = (C % 4 == 0) && aligned16(A) && aligned16(out);
  const int lpr = pick_lpr(vec ? C / 4 : C);
  LAUNCH_ROW(k_dense_renorm_fwd, vec, lpr, R, as_stream(stream), A, R, C, lpr, p, out);
  CGC_RETURN_IF_LAUNCH_FAILED();
  return 0;
}
extern "C" int cgc_dense_renorm_bwd(const float* A, const float* dOut, int R, int C, float p, float* dA, cgc_stream_t stream) {
  if (R <= 0 || C <= 0) return 0;
  const bool vec = (C % 4 == 0) && aligned16(A) && aligned16(dOut) && aligned16(dA);
  const int lpr = pick_lpr(vec ? C / 4 : C);
  LAUNCH_ROW(k_dense_renorm_bwd, vec, lpr, R, as_stream(stream), A, dOut, R, C, lpr, p, dA);
  CGC_RETURN_IF_LAUNCH_FAILED();
  return 0;
}

extern "C" int cgc_adj_prep_fwd(const float* A, int R, int C, float p, float* At, float* An, float* invd, float* ge1, cgc_stream_t stream) {
  if (R <= 0 || C <= 0) return 0;
  if (p >= 0.f && At == nullptr) return CGC_EINVAL;
  const bool vec = (C % 4 == 0) && aligned16(A) && aligned16(An) && (At == nullptr || aligned16(At));
  ColCfg cfg = col_cfg(R, C, vec);
  if (cfg.ok && cfg.maxj <= 8) {
    cfg.blocks = row_blocks(R, cfg.lpr);
    DISPATCH_COL(k_adj_prep_fwd_reg, cfg, 0, as_stream(stream), A, R, C, cfg.lpr, p, At, An, invd, ge1);
    CGC_RETURN_IF_LAUNCH_FAILED();
    return 0;
  }
  const int lpr = pick_lpr(vec ? C / 4 : C);
  LAUNCH_ROW(k_adj_prep_fwd, vec, lpr, R, as_stream(stream), A, R, C, lpr, p, At, An, invd, ge1);
  CGC_RETURN_IF_LAUNCH_FAILED();
  return 0;
}
extern "C" int cgc_adj_prep_bwd(const float* A, const float* An, const float* invd, const float* ge1, const float* gAn, const float* gAt,
                                int R, int C, float p, float* dA, cgc_stream_t stream) {
  if (R <= 0 || C <= 0) return 0;
  const bool vec = (C % 4 == 0) && aligned16(A) && aligned16(An) && aligned16(gAn) && aligned16(dA) && (gAt == nullptr || aligned16(gAt));
  ColCfg cfg = col_cfg(R, C, vec);
  if (cfg.ok && cfg.maxj <= 8) {
    cfg.blocks = row_blocks(R, cfg.lpr);
    DISPATCH_COL(k_adj_prep_bwd_reg, cfg, 0, as_stream(stream), A, An, invd, ge1, gAn, gAt, R, C, cfg.lpr, p, dA);
    CGC_RETURN_IF_LAUNCH_FAILED();
    return 0;
  }
  const int lpr = pick_lpr(vec ? C / 4 : C);
  LAUNCH_ROW(k_adj_prep_bwd, vec, lpr, R, as_stream(stream), A, An, invd, ge1, gAn, gAt, R, C, lpr, p, dA);
  CGC_RETURN_IF_LAUNCH_FAILED();
  return 0;
}
