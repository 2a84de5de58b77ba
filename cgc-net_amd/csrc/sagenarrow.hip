// Backward of a NARROW SAGE projection (hidden width 20 -> 20) as one kernel:  given d y of  y = BN(act(l2norm(agg W + b)))
// it forms  d h  (BatchNorm, activation and L2-norm backward, as cgc_bn_act_l2_bwd)  and, without writing it,
//   d agg [n, fin] = d h W^T,     d W [fin, F] = agg^T d h,     d b [F] = colsum(d h).
//
// Reference: the backward of DenseSAGEConv + relu + BatchNorm inside GNN_Module (model/network.py:109-125) for the thirteen
// narrow layers of a step.  Unfused, each of them cost cgc_bn_act_l2_bwd (writes d h), a column-sum reduction, a skinny GEMM
// for d agg, and -- the expensive part -- a 20 x 20 weight gradient contracted over 57.7 k rows as a split-K GEMM on 128 x 32
// tiles (3 % of the tile is output) plus its deterministic combine: 7 launches, ~55 us.  Here: 1 launch + 1 slot reduction.
//
// One wave owns 32 rows.  Layout: lane (f = lane & 31, half = lane >> 5) holds column f of the rows 2s + half, s = 0..15 (16
// registers per operand) -- exactly the B operand of v_mfma_f32_32x32x2_f32 for a contraction over ROWS, so
//   d W [k][f] += sum_rows agg[row][k] * d h[row][f]   runs on the matrix core with agg loaded in the same row-pair layout (lane = k)
// and accumulates in 16 registers per wave over all its row tiles; d b is a per-lane sum.  The per-row dot <hn, d hn> of the
// L2-norm backward is a 5-step shuffle reduction inside each half.  For d agg = d h W^T the tile takes one trip through a
// wave-private LDS strip (rows become lanes: the A operand), W^T sits in registers as B fragments.  Each workgroup leaves one slot
// row [d W | d b] (its four waves folded through LDS); cgc's fixed-order slot reduction folds the slots (deterministic).
#include "common.hpp"
#include "groups.hpp"

#define L2_EPS 1e-12f
typedef float floatx16 __attribute__((ext_vector_type(16)));

// Sum over the 32 lanes of a half (lanes 0-31 / 32-63) delivered to every lane: four DPP steps inside each row of 16 lanes (quad
// swaps, half mirror, mirror -- register-to-register, no LDS round trip) and one cross-row exchange.  __shfl_xor compiles to
// ds_bpermute_b32 for every step; with 16 row values per tile that was 80 dependent LDS round trips per wave.
template <int CTRL>
__device__ __forceinline__ float dpp_step(float v) {
  return v + __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float half_sum(float v) {
  v = dpp_step<0xB1>(v);      // quad_perm [1,0,3,2]
  v = dpp_step<0x4E>(v);      // quad_perm [2,3,0,1]
  v = dpp_step<0x141>(v);     // row_half_mirror
  v = dpp_step<0x140>(v);     // row_mirror
  return v + __shfl_xor(v, 16);
}

#define SN_LDT 33       // LDS row stride of the transposition strip (words): conflict-free column reads

template <int FS, int ACT>       // FS = ceil(F / 2): MFMA steps of the d agg product; ACT: activation code (compile time: no per-element branch)
__global__ __launch_bounds__(256) void k_sage_narrow_bwd(const SnBwdPtrs p0, const SnBwdPtrs p1, int ldy, int n, int F, int /*act*/,
                                                         int normalize, int mode, float inv_count, int lda, int fin, int ldd, int tiles) {
  const bool second = blockIdx.y != 0;
  const float* __restrict__ dy = second ? p1.dy : p0.dy;
  const float* __restrict__ hn = second ? p1.hn : p0.hn;
  const float* __restrict__ rinv = second ? p1.rinv : p0.rinv;
  const float* __restrict__ mean = second ? p1.mean : p0.mean;
  const float* __restrict__ istd = second ? p1.istd : p0.istd;
  const float* __restrict__ gamma = second ? p1.gamma : p0.gamma;
  const float* __restrict__ sums = second ? p1.sums : p0.sums;
  const float* __restrict__ agg = second ? p1.agg : p0.agg;
  const float* __restrict__ W = second ? p1.W : p0.W;
  float* __restrict__ dagg = second ? p1.dagg : p0.dagg;
  float* __restrict__ ws = second ? p1.ws : p0.ws;
  __shared__ float strip[4][17 * 64];              // per wave: the 32 x 33 transposition strip; at the end the [17][64] exchange buffer
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int f = lane & 31, half = lane >> 5;
  const bool fok = f < F;
  // per-column constants of the BatchNorm backward: do = ca*dy - cb - xhat*cc
  float ca = 1.f, cb = 0.f, cc = 0.f, mu = 0.f, is = 0.f;
  if (mode != 0 && fok) {
    ca = gamma[f] * istd[f];
    if (mode == 2) {
      cb = ca * sums[f] * inv_count;
      cc = ca * sums[F + f] * inv_count;
      mu = mean[f];
      is = istd[f];
    }
  }
  // B fragments of d agg = d h W^T: lane (k = f index, half) holds W[k][2t + half]
  float wb[FS];
#pragma unroll
  for (int t = 0; t < FS; ++t) {
    const int ff = 2 * t + half;
    wb[t] = (dagg != nullptr && f < fin && ff < F) ? W[(size_t)f * F + ff] : 0.f;
  }
  floatx16 dw;
#pragma unroll
  for (int r = 0; r < 16; ++r) dw[r] = 0.f;
  float db = 0.f;
  float* __restrict__ st = strip[wave];

  const int fcl = fok ? f : 0, facl = f < fin ? f : 0;
  for (int tile = blockIdx.x * 4 + wave; tile < tiles; tile += gridDim.x * 4) {
    const int row0 = tile * 32;
    float dh[16], ag[16];
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      // (every load unconditional, from a clamped in-range address, then masked: a predicated load compiles to a branch per load
      // and the 48 loads of a tile no longer travel together)
      const int row = row0 + 2 * s + half;
      const bool ok = row < n;
      const size_t rc = ok ? row : n - 1;
      const float g_ = dy[rc * ldy + fcl], x_ = hn[rc * F + fcl], a_ = agg[rc * lda + facl];
      const float g = (ok && fok) ? g_ : 0.f;
      const float x = (ok && fok) ? x_ : 0.f;
      ag[s] = (ok && f < fin) ? a_ : 0.f;
      float go = ca * g;
      if (mode == 2) go = go - cb - (act_fwd(x, ACT) - mu) * is * cc;
      go *= act_bwd(x, ACT);
      if (!(ok && fok)) go = 0.f;
      float o = go;
      if (normalize) {
        const float dot = half_sum(x * go);                                    // over the 32 columns of this row (inside the half)
        const float r_ = rinv[rc];
        const float r = ok ? r_ : 1.f;
        const bool clamped = !(r < 1.f / L2_EPS);                              // ||h|| <= eps: F.normalize divided by eps
        o = clamped ? go * (1.f / L2_EPS) : r * (go - x * dot);
      }
      dh[s] = o;
      db += o;
    }
    // d W [k][f] += agg^T d h: contraction over the 32 rows, two per MFMA
#pragma unroll
    for (int s = 0; s < 16; ++s) dw = __builtin_amdgcn_mfma_f32_32x32x2f32(ag[s], dh[s], dw, 0, 0, 0);
    if (dagg != nullptr) {
      // rows become lanes: park the tile ([row][f], row stride 33 words) and read the A fragments d h[row][2t + half]
#pragma unroll
      for (int s = 0; s < 16; ++s) st[(2 * s + half) * SN_LDT + f] = dh[s];
      __builtin_amdgcn_wave_barrier();
      floatx16 da;
#pragma unroll
      for (int r = 0; r < 16; ++r) da[r] = 0.f;
#pragma unroll
      for (int t = 0; t < FS; ++t) da = __builtin_amdgcn_mfma_f32_32x32x2f32(st[f * SN_LDT + 2 * t + half], wb[t], da, 0, 0, 0);
      __builtin_amdgcn_wave_barrier();
      // D[row][k]: lane = k, register r = row (r&3) + 8(r>>2) + 4*half
      if (f < fin) {
        if (row0 + 32 <= n) {        // whole tile inside the rows (wave-uniform): no per-store predicate
#pragma unroll
          for (int r = 0; r < 16; ++r) dagg[(size_t)(row0 + (r & 3) + 8 * (r >> 2) + 4 * half) * ldd + f] = da[r];
        } else {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int row = row0 + (r & 3) + 8 * (r >> 2) + 4 * half;
            if (row < n) dagg[(size_t)row * ldd + f] = da[r];
          }
        }
      }
    }
  }
  // the four waves' partial d W / d b are folded through LDS (fixed order); one slot row per workgroup:
  // [d W (fin x F, row-major) | d b (F)]
  __syncthreads();                                   // the strips are free now: reuse them as the exchange buffer
  float* __restrict__ xch = &strip[0][0];            // [wave][17][64]
  static_assert(32 * SN_LDT <= 17 * 64, "transposition strip exceeds the per-wave LDS");
#pragma unroll
  for (int r = 0; r < 16; ++r) xch[(wave * 17 + r) * 64 + lane] = dw[r];
  xch[(wave * 17 + 16) * 64 + lane] = db;
  __syncthreads();
  if (wave == 0) {
    float* __restrict__ slot = ws + (size_t)blockIdx.x * (fin * F + F);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float v = (xch[(0 * 17 + r) * 64 + lane] + xch[(1 * 17 + r) * 64 + lane]) + (xch[(2 * 17 + r) * 64 + lane] + xch[(3 * 17 + r) * 64 + lane]);
      const int k = (r & 3) + 8 * (r >> 2) + 4 * half;
      if (fok && k < fin) slot[k * F + f] = v;
    }
    float d = (xch[(0 * 17 + 16) * 64 + lane] + xch[(1 * 17 + 16) * 64 + lane]) + (xch[(2 * 17 + 16) * 64 + lane] + xch[(3 * 17 + 16) * 64 + lane]);
    d += __shfl_xor(d, 32);
    if (fok && half == 0) slot[fin * F + f] = d;
  }
}

int launch_reduce_slots_f32(const float* ws, int slots, int width, float* out, hipStream_t stream);   // rowops.hip

static int sn_grid(int n) {
  const int tiles = ceil_div(n, 32);
  int g = ceil_div(tiles, 4);
  return g < 1024 ? g : 1024;
}

extern "C" int64_t cgc_sage_narrow_ws_floats(int n, int fin, int F) { return (int64_t)sn_grid(n > 0 ? n : 1) * (fin * F + F); }

// dy [n,F] (row stride ldy), hn [n,F], rinv [n], BatchNorm vectors and sums [2,F] exactly as cgc_bn_act_l2_bwd (mode 0 / 1 / 2);
// agg [n,fin] (row stride lda), W [fin,F].  Out: dagg [n,fin] (NULL: not needed), dwdb [fin*F + F] = d W row-major followed by d b.
// ws: cgc_sage_narrow_ws_floats(n, fin, F) floats.  Envelope fin <= 32, F <= 32; otherwise CGC_EINVAL, nothing launched.
extern "C" int cgc_sage_narrow_bwd(const float* dy, int ldy, const float* hn, const float* rinv, int n, int F, int act, int normalize,
                                   int mode, const float* mean, const float* istd, const float* gamma, const float* sums, double count,
                                   const float* agg, int lda, int fin, const float* W, float* dagg, float* dwdb, float* ws,
                                   cgc_stream_t stream_) {
  return cgc_sage_narrow_bwd_ld(dy, ldy, hn, rinv, n, F, act, normalize, mode, mean, istd, gamma, sums, count, agg, lda, fin, W, dagg, fin,
                                dwdb, ws, stream_);
}

// the same with a row stride for dagg (>= fin): the two blocks of a level write their halves of one [n, fin_e + fin_p] gradient
// One launch for up to two layers of equal shape (the embedding and the assignment block of a level run in lockstep): sets g[0..ng)
int sage_narrow_bwd_groups(const SnBwdPtrs* g, float* const* dwdb, int ng, int ldy, int n, int F, int act, int normalize, int mode, double count,
                           int lda, int fin, int ldd, hipStream_t st) {
  if (F <= 0 || fin <= 0) return 0;
  if (F > 32 || fin > 32 || ng < 1 || ng > 2) return CGC_EINVAL;
  for (int i = 0; i < ng; ++i)
    if (g[i].ws == nullptr || dwdb[i] == nullptr || (g[i].dagg != nullptr && ldd < fin)) return CGC_EINVAL;
  const int width = fin * F + F;
  if (n <= 0) {
    for (int i = 0; i < ng; ++i) (void)hipMemsetAsync(dwdb[i], 0, sizeof(float) * width, st);
    return 0;
  }
  const int tiles = ceil_div(n, 32), grid = sn_grid(n);
  const float inv_count = (float)(1.0 / count);
  const int fs = (F + 1) / 2;
  const SnBwdPtrs& g1 = g[ng - 1];
#define SN_LAUNCH(FS_, ACT_)                                                                                                       \
  hipLaunchKernelGGL((k_sage_narrow_bwd<FS_, ACT_>), dim3(grid, ng), dim3(256), 0, st, g[0], g1, ldy, n, F, act, normalize, mode, inv_count, \
                     lda, fin, ldd, tiles)
#define SN_ACT(FS_)                                                                                        \
  switch (act) {                                                                                           \
    case CGC_ACT_RELU: SN_LAUNCH(FS_, CGC_ACT_RELU); break;                                                \
    case CGC_ACT_ELU: SN_LAUNCH(FS_, CGC_ACT_ELU); break;                                                  \
    case CGC_ACT_LEAKYRELU: SN_LAUNCH(FS_, CGC_ACT_LEAKYRELU); break;                                      \
    default: SN_LAUNCH(FS_, CGC_ACT_IDENTITY); break;                                                      \
  }
  if (fs <= 8) { SN_ACT(8); } else if (fs <= 10) { SN_ACT(10); } else { SN_ACT(16); }
#undef SN_ACT
#undef SN_LAUNCH
  CGC_RETURN_IF_LAUNCH_FAILED();
  return launch_reduce_slots_f32_pair(g[0].ws, dwdb[0], g1.ws, dwdb[ng - 1], ng, grid, width, st);
}

extern "C" int cgc_sage_narrow_bwd_ld(const float* dy, int ldy, const float* hn, const float* rinv, int n, int F, int act, int normalize,
                                      int mode, const float* mean, const float* istd, const float* gamma, const float* sums, double count,
                                      const float* agg, int lda, int fin, const float* W, float* dagg, int ldd, float* dwdb, float* ws,
                                      cgc_stream_t stream_) {
  const SnBwdPtrs g{dy, hn, rinv, mean, istd, gamma, sums, agg, W, dagg, ws};
  float* const out[1] = {dwdb};
  return sage_narrow_bwd_groups(&g, out, 1, ldy, n, F, act, normalize, mode, count, lda, fin, ldd, as_stream(stream_));
}

// ------------------------------------------------------------------------------------------------------------------------
// Forward of a narrow SAGE projection: hn = l2norm(agg W + b) + the BatchNorm statistics of act(hn), one kernel (was a
// short-K GEMM + cgc_l2norm_act_stats).  One wave owns 32 rows: the product runs as K/2 MFMAs with the rows as M (agg loaded
// row-per-lane: the A operand) and lands in the lane = column layout, where the row norm is a 5-step shuffle reduction per
// register and the column statistics are per-lane sums.  One slot row [2, F] per workgroup (four waves folded through LDS).
template <int KS, int ACT>
__global__ __launch_bounds__(256) void k_sage_narrow_fwd(const SnFwdPtrs p0, const SnFwdPtrs p1, int lda, int n, int K, int F, int normalize,
                                                         int /*act*/, int tiles) {
  const bool second = blockIdx.y != 0;
  const float* __restrict__ agg = second ? p1.agg : p0.agg;
  const float* __restrict__ W = second ? p1.W : p0.W;
  const float* __restrict__ bias = second ? p1.bias : p0.bias;
  float* __restrict__ hn = second ? p1.hn : p0.hn;
  float* __restrict__ rinv_out = second ? p1.rinv : p0.rinv;
  float* __restrict__ ws = second ? p1.ws : p0.ws;
  __shared__ double xch[4][2][32];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int f = lane & 31, half = lane >> 5;
  const bool fok = f < F;
  float bw[KS];
#pragma unroll
  for (int s = 0; s < KS; ++s) {
    const int k = 2 * s + half;
    bw[s] = (fok && k < K) ? W[(size_t)k * F + f] : 0.f;
  }
  const float bia = (fok && bias != nullptr) ? bias[f] : 0.f;
  double s1 = 0.0, s2 = 0.0;     // BatchNorm statistics in double from the first addition on (rowops.hip: col_reduce_store_f64)
  for (int tile = blockIdx.x * 4 + wave; tile < tiles; tile += gridDim.x * 4) {
    const int row0 = tile * 32;
    const float* __restrict__ a = agg + (size_t)min(row0 + f, n - 1) * lda;        // A operand: lane = row
    float av[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const int k = 2 * s + half;
      const float ld = a[k < K ? k : 0];             // (unconditional load + mask: a predicated load is a branch)
      av[s] = k < K ? ld : 0.f;
    }
    floatx16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = bia;
#pragma unroll
    for (int s = 0; s < KS; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s], bw[s], acc, 0, 0, 0);
    // lane = column f, register r = row (r&3) + 8(r>>2) + 4*half
    float vv[16], rr[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      float v = fok ? acc[r] : 0.f;
      float rin = 1.f;
      if (normalize) {
        const float q = half_sum(v * v);
        rin = 1.f / fmaxf(sqrtf(q), L2_EPS);
        v *= rin;
      }
      vv[r] = v;
      rr[r] = rin;
    }
    if (row0 + 32 <= n) {          // whole tile inside the rows (wave-uniform): two lane predicates for 32 stores instead of 32
      if (fok) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = row0 + (r & 3) + 8 * (r >> 2) + 4 * half;
          const double o = (double)act_fwd(vv[r], ACT);
          s1 += o;
          s2 = fma(o, o, s2);
          hn[(size_t)row * F + f] = vv[r];
        }
      }
      if (f == 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) rinv_out[row0 + (r & 3) + 8 * (r >> 2) + 4 * half] = rr[r];
      }
    } else {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = row0 + (r & 3) + 8 * (r >> 2) + 4 * half;
        if (row < n) {
          if (fok) {
            const double o = (double)act_fwd(vv[r], ACT);
            s1 += o;
            s2 = fma(o, o, s2);
            hn[(size_t)row * F + f] = vv[r];
          }
          if (f == 0) rinv_out[row] = rr[r];
        }
      }
    }
  }
  if (ws == nullptr) return;
  s1 += __shfl_xor(s1, 32);
  s2 += __shfl_xor(s2, 32);
  if (half == 0) { xch[wave][0][f] = s1; xch[wave][1][f] = s2; }
  __syncthreads();
  if (wave == 0 && half == 0 && fok) {
    double* __restrict__ slot = reinterpret_cast<double*>(ws) + (size_t)blockIdx.x * 2 * F;
    slot[f] = (xch[0][0][f] + xch[1][0][f]) + (xch[2][0][f] + xch[3][0][f]);
    slot[F + f] = (xch[0][1][f] + xch[1][1][f]) + (xch[2][1][f] + xch[3][1][f]);
  }
}

extern "C" int cgc_stats_blocks(int n, int F);
// hn = l2norm(agg W + bias) (+ BatchNorm statistics) for up to two layers of equal shape in one launch each
int sage_narrow_fwd_groups(const SnFwdPtrs* g, const SnFwdBn* bn, int ng, int lda, int n, int K, int F, int normalize, int act, int stats,
                           double count, hipStream_t st) {
  if (F <= 0) return 0;
  if (K <= 0 || K > 32 || F > 32 || ng < 1 || ng > 2) return CGC_EINVAL;
  for (int i = 0; i < ng; ++i)
    if (stats && (g[i].ws == nullptr || bn[i].mean == nullptr || bn[i].istd == nullptr)) return CGC_EINVAL;
  int slots = 0;
  if (n > 0) {
    const int tiles = ceil_div(n, 32);
    int grid = ceil_div(tiles, 4);
    const int cap = cgc_stats_blocks(n, F);
    if (grid > 1024) grid = 1024;
    if (stats && grid > cap) grid = cap > 0 ? cap : 1;
    SnFwdPtrs a0 = g[0], a1 = g[ng - 1];
    if (!stats) a0.ws = a1.ws = nullptr;
    const int ks = (K + 1) / 2;
#define SNF_LAUNCH(KS_, ACT_) \
  hipLaunchKernelGGL((k_sage_narrow_fwd<KS_, ACT_>), dim3(grid, ng), dim3(256), 0, st, a0, a1, lda, n, K, F, normalize, act, tiles)
#define SNF_ACT(KS_)                                                                                       \
  switch (act) {                                                                                           \
    case CGC_ACT_RELU: SNF_LAUNCH(KS_, CGC_ACT_RELU); break;                                               \
    case CGC_ACT_ELU: SNF_LAUNCH(KS_, CGC_ACT_ELU); break;                                                 \
    case CGC_ACT_LEAKYRELU: SNF_LAUNCH(KS_, CGC_ACT_LEAKYRELU); break;                                     \
    default: SNF_LAUNCH(KS_, CGC_ACT_IDENTITY); break;                                                     \
  }
    if (ks <= 8) { SNF_ACT(8); } else if (ks <= 10) { SNF_ACT(10); } else { SNF_ACT(16); }
#undef SNF_ACT
#undef SNF_LAUNCH
    CGC_RETURN_IF_LAUNCH_FAILED();
    slots = grid;
  }
  if (stats) {
    StatsFinPtrs f[2];
    for (int i = 0; i < ng; ++i)
      f[i] = StatsFinPtrs{g[i].ws, bn[i].running_mean, bn[i].running_var, bn[i].mean, bn[i].istd, reinterpret_cast<long long*>(bn[i].nbt),
                          bn[i].eps, bn[i].momentum};
    return launch_stats_finalize_groups(f, ng, slots, F, count, st);
  }
  return 0;
}

// hn [n,F] (contiguous) = l2norm(agg [n,K] (row stride lda) @ W [K,F] + bias), rinv [n]; stats != 0: mean / istd / running statistics /
// num_batches_tracked as cgc_l2norm_act_bn (ws: its slot area).  Envelope K <= 32, F <= 32; otherwise CGC_EINVAL, nothing launched.
extern "C" int cgc_sage_narrow_fwd(const float* agg, int lda, const float* W, const float* bias, int n, int K, int F, int normalize,
                                   int act, float* hn, float* rinv, int stats, float* ws, double count, float eps, float momentum,
                                   float* running_mean, float* running_var, int64_t* num_batches_tracked, float* mean, float* istd,
                                   cgc_stream_t stream_) {
  const SnFwdPtrs g{agg, W, bias, hn, rinv, ws};
  const SnFwdBn bn{eps, momentum, running_mean, running_var, num_batches_tracked, mean, istd};
  return sage_narrow_fwd_groups(&g, &bn, 1, lda, n, K, F, normalize, act, stats, count, as_stream(stream_));
}
