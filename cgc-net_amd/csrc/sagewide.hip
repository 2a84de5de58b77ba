// The wide SAGE projection of the assignment block as ONE kernel:  hn = l2norm(agg W + b) (+ the activation statistics of the
// BatchNorm that follows) for a NARROW input (K = hidden width, 20) and a WIDE output (F = cluster count, 1140 / 1600).
//
// Reference: DenseSAGEConv(assign_hidden, assign_dim) -> relu -> BatchNorm inside GNN_Module (model/network.py:114-116;
// torch-geometric 1.2.1 DenseSAGEConv: out = agg @ W + b, F.normalize(out, p=2, dim=-1)).  Unfused this is a short-K GEMM that
// writes h [Ntot, F] (263 MB at C3) plus cgc_l2norm_act_stats, which reads it back and writes hn: 790 MB of HBM traffic for
// 2.6 GFLOP (227 us).  Here a workgroup owns 32 ROWS AND ALL F COLUMNS, so the row norm is available before anything is stored
// and only hn is ever written (263 MB).
//
//   * 4 waves; wave w owns NTW column tiles of 32 (F <= 4*NTW*32).  Its slice of W sits in registers as MFMA B fragments for
//     the whole kernel (K/2 x NTW registers); the 32 x K tile of agg is the A fragment (K/2 registers, reloaded per row tile).
//   * v_mfma_f32_32x32x2_f32, accumulator layout lane = column, register r = row (r&3)+8(r>>2)+4(lane>>5), initialised with
//     the bias.  The rank-K product is so cheap (K/2 MFMAs per 32x32 tile) that it is computed TWICE instead of being kept:
//     pass 1 only accumulates the squared row norms (16 registers), one cross-lane + cross-wave reduction per row tile gives
//     1/||h||, pass 2 recomputes every tile, scales it, adds act(hn), act(hn)^2 to per-lane COLUMN sums (the lane owns the
//     column: no cross-lane work) and stores 128-byte row segments (parking the tile in LDS for 16-byte stores was measured:
//     the extra address arithmetic costs the second wave per SIMD, 186 us instead of 125 us).
//   * persistent workgroups stride over the row tiles; at the end each leaves one slot row [2, F] of column sums, folded in a
//     fixed order in fp64 by k_stats_finalize (rowops.hip): deterministic.
#include "common.hpp"

#define L2_EPS 1e-12f
typedef float floatx16 __attribute__((ext_vector_type(16)));

template <int KS, int NTW>
__global__ __launch_bounds__(256, 1) void k_sage_wide_fwd(const float* __restrict__ agg, int lda, const float* __restrict__ W,
                                                          const float* __restrict__ bias, int n, int K, int F, int normalize, int act,
                                                          float* __restrict__ hn, int ldh, float* __restrict__ rinv_out,
                                                          float* __restrict__ ws, int row_tiles) {
  __shared__ float red[4][32];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int l31 = lane & 31, lhi = lane >> 5;
  const int c_base = wave * NTW * 32;

  // BatchNorm statistics without the E[o^2] - E[o]^2 cancellation (rowops.hip: col_reduce_store_f64 tells why it matters).  Adding
  // every o and o^2 in double costs this kernel its second wave per SIMD (36 more registers: 244 -> 260; 123 -> 180 us), so the
  // sums here are SHIFTED fp32 sums: d = o - c with c = the column's mean over the workgroup's first row tile, s1 = sum d,
  // s2 = sum d^2 -- deviations are formed directly, nothing large cancels -- and only the last step is in double: a lane's
  // (count, mean = c + s1 / count, M2 = s2 - s1^2 / count) is turned into the slot format (sum o, sum o^2).
  float bw[NTW][KS], bia[NTW], s1[NTW], s2[NTW], shift[NTW];
  int seen = 0;                  // rows this lane has folded in (the same for all its column tiles)
#pragma unroll
  for (int t = 0; t < NTW; ++t) {
    const int col = c_base + t * 32 + l31;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const int k = 2 * s + lhi;
      bw[t][s] = (col < F && k < K) ? W[(size_t)k * F + col] : 0.f;
    }
    bia[t] = (col < F && bias != nullptr) ? bias[col] : 0.f;
    s1[t] = s2[t] = shift[t] = 0.f;
  }

  for (int rt = blockIdx.x; rt < row_tiles; rt += gridDim.x) {
    const int row0 = rt * 32;
    float av[KS];
    {
      const float* __restrict__ a = agg + (size_t)min(row0 + l31, n - 1) * lda;
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        const int k = 2 * s + lhi;
        av[s] = k < K ? a[k] : 0.f;
      }
    }
    float rin[16];
    if (normalize) {
      float q[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) q[r] = 0.f;
#pragma unroll
      for (int t = 0; t < NTW; ++t) {
        floatx16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = bia[t];
#pragma unroll
        for (int s = 0; s < KS; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s], bw[t][s], acc, 0, 0, 0);
        const bool colok = c_base + t * 32 + l31 < F;
#pragma unroll
        for (int r = 0; r < 16; ++r) q[r] = colok ? fmaf(acc[r], acc[r], q[r]) : q[r];
        __builtin_amdgcn_sched_barrier(0);            // one tile at a time: keeps a single accumulator live (no spills)
      }
#pragma unroll
      for (int r = 0; r < 16; ++r)
        for (int o = 16; o > 0; o >>= 1) q[r] += __shfl_xor(q[r], o);
      if (l31 == 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) red[wave][(r & 3) + 8 * (r >> 2) + 4 * lhi] = q[r];
      }
      __syncthreads();
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * lhi;
        const float tot = (red[0][row] + red[1][row]) + (red[2][row] + red[3][row]);
        rin[r] = 1.f / fmaxf(sqrtf(tot), L2_EPS);
      }
      __syncthreads();                               // red is rewritten by the next row tile
    } else {
#pragma unroll
      for (int r = 0; r < 16; ++r) rin[r] = 1.f;
    }
    if (wave == 0 && l31 == 0) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = row0 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
        if (row < n) rinv_out[row] = rin[r];
      }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) seen += (row0 + (r & 3) + 8 * (r >> 2) + 4 * lhi) < n ? 1 : 0;
#pragma unroll
    for (int t = 0; t < NTW; ++t) {
      floatx16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = bia[t];
#pragma unroll
      for (int s = 0; s < KS; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s], bw[t][s], acc, 0, 0, 0);
      const int col = c_base + t * 32 + l31;
      if (col < F) {           // (both halves of the wave hold the same column: the exchange below has its partner)
        if (rt == (int)blockIdx.x) {      // the workgroup's first row tile fixes the shift: this column's mean over the tile's rows
          float t1 = 0.f;
#pragma unroll
          for (int r = 0; r < 16; ++r) t1 += (row0 + (r & 3) + 8 * (r >> 2) + 4 * lhi) < n ? act_fwd(acc[r] * rin[r], act) : 0.f;
          t1 += __shfl_xor(t1, 32);
          shift[t] = t1 / (float)min(32, n - row0);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = row0 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
          if (row < n) {
            const float v = acc[r] * rin[r];
            const float d = act_fwd(v, act) - shift[t];
            s1[t] += d;
            s2[t] = fmaf(d, d, s2[t]);
            hn[(size_t)row * ldh + col] = v;
          }
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  if (ws != nullptr) {
    double* slot = reinterpret_cast<double*>(ws) + (size_t)blockIdx.x * 2 * F;
#pragma unroll
    for (int t = 0; t < NTW; ++t) {
      // this half's rows as (count, mean, M2) in double, then sum o = count * mean and sum o^2 = M2 + count * mean^2
      const double cn = (double)seen, c = (double)shift[t], a1 = (double)s1[t], a2 = (double)s2[t];
      const double mean = seen > 0 ? c + a1 / cn : 0.0, M2 = seen > 0 ? a2 - a1 * a1 / cn : 0.0;
      double so = cn * mean, soo = M2 + cn * mean * mean;
      so += __shfl_xor(so, 32);
      soo += __shfl_xor(soo, 32);
      const int col = c_base + t * 32 + l31;
      if (lhi == 0 && col < F) {
        slot[col] = so;
        slot[F + col] = soo;
      }
    }
  }
}

extern "C" int cgc_stats_blocks(int n, int F);
int launch_stats_finalize(const float* ws, int slots, int F, double count, float eps, float momentum, float* running_mean,
                          float* running_var, float* mean, float* istd, int64_t* nbt, hipStream_t stream);   // rowops.hip

// hn [n,F] (row stride ldh) = l2norm(agg [n,K] (row stride lda) @ W [K,F] + bias) ; rinv [n] = 1/max(||.||,1e-12) (1 if !normalize);
// stats != 0: additionally mean / istd / running statistics / num_batches_tracked as cgc_l2norm_act_bn (ws: its slot area).
// Returns CGC_EINVAL (nothing launched) for shapes outside the kernel's envelope (K > 32 or F > 1664): the caller then runs
// cgc_gemm_f32 + cgc_l2norm_act_bn.
extern "C" int cgc_sage_wide_fwd(const float* agg, int lda, const float* W, const float* bias, int n, int K, int F, int normalize, int act,
                                 float* hn, int ldh, float* rinv, int stats, float* ws, double count, float eps, float momentum,
                                 float* running_mean, float* running_var, int64_t* num_batches_tracked, float* mean, float* istd,
                                 cgc_stream_t stream_) {
  hipStream_t st = as_stream(stream_);
  if (F <= 0) return 0;
  if (K <= 0 || K > 32 || F > 4 * 13 * 32 || ldh < F) return CGC_EINVAL;
  if (stats && (ws == nullptr || mean == nullptr || istd == nullptr)) return CGC_EINVAL;
  int slots = 0;
  if (n > 0) {
    const int row_tiles = ceil_div(n, 32);
    int wgs = row_tiles < 512 ? row_tiles : 512;
    const int cap = cgc_stats_blocks(n, F);                 // the caller's slot area holds this many slot rows
    if (stats && wgs > cap) wgs = cap > 0 ? cap : 1;
    float* wsp = stats ? ws : nullptr;
    const int ks = K <= 16 ? 8 : K <= 20 ? 10 : 16;
    const bool narrow = F <= 4 * 9 * 32;
#define SW_LAUNCH(KS_, NTW_)                                                                                              \
  hipLaunchKernelGGL((k_sage_wide_fwd<KS_, NTW_>), dim3(wgs), dim3(256), 0, st, agg, lda, W, bias, n, K, F, normalize, act, hn, ldh, \
                     rinv, wsp, row_tiles)
    if (narrow) {
      if (ks == 8) SW_LAUNCH(8, 9); else if (ks == 10) SW_LAUNCH(10, 9); else SW_LAUNCH(16, 9);
    } else {
      if (ks == 8) SW_LAUNCH(8, 13); else if (ks == 10) SW_LAUNCH(10, 13); else SW_LAUNCH(16, 13);
    }
#undef SW_LAUNCH
    CGC_RETURN_IF_LAUNCH_FAILED();
    slots = wgs;
  }
  if (stats) return launch_stats_finalize(ws, slots, F, count, eps, momentum, running_mean, running_var, mean, istd, num_batches_tracked, st);
  return 0;
}
