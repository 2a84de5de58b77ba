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
#include <stdlib.h>

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

// ------------------------------------------------------------------------------------------------------------------------
// Round 4: the same operator as ONE pass of the matrix cores.  Timing ablations of the kernel above (tools/variant_lib.sh,
// -DCGC_X_SW_*): 149 us with everything, 133 without the stores, 126 without the statistics, 103 without either -- it is bound by
// its own arithmetic: two rank-K passes of dependent 10-MFMA chains with one accumulator live (W in 90 registers leaves room for
// no more), not by the 263 MB it writes.  Here W lives in LDS instead ([K][F] k-major: the B fragment of MFMA step s is one
// conflict-free ds_read_b32), EIGHT waves share it and a 32-row tile (wave w owns NTW column tiles; two waves per SIMD inside one
// workgroup, one workgroup per CU), and the freed registers hold ALL of the wave's NTW accumulators: the product is computed once,
// the NTW chains are independent (MFMAs issue back to back), the row norms come from the kept accumulators, and the scaled values are
// stored from them.  Statistics as above (shifted fp32 sums, the last step in double).
template <int KS, int NTW, int ACT>      // ACT: activation code at compile time (a run-time switch per element splits every basic block)
__global__ __launch_bounds__(512, 1) void k_sage_wide_fwd8(const float* __restrict__ agg, int lda, const float* __restrict__ W,
                                                           const float* __restrict__ bias, int n, int K, int F, int Fp, int normalize,
                                                           int /*act*/, float* __restrict__ hn, int ldh, float* __restrict__ rinv_out,
                                                           float* __restrict__ ws, int row_tiles) {
  extern __shared__ __attribute__((aligned(16))) float sm8[];
  float* __restrict__ Wl = sm8;                          // [2 KS][Fp]: rows k >= K and columns >= F are zero
  float* __restrict__ red = sm8 + (size_t)2 * KS * Fp;   // [2][8][32]: squared-norm partials, double buffered over row tiles
  float* __restrict__ rstrip = red + 512;                // [8][32]: 1 / ||h|| of the tile's rows, one private copy per wave
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int l31 = lane & 31, lhi = lane >> 5;
  const int c_base = wave * NTW * 32;
  if ((F & 3) == 0 && (reinterpret_cast<uintptr_t>(W) & 15u) == 0) {
    // W -> LDS with every load of a thread in flight at once (a load -> store loop of 50 dependent L2 round trips per thread was
    // a third of the kernel): 16-byte units, then the zero padding (columns F..Fp, rows K..2 KS)
    constexpr int MAXU = (2 * KS * 8 * NTW * 8 + 511) / 512;      // ceil(2 KS * Fp / 4 / 512)
    const int F4 = F >> 2, U = K * F4;
    float4 tmp[MAXU];
#pragma unroll
    for (int j = 0; j < MAXU; ++j) {
      const int u = min((int)threadIdx.x + j * 512, U - 1);
      tmp[j] = reinterpret_cast<const float4*>(W)[u];
    }
#pragma unroll
    for (int j = 0; j < MAXU; ++j) {
      const int u = threadIdx.x + j * 512;
      if (u < U) {
        const int k = u / F4, c = u - k * F4;
        *reinterpret_cast<float4*>(&Wl[(size_t)k * Fp + 4 * c]) = tmp[j];
      }
    }
    for (int i = threadIdx.x; i < 2 * KS * (Fp - F); i += 512) {
      const int k = i / (Fp - F), c = F + i - k * (Fp - F);
      Wl[(size_t)k * Fp + c] = 0.f;
    }
    for (int i = threadIdx.x; i < (2 * KS - K) * F; i += 512) Wl[(size_t)(K + i / F) * Fp + i % F] = 0.f;
  } else {
    for (int i = threadIdx.x; i < 2 * KS * Fp; i += 512) {
      const int k = i / Fp, c = i - k * Fp;
      Wl[i] = (k < K && c < F) ? W[(size_t)k * F + c] : 0.f;
    }
  }
  float bia[NTW], s1[NTW], s2[NTW], shift[NTW];
#pragma unroll
  for (int t = 0; t < NTW; ++t) {
    const int col = c_base + t * 32 + l31;
    bia[t] = (col < F && bias != nullptr) ? bias[col] : 0.f;
    s1[t] = s2[t] = shift[t] = 0.f;
  }
  __syncthreads();
  int seen = 0, par = 0;
  const __amdgpu_buffer_rsrc_t rsrc_hn = __builtin_amdgcn_make_buffer_rsrc(hn, 0, (int)((size_t)n * ldh * 4), 0x00020000);
  float av_next[KS];           // the next row tile's A fragments travel while this one is computed
  {
    const float* __restrict__ a = agg + (size_t)min((int)blockIdx.x * 32 + l31, n - 1) * lda;
#pragma unroll
    for (int s = 0; s < KS; ++s) av_next[s] = a[min(2 * s + lhi, K - 1)];
  }
  for (int rt = blockIdx.x; rt < row_tiles; rt += gridDim.x, par ^= 1) {
    const int row0 = rt * 32;
    float av[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) av[s] = (2 * s + lhi) < K ? av_next[s] : 0.f;
    floatx16 acc[NTW];
#pragma unroll
    for (int t = 0; t < NTW; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][r] = bia[t];
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const float* __restrict__ wrow = Wl + (size_t)(2 * s + lhi) * Fp + c_base + l31;
#pragma unroll
#ifndef CGC_X8_NOMFMA
      for (int t = 0; t < NTW; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s], wrow[t * 32], acc[t], 0, 0, 0);
#else
      for (int t = 0; t < NTW; ++t) acc[t][s] += av[s] * wrow[t * 32];
#endif
    }
    // The next row tile's A fragments are requested HERE -- behind this tile's MFMAs, in front of the norm reduction -- and waited for
    // in front of this tile's stores (the empty asm below).  vmcnt counts loads and stores in issue order: a wait for loads that
    // were issued before ~80 conditional stores (or after them) can only be written as vmcnt(0), i.e. it drains the stores too --
    // at the top of the next tile that exposed the whole store latency once per row tile, in this kernel and in its predecessor
    // (whose 149 us no change to its arithmetic or to its stores would move).  Here the drain finds the previous tile's stores
    // long retired and the fresh loads have the reduction + barrier to arrive.
    {
      const int nrt = rt + (int)gridDim.x < row_tiles ? rt + (int)gridDim.x : rt;
      const float* __restrict__ a = agg + (size_t)min(nrt * 32 + l31, n - 1) * lda;
#pragma unroll
      for (int s = 0; s < KS; ++s) av_next[s] = a[min(2 * s + lhi, K - 1)];
    }
    float rin[16];
    if (normalize) {
      float q[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) q[r] = 0.f;
#pragma unroll
      for (int t = 0; t < NTW; ++t) {
        const bool colok = c_base + t * 32 + l31 < F;
#pragma unroll
        for (int r = 0; r < 16; ++r) q[r] = colok ? fmaf(acc[t][r], acc[t][r], q[r]) : q[r];
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) q[r] = group_sum(q[r], 32);      // over the 32 columns a half wave holds (DPP + one cross-row step)
      float* __restrict__ rd = red + par * 256;
      if (l31 == 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) rd[wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi] = q[r];
      }
      __syncthreads();                              // (one barrier per row tile: the next tile writes the other half of `red`)
      // 1 / ||h|| once per ROW (correctly rounded sqrt and division are ~70 instructions: 16 of them per lane were a third of the
      // kernel): lane l of every wave does row l & 31 and parks it in the wave's own strip; LDS operations of a wave are ordered
      float* __restrict__ rs = rstrip + wave * 32;
      {
        const float tot = ((rd[l31] + rd[32 + l31]) + (rd[64 + l31] + rd[96 + l31])) + ((rd[128 + l31] + rd[160 + l31]) + (rd[192 + l31] + rd[224 + l31]));
        if (lhi == 0) rs[l31] = 1.f / fmaxf(sqrtf(tot), L2_EPS);
      }
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int r = 0; r < 16; ++r) rin[r] = rs[(r & 3) + 8 * (r >> 2) + 4 * lhi];
      __builtin_amdgcn_wave_barrier();
    } else {
#pragma unroll
      for (int r = 0; r < 16; ++r) rin[r] = 1.f;
    }
#pragma unroll
    for (int s = 0; s < KS; ++s) asm volatile("" : "+v"(av_next[s]));      // (the prefetch has to have landed before the first store)
    if (wave == 0 && l31 == 0) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = row0 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
        if (row < n) rinv_out[row] = rin[r];
      }
    }
    // Stores: BUFFER stores off one descriptor over hn (num_records = n * ldh floats: rows past n are dropped by the bounds check, lanes
    // whose column is past F carry an out-of-range offset) -- a lane contributes a loop-invariant byte offset, the row is the
    // instruction's scalar offset: no per-element predicate, no branch, no 64-bit address arithmetic (the flat form cost a compare, an
    // exec-mask branch and a 64-bit multiply-add per stored element: ~120 branches per row tile and wave).
    const bool full = row0 + 32 <= n;            // (all but the last row tile: the statistics need no row mask either)
#pragma unroll
    for (int t = 0; t < NTW; ++t) {
      const int col = c_base + t * 32 + l31;
      const bool colok = col < F;
      const unsigned voff = colok ? (unsigned)(4 * lhi * ldh + col) * 4u : 0x80000000u;
      if (ws != nullptr && rt == (int)blockIdx.x) {     // the workgroup's first row tile fixes the shift (see k_sage_wide_fwd)
        float t1 = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) t1 += (row0 + (r & 3) + 8 * (r >> 2) + 4 * lhi) < n ? act_fwd(acc[t][r] * rin[r], ACT) : 0.f;
        t1 += __shfl_xor(t1, 32);
        shift[t] = t1 / (float)min(32, n - row0);
      }
      float a1 = 0.f, a2 = 0.f;
      if (full) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float v = acc[t][r] * rin[r];
#ifndef CGC_X8_NOSTATS
          const float d = act_fwd(v, ACT) - shift[t];
          a1 += d;
          a2 = fmaf(d, d, a2);
#endif
#ifndef CGC_X8_NOSTORE
          __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rsrc_hn, voff,
                                                (unsigned)(row0 + (r & 3) + 8 * (r >> 2)) * (unsigned)ldh * 4u, 0);
#endif
        }
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float v = acc[t][r] * rin[r];
          const bool ok = row0 + (r & 3) + 8 * (r >> 2) + 4 * lhi < n;
          const float d = ok ? act_fwd(v, ACT) - shift[t] : 0.f;
          a1 += d;
          a2 = fmaf(d, d, a2);
          // (the partial last tile: rows past n are dropped by the LANE's own offset, not by the scalar row offset -- whether the
          // bounds check of a raw buffer includes soffset is not something to rely on)
          __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rsrc_hn, ok ? voff : 0x80000000u,
                                                (unsigned)(row0 + (r & 3) + 8 * (r >> 2)) * (unsigned)ldh * 4u, 0);
        }
      }
      if (colok) { s1[t] += a1; s2[t] += a2; }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) seen += (row0 + (r & 3) + 8 * (r >> 2) + 4 * lhi) < n ? 1 : 0;
  }
  if (ws != nullptr) {
    double* slot = reinterpret_cast<double*>(ws) + (size_t)blockIdx.x * 2 * F;
#pragma unroll
    for (int t = 0; t < NTW; ++t) {
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

// ------------------------------------------------------------------------------------------------------------------------
// Round 5: the row norm WITHOUT the wide product.  ||agg_i W + b||^2 = agg_i^T (W W^T) agg_i + 2 agg_i . (W b) + b . b is a K x K
// quadratic form per row (K = 20: 230 multiply-adds in double, 13 MFLOP for a whole C3 batch), so 1 / ||h|| of every row is known
// BEFORE the projection runs and the projection needs no exchange between the waves that share a row: no LDS, no barrier, no
// workgroup-wide W.  Three launches:
//   k_sage_gram   G = upper triangle of W W^T (off-diagonal entries doubled) | 2 W b | b.b in double -- one wave per entry;
//   k_sage_rinv   one thread per row evaluates the form in double from G (LDS, broadcast reads) and stores 1 / max(||h||, eps);
//   k_sage_wide_cols  a WAVE owns 32 rows x 96 columns: its three 32-column slices of W stay in registers as MFMA B fragments
//                 (3 KS registers), three independent accumulator chains, scaled by 1 / ||h|| and stored straight from the
//                 accumulators (128-byte row segments, buffer stores), BatchNorm column sums per lane as in the kernels above.
//                 The waves of a workgroup take neighbouring column groups of the same rows; three workgroups per CU, so one
//                 wave's stores and statistics run under the other waves' MFMA chains (the 8-wave kernel above serialises them:
//                 42 + 32 + 23 + 6 us).
// G travels in the first bytes of hn itself (written by the first launch, read by the second, overwritten by the third: stream order).
#define SWC_TILES 3
__global__ __launch_bounds__(256) void k_sage_gram(const float* __restrict__ W, const float* __restrict__ bias, int K, int F,
                                                   double* __restrict__ G) {
  // entry e: e < K (K + 1) / 2 -> (i, j >= i) of the upper triangle in row-major order; then K entries 2 (W b)_i; then b . b.
  // One workgroup per entry, a fixed summation tree (thread -> wave -> workgroup): the same bits on every run.
  __shared__ double part[4];
  const int tri = K * (K + 1) / 2, e = blockIdx.x;
  const float *x, *y;
  double scale = 1.0;
  if (e < tri) {
    int i = 0, rem = e;
    while (rem >= K - i) { rem -= K - i; ++i; }
    const int j = i + rem;
    x = W + (size_t)i * F;
    y = W + (size_t)j * F;
    scale = i == j ? 1.0 : 2.0;
  } else if (e < tri + K) {
    x = W + (size_t)(e - tri) * F;
    y = bias;
    scale = 2.0;
  } else {
    x = y = bias;
  }
  double s = 0.0;
  if (y != nullptr) {
    for (int c0 = 0; c0 < F; c0 += 256 * 8) {
      float xv[8], yv[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {                  // (every load of the pass in flight before the first use)
        const int c = c0 + u * 256 + (int)threadIdx.x;
        xv[u] = c < F ? x[c] : 0.f;
        yv[u] = c < F ? y[c] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) s = fma((double)xv[u], (double)yv[u], s);
    }
  }
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) G[e] = ((part[0] + part[1]) + (part[2] + part[3])) * scale;
}

template <int KK>
__global__ __launch_bounds__(256) void k_sage_rinv(const float* __restrict__ agg, int lda, int n, int K, const double* __restrict__ G,
                                                   float* __restrict__ rinv) {
  __shared__ double g[KK * (KK + 1) / 2 + KK + 1];       // the same packing as G, for width KK (entries with an index >= K: zero)
  const int tri = K * (K + 1) / 2;
  const int row = blockIdx.x * 256 + threadIdx.x;
  float xf[KK];                                           // the row travels while G is staged
  {
    const float* __restrict__ a = agg + (size_t)min(row, n - 1) * lda;
#pragma unroll
    for (int k = 0; k < KK; ++k) xf[k] = k < K ? a[k] : 0.f;
  }
  for (int e = threadIdx.x; e < KK * (KK + 1) / 2 + KK + 1; e += 256) {
    double v = 0.0;
    if (e < KK * (KK + 1) / 2) {
      int i = 0, rem = e;
      while (rem >= KK - i) { rem -= KK - i; ++i; }
      const int j = i + rem;
      if (j < K) v = G[i * K - i * (i - 1) / 2 + (j - i)];
    } else if (e < KK * (KK + 1) / 2 + KK) {
      const int i = e - KK * (KK + 1) / 2;
      if (i < K) v = G[tri + i];
    } else {
      v = G[tri + K];
    }
    g[e] = v;
  }
  __syncthreads();
  if (row >= n) return;
  double x[KK];
#pragma unroll
  for (int k = 0; k < KK; ++k) x[k] = (double)xf[k];
  double q = g[KK * (KK + 1) / 2 + KK];
  int e = 0;
#pragma unroll
  for (int i = 0; i < KK; ++i) {
    double t = g[KK * (KK + 1) / 2 + i];
#pragma unroll
    for (int j = i; j < KK; ++j) t = fma(g[e++], x[j], t);
    q = fma(t, x[i], q);
  }
  const double nrm = sqrt(q > 0.0 ? q : 0.0);
  rinv[row] = (float)(1.0 / (nrm > (double)L2_EPS ? nrm : (double)L2_EPS));
}

// the 16 row factors a lane needs for the row tile at row0 (accumulator register r <-> row (r & 3) + 8 (r >> 2) + 4 lhi): four 16-byte
// loads for a whole tile; the partial last tile reads element by element (whether a 16-byte buffer load that straddles num_records
// returns its in-range part is not something to rely on)
__device__ __forceinline__ void load_rows(__amdgpu_buffer_rsrc_t rsrc, int row0, int n, int lhi, float4 (&out)[4]) {
  if (row0 + 32 <= n) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
      out[j] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, (unsigned)(row0 + 8 * j + 4 * lhi) * 4u, 0, 0));
  } else {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const unsigned o = (unsigned)(row0 + 8 * j + 4 * lhi) * 4u;
      out[j].x = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, o, 0, 0));
      out[j].y = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, o + 4u, 0, 0));
      out[j].z = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, o + 8u, 0, 0));
      out[j].w = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, o + 12u, 0, 0));
    }
  }
}

template <int KS, int ACT>
#ifndef CGC_SWC_WAVES
#define CGC_SWC_WAVES 3
#endif
__global__ __launch_bounds__(256, CGC_SWC_WAVES) void k_sage_wide_cols(const float* __restrict__ agg, int lda, const float* __restrict__ W,
                                                           const float* __restrict__ bias, int n, int K, int F,
                                                           const float* __restrict__ rinv, float* __restrict__ hn, int ldh,
                                                           float* __restrict__ ws, int row_tiles, int chunks, int ngroups) {
  constexpr int NT = SWC_TILES;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int l31 = lane & 31, lhi = lane >> 5;
  const int rc = blockIdx.x % chunks, grp = (blockIdx.x / chunks) * 4 + wave;
  if (grp >= ngroups) return;                        // (no barrier anywhere in this kernel)
  const int c_base = grp * NT * 32;
  // B fragments: rows k < K of W, and the BIAS as row K (the A fragment carries a 1 there): the accumulators start from an inline
  // zero -- a bias-initialised set of accumulators kept for every row tile costs NT x 16 registers -- and the bias is added last, as
  // in the reference's matmul + bias.  Buffer loads: entries past F columns read as zero through the bounds check, no branches.
  float bw[NT][KS], s1[NT], s2[NT], shift[NT];
  {
    const __amdgpu_buffer_rsrc_t rsrc_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(W), 0, K * F * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_b = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(bias), 0, bias != nullptr ? F * 4 : 0, 0x00020000);
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int col = c_base + t * 32 + l31;
      const float bv = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc_b, (unsigned)col * 4u, 0, 0));
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        const int k = 2 * s + lhi;
        const float wv = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc_w, (col < F && k < K) ? (unsigned)(k * F + col) * 4u : 0x80000000u, 0, 0));
        bw[t][s] = k == K ? bv : wv;
      }
      s1[t] = s2[t] = shift[t] = 0.f;
    }
  }
  int seen = 0;
  const __amdgpu_buffer_rsrc_t rsrc_hn = __builtin_amdgcn_make_buffer_rsrc(hn, 0, (int)((size_t)n * ldh * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_ri = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(rinv), 0, n * 4, 0x00020000);
  // the next row tile's A fragments travel while this one is computed (rows past n: clamped)
  // (one descriptor over agg, one lane offset per row tile, the k step as the instruction's immediate offset: a flat load per
  // element needs a 64-bit address each; entries with k >= K are replaced below, reads past the last row's K entries return zero)
  const __amdgpu_buffer_rsrc_t rsrc_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(agg), 0, (int)(((size_t)(n - 1) * lda + K) * 4), 0x00020000);
  float av_next[KS];
  {
    const unsigned ao = (unsigned)(min(rc * 32 + l31, n - 1) * lda + lhi) * 4u;
#pragma unroll
    for (int s = 0; s < KS; ++s) av_next[s] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc_a, ao + 8u * s, 0, 0));
  }
  for (int rt = rc; rt < row_tiles; rt += chunks) {
    const int row0 = rt * 32;
    float av[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) av[s] = (2 * s + lhi) < K ? av_next[s] : (2 * s + lhi) == K ? 1.f : 0.f;
    floatx16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
#pragma unroll
    for (int s = 0; s < KS; ++s)
#pragma unroll
#ifndef CGC_SWC_NOMFMA
      for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s], bw[t][s], acc[t], 0, 0, 0);
#else
      for (int t = 0; t < NT; ++t) acc[t][s % 16] = av[s] + bw[t][s];        // (timing ablation: no matrix instruction, no chain)
#endif
    // This tile's row factors and the next tile's A fragments are requested behind this tile's MFMAs and waited for in front of its
    // stores (vmcnt counts loads and stores in issue order: see k_sage_wide_fwd8).  (Requesting them INSIDE the chains -- step s's
    // fragment as soon as step s has consumed the current one -- was measured: 67 -> 71 us.)
    float4 rq[4];
    load_rows(rsrc_ri, row0, n, lhi, rq);
    {
      const int nrt = rt + chunks < row_tiles ? rt + chunks : rt;
      const unsigned ao = (unsigned)(min(nrt * 32 + l31, n - 1) * lda + lhi) * 4u;
#pragma unroll
      for (int s = 0; s < KS; ++s) av_next[s] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc_a, ao + 8u * s, 0, 0));
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int s = 0; s < KS; ++s) asm volatile("" : "+v"(av_next[s]));
    float rin[16];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      asm volatile("" : "+v"(rq[j].x), "+v"(rq[j].y), "+v"(rq[j].z), "+v"(rq[j].w));
      rin[4 * j] = rq[j].x; rin[4 * j + 1] = rq[j].y; rin[4 * j + 2] = rq[j].z; rin[4 * j + 3] = rq[j].w;
    }
    const bool full = row0 + 32 <= n;
    if (rt == rc) {
      // The shift of the statistics: any value near the column's mean does (deviations from it are summed, nothing large cancels);
      // each half wave takes its own first row of its first tile (the halves are folded as (count, mean, M2) at the end)
#pragma unroll
      for (int t = 0; t < NT; ++t) shift[t] = act_fwd(acc[t][0] * rin[0], ACT);
    }
    __builtin_amdgcn_sched_barrier(0);
    // Stores: buffer stores off one descriptor over hn, a lane contributes a loop-invariant byte offset, the row is the instruction's
    // scalar offset (k_sage_wide_fwd8).  Parking each 32 x 32 tile in a per-wave LDS patch and storing it as 16-byte pieces of whole
    // 128-byte row segments (4 store instructions per tile instead of 16) was measured: 67 us either way -- and a 16-byte buffer
    // store whose row offset sits in an SGPR left one piece in ~10^4 with its first word replaced by what the NEXT vector
    // instruction wrote to that data register (the data is not read at issue in that form; with the whole offset in the VGPR it was
    // correct).  Dword stores have no such hazard.
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int col = c_base + t * 32 + l31;
      const bool colok = col < F;
      const unsigned voff = colok ? (unsigned)(4 * lhi * ldh + col) * 4u : 0x80000000u;
      float a1 = 0.f, a2 = 0.f;
      if (full) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float v = acc[t][r] * rin[r];
#ifndef CGC_SWC_NOSTATS
          const float d = act_fwd(v, ACT) - shift[t];
          a1 += d;
          a2 = fmaf(d, d, a2);
#endif
#ifndef CGC_SWC_NOSTORE
          __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rsrc_hn, voff,
                                                (unsigned)(row0 + (r & 3) + 8 * (r >> 2)) * (unsigned)ldh * 4u, 0);
#else
          asm volatile("" ::"v"(v));          // (timing ablation: the value stays computed, nothing is stored)
#endif
        }
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float v = acc[t][r] * rin[r];
          const bool ok = row0 + (r & 3) + 8 * (r >> 2) + 4 * lhi < n;
          const float d = ok ? act_fwd(v, ACT) - shift[t] : 0.f;
          a1 += d;
          a2 = fmaf(d, d, a2);
          // (rows past n are dropped by the LANE's own offset: the bounds check of a raw buffer does not include the scalar offset)
          __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rsrc_hn, ok ? voff : 0x80000000u,
                                                (unsigned)(row0 + (r & 3) + 8 * (r >> 2)) * (unsigned)ldh * 4u, 0);
        }
      }
      if (colok) { s1[t] += a1; s2[t] += a2; }
      __builtin_amdgcn_sched_barrier(0);          // one column tile at a time (keeps the scaled copies of one accumulator live, not three)
    }
    if (full) {
      seen += 16;
    } else {
#pragma unroll
      for (int r = 0; r < 16; ++r) seen += (row0 + (r & 3) + 8 * (r >> 2) + 4 * lhi) < n ? 1 : 0;
    }
  }
  if (ws != nullptr) {
    double* slot = reinterpret_cast<double*>(ws) + (size_t)rc * 2 * F;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
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
    // the single-pass 8-wave kernel when W fits the LDS next to nothing else ([2 ks][Fp] floats + 2 KB): one workgroup per CU
    // round 5: row norms from the K x K quadratic form, then one wave per 32 rows x 96 columns (see k_sage_wide_cols)
    static const int use_cols = getenv("CGC_SAGE_WIDE_COLS") ? atoi(getenv("CGC_SAGE_WIDE_COLS")) : 1;
    static const int cols_chunks = getenv("CGC_SAGE_WIDE_CHUNKS") ? atoi(getenv("CGC_SAGE_WIDE_CHUNKS")) : 256;
    // (small launches keep the one-kernel form: at 7200 rows -- a 4-graph shard -- two more launches cost more than the wave-level kernel saves)
    static const int cols_min_rows = getenv("CGC_SAGE_WIDE_COLS_MIN") ? atoi(getenv("CGC_SAGE_WIDE_COLS_MIN")) : 12288;
    if (use_cols && normalize && K <= 21 && n >= cols_min_rows && n >= 64 && (long long)n * ldh * 4 < (1LL << 31) && (reinterpret_cast<uintptr_t>(hn) & 7u) == 0 &&
        aligned16(rinv) && (size_t)n * ldh * 4 >= sizeof(double) * (size_t)(K * (K + 1) / 2 + K + 1) && ((long long)(n - 1) * lda + K) * 4 < (1LL << 31) &&
        // G is parked in the first bytes of hn and overwritten by the projection: with hn a column window of a wider buffer (ldh > F)
        // it must fit row 0's own F floats, or it would land in columns [F, ldh) that belong to the enclosing buffer
        (ldh == F || (size_t)F * 4 >= sizeof(double) * (size_t)(K * (K + 1) / 2 + K + 1))) {
      int per = ceil_div(row_tiles, cols_chunks > 0 ? cols_chunks : 256);
      int chunks = ceil_div(row_tiles, per);
      if (stats && chunks > cap) { chunks = cap > 0 ? cap : 1; }
      const int ngroups = ceil_div(ceil_div(F, 32), SWC_TILES);
      double* G = reinterpret_cast<double*>(hn);
      hipLaunchKernelGGL(k_sage_gram, dim3(K * (K + 1) / 2 + K + 1), dim3(256), 0, st, W, bias, K, F, G);
      if (ks == 8) hipLaunchKernelGGL((k_sage_rinv<16>), dim3(ceil_div(n, 256)), dim3(256), 0, st, agg, lda, n, K, G, rinv);
      else if (ks == 10) hipLaunchKernelGGL((k_sage_rinv<20>), dim3(ceil_div(n, 256)), dim3(256), 0, st, agg, lda, n, K, G, rinv);
      else hipLaunchKernelGGL((k_sage_rinv<32>), dim3(ceil_div(n, 256)), dim3(256), 0, st, agg, lda, n, K, G, rinv);
      const dim3 grid(chunks * ceil_div(ngroups, 4));
#define SWC_ONE(KS_, ACT_)                                                                                                         \
  hipLaunchKernelGGL((k_sage_wide_cols<KS_, ACT_>), grid, dim3(256), 0, st, agg, lda, W, bias, n, K, F, rinv, hn, ldh, wsp, row_tiles, \
                     chunks, ngroups)
#define SWC_LAUNCH(KS_)                                             \
  do {                                                              \
    if (act == CGC_ACT_RELU) SWC_ONE(KS_, CGC_ACT_RELU);            \
    else if (act == CGC_ACT_ELU) SWC_ONE(KS_, CGC_ACT_ELU);         \
    else if (act == CGC_ACT_LEAKYRELU) SWC_ONE(KS_, CGC_ACT_LEAKYRELU); \
    else SWC_ONE(KS_, CGC_ACT_IDENTITY);                            \
  } while (0)
      if (K <= 17) SWC_LAUNCH(9); else SWC_LAUNCH(11);      // K + 1 (the bias row) <= 2 KS
#undef SWC_LAUNCH
#undef SWC_ONE
      CGC_RETURN_IF_LAUNCH_FAILED();
      if (stats) return launch_stats_finalize(ws, chunks, F, count, eps, momentum, running_mean, running_var, mean, istd, num_batches_tracked, st);
      return 0;
    }
    static const int use8 = getenv("CGC_SAGE_WIDE8") ? atoi(getenv("CGC_SAGE_WIDE8")) : 1;
    const int ntw8 = F <= 8 * 5 * 32 ? 5 : 7, Fp = 8 * ntw8 * 32;
    const size_t lds8 = sizeof(float) * ((size_t)2 * ks * Fp + 512 + 256);
    if (use8 && F <= 8 * 7 * 32 && lds8 <= 156 * 1024 && n >= 64 && (long long)n * ldh * 4 < (1LL << 31)) {
      int wg8 = row_tiles < 256 ? row_tiles : 256;
      if (stats && wg8 > cap) wg8 = cap > 0 ? cap : 1;
#define SW8_ONE(KS_, NTW_, ACT_)                                                                                                       \
  do {                                                                                                                                 \
    static bool attr__[CGC_MAX_DEVICES] = {};                                                                                          \
    cgc_allow_lds(reinterpret_cast<const void*>(&k_sage_wide_fwd8<KS_, NTW_, ACT_>), 156 * 1024, attr__);                              \
    hipLaunchKernelGGL((k_sage_wide_fwd8<KS_, NTW_, ACT_>), dim3(wg8), dim3(512), lds8, st, agg, lda, W, bias, n, K, F, Fp, normalize, \
                       act, hn, ldh, rinv, wsp, row_tiles);                                                                            \
  } while (0)
#define SW8_LAUNCH(KS_, NTW_)                                                                                                          \
  do {                                                                                                                                 \
    if (act == CGC_ACT_RELU) SW8_ONE(KS_, NTW_, CGC_ACT_RELU);                                                                         \
    else if (act == CGC_ACT_ELU) SW8_ONE(KS_, NTW_, CGC_ACT_ELU);                                                                      \
    else if (act == CGC_ACT_LEAKYRELU) SW8_ONE(KS_, NTW_, CGC_ACT_LEAKYRELU);                                                          \
    else SW8_ONE(KS_, NTW_, CGC_ACT_IDENTITY);                                                                                         \
  } while (0)
      if (ntw8 == 5) {
        if (ks == 8) SW8_LAUNCH(8, 5); else if (ks == 10) SW8_LAUNCH(10, 5); else SW8_LAUNCH(16, 5);
      } else {
        if (ks == 8) SW8_LAUNCH(8, 7); else if (ks == 10) SW8_LAUNCH(10, 7); else SW8_LAUNCH(16, 7);
      }
#undef SW8_LAUNCH
#undef SW8_ONE
      CGC_RETURN_IF_LAUNCH_FAILED();
      if (stats) return launch_stats_finalize(ws, wg8, F, count, eps, momentum, running_mean, running_var, mean, istd, num_batches_tracked, st);
      return 0;
    }
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
