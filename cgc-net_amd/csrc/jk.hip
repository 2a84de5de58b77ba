// Jumping-knowledge attention of CGC-Net (DenseJK, model/network.py:11-55): per node, a bidirectional LSTM (input C, hidden
// H = 3C/2) runs over the node's THREE layer embeddings, a Linear(2H -> 1) scores each step, a softmax over the three
// scores weights the embeddings.  Rows are independent and the recurrence is 3 steps long, so the whole operator is one
// kernel per direction of autograd: TWO lanes per node (one per LSTM direction), the 2 x 4H x (C+H) LSTM weights live in LDS as float4 {i,f,g,o}
// per (hidden unit, input) and are read as wave-wide broadcasts (conflict-free), 200 FMAs per ds_read_b128 x 50.
// MIOpen's generic RNN path spends ~10 ms per training step on this (rocBLAS calls per time step); this takes < 0.2 ms.
//
// Forward keeps h_t, c_t of both directions ([2*3*H][npad], slot-major => coalesced) for the backward pass, which
// recomputes the gate activations, back-propagates through time in registers, and writes the gate gradients and the cell
// inputs TRANSPOSED ([4H+1][3*npad], [C+2H+1][3*npad]) so that every weight / bias / attention gradient falls out of one
// batched NT GEMM per direction (cgc_gemm_f32) followed by the deterministic slice reduction.
#include <stdlib.h>

#include "jk.hpp"

template <int C>
__device__ __forceinline__ void jk_fill_lds(const JkWeights& w, float4* Wt, float4* B4, float* watt, int nthreads) {
  constexpr int H = JkDims<C>::H, KIN = JkDims<C>::KIN;
  for (int idx = threadIdx.x; idx < 2 * H * KIN; idx += nthreads) {
    const int d = idx / (H * KIN), rem = idx - d * H * KIN, j = rem / KIN, k = rem - j * KIN;
    float v[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int row = g * H + j;
      v[g] = k < C ? w.w_ih[d][row * C + k] : w.w_hh[d][row * H + (k - C)];
    }
    Wt[idx] = make_float4(v[0], v[1], v[2], v[3]);
  }
  for (int idx = threadIdx.x; idx < 2 * H; idx += nthreads) {
    const int d = idx / H, j = idx - d * H;
    float v[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) v[g] = w.b_ih[d][g * H + j] + w.b_hh[d][g * H + j];
    B4[idx] = make_float4(v[0], v[1], v[2], v[3]);
    watt[idx] = w.w_att[idx];
  }
  if (threadIdx.x == 0) watt[2 * H] = w.b_att[0];
}

#define JK_THREADS 128

template <int C>
__global__ __launch_bounds__(JK_THREADS) void k_jk_fwd(const float* __restrict__ xs, int n, int npad, const JkWeights w,
                                                       float* __restrict__ out, float* HS, float* CS) {
  constexpr int H = JkDims<C>::H, KIN = JkDims<C>::KIN;
  extern __shared__ __attribute__((aligned(16))) float4 lds4[];
  float4* Wt = lds4;
  float4* B4 = lds4 + 2 * H * KIN;
  float* watt = reinterpret_cast<float*>(B4 + 2 * H);
  jk_fill_lds<C>(w, Wt, B4, watt, JK_THREADS);
  __syncthreads();
  // two adjacent lanes per node: the even lane runs the forward-direction LSTM, the odd lane the reverse one (the two
  // recurrences are independent); they meet once, through a shuffle, for the attention scores
  const int gid = blockIdx.x * JK_THREADS + threadIdx.x;
  const int row = gid >> 1, d = gid & 1;
  const bool valid = row < n;          // keep invalid lanes alive for the shuffle below
  const int r = valid ? row : 0;

  float x[3][C];
#pragma unroll
  for (int t = 0; t < 3; ++t)
#pragma unroll
    for (int k = 0; k < C; ++k) x[t][k] = xs[(size_t)r * 3 * C + t * C + k];
  float score[3];
#pragma unroll
  for (int t = 0; t < 3; ++t) score[t] = 0.f;

  const float4* Wd = Wt + d * H * KIN;
#pragma unroll
  for (int s = 0; s < 3; ++s) {
    float hprev[H];
    // time index of step s for this lane's direction (runtime d): forward s, reverse 2-s
    const int t = d == 0 ? s : 2 - s;
    const int tprev = d == 0 ? t - 1 : t + 1;
#pragma unroll
    for (int k = 0; k < H; ++k) hprev[k] = (s > 0 && valid) ? HS[(size_t)((d * 3 + tprev) * H + k) * npad + r] : 0.f;
    float xt[C];
#pragma unroll
    for (int k = 0; k < C; ++k) xt[k] = d == 0 ? x[s][k] : x[2 - s][k];
    float sc = 0.f;
    for (int j = 0; j < H; ++j) {     // rolled: keeps the code in the instruction cache
      const float4* wj = Wd + j * KIN;
      float4 acc = B4[d * H + j];
#pragma unroll
      for (int k = 0; k < C; ++k) {
        const float4 q = wj[k];
        acc.x = fmaf(q.x, xt[k], acc.x); acc.y = fmaf(q.y, xt[k], acc.y);
        acc.z = fmaf(q.z, xt[k], acc.z); acc.w = fmaf(q.w, xt[k], acc.w);
      }
      if (s > 0) {          // first step of a direction: h_prev = 0, the recurrent 3/5 of the dot product vanishes
#pragma unroll
        for (int k = 0; k < H; ++k) {
          const float4 q = wj[C + k];
          acc.x = fmaf(q.x, hprev[k], acc.x); acc.y = fmaf(q.y, hprev[k], acc.y);
          acc.z = fmaf(q.z, hprev[k], acc.z); acc.w = fmaf(q.w, hprev[k], acc.w);
        }
      }
      const float gi = sigmoidf_(acc.x), gf = sigmoidf_(acc.y), gg = tanhf(acc.z), go = sigmoidf_(acc.w);
      const float cprev = (s > 0 && valid) ? CS[(size_t)((d * 3 + tprev) * H + j) * npad + r] : 0.f;
      const float c = gf * cprev + gi * gg;
      const float h = go * tanhf(c);
      if (valid) {
        CS[(size_t)((d * 3 + t) * H + j) * npad + r] = c;
        HS[(size_t)((d * 3 + t) * H + j) * npad + r] = h;
      }
      sc = fmaf(watt[d * H + j], h, sc);
    }
    // score of TIME t: forward lane s -> t = s; reverse lane s -> t = 2-s
    if (d == 0) score[s] = sc;
    else score[2 - s] = sc;
  }
#pragma unroll
  for (int t = 0; t < 3; ++t) score[t] += __shfl_xor(score[t], 1) + watt[2 * H];
  const float m = fmaxf(score[0], fmaxf(score[1], score[2]));
  float a[3];
  float den = 0.f;
#pragma unroll
  for (int t = 0; t < 3; ++t) { a[t] = expf(score[t] - m); den += a[t]; }
  const float inv = 1.f / den;
  if (valid && d == 0) {
#pragma unroll
    for (int k = 0; k < C; ++k) out[(size_t)row * C + k] = (a[0] * x[0][k] + a[1] * x[1][k] + a[2] * x[2][k]) * inv;
  }
}

// Backward.  DGT: [2][4H+1][3*npad]  (rows g*H+j = d loss / d pre-activation gate, row 4H = d loss / d attention score)
//            INT: [2][C+2H+1][3*npad] (rows: x_t (C), h_{t-1} (H), ones (1), h_t (H)); column = t*npad + row.
//            DHC: [2][2][H][npad] scratch (recurrent dh and dc carries of each direction).
template <int C>
__global__ __launch_bounds__(JK_THREADS) void k_jk_bwd(const float* __restrict__ xs, const float* __restrict__ dout, int n, int npad,
                                                       const JkWeights w, const float* HS, const float* CS,
                                                       float* __restrict__ dxs, float* __restrict__ DGT, float* __restrict__ INT,
                                                       float* DHC) {
  constexpr int H = JkDims<C>::H, KIN = JkDims<C>::KIN;
  constexpr int NG = 4 * H + 1, NI = C + 2 * H + 1;
  extern __shared__ __attribute__((aligned(16))) float4 lds4[];
  float4* Wt = lds4;
  float4* B4 = lds4 + 2 * H * KIN;
  float* watt = reinterpret_cast<float*>(B4 + 2 * H);
  jk_fill_lds<C>(w, Wt, B4, watt, JK_THREADS);
  __syncthreads();
  // two adjacent lanes per node, one per LSTM direction (as in the forward kernel); partial input gradients and partial
  // attention scores are exchanged with one shuffle each
  const int gid = blockIdx.x * JK_THREADS + threadIdx.x;
  const int row = gid >> 1, d = gid & 1;
  const size_t ktot = (size_t)3 * npad;
  float* dgt = DGT + (size_t)d * NG * ktot;
  float* inT = INT + (size_t)d * NI * ktot;
  if (row >= n) {            // padding columns of the transposed buffers must be zero: they take part in the GEMM
    if (row < npad)          // (no shuffle partner needed: both lanes of a padding node take this branch)
      for (int t = 0; t < 3; ++t) {
        const size_t col = (size_t)t * npad + row;
        for (int r = 0; r < NG; ++r) dgt[(size_t)r * ktot + col] = 0.f;
        for (int r = 0; r < NI; ++r) inT[(size_t)r * ktot + col] = 0.f;
      }
    return;
  }

  float x[3][C], dx[3][C];
  float ds[3];
  {
    float dy[C];
#pragma unroll
    for (int k = 0; k < C; ++k) dy[k] = dout[(size_t)row * C + k];
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
      for (int k = 0; k < C; ++k) x[t][k] = xs[(size_t)row * 3 * C + t * C + k];
    // attention weights again: this lane's half of every score from its direction's saved hidden states, partner's half by shuffle
    float score[3];
#pragma unroll
    for (int t = 0; t < 3; ++t) {
      float sc = 0.f;
      for (int j = 0; j < H; ++j) sc = fmaf(watt[d * H + j], HS[(size_t)((d * 3 + t) * H + j) * npad + row], sc);
      score[t] = sc + __shfl_xor(sc, 1) + watt[2 * H];
    }
    const float m = fmaxf(score[0], fmaxf(score[1], score[2]));
    float a[3], den = 0.f;
#pragma unroll
    for (int t = 0; t < 3; ++t) { a[t] = expf(score[t] - m); den += a[t]; }
    float da[3], mean = 0.f;
#pragma unroll
    for (int t = 0; t < 3; ++t) {
      a[t] /= den;
      float v = 0.f;
#pragma unroll
      for (int k = 0; k < C; ++k) v = fmaf(dy[k], x[t][k], v);
      da[t] = v;
      mean = fmaf(a[t], v, mean);
    }
#pragma unroll
    for (int t = 0; t < 3; ++t) {
      ds[t] = a[t] * (da[t] - mean);
#pragma unroll
      for (int k = 0; k < C; ++k) dx[t][k] = d == 0 ? a[t] * dy[k] : 0.f;     // the attention term is counted once
    }
  }

  float* DH = DHC + (size_t)((d * 2 + 0) * H) * npad;       // [H][npad] recurrent carries of this direction
  float* DC = DHC + (size_t)((d * 2 + 1) * H) * npad;
  const float4* Wd = Wt + d * H * KIN;
#pragma unroll
  for (int s = 2; s >= 0; --s) {
    const int t = d == 0 ? s : 2 - s;
    const int tprev = d == 0 ? t - 1 : t + 1;
    const size_t col = (size_t)t * npad + row;
    float xt[C], dxt[C];
#pragma unroll
    for (int k = 0; k < C; ++k) { xt[k] = d == 0 ? x[s][k] : x[2 - s][k]; dxt[k] = 0.f; }
    const float dst = d == 0 ? ds[s] : ds[2 - s];
    float hprev[H], dhp[H];
#pragma unroll
    for (int k = 0; k < H; ++k) {
      hprev[k] = s > 0 ? HS[(size_t)((d * 3 + tprev) * H + k) * npad + row] : 0.f;
      dhp[k] = 0.f;
      inT[(size_t)(C + k) * ktot + col] = hprev[k];
    }
#pragma unroll
    for (int k = 0; k < C; ++k) inT[(size_t)k * ktot + col] = xt[k];
    inT[(size_t)(C + H) * ktot + col] = 1.f;
    dgt[(size_t)(4 * H) * ktot + col] = dst;
    for (int j = 0; j < H; ++j) {
      const float4* wj = Wd + j * KIN;
      float4 acc = B4[d * H + j];
#pragma unroll
      for (int k = 0; k < C; ++k) {
        const float4 q = wj[k];
        acc.x = fmaf(q.x, xt[k], acc.x); acc.y = fmaf(q.y, xt[k], acc.y);
        acc.z = fmaf(q.z, xt[k], acc.z); acc.w = fmaf(q.w, xt[k], acc.w);
      }
      if (s > 0) {          // first step of a direction: h_prev = 0, the recurrent 3/5 of the dot product vanishes
#pragma unroll
        for (int k = 0; k < H; ++k) {
          const float4 q = wj[C + k];
          acc.x = fmaf(q.x, hprev[k], acc.x); acc.y = fmaf(q.y, hprev[k], acc.y);
          acc.z = fmaf(q.z, hprev[k], acc.z); acc.w = fmaf(q.w, hprev[k], acc.w);
        }
      }
      const float gi = sigmoidf_(acc.x), gf = sigmoidf_(acc.y), gg = tanhf(acc.z), go = sigmoidf_(acc.w);
      const size_t slot = (size_t)((d * 3 + t) * H + j) * npad + row;
      const float cprev = s > 0 ? CS[(size_t)((d * 3 + tprev) * H + j) * npad + row] : 0.f;
      const float th = tanhf(CS[slot]);
      inT[(size_t)(C + H + 1 + j) * ktot + col] = HS[slot];                   // h_t: pairs with the score-gradient row
      float dh = dst * watt[d * H + j];
      float dc = 0.f;
      if (s < 2) { dh += DH[(size_t)j * npad + row]; dc = DC[(size_t)j * npad + row]; }
      dc = fmaf(dh * go, 1.f - th * th, dc);
      float4 q;                                                               // d loss / d pre-activation (i, f, g, o)
      q.x = dc * gg * gi * (1.f - gi);
      q.y = dc * cprev * gf * (1.f - gf);
      q.z = dc * gi * (1.f - gg * gg);
      q.w = dh * th * go * (1.f - go);
      if (s > 0) DC[(size_t)j * npad + row] = dc * gf;
      dgt[(size_t)(0 * H + j) * ktot + col] = q.x;
      dgt[(size_t)(1 * H + j) * ktot + col] = q.y;
      dgt[(size_t)(2 * H + j) * ktot + col] = q.z;
      dgt[(size_t)(3 * H + j) * ktot + col] = q.w;
#pragma unroll
      for (int k = 0; k < C; ++k) {
        const float4 v = wj[k];
        dxt[k] += v.x * q.x + v.y * q.y + v.z * q.z + v.w * q.w;
      }
      if (s > 0) {          // no earlier step to hand a hidden-state gradient to
#pragma unroll
        for (int k = 0; k < H; ++k) {
          const float4 v = wj[C + k];
          dhp[k] += v.x * q.x + v.y * q.y + v.z * q.z + v.w * q.w;
        }
      }
    }
    if (s > 0) {
#pragma unroll
      for (int k = 0; k < H; ++k) DH[(size_t)k * npad + row] = dhp[k];
    }
    // fold this step's input gradient into time slot t (static index per direction)
#pragma unroll
    for (int k = 0; k < C; ++k) {
      dx[s][k] += d == 0 ? dxt[k] : 0.f;
      dx[2 - s][k] += d == 0 ? 0.f : dxt[k];
    }
  }
#pragma unroll
  for (int t = 0; t < 3; ++t)
#pragma unroll
    for (int k = 0; k < C; ++k) {
      const float tot = dx[t][k] + __shfl_xor(dx[t][k], 1);
      if (d == 0) dxs[(size_t)row * 3 * C + t * C + k] = tot;
    }
}

static void fill_weights(JkWeights& w, const float* const* lstm, const float* w_att, const float* b_att) {
  for (int d = 0; d < 2; ++d) {
    w.w_ih[d] = lstm[4 * d + 0];
    w.w_hh[d] = lstm[4 * d + 1];
    w.b_ih[d] = lstm[4 * d + 2];
    w.b_hh[d] = lstm[4 * d + 3];
  }
  w.w_att = w_att;
  w.b_att = b_att;
}

// Channel counts the fused kernels are compiled for: every even C up to 32 (hidden H = 3C/2 <= 48; model/network.py:27-33 needs C
// even for the bidirectional split).  C in {4, 8, 12, 16, 20} (C % 4 == 0, H <= 32) run on the matrix-core kernels (jk_mfma.hip),
// the others on the thread-per-direction kernels of this file.
extern "C" int cgc_jk_supported(int C) { return C >= 2 && C <= 32 && C % 2 == 0; }
extern "C" int cgc_jk_matrix_core(int C) { return C == 4 || C == 8 || C == 12 || C == 16 || C == 20; }

#define JK_FOR_EACH_C(X) X(2) X(4) X(6) X(8) X(10) X(12) X(14) X(16) X(18) X(20) X(22) X(24) X(26) X(28) X(30) X(32)

template <int C>
static void jk_allow_lds() {      // the LDS image of the weights exceeds the 64 KB default for C >= 24
  static bool done_f[CGC_MAX_DEVICES] = {}, done_b[CGC_MAX_DEVICES] = {};
  cgc_allow_lds(reinterpret_cast<const void*>(&k_jk_fwd<C>), (int)JkDims<C>::lds_bytes, done_f);
  cgc_allow_lds(reinterpret_cast<const void*>(&k_jk_bwd<C>), (int)JkDims<C>::lds_bytes, done_b);
}

template <int C>
static int launch_jk_fwd(const float* xs, int n, int npad, const JkWeights& w, float* out, float* HS, float* CS, hipStream_t st) {
  jk_allow_lds<C>();
  hipLaunchKernelGGL(k_jk_fwd<C>, dim3(ceil_div(2 * n, JK_THREADS)), dim3(JK_THREADS), JkDims<C>::lds_bytes, st, xs, n, npad, w, out, HS, CS);
  CGC_RETURN_IF_LAUNCH_FAILED();
  return 0;
}
template <int C>
static int launch_jk_bwd(const float* xs, const float* dout, int n, int npad, const JkWeights& w, const float* HS, const float* CS,
                         float* dxs, float* DGT, float* INT, float* DHC, hipStream_t st) {
  jk_allow_lds<C>();
  hipLaunchKernelGGL(k_jk_bwd<C>, dim3(ceil_div(2 * npad, JK_THREADS)), dim3(JK_THREADS), JkDims<C>::lds_bytes, st, xs, dout, n, npad, w,
                     HS, CS, dxs, DGT, INT, DHC);
  CGC_RETURN_IF_LAUNCH_FAILED();
  return 0;
}

extern "C" int cgc_jk_lstm_fwd(const float* xs, int n, int npad, int C, const float* const* lstm, const float* w_att,
                               const float* b_att, float* out, float* HS, float* CS, cgc_stream_t stream) {
  if (n <= 0) return 0;
  if (npad < n) return CGC_EINVAL;
  JkWeights w;
  fill_weights(w, lstm, w_att, b_att);
  static const int k_mfma = getenv("CGC_JK_MFMA") != nullptr ? atoi(getenv("CGC_JK_MFMA")) : 1;
  if (k_mfma) {
    const int rc = jk_mfma_fwd(xs, n, npad, C, w, out, HS, CS, as_stream(stream));
    if (rc != CGC_EINVAL) return rc;          // unaligned buffers: the thread-per-direction kernel takes any alignment
  }
  switch (C) {
#define X(C_) case C_: return launch_jk_fwd<C_>(xs, n, npad, w, out, HS, CS, as_stream(stream));
    JK_FOR_EACH_C(X)
#undef X
    default: return CGC_EINVAL;
  }
}

extern "C" int cgc_jk_lstm_bwd(const float* xs, const float* dout, int n, int npad, int C, const float* const* lstm,
                               const float* w_att, const float* b_att, const float* HS, const float* CS, float* dxs, float* DGT,
                               float* INT, float* DHC, cgc_stream_t stream) {
  if (n <= 0) return 0;
  if (npad < n) return CGC_EINVAL;
  JkWeights w;
  fill_weights(w, lstm, w_att, b_att);
  static const int k_mfma = getenv("CGC_JK_MFMA") != nullptr ? atoi(getenv("CGC_JK_MFMA")) : 1;
  if (k_mfma) {
    const int rc = jk_mfma_bwd(xs, dout, n, npad, C, w, HS, CS, dxs, DGT, INT, as_stream(stream));
    if (rc != CGC_EINVAL) return rc;
  }
  switch (C) {
#define X(C_) case C_: return launch_jk_bwd<C_>(xs, dout, n, npad, w, HS, CS, dxs, DGT, INT, DHC, as_stream(stream));
    JK_FOR_EACH_C(X)
#undef X
    default: return CGC_EINVAL;
  }
}

extern "C" int64_t cgc_jk_bwd_ws_floats(int C) { return jk_mfma_bwd_ws_floats(C); }

// Backward with the parameter gradients accumulated in-kernel (matrix-core path only; CGC_EINVAL when the buffers are not
// 16-byte aligned or C is unsupported -- the caller then uses cgc_jk_lstm_bwd + one GEMM per direction).
extern "C" int cgc_jk_lstm_bwd_params(const float* xs, const float* dout, int n, int npad, int C, const float* const* lstm,
                                      const float* w_att, const float* b_att, const float* HS, const float* CS, float* dxs,
                                      float* G, float* ws, cgc_stream_t stream) {
  if (n <= 0) return 0;
  if (npad < n) return CGC_EINVAL;
  JkWeights w;
  fill_weights(w, lstm, w_att, b_att);
  return jk_mfma_bwd_params(xs, dout, n, npad, C, w, HS, CS, dxs, G, ws, as_stream(stream));
}

// cgc_jk_lstm_bwd_params + cgc_jk_unpack_param_grads in one: flat = cgc_jk_param_grad_floats(C) floats in parameter order (no G).
extern "C" int cgc_jk_lstm_bwd_flat(const float* xs, const float* dout, int n, int npad, int C, const float* const* lstm,
                                    const float* w_att, const float* b_att, const float* HS, const float* CS, float* dxs,
                                    float* flat, float* ws, cgc_stream_t stream) {
  if (n <= 0) return 0;
  if (npad < n) return CGC_EINVAL;
  JkWeights w;
  fill_weights(w, lstm, w_att, b_att);
  return jk_mfma_bwd_flat(xs, dout, n, npad, C, w, HS, CS, dxs, flat, ws, as_stream(stream));
}

// G [2][4H+1][C+2H+1] -> the parameter gradients of DenseJK in ONE contiguous buffer, in torch.nn.LSTM / nn.Linear order:
// per direction d: dW_ih [4H,C] | dW_hh [4H,H] | db_ih [4H] | db_hh [4H] (= db_ih), then d att.weight [2H] and d att.bias [1].
// Every gradient is then a contiguous slice that autograd can hand to the parameter as it is -- returned as strided windows of
// G they were each cloned by AccumulateGrad (24 extra copy kernels per step for the three DenseJK modules).
__global__ void k_jk_unpack(const float* __restrict__ G, int C, int H, float* __restrict__ flat) {
  const int NI = C + 2 * H + 1, NG = 4 * H + 1;
  const int per_dir = 4 * H * C + 4 * H * H + 8 * H, total = 2 * per_dir + 2 * H + 1;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    float v;
    if (i < 2 * per_dir) {
      const int d = i / per_dir;
      int e = i - d * per_dir;
      const float* Gd = G + (size_t)d * NG * NI;
      if (e < 4 * H * C) v = Gd[(e / C) * NI + (e % C)];
      else if ((e -= 4 * H * C) < 4 * H * H) v = Gd[(e / H) * NI + C + (e % H)];
      else { e -= 4 * H * H; v = Gd[(e % (4 * H)) * NI + C + H]; }
    } else {
      const int e = i - 2 * per_dir;
      if (e < 2 * H) v = G[((size_t)(e / H) * NG + 4 * H) * NI + C + H + 1 + (e % H)];
      else v = G[(size_t)(4 * H) * NI + C + H];
    }
    flat[i] = v;
  }
}

extern "C" int64_t cgc_jk_param_grad_floats(int C) {
  const int64_t H = 3 * (int64_t)C / 2;
  return 2 * (4 * H * C + 4 * H * H + 8 * H) + 2 * H + 1;
}

extern "C" int cgc_jk_unpack_param_grads(const float* G, int C, float* flat, cgc_stream_t stream) {
  if (C <= 0) return CGC_EINVAL;
  const int H = 3 * C / 2;
  const int total = (int)cgc_jk_param_grad_floats(C);
  hipLaunchKernelGGL(k_jk_unpack, dim3(ceil_div(total, 256)), dim3(256), 0, as_stream(stream), G, C, H, flat);
  CGC_RETURN_IF_LAUNCH_FAILED();
  return 0;
}
