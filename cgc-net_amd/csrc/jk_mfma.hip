// DenseJK (model/network.py:11-55) on the matrix cores.
//
// The bidirectional LSTM over a node's three layer embeddings is, per step, gates[4H] = W [4H x (C+H)] . [x_t | h_{t-1}].
// One WAVE owns 32 nodes and one direction and computes the TRANSPOSED product  gates^T[(g,j) x node] = W . [x|h]^T  with
// v_mfma_f32_32x32x2_f32: the M dimension is (gate g = one 32-row tile each, hidden unit j), the N dimension is the 32
// nodes, K runs over the inputs.  In the accumulator layout of that instruction a lane holds ONE node (column = lane&31)
// and, in its 16 registers, the units j = (r&3) + 8(r>>2) + 4(lane>>5) -- all four gates of a (node, unit) pair sit in the
// same lane and register index, so the LSTM cell is plain per-register arithmetic, and the new hidden state h_t[r] is
// ALREADY the B operand (k = 8q + 4(lane>>5) + t  <->  r = 4q + t) of the next step's recurrent product: the recurrence
// never leaves the registers.  x_t fragments are 16-byte global loads with the same k permutation; the weights (A operand,
// row = unit) are staged once per workgroup in LDS in fragment order and read back as conflict-free ds_read_b128.
// fp32 MFMA keeps the exact fp32 FMA chain (1e-4 parity budget); the thread-per-direction kernels in jk.hip ran at ~7 %
// of the fp32 peak (1 wave/SIMD, LDS-broadcast bound), these run the same arithmetic at matrix-core rate.
#include "jk.hpp"

typedef float floatx16 __attribute__((ext_vector_type(16)));

#ifndef JKM_WAVES
#define JKM_WAVES 2      // workgroups per CU the register allocation aims for (2 x 57 KB of LDS)
#endif
#ifndef JKM_FWD_UNROLL
#define JKM_FWD_UNROLL 1    // recurrence rolled: 10-15 % faster than fully unrolled (less register pressure, same MFMA stream)
#endif
#ifndef JKB_TILES
#define JKB_TILES 2          // node tiles per backward workgroup (x 2 directions = 4 waves, one per SIMD; 4 tiles = 2 waves per
                             // SIMD under a 256-register cap measured 10-100 % slower: spills and coarser work units)
#endif
#define JKB_THREADS (JKB_TILES * 128)

template <int C>
struct JkM {
  static constexpr int H = 3 * C / 2;
  static constexpr int XG = (C + 7) / 8;      // 8-wide k groups of the x part (zero padded)
  static constexpr int HG = (H + 7) / 8;      // ... of the recurrent part
  static constexpr int NQ = XG + HG;
  static constexpr int W_FLOATS = 2 * 4 * NQ * 32 * 8;        // [d][g][q][unit 32][lane half 2][t 4]
  static constexpr int B_OFF = W_FLOATS;                       // bias  [d][g][32]
  static constexpr int A_OFF = B_OFF + 2 * 4 * 32;             // w_att [d][32], then b_att
  static constexpr int S_OFF = A_OFF + 2 * 32 + 4;             // score exchange [node half 2][d 2][t 3][32]
  static constexpr int TOTAL = S_OFF + 2 * 2 * 3 * 32;
  static constexpr size_t lds_bytes = sizeof(float) * TOTAL;
  static constexpr int PG_FLOATS = (8 * 16 + 16 + 1) * 64;        // parameter-gradient partial of one wave
  static_assert(H <= 32 && C % 4 == 0, "one 32-row tile per gate; 16-byte x fragments");
};

#ifdef CGC_JK_PRECISE
__device__ __forceinline__ float fast_sigmoid(float x) { return 1.f / (1.f + expf(-x)); }
__device__ __forceinline__ float fast_tanh(float x) { return tanhf(x); }
#define JK_EXP expf
#else
// Reciprocal of d in [1, inf]: v_rcp_f32 (1 ulp) instead of the correctly rounded division sequence the library's compile flags
// turn __frcp_rn / `1.f / d` into (v_div_scale x 2, v_rcp, five FMAs, v_div_fmas, v_div_fixup: ten instructions, five times per
// hidden unit and recurrence step in kernels that issue 8 vector instructions per MFMA).  The error stays RELATIVE (what the
// medium_shipped margins are sensitive to: see fast_tanh) and below that of the __expf in front of it.  JK_RCP_NEWTON adds one
// Newton step (two FMAs, ~0.5 ulp; the argument is clamped so that d stays finite: rcp(inf) = 0 and 0 * inf would poison it).
#ifndef JK_RCP_NEWTON
#define JK_RCP_NEWTON 1      // round 6: on.  Plain v_rcp_f32 (1 ulp) moved medium_shipped's worst margin against the reference's float64 gradients from
#endif                       // 3.1e-5 to 9.9e-5 (GCN_embed_3.gcn1.bias amplifies a relative error of the gates a thousandfold); with the Newton step: see DESIGN
#if JK_RCP_NEWTON
__device__ __forceinline__ float jk_rcp(float d) {
  const float r = __builtin_amdgcn_rcpf(d);
  return fmaf(fmaf(-d, r, 1.f), r, r);
}
__device__ __forceinline__ float fast_sigmoid(float x) { return jk_rcp(1.f + __expf(fminf(-x, 87.f))); }
#else
__device__ __forceinline__ float jk_rcp(float d) { return __builtin_amdgcn_rcpf(d); }
__device__ __forceinline__ float fast_sigmoid(float x) { return jk_rcp(1.f + __expf(-x)); }
#endif
// tanh with a few ulp of RELATIVE error everywhere.  The one-line form 2 / (1 + exp(-2x)) - 1 has ~2e-7 of ABSOLUTE error, i.e.
// 2e-6 relative at |x| = 0.1 and 2e-4 at 1e-3 -- cell states and gate inputs of a freshly initialised LSTM are that small, and on
// the reference-generated medium_shipped fixture the network amplifies it a thousandfold (a one-ulp perturbation of the parameters
// moves d loss / d GCN_embed_3.gcn1.bias by 2.5e-4 of its max-norm): 1.5e-4 against the reference in float64, 4e-5 with this form
// (same as libm's tanhf).  |x| >= 1/4: (1 - e) / (1 + e) with e = exp(-2|x|) <= 0.61 (no cancellation); below: the odd Taylor
// polynomial to x^9 (truncation 9e-9 relative at 1/4).  Branch-free: both are computed, six FMAs more than before.
__device__ __forceinline__ float fast_tanh(float x) {
  const float ax = fabsf(x), x2 = x * x;
  const float e = __expf(-2.f * ax);
  const float big = (1.f - e) * jk_rcp(1.f + e);
  const float small = ax * fmaf(x2, fmaf(x2, fmaf(x2, fmaf(x2, 62.f / 2835.f, -17.f / 315.f), 2.f / 15.f), -1.f / 3.f), 1.f);
  return copysignf(ax < 0.25f ? small : big, x);
}
#define JK_EXP __expf
#endif

template <int C>
__device__ __forceinline__ void jkm_fill(const JkWeights& w, float* lds, int nthreads) {
  using M = JkM<C>;
  constexpr int H = M::H, XG = M::XG, NQ = M::NQ;
  // 256 threads = one (d, g, q) block of 32 units x 2 lane halves x 4 k per pass: (unit, half, k) are fixed per thread and
  // (d, g, q) are compile-time per pass, so the 8*NQ global loads of a thread are independent and issue back to back
  {
    const int tt = threadIdx.x & 3, lhi = (threadIdx.x >> 2) & 1, j = (threadIdx.x >> 3) & 31;
    float v[2 * 4 * NQ];
#pragma unroll
    for (int it = 0; it < 2 * 4 * NQ; ++it) {
      const int q = it % NQ, g = (it / NQ) & 3, d = it / (4 * NQ);
      v[it] = 0.f;
      if (q < XG) {
        const int k = 8 * q + 4 * lhi + tt;
        if (j < H && k < C) v[it] = w.w_ih[d][(g * H + j) * C + k];
      } else {
        const int k = 8 * (q - XG) + 4 * lhi + tt;
        if (j < H && k < H) v[it] = w.w_hh[d][(g * H + j) * H + k];
      }
    }
#pragma unroll
    for (int it = 0; it < 2 * 4 * NQ; ++it) lds[it * 256 + threadIdx.x] = v[it];
  }
  for (int idx = threadIdx.x; idx < 2 * 4 * 32; idx += nthreads) {
    const int j = idx & 31, g = (idx >> 5) & 3, d = idx >> 7;
    lds[M::B_OFF + idx] = j < H ? w.b_ih[d][g * H + j] + w.b_hh[d][g * H + j] : 0.f;
  }
  for (int idx = threadIdx.x; idx < 2 * 32; idx += nthreads) {
    const int j = idx & 31, d = idx >> 5;
    lds[M::A_OFF + idx] = j < H ? w.w_att[d * H + j] : 0.f;
  }
  if (threadIdx.x == 0) lds[M::A_OFF + 64] = w.b_att[0];
}

template <int C>
__global__ __launch_bounds__(256, JKM_WAVES) void k_jk_fwd_mfma(const float* __restrict__ xs, int n, int npad, const JkWeights w,
                                                     float* __restrict__ out, float* __restrict__ HS, float* __restrict__ CS) {
  using M = JkM<C>;
  constexpr int H = M::H, XG = M::XG, HG = M::HG, NQ = M::NQ;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  jkm_fill<C>(w, lds, 256);
  __syncthreads();
  const int lane = threadIdx.x & 63, l31 = lane & 31, lhi = lane >> 5;
  const int wave = threadIdx.x >> 6, d = wave & 1, half = wave >> 1;
  const float4* Wl = reinterpret_cast<const float4*>(lds) + (size_t)d * 4 * NQ * 64 + l31 * 2 + lhi;   // + (g*NQ + q)*64
  const float* Bl = lds + M::B_OFF + d * 128 + 4 * lhi;          // + g*32 + 8q'
  const float* Al = lds + M::A_OFF + d * 32 + 4 * lhi;           // + 8q'
  float* Sx = lds + M::S_OFF + half * 192;                       // [d][t][32]
  const int ntiles = (n + 31) / 32;

  for (int base = blockIdx.x * 2; base < ntiles; base += gridDim.x * 2) {
    const int node = (base + half) * 32 + l31;
    const bool valid = node < n, keep = node < npad && base + half < ntiles;
    const float* xrow = xs + (size_t)(valid ? node : 0) * 3 * C + 4 * lhi;
    floatx16 cst, hst;
#pragma unroll
    for (int r = 0; r < 16; ++r) { cst[r] = 0.f; hst[r] = 0.f; }

#pragma unroll JKM_FWD_UNROLL
    for (int s = 0; s < 3; ++s) {
      const int t = d ? 2 - s : s;
      floatx16 acc[4];
#pragma unroll
      for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float4 b = *reinterpret_cast<const float4*>(Bl + g * 32 + 8 * q);
          acc[g][4 * q + 0] = b.x; acc[g][4 * q + 1] = b.y; acc[g][4 * q + 2] = b.z; acc[g][4 * q + 3] = b.w;
        }
      // input part: B operand = x_t fragment of this lane's node
#pragma unroll
      for (int q = 0; q < XG; ++q) {
        float4 xf = make_float4(0.f, 0.f, 0.f, 0.f);
        if (valid && (8 * q + 4 * lhi < C)) xf = *reinterpret_cast<const float4*>(xrow + t * C + 8 * q);
        float4 a[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) a[g] = Wl[(g * NQ + q) * 64];
#pragma unroll
        for (int g = 0; g < 4; ++g) acc[g] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[g].x, xf.x, acc[g], 0, 0, 0);
#pragma unroll
        for (int g = 0; g < 4; ++g) acc[g] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[g].y, xf.y, acc[g], 0, 0, 0);
#pragma unroll
        for (int g = 0; g < 4; ++g) acc[g] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[g].z, xf.z, acc[g], 0, 0, 0);
#pragma unroll
        for (int g = 0; g < 4; ++g) acc[g] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[g].w, xf.w, acc[g], 0, 0, 0);
      }
      // recurrent part: B operand = h_{t-1}, straight from the registers the previous step left it in (zero at s = 0)
      if (s > 0) {
#pragma unroll
        for (int q = 0; q < HG; ++q) {
          float4 a[4];
#pragma unroll
          for (int g = 0; g < 4; ++g) a[g] = Wl[(g * NQ + XG + q) * 64];
#pragma unroll
          for (int g = 0; g < 4; ++g) acc[g] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[g].x, hst[4 * q + 0], acc[g], 0, 0, 0);
#pragma unroll
          for (int g = 0; g < 4; ++g) acc[g] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[g].y, hst[4 * q + 1], acc[g], 0, 0, 0);
#pragma unroll
          for (int g = 0; g < 4; ++g) acc[g] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[g].z, hst[4 * q + 2], acc[g], 0, 0, 0);
#pragma unroll
          for (int g = 0; g < 4; ++g) acc[g] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[g].w, hst[4 * q + 3], acc[g], 0, 0, 0);
        }
      }
      // LSTM cell, register by register (unit j = (r&3) + 8(r>>2) + 4*lhi of this lane's node)
      float p = 0.f;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float4 wa = *reinterpret_cast<const float4*>(Al + 8 * q);
        const float was[4] = {wa.x, wa.y, wa.z, wa.w};
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int r = 4 * q + u, j = u + 8 * q + 4 * lhi;
          const float gi = fast_sigmoid(acc[0][r]), gf = fast_sigmoid(acc[1][r]);
          const float gg = fast_tanh(acc[2][r]), go = fast_sigmoid(acc[3][r]);
          const float c = gf * cst[r] + gi * gg;
          const float h = go * fast_tanh(c);
          cst[r] = c;
          hst[r] = h;
          p = fmaf(was[u], h, p);
          if (8 * q + u < H && j < H && keep) {      // (first clause prunes whole padded groups at compile time)
            const size_t slot = (size_t)((d * 3 + t) * H + j) * npad + node;
            HS[slot] = h;
            CS[slot] = c;
          }
        }
      }
      p += __shfl_xor(p, 32);
      if (lhi == 0) Sx[(d * 3 + t) * 32 + l31] = p;
    }
    __syncthreads();
    float sc[3];
#pragma unroll
    for (int t = 0; t < 3; ++t) sc[t] = Sx[t * 32 + l31] + Sx[(3 + t) * 32 + l31] + lds[M::A_OFF + 64];
    const float m = fmaxf(sc[0], fmaxf(sc[1], sc[2]));
    float a3[3], den = 0.f;
#pragma unroll
    for (int t = 0; t < 3; ++t) { a3[t] = JK_EXP(sc[t] - m); den += a3[t]; }
    const float inv = 1.f / den;
    if (d == 0 && valid) {
#pragma unroll
      for (int q = 0; q < XG; ++q) {
        if (8 * q + 4 * lhi < C) {
          const float4 x0 = *reinterpret_cast<const float4*>(xrow + 0 * C + 8 * q);
          const float4 x1 = *reinterpret_cast<const float4*>(xrow + 1 * C + 8 * q);
          const float4 x2 = *reinterpret_cast<const float4*>(xrow + 2 * C + 8 * q);
          float4 o;
          o.x = (a3[0] * x0.x + a3[1] * x1.x + a3[2] * x2.x) * inv;
          o.y = (a3[0] * x0.y + a3[1] * x1.y + a3[2] * x2.y) * inv;
          o.z = (a3[0] * x0.z + a3[1] * x1.z + a3[2] * x2.z) * inv;
          o.w = (a3[0] * x0.w + a3[1] * x1.w + a3[2] * x2.w) * inv;
          *reinterpret_cast<float4*>(out + (size_t)node * C + 8 * q + 4 * lhi) = o;
        }
      }
    }
    __syncthreads();          // the score exchange buffer is rewritten by the next tile
  }
}

// Backward on the matrix cores.  Same tiling (one wave = 32 nodes x one direction, lane = node).  Per recurrence step, last
// to first: the gate pre-activations are recomputed with the forward's MFMAs (h_{t-1}, c_{t-1} come back from HS / CS as
// coalesced loads already in operand layout), the cell backward runs per register, and the four gate-gradient tiles q_g --
// still "lane = node, register = unit" -- are directly the B operand of  d[h_{t-1} | x_t]^T = W^T . q  (A operand = W^T,
// read out of the SAME LDS image the forward uses, strided: 4-way bank conflicts, irrelevant next to 128 MFMAs), whose
// accumulator layout hands dh_{t-1} to the next step in registers and dx_t as 16-byte fragments.  Gate gradients and cell
// inputs are written transposed (DGT / INT, see jk.hip) for the parameter-gradient GEMM.
template <int C, bool PG>
__global__ __launch_bounds__(JKB_THREADS) void k_jk_bwd_mfma(const float* __restrict__ xs, const float* __restrict__ dout, int n, int npad,
                                                     const JkWeights w, const float* __restrict__ HS, const float* __restrict__ CS,
                                                     float* __restrict__ dxs, float* __restrict__ DGT, float* __restrict__ INT,
                                                     float* __restrict__ PART) {
  using M = JkM<C>;
  constexpr int H = M::H, XG = M::XG, HG = M::HG, NQ = M::NQ;
  constexpr int NG = 4 * H + 1, NI = C + 2 * H + 1;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  if (threadIdx.x < 256) jkm_fill<C>(w, lds, 256);
  __syncthreads();
  const int lane = threadIdx.x & 63, l31 = lane & 31, lhi = lane >> 5;
  const int wave = threadIdx.x >> 6, d = wave & 1, half = wave >> 1;          // half = node tile of this workgroup pass (0..JKB_TILES-1)
  const float4* Wl = reinterpret_cast<const float4*>(lds) + (size_t)d * 4 * NQ * 64 + l31 * 2 + lhi;   // forward fragments
  const float* Bl = lds + M::B_OFF + d * 128 + 4 * lhi;
  const float* Al = lds + M::A_OFF + d * 32 + 4 * lhi;
  float* Sx = lds + M::TOTAL + half * 192;                                                           // [d][t][32]
  float4* Ex = reinterpret_cast<float4*>(lds + M::TOTAL + JKB_TILES * 192) + (size_t)(half * 2 + d) * 3 * XG * 64 + lane;   // + (t*XG + q)*64
  // W^T fragments out of the forward image: row kin = l31 of the (h | x) tile, k = (g, unit)
  const int kh = l31 < 8 * HG ? l31 : 8 * HG - 1, kx = l31 < 8 * XG ? l31 : 8 * XG - 1;
  const float* WtH = lds + (size_t)d * 4 * NQ * 256 + (XG + kh / 8) * 256 + (kh & 7) + 32 * lhi;   // + g*NQ*256 + (8q'+tt)*8
  const float* WtX = lds + (size_t)d * 4 * NQ * 256 + (kx / 8) * 256 + (kx & 7) + 32 * lhi;
  const size_t ktot = (size_t)3 * npad;
  float* dgt = DGT + (size_t)d * NG * ktot;
  float* inT = INT + (size_t)d * NI * ktot;
  const int ntiles = npad / 32;
  // PG: parameter gradients accumulated here.  dW[(g,j)][kin] = sum_node q_g[node][j] * in[node][kin] contracts over the
  // LANE dimension of the layout above, so q and the cell inputs take one trip through a per-wave LDS tile ([node][36]) to
  // come back with lane = row / column and the node pair as the MFMA k index.  8 persistent accumulator tiles per wave:
  // (gate g) x (inputs: h part with the bias column at 31 | x part); attention-weight gradients per lane.
  float* Qw = lds + M::TOTAL + JKB_TILES * 192 + JKB_TILES * 2 * 3 * XG * 64 * 4 + (size_t)wave * 3 * 32 * 36;
  float* Iw = Qw + 32 * 36;                       // two input tiles
  floatx16 dW[PG ? 4 : 1][2];
  floatx16 wacc;
  float bacc = 0.f;
  if (PG) {
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int r = 0; r < 16; ++r) { dW[g][0][r] = 0.f; dW[g][1][r] = 0.f; }
#pragma unroll
    for (int r = 0; r < 16; ++r) wacc[r] = 0.f;
  }

  for (int base = blockIdx.x * JKB_TILES; base < ntiles; base += gridDim.x * JKB_TILES) {
    const int node = (base + half) * 32 + l31;
    if (PG && base * 32 >= n) break;            // padding tiles contribute nothing
    if (base * 32 >= n) {            // all tiles of this workgroup pass are padding columns: they must read as zero in the GEMM
      for (int t = 0; t < 3; ++t) {
        const size_t col = (size_t)t * npad + node;
        for (int r = lhi; r < NG; r += 2) dgt[(size_t)r * ktot + col] = 0.f;
        for (int r = lhi; r < NI; r += 2) inT[(size_t)r * ktot + col] = 0.f;
      }
      continue;
    }
    const bool valid = node < n;
    const float* xrow = xs + (size_t)(valid ? node : 0) * 3 * C + 4 * lhi;

    // ---- attention: scores from the saved hidden states (own direction; the partner wave supplies the other half)
#pragma unroll
    for (int t = 0; t < 3; ++t) {
      float p = 0.f;
#pragma unroll
      for (int q = 0; q < HG; ++q) {
        const float4 wa = *reinterpret_cast<const float4*>(Al + 8 * q);
        const float was[4] = {wa.x, wa.y, wa.z, wa.w};
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int j = u + 8 * q + 4 * lhi;
          if (8 * q + u < H && j < H && valid) p = fmaf(was[u], HS[(size_t)((d * 3 + t) * H + j) * npad + node], p);
        }
      }
      p += __shfl_xor(p, 32);
      if (lhi == 0) Sx[(d * 3 + t) * 32 + l31] = p;
    }
    __syncthreads();
    float a3[3], ds3[3];
    {
      float sc[3];
#pragma unroll
      for (int t = 0; t < 3; ++t) sc[t] = Sx[t * 32 + l31] + Sx[(3 + t) * 32 + l31] + lds[M::A_OFF + 64];
      const float m = fmaxf(sc[0], fmaxf(sc[1], sc[2]));
      float den = 0.f;
#pragma unroll
      for (int t = 0; t < 3; ++t) { a3[t] = JK_EXP(sc[t] - m); den += a3[t]; }
      const float inv = 1.f / den;
      float da[3] = {0.f, 0.f, 0.f}, mean = 0.f;
#pragma unroll
      for (int q = 0; q < XG; ++q) {
        if (valid && (8 * q + 4 * lhi < C)) {
          const float4 dy = *reinterpret_cast<const float4*>(dout + (size_t)node * C + 8 * q + 4 * lhi);
#pragma unroll
          for (int t = 0; t < 3; ++t) {
            const float4 x = *reinterpret_cast<const float4*>(xrow + t * C + 8 * q);
            da[t] += dy.x * x.x + dy.y * x.y + dy.z * x.z + dy.w * x.w;
          }
        }
      }
#pragma unroll
      for (int t = 0; t < 3; ++t) {
        da[t] += __shfl_xor(da[t], 32);
        a3[t] *= inv;
        mean = fmaf(a3[t], da[t], mean);
      }
#pragma unroll
      for (int t = 0; t < 3; ++t) ds3[t] = valid ? a3[t] * (da[t] - mean) : 0.f;
    }

    floatx16 dhc, dcc;                 // recurrent carries d loss / d h_{t}, d c_{t} arriving from the later step
#pragma unroll
    for (int r = 0; r < 16; ++r) { dhc[r] = 0.f; dcc[r] = 0.f; }

    // (the staged variant is ~7 % faster with the recurrence rolled, the in-kernel-gradient variant ~20 % slower)
#ifndef JKB_PG_UNROLL
#define JKB_PG_UNROLL 3
#endif
#pragma clang loop unroll_count(PG ? JKB_PG_UNROLL : 1)
    for (int s = 2; s >= 0; --s) {
      const int t = d ? 2 - s : s, tprev = d ? t + 1 : t - 1;
      const size_t col = (size_t)t * npad + node;
      const float dst = t == 0 ? ds3[0] : (t == 1 ? ds3[1] : ds3[2]);
      float4 xt[XG];
#pragma unroll
      for (int q = 0; q < XG; ++q)
        xt[q] = (valid && (8 * q + 4 * lhi < C)) ? *reinterpret_cast<const float4*>(xrow + t * C + 8 * q) : make_float4(0.f, 0.f, 0.f, 0.f);
      floatx16 hprev, cprev;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int j = (r & 3) + 8 * (r >> 2) + 4 * lhi;
        hprev[r] = 0.f;
        cprev[r] = 0.f;
        if (s > 0 && (r & 3) + 8 * (r >> 2) < H && j < H && valid) {
          const size_t slot = (size_t)((d * 3 + tprev) * H + j) * npad + node;
          hprev[r] = HS[slot];
          cprev[r] = CS[slot];
        }
      }
      // ---- gate pre-activations again (the forward's products)
      floatx16 acc[4];
#pragma unroll
      for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float4 b = *reinterpret_cast<const float4*>(Bl + g * 32 + 8 * q);
          acc[g][4 * q + 0] = b.x; acc[g][4 * q + 1] = b.y; acc[g][4 * q + 2] = b.z; acc[g][4 * q + 3] = b.w;
        }
#pragma unroll
      for (int q = 0; q < XG; ++q) {
        float4 a[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) a[g] = Wl[(g * NQ + q) * 64];
#pragma unroll
        for (int g = 0; g < 4; ++g) acc[g] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[g].x, xt[q].x, acc[g], 0, 0, 0);
#pragma unroll
        for (int g = 0; g < 4; ++g) acc[g] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[g].y, xt[q].y, acc[g], 0, 0, 0);
#pragma unroll
        for (int g = 0; g < 4; ++g) acc[g] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[g].z, xt[q].z, acc[g], 0, 0, 0);
#pragma unroll
        for (int g = 0; g < 4; ++g) acc[g] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[g].w, xt[q].w, acc[g], 0, 0, 0);
      }
      if (s > 0) {
#pragma unroll
        for (int q = 0; q < HG; ++q) {
          float4 a[4];
#pragma unroll
          for (int g = 0; g < 4; ++g) a[g] = Wl[(g * NQ + XG + q) * 64];
#pragma unroll
          for (int g = 0; g < 4; ++g) acc[g] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[g].x, hprev[4 * q + 0], acc[g], 0, 0, 0);
#pragma unroll
          for (int g = 0; g < 4; ++g) acc[g] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[g].y, hprev[4 * q + 1], acc[g], 0, 0, 0);
#pragma unroll
          for (int g = 0; g < 4; ++g) acc[g] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[g].z, hprev[4 * q + 2], acc[g], 0, 0, 0);
#pragma unroll
          for (int g = 0; g < 4; ++g) acc[g] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[g].w, hprev[4 * q + 3], acc[g], 0, 0, 0);
        }
      }
      // ---- cell backward per (node, unit); acc[g] is overwritten with d loss / d pre-activation of gate g
      if (!PG && lhi == 0) {
        dgt[(size_t)(4 * H) * ktot + col] = dst;
        inT[(size_t)(C + H) * ktot + col] = 1.f;
      }
#pragma unroll
      for (int q = 0; q < XG; ++q) {
        if (!PG && 8 * q + 4 * lhi < C) {
          inT[(size_t)(8 * q + 4 * lhi + 0) * ktot + col] = xt[q].x;
          inT[(size_t)(8 * q + 4 * lhi + 1) * ktot + col] = xt[q].y;
          inT[(size_t)(8 * q + 4 * lhi + 2) * ktot + col] = xt[q].z;
          inT[(size_t)(8 * q + 4 * lhi + 3) * ktot + col] = xt[q].w;
        }
      }
#pragma unroll
      for (int q = 0; q < HG; ++q) {
        const float4 wa = *reinterpret_cast<const float4*>(Al + 8 * q);
        const float was[4] = {wa.x, wa.y, wa.z, wa.w};
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int r = 4 * q + u, j = u + 8 * q + 4 * lhi;
          const float gi = fast_sigmoid(acc[0][r]), gf = fast_sigmoid(acc[1][r]);
          const float gg = fast_tanh(acc[2][r]), go = fast_sigmoid(acc[3][r]);
          const float c = gf * cprev[r] + gi * gg;
          const float th = fast_tanh(c);
          const float dh = fmaf(dst, was[u], dhc[r]);
          const float dc = fmaf(dh * go, 1.f - th * th, dcc[r]);
          dcc[r] = dc * gf;
          const float qi = dc * gg * gi * (1.f - gi), qf = dc * cprev[r] * gf * (1.f - gf);
          const float qg = dc * gi * (1.f - gg * gg), qo = dh * th * go * (1.f - go);
          acc[0][r] = qi; acc[1][r] = qf; acc[2][r] = qg; acc[3][r] = qo;
          if (PG) wacc[r] = fmaf(dst, go * th, wacc[r]);          // d w_att[j] += ds_t * h_t[j]
          if (!PG && 8 * q + u < H && j < H) {
            dgt[(size_t)(0 * H + j) * ktot + col] = qi;
            dgt[(size_t)(1 * H + j) * ktot + col] = qf;
            dgt[(size_t)(2 * H + j) * ktot + col] = qg;
            dgt[(size_t)(3 * H + j) * ktot + col] = qo;
            inT[(size_t)(C + j) * ktot + col] = hprev[r];
            inT[(size_t)(C + H + 1 + j) * ktot + col] = go * th;
          }
        }
      }
      if (PG) {
        if (lhi == 0) bacc += dst;                                // d b_att (one lane half per node)
        // inputs of this step, node-major: tile 0 = h_{t-1} (+ 1.0 at column 31: the bias column), tile 1 = x_t
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
          if (q < HG) v = make_float4(hprev[4 * q + 0], hprev[4 * q + 1], hprev[4 * q + 2], hprev[4 * q + 3]);
          if (q == 3 && lhi == 1) v.w = 1.f;
          *reinterpret_cast<float4*>(Iw + l31 * 36 + 8 * q + 4 * lhi) = v;
          float4 xv = make_float4(0.f, 0.f, 0.f, 0.f);
          if (q < XG) xv = xt[q];
          *reinterpret_cast<float4*>(Iw + 32 * 36 + l31 * 36 + 8 * q + 4 * lhi) = xv;
        }
        __builtin_amdgcn_wave_barrier();
#ifndef JKB_BF_RELOAD
#define JKB_BF_RELOAD 0      // 1: the input fragments are read from the LDS tile again for every gate (32 registers less held)
#endif
        float bf[2][16];
        if (!JKB_BF_RELOAD) {
#pragma unroll
          for (int kk = 0; kk < 16; ++kk) {
            bf[0][kk] = Iw[(2 * kk + lhi) * 36 + l31];
            bf[1][kk] = Iw[32 * 36 + (2 * kk + lhi) * 36 + l31];
          }
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          if (JKB_BF_RELOAD) {
#pragma unroll
            for (int kk = 0; kk < 16; ++kk) {
              bf[0][kk] = Iw[(2 * kk + lhi) * 36 + l31];
              bf[1][kk] = Iw[32 * 36 + (2 * kk + lhi) * 36 + l31];
            }
          }
          __builtin_amdgcn_wave_barrier();
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (q < HG) v = make_float4(acc[g][4 * q + 0], acc[g][4 * q + 1], acc[g][4 * q + 2], acc[g][4 * q + 3]);
            *reinterpret_cast<float4*>(Qw + l31 * 36 + 8 * q + 4 * lhi) = v;
          }
          __builtin_amdgcn_wave_barrier();
          float af[16];
#pragma unroll
          for (int kk = 0; kk < 16; ++kk) af[kk] = Qw[(2 * kk + lhi) * 36 + l31];
#pragma unroll
          for (int kk = 0; kk < 16; ++kk) {
            dW[g][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[kk], bf[0][kk], dW[g][0], 0, 0, 0);
            dW[g][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[kk], bf[1][kk], dW[g][1], 0, 0, 0);
          }
        }
      }
      // ---- d[h_{t-1} | x_t]^T = W^T q : A = W^T rows (h unit | x feature) = l31, B = q straight from the registers
      floatx16 dh2, dx2;
#pragma unroll
      for (int r = 0; r < 16; ++r) { dh2[r] = 0.f; dx2[r] = 0.f; }
#pragma unroll
      for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int q = 0; q < HG; ++q) {
          float ah[4], ax[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            ax[u] = WtX[g * NQ * 256 + (8 * q + u) * 8];
            ah[u] = s > 0 ? WtH[g * NQ * 256 + (8 * q + u) * 8] : 0.f;
          }
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            dx2 = __builtin_amdgcn_mfma_f32_32x32x2f32(ax[u], acc[g][4 * q + u], dx2, 0, 0, 0);
            if (s > 0) dh2 = __builtin_amdgcn_mfma_f32_32x32x2f32(ah[u], acc[g][4 * q + u], dh2, 0, 0, 0);
          }
        }
      if (s > 0) dhc = dh2;
#pragma unroll
      for (int q = 0; q < XG; ++q) Ex[(t * XG + q) * 64] = make_float4(dx2[4 * q + 0], dx2[4 * q + 1], dx2[4 * q + 2], dx2[4 * q + 3]);
    }

    // ---- input gradient: attention term + both directions' LSTM terms (parked in LDS per time slot)
    __syncthreads();
    if (d == 0 && valid) {
      const float4* Eo = Ex + 3 * XG * 64;           // the reverse-direction wave of the same node tile
#pragma unroll
      for (int q = 0; q < XG; ++q) {
        if (8 * q + 4 * lhi < C) {
          const float4 dy = *reinterpret_cast<const float4*>(dout + (size_t)node * C + 8 * q + 4 * lhi);
#pragma unroll
          for (int t = 0; t < 3; ++t) {
            const float4 p0 = Ex[(t * XG + q) * 64], p1 = Eo[(t * XG + q) * 64];
            float4 v;
            v.x = fmaf(a3[t], dy.x, p0.x + p1.x); v.y = fmaf(a3[t], dy.y, p0.y + p1.y);
            v.z = fmaf(a3[t], dy.z, p0.z + p1.z); v.w = fmaf(a3[t], dy.w, p0.w + p1.w);
            *reinterpret_cast<float4*>(dxs + (size_t)node * 3 * C + t * C + 8 * q + 4 * lhi) = v;
          }
        }
      }
    }
    __syncthreads();
  }
  if (PG) {       // this wave's partial parameter gradients: [8 tiles x 16 registers | 16 attention registers | bias] x 64 lanes
    float* pw = PART + ((size_t)d * gridDim.x * JKB_TILES + (size_t)blockIdx.x * JKB_TILES + half) * JkM<C>::PG_FLOATS + lane;
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) pw[((g * 2 + m) * 16 + r) * 64] = dW[g][m][r];
#pragma unroll
    for (int r = 0; r < 16; ++r) pw[(128 + r) * 64] = wacc[r];
    pw[144 * 64] = bacc;
  }
}

// parameter-gradient partials -> G [2][4H+1][C+2H+1] (the layout ops.py reads: rows = gate pre-activations then the
// attention score, columns = x (C) | h_{t-1} (H) | 1 | h_t (H)), two deterministic stages
__global__ void k_jk_pg_fold(const float* __restrict__ part, int P, int P2, int per, float* __restrict__ tmp) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x, p2 = blockIdx.y, d = blockIdx.z;
  if (e >= per) return;
  const float* src = part + ((size_t)d * P + (size_t)p2 * 16) * per + e;
  const int cnt = min(16, P - p2 * 16);
  float a = 0.f;
  for (int k = 0; k < cnt; ++k) a += src[(size_t)k * per];
  tmp[((size_t)d * P2 + p2) * per + e] = a;
}

template <int C>
__global__ void k_jk_pg_finish(const float* __restrict__ tmp, int P2, float* __restrict__ G) {
  constexpr int H = JkM<C>::H, per = JkM<C>::PG_FLOATS, NG = 4 * H + 1, NI = C + 2 * H + 1;
  const int e = blockIdx.x * blockDim.x + threadIdx.x, d = blockIdx.y;
  if (e >= per) return;
  const int slot = e >> 6, lane = e & 63, lhi = lane >> 5, kin = lane & 31;
  const float* src = tmp + (size_t)d * P2 * per;
  if (slot < 128) {
    const int tile = slot >> 4, r = slot & 15, g = tile >> 1, m = tile & 1;
    const int j = (r & 3) + 8 * (r >> 2) + 4 * lhi;
    int col = -1;
    if (m == 0) col = kin < H ? C + kin : (kin == 31 ? C + H : -1);
    else col = kin < C ? kin : -1;
    if (j >= H || col < 0) return;
    float a = 0.f;
    for (int k = 0; k < P2; ++k) a += src[(size_t)k * per + e];
    G[((size_t)d * NG + g * H + j) * NI + col] = a;
  } else if (kin == 0) {                        // per-lane (= per-node) sums: fold the 32 lanes of this half
    const int r = slot - 128;
    if (slot < 144) {
      const int j = (r & 3) + 8 * (r >> 2) + 4 * lhi;
      if (j >= H) return;
      float a = 0.f;
      for (int k = 0; k < P2; ++k)
        for (int l = 0; l < 32; ++l) a += src[(size_t)k * per + slot * 64 + lhi * 32 + l];
      G[((size_t)d * NG + 4 * H) * NI + C + H + 1 + j] = a;
    } else if (lhi == 0 && d == 0) {
      float a = 0.f;
      for (int k = 0; k < P2; ++k)
        for (int l = 0; l < 32; ++l) a += src[(size_t)k * per + slot * 64 + l];
      G[((size_t)4 * H) * NI + C + H] = a;
    }
  }
}

// the same second stage writing the parameter gradients straight into the flat buffer of cgc_jk_unpack_param_grads (per direction
// dW_ih [4H,C] | dW_hh [4H,H] | db_ih [4H] | db_hh [4H], then d att.weight [2H], d att.bias): no G, no memset, no unpack kernel.
// Same summation order per element as k_jk_pg_finish.
template <int C>
__global__ void k_jk_pg_finish_flat(const float* __restrict__ tmp, int P2, float* __restrict__ flat) {
  constexpr int H = JkM<C>::H, per = JkM<C>::PG_FLOATS;
  constexpr int per_dir = 4 * H * C + 4 * H * H + 8 * H, total = 2 * per_dir + 2 * H + 1;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  int d, row = 0, col = 0, att = -1;               // G coordinates of flat element i: (d, row = g*H + j, col) or an attention entry
  if (i < 2 * per_dir) {
    d = i / per_dir;
    int e = i - d * per_dir;
    if (e < 4 * H * C) { row = e / C; col = e % C; }
    else if ((e -= 4 * H * C) < 4 * H * H) { row = e / H; col = C + e % H; }
    else { e -= 4 * H * H; row = e % (4 * H); col = C + H; }
  } else {
    const int e = i - 2 * per_dir;
    if (e < 2 * H) { d = e / H; att = e % H; }
    else { d = 0; att = H; }                        // attention bias
  }
  const float* src = tmp + (size_t)d * P2 * per;
  float a = 0.f;
  if (att < 0) {
    const int g = row / H, j = row - g * H;
    const int m = col < C ? 1 : 0;                 // tile: m = 1 the x part, m = 0 the h part with the bias column at lane 31
    const int kin = col < C ? col : (col == C + H ? 31 : col - C);
    const int lhi = (j >> 2) & 1, r = (j & 3) + 4 * (j >> 3);
    const int e = ((g * 2 + m) * 16 + r) * 64 + lhi * 32 + kin;
    for (int k = 0; k < P2; ++k) a += src[(size_t)k * per + e];
  } else if (att < H) {
    const int j = att, lhi = (j >> 2) & 1, r = (j & 3) + 4 * (j >> 3), slot = 128 + r;
    for (int k = 0; k < P2; ++k)
      for (int l = 0; l < 32; ++l) a += src[(size_t)k * per + slot * 64 + lhi * 32 + l];
  } else {
    for (int k = 0; k < P2; ++k)
      for (int l = 0; l < 32; ++l) a += src[(size_t)k * per + 144 * 64 + l];
  }
  flat[i] = a;
}


// =====================================================================================================================
// Unit-split kernels (round 3).  The kernels above give one wave a whole (32-node tile, direction): 368 MFMAs per recurrence
// step in the backward, 8 persistent accumulator tiles, 213 spilled registers, and ~100 us of strictly serial work per tile --
// the launch is as long as its longest wave however few nodes there are (3 x 107 us at 4 graphs per GPU).  Here EIGHT waves
// share a tile: wave (d, w) owns direction d and the hidden units 8w .. 8w+7.  Its M tile of the transposed gate product is
// rows rho = 8 g + u (gate g, own unit u): in the accumulator layout register 4g + i of lane (node, half) is gate g of unit
// 8w + 4 half + i, so a lane still holds all four gates of its (node, unit) pairs and the cell stays per-register arithmetic.
// What no longer stays in registers is exchanged through LDS once per step: the new h (forward: 16 bytes per lane, double
// buffered, one barrier), and in the backward the partial products of d[h|x]^T = W^T q, which contract over ALL (gate, unit)
// rows (two barriers).  Per wave and step: 28 + 32 + 32 MFMAs instead of 112 + 128 + 128, two accumulator tiles for the
// parameter gradients instead of eight, no spills.
// =====================================================================================================================
template <int C>
struct JkU {
  static constexpr int H = 3 * C / 2, XG = (C + 7) / 8, HG = (H + 7) / 8, NQ = XG + HG;
  static constexpr int W_FLOATS = 2 * 4 * NQ * 256;            // [d][w][q][rho 32][lane half 2][t 4]
  static constexpr int B_OFF = W_FLOATS;                       // bias [d][w][rho 32]
  static constexpr int A_OFF = B_OFF + 256;                    // w_att [d][32], b_att
  static constexpr int S_OFF = A_OFF + 68;                     // score partials [wave 8][t 3][32]
  static constexpr int X_OFF = S_OFF + 768;                    // forward: h exchange [buffer 2][wave 8][64] float4
  static constexpr int FWD_TOTAL = X_OFF + 2 * 8 * 256;
  static constexpr int Q_OFF = S_OFF + 768;                    // backward, per wave: q transposed [node 32][36]; then its dh partials
  static constexpr int I_OFF = Q_OFF + 8 * 1152;               // cell inputs, node-major [d][h tile | x tile][32][36]
  static constexpr int P_OFF = I_OFF + 4 * 1152;               // dx partials [wave 8][XG][64] float4; at the end [d][q][t][64]
  static constexpr int BWD_TOTAL = P_OFF + 8 * XG * 256;
  static constexpr int PG_SLOTS = 37;                          // per wave: 2 x 16 accumulator registers, 4 attention registers, bias
  static constexpr int PG_FLOATS = 8 * PG_SLOTS * 64;          // parameter-gradient partial of one workgroup
  static_assert(H <= 30 && C % 4 == 0, "column 31 of the h tile is the bias column; 16-byte x fragments");
};

template <int C>
__device__ __forceinline__ void jku_fill(const JkWeights& w, float* lds) {
  using U = JkU<C>;
  constexpr int H = U::H, XG = U::XG, NQ = U::NQ;
  const int tid = threadIdx.x, d = tid >> 8;
  const float* wih = d ? w.w_ih[1] : w.w_ih[0];
  const float* whh = d ? w.w_hh[1] : w.w_hh[0];
  const float* bih = d ? w.b_ih[1] : w.b_ih[0];
  const float* bhh = d ? w.b_hh[1] : w.b_hh[0];
  const int tt = tid & 3, lhi = (tid >> 2) & 1, rho = (tid >> 3) & 31, g = rho >> 3, u = rho & 7;
  float v[4 * NQ];
#pragma unroll
  for (int it = 0; it < 4 * NQ; ++it) {
    const int q = it % NQ, ww = it / NQ, j = 8 * ww + u;
    const int jr = j < H ? j : H - 1;                  // unconditional loads from clamped addresses + select (no branch per load)
    if (q < XG) {
      const int k = 8 * q + 4 * lhi + tt;
      const float ld = wih[(g * H + jr) * C + (k < C ? k : C - 1)];
      v[it] = (j < H && k < C) ? ld : 0.f;
    } else {
      const int k = 8 * (q - XG) + 4 * lhi + tt;
      const float ld = whh[(g * H + jr) * H + (k < H ? k : H - 1)];
      v[it] = (j < H && k < H) ? ld : 0.f;
    }
  }
#pragma unroll
  for (int it = 0; it < 4 * NQ; ++it) lds[(d * 4 * NQ + it) * 256 + (tid & 255)] = v[it];
  if ((tid & 255) < 128) {
    const int r2 = tid & 31, ww = (tid >> 5) & 3, j = 8 * ww + (r2 & 7), gg = r2 >> 3;
    lds[U::B_OFF + (d * 4 + ww) * 32 + r2] = j < H ? bih[gg * H + j] + bhh[gg * H + j] : 0.f;
  } else if ((tid & 255) < 160) {
    const int j = tid & 31;
    lds[U::A_OFF + d * 32 + j] = j < H ? w.w_att[d * H + j] : 0.f;
  }
  if (tid == 0) lds[U::A_OFF + 64] = w.b_att[0];
}

template <int C>
__global__ __launch_bounds__(512) void k_jku_fwd(const float* __restrict__ xs, int n, int npad, const JkWeights w,
                                                 float* __restrict__ out, float* __restrict__ HS, float* __restrict__ CS) {
  using U = JkU<C>;
  constexpr int H = U::H, XG = U::XG, HG = U::HG, NQ = U::NQ;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  jku_fill<C>(w, lds);
  __syncthreads();
  const int lane = threadIdx.x & 63, l31 = lane & 31, lhi = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), d = wave & 1, ww = wave >> 1;      // (wave-uniform: SGPRs)
  const float4* Wl = reinterpret_cast<const float4*>(lds) + (size_t)(d * 4 + ww) * NQ * 64 + l31 * 2 + lhi;   // + q*64
  const float* Bl = lds + U::B_OFF + (d * 4 + ww) * 32 + 4 * lhi;                                             // + 8g
  const float4 wa4 = *reinterpret_cast<const float4*>(lds + U::A_OFF + d * 32 + 8 * ww + 4 * lhi);
  const float was[4] = {wa4.x, wa4.y, wa4.z, wa4.w};
  float* Sx = lds + U::S_OFF;
  float4* Hx = reinterpret_cast<float4*>(lds + U::X_OFF);
  const int ntiles = (n + 31) / 32;

  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int node = tile * 32 + l31;
    const bool valid = node < n, keep = node < npad;
    const float* xrow = xs + (size_t)(valid ? node : 0) * 3 * C + 4 * lhi;
    float cst[4] = {0.f, 0.f, 0.f, 0.f};
    float4 hq[HG];
#pragma unroll
    for (int q = 0; q < HG; ++q) hq[q] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int s = 0; s < 3; ++s) {
      const int t = d ? 2 - s : s;
      floatx16 acc;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float4 b = *reinterpret_cast<const float4*>(Bl + 8 * g);
        acc[4 * g + 0] = b.x; acc[4 * g + 1] = b.y; acc[4 * g + 2] = b.z; acc[4 * g + 3] = b.w;
      }
#pragma unroll
      for (int q = 0; q < XG; ++q) {
        // (unconditional load from a clamped address + select: a predicated load is a branch)
        const bool in = 8 * q + 4 * lhi < C;
        const float4 ld = *reinterpret_cast<const float4*>(xrow + t * C + (in ? 8 * q : 8 * q - 4));
        const float4 xf = (valid && in) ? ld : make_float4(0.f, 0.f, 0.f, 0.f);
        const float4 a = Wl[q * 64];
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, xf.x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, xf.y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, xf.z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, xf.w, acc, 0, 0, 0);
      }
      if (s > 0) {
#pragma unroll
        for (int q = 0; q < HG; ++q) {
          const float4 a = Wl[(XG + q) * 64];
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, hq[q].x, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, hq[q].y, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, hq[q].z, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, hq[q].w, acc, 0, 0, 0);
        }
      }
      float p = 0.f, hn[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int j = 8 * ww + 4 * lhi + i;
        const float gi = fast_sigmoid(acc[i]), gf = fast_sigmoid(acc[4 + i]);
        const float gg = fast_tanh(acc[8 + i]), go = fast_sigmoid(acc[12 + i]);
        const float c = gf * cst[i] + gi * gg;
        const float h = go * fast_tanh(c);
        cst[i] = c;
        hn[i] = h;
        p = fmaf(was[i], h, p);
        if (j < H && keep) {
          const size_t slot = (size_t)((d * 3 + t) * H + j) * npad + node;
          HS[slot] = h;
          CS[slot] = c;
        }
      }
      p += __shfl_xor(p, 32);
      if (lhi == 0) Sx[(wave * 3 + t) * 32 + l31] = p;
      if (s < 2) {        // h_t of all 32 units = the B operand of the next step's recurrent product: one 16-byte piece per wave
        Hx[((s & 1) * 8 + wave) * 64 + lane] = make_float4(hn[0], hn[1], hn[2], hn[3]);
        __syncthreads();
#pragma unroll
        for (int q = 0; q < HG; ++q) hq[q] = Hx[((s & 1) * 8 + q * 2 + d) * 64 + lane];
      }
    }
    __syncthreads();
    if (d == 0 && ww < XG) {
      float sc[3];
#pragma unroll
      for (int t = 0; t < 3; ++t) {
        float a = lds[U::A_OFF + 64];
#pragma unroll
        for (int v = 0; v < 8; ++v) a += Sx[(v * 3 + t) * 32 + l31];
        sc[t] = a;
      }
      const float m = fmaxf(sc[0], fmaxf(sc[1], sc[2]));
      float a3[3], den = 0.f;
#pragma unroll
      for (int t = 0; t < 3; ++t) { a3[t] = JK_EXP(sc[t] - m); den += a3[t]; }
      const float inv = 1.f / den;
      const int q = ww;
      if (valid && 8 * q + 4 * lhi < C) {
        const float4 x0 = *reinterpret_cast<const float4*>(xrow + 0 * C + 8 * q);
        const float4 x1 = *reinterpret_cast<const float4*>(xrow + 1 * C + 8 * q);
        const float4 x2 = *reinterpret_cast<const float4*>(xrow + 2 * C + 8 * q);
        float4 o;
        o.x = (a3[0] * x0.x + a3[1] * x1.x + a3[2] * x2.x) * inv;
        o.y = (a3[0] * x0.y + a3[1] * x1.y + a3[2] * x2.y) * inv;
        o.z = (a3[0] * x0.z + a3[1] * x1.z + a3[2] * x2.z) * inv;
        o.w = (a3[0] * x0.w + a3[1] * x1.w + a3[2] * x2.w) * inv;
        *reinterpret_cast<float4*>(out + (size_t)node * C + 8 * q + 4 * lhi) = o;
      }
    }
    __syncthreads();          // the score partials are rewritten by the next tile
  }
}

// Backward, unit-split (parameter gradients accumulated in-kernel; the staged variant stays with k_jk_bwd_mfma<C, false>).
template <int C>
__global__ __launch_bounds__(512) void k_jku_bwd(const float* __restrict__ xs, const float* __restrict__ dout, int n, int npad,
                                                 const JkWeights w, const float* __restrict__ HS, const float* __restrict__ CS,
                                                 float* __restrict__ dxs, float* __restrict__ PART) {
  using U = JkU<C>;
  constexpr int H = U::H, XG = U::XG, HG = U::HG, NQ = U::NQ;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  jku_fill<C>(w, lds);
  __syncthreads();
  const int lane = threadIdx.x & 63, l31 = lane & 31, lhi = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), d = wave & 1, ww = wave >> 1;      // (wave-uniform: SGPRs)
  const float4* Wl = reinterpret_cast<const float4*>(lds) + (size_t)(d * 4 + ww) * NQ * 64 + l31 * 2 + lhi;
  const float* Bl = lds + U::B_OFF + (d * 4 + ww) * 32 + 4 * lhi;
  const float4 wa4 = *reinterpret_cast<const float4*>(lds + U::A_OFF + d * 32 + 8 * ww + 4 * lhi);
  const float was[4] = {wa4.x, wa4.y, wa4.z, wa4.w};
  float* Sx = lds + U::S_OFF;
  float* Qw = lds + U::Q_OFF + wave * 1152;
  float* Iw = lds + U::I_OFF + d * 2 * 1152;
  float4* Dx = reinterpret_cast<float4*>(lds + U::P_OFF);
  // W^T fragments out of the forward image: row kin = l31 of the (h | x) tile, k = this wave's (gate, unit) rows
  const int kh = l31 < 8 * HG ? l31 : 8 * HG - 1, kx = l31 < 8 * XG ? l31 : 8 * XG - 1;
  const float* WtH = lds + (size_t)((d * 4 + ww) * NQ + XG + kh / 8) * 256 + (kh & 7) + 32 * lhi;   // + (r>>2)*64 + (r&3)*8
  const float* WtX = lds + (size_t)((d * 4 + ww) * NQ + kx / 8) * 256 + (kx & 7) + 32 * lhi;
  const int ntiles = (n + 31) / 32;
  floatx16 dW0, dW1;
#pragma unroll
  for (int r = 0; r < 16; ++r) { dW0[r] = 0.f; dW1[r] = 0.f; }
  float wacc[4] = {0.f, 0.f, 0.f, 0.f}, bacc = 0.f;

  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int node = tile * 32 + l31;
    const bool valid = node < n;
    const int nd = valid ? node : 0;
    // saved states are read as  scalar row base + one of two lane offsets  (unit j = 8 q + 4 half + i: half 1 is dropped for the
    // units past H, whose value is discarded anyway): no per-load address arithmetic, no predicated loads
    const size_t off0 = (size_t)nd, off1 = (size_t)nd + (size_t)(4 * lhi) * npad;
    const float* xrow = xs + (size_t)(valid ? node : 0) * 3 * C + 4 * lhi;
    // ---- attention: score partials over this wave's units, d out . x_t per time slot
    float da[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < 3; ++t) {
      float p = 0.f;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        // (unconditional loads from clamped addresses + select: a predicated load is a branch per load)
        const int j = 8 * ww + 4 * lhi + i, jr = 8 * ww + i < H ? 8 * ww + i : H - 1;
        const float hv = (HS + (size_t)((d * 3 + t) * H + jr) * npad)[8 * ww + 4 + i < H ? off1 : off0];
        p = fmaf(was[i], (j < H && valid) ? hv : 0.f, p);
      }
      p += __shfl_xor(p, 32);
      if (lhi == 0) Sx[(wave * 3 + t) * 32 + l31] = p;
    }
#pragma unroll
    for (int q = 0; q < XG; ++q) {
      if (valid && (8 * q + 4 * lhi < C)) {
        const float4 dy = *reinterpret_cast<const float4*>(dout + (size_t)node * C + 8 * q + 4 * lhi);
#pragma unroll
        for (int t = 0; t < 3; ++t) {
          const float4 x = *reinterpret_cast<const float4*>(xrow + t * C + 8 * q);
          da[t] += dy.x * x.x + dy.y * x.y + dy.z * x.z + dy.w * x.w;
        }
      }
    }
    __syncthreads();
    float a3[3], ds3[3];
    {
      float sc[3];
#pragma unroll
      for (int t = 0; t < 3; ++t) {
        float a = lds[U::A_OFF + 64];
#pragma unroll
        for (int v = 0; v < 8; ++v) a += Sx[(v * 3 + t) * 32 + l31];
        sc[t] = a;
      }
      const float m = fmaxf(sc[0], fmaxf(sc[1], sc[2]));
      float den = 0.f, mean = 0.f;
#pragma unroll
      for (int t = 0; t < 3; ++t) { a3[t] = JK_EXP(sc[t] - m); den += a3[t]; }
      const float inv = 1.f / den;
#pragma unroll
      for (int t = 0; t < 3; ++t) {
        da[t] += __shfl_xor(da[t], 32);
        a3[t] *= inv;
        mean = fmaf(a3[t], da[t], mean);
      }
#pragma unroll
      for (int t = 0; t < 3; ++t) ds3[t] = valid ? a3[t] * (da[t] - mean) : 0.f;
    }
    float dhc[4] = {0.f, 0.f, 0.f, 0.f}, dcc[4] = {0.f, 0.f, 0.f, 0.f};
    float4 dxs0 = make_float4(0.f, 0.f, 0.f, 0.f), dxs1 = dxs0, dxs2 = dxs0;       // d x_t pieces by recurrence step

    // operands of one recurrence step: x_t, h_{t-1} (all units: the B operand of the recurrent product), c_{t-1} (own units).
    // Loaded for the first step here and for every later one right after the previous step's first barrier, when these
    // registers are dead: the loads travel while the parameter-gradient MFMAs run instead of in front of the gate product.
    float4 xt[XG], hq[HG];
    float cprev[4];
#define JKU_LOAD_STEP(S_)                                                                                                   \
    {                                                                                                                       \
      const int s_ = (S_), t_ = d ? 2 - s_ : s_, tp_ = d ? t_ + 1 : t_ - 1;                                                  \
      _Pragma("unroll") for (int q = 0; q < XG; ++q)                                                                        \
      {                                                                                                                     \
        const bool in_ = 8 * q + 4 * lhi < C;                                                                               \
        const float4 ld_ = *reinterpret_cast<const float4*>(xs + (size_t)nd * 3 * C + t_ * C + (in_ ? 8 * q + 4 * lhi : 0)); \
        xt[q] = (valid && in_) ? ld_ : make_float4(0.f, 0.f, 0.f, 0.f);                                                     \
      }                                                                                                                     \
      _Pragma("unroll") for (int q = 0; q < HG; ++q) {                                                                      \
        float hv[4] = {0.f, 0.f, 0.f, 0.f};                                                                                 \
        if (s_ > 0) {                                                                                                       \
          _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                                   \
            const int j = 8 * q + 4 * lhi + i;                                                                              \
            const float ld_ = (HS + (size_t)((d * 3 + tp_) * H + (8 * q + i < H ? 8 * q + i : H - 1)) * npad)               \
                [8 * q + 4 + i < H ? off1 : off0];                                                                          \
            hv[i] = (j < H && valid) ? ld_ : 0.f;                                                                           \
          }                                                                                                                 \
        }                                                                                                                   \
        hq[q] = make_float4(hv[0], hv[1], hv[2], hv[3]);                                                                    \
      }                                                                                                                     \
      _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                                       \
        const int j = 8 * ww + 4 * lhi + i;                                                                                 \
        const int jr_ = 8 * ww + i < H ? 8 * ww + i : H - 1;                                                                \
        const float ld_ = (CS + (size_t)((d * 3 + (s_ > 0 ? tp_ : t_)) * H + jr_) * npad)[8 * ww + 4 + i < H ? off1 : off0]; \
        cprev[i] = (s_ > 0 && j < H && valid) ? ld_ : 0.f;                                                                  \
      }                                                                                                                     \
    }
    JKU_LOAD_STEP(2)
    // (recurrence rolled: unrolled, the scheduler hoists the loads of all three steps and spills 81 registers)
#pragma unroll 1
    for (int s = 2; s >= 0; --s) {
      const int t = d ? 2 - s : s;
      const float dst = t == 0 ? ds3[0] : (t == 1 ? ds3[1] : ds3[2]);
      if (ww == 0) {      // the step's cell inputs, node-major: tile 0 = h_{t-1} (+ 1.0 at column 31: the bias column), tile 1 = x_t
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
          if (q < HG) v = hq[q < HG ? q : 0];
          if (q == 3 && lhi == 1) v.w = 1.f;
          *reinterpret_cast<float4*>(Iw + l31 * 36 + 8 * q + 4 * lhi) = v;
          float4 xv = make_float4(0.f, 0.f, 0.f, 0.f);
          if (q < XG) xv = xt[q < XG ? q : 0];
          *reinterpret_cast<float4*>(Iw + 1152 + l31 * 36 + 8 * q + 4 * lhi) = xv;
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      // ---- gate pre-activations again
      floatx16 acc;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float4 b = *reinterpret_cast<const float4*>(Bl + 8 * g);
        acc[4 * g + 0] = b.x; acc[4 * g + 1] = b.y; acc[4 * g + 2] = b.z; acc[4 * g + 3] = b.w;
      }
#pragma unroll
      for (int q = 0; q < XG; ++q) {
        const float4 a = Wl[q * 64];
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, xt[q].x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, xt[q].y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, xt[q].z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, xt[q].w, acc, 0, 0, 0);
      }
      if (s > 0) {
#pragma unroll
        for (int q = 0; q < HG; ++q) {
          const float4 a = Wl[(XG + q) * 64];
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, hq[q].x, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, hq[q].y, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, hq[q].z, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, hq[q].w, acc, 0, 0, 0);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      // ---- cell backward for the own (node, unit) pairs; acc becomes d loss / d pre-activation
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float gi = fast_sigmoid(acc[i]), gf = fast_sigmoid(acc[4 + i]);
        const float gg = fast_tanh(acc[8 + i]), go = fast_sigmoid(acc[12 + i]);
        const float c = gf * cprev[i] + gi * gg;
        const float th = fast_tanh(c);
        const float dh = fmaf(dst, was[i], dhc[i]);
        const float dc = fmaf(dh * go, 1.f - th * th, dcc[i]);
        dcc[i] = dc * gf;
        acc[i] = dc * gg * gi * (1.f - gi);
        acc[4 + i] = dc * cprev[i] * gf * (1.f - gf);
        acc[8 + i] = dc * gi * (1.f - gg * gg);
        acc[12 + i] = dh * th * go * (1.f - go);
        wacc[i] = fmaf(dst, go * th, wacc[i]);          // d w_att[j] += ds_t * h_t[j]
      }
      if (wave == 0 && lhi == 0) bacc += dst;            // d b_att
      __builtin_amdgcn_sched_barrier(0);
      // q transposed through the wave's tile: rows (gate, unit) become the lane index, node pairs the MFMA k index
#pragma unroll
      for (int g = 0; g < 4; ++g)
        *reinterpret_cast<float4*>(Qw + l31 * 36 + 8 * g + 4 * lhi) = make_float4(acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]);
      __builtin_amdgcn_wave_barrier();
      float af[16];
#pragma unroll
      for (int kk = 0; kk < 16; ++kk) af[kk] = Qw[(2 * kk + lhi) * 36 + l31];
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_sched_barrier(0);
      // ---- this wave's share of d[h_{t-1} | x_t]^T = W^T q (contraction over its 32 (gate, unit) rows)
      floatx16 dh2, dx2;
#pragma unroll
      for (int r = 0; r < 16; ++r) { dh2[r] = 0.f; dx2[r] = 0.f; }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float ax = WtX[(r >> 2) * 64 + (r & 3) * 8];
        dx2 = __builtin_amdgcn_mfma_f32_32x32x2f32(ax, acc[r], dx2, 0, 0, 0);
        if (s > 0) {
          const float ah = WtH[(r >> 2) * 64 + (r & 3) * 8];
          dh2 = __builtin_amdgcn_mfma_f32_32x32x2f32(ah, acc[r], dh2, 0, 0, 0);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      float4* Pw = reinterpret_cast<float4*>(Qw);       // the transposed tile is consumed (af): its space takes the dh partials
      if (s > 0) {
#pragma unroll
        for (int q = 0; q < 4; ++q) Pw[q * 64 + lane] = make_float4(dh2[4 * q], dh2[4 * q + 1], dh2[4 * q + 2], dh2[4 * q + 3]);
      }
#pragma unroll
      for (int q = 0; q < XG; ++q) Dx[(wave * XG + q) * 64 + lane] = make_float4(dx2[4 * q], dx2[4 * q + 1], dx2[4 * q + 2], dx2[4 * q + 3]);
      __syncthreads();
      if (s > 0) JKU_LOAD_STEP(s - 1)
      __builtin_amdgcn_sched_barrier(0);
      // ---- parameter gradients of the own rows: dW[(g,u)][kin] += sum_node q . in
      {
        float bf0[16], bf1[16];
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) {
          bf0[kk] = Iw[(2 * kk + lhi) * 36 + l31];
          bf1[kk] = Iw[1152 + (2 * kk + lhi) * 36 + l31];
        }
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) {
          dW0 = __builtin_amdgcn_mfma_f32_32x32x2f32(af[kk], bf0[kk], dW0, 0, 0, 0);
          dW1 = __builtin_amdgcn_mfma_f32_32x32x2f32(af[kk], bf1[kk], dW1, 0, 0, 0);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      if (s > 0) {         // d h_{t-1} of the own units: the four waves' partials, added in wave order
        const float4* P0 = reinterpret_cast<const float4*>(lds + U::Q_OFF) + ww * 64 + lane;
        float4 a = P0[(0 * 2 + d) * 288];
        const float4 b = P0[(1 * 2 + d) * 288], c2 = P0[(2 * 2 + d) * 288], e = P0[(3 * 2 + d) * 288];
        a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
        a.x += c2.x; a.y += c2.y; a.z += c2.z; a.w += c2.w;
        a.x += e.x; a.y += e.y; a.z += e.z; a.w += e.w;
        dhc[0] = a.x; dhc[1] = a.y; dhc[2] = a.z; dhc[3] = a.w;
      }
      if (ww < XG) {       // d x_t piece q = ww of this direction
        float4 a = Dx[((0 * 2 + d) * XG + ww) * 64 + lane];
#pragma unroll
        for (int v = 1; v < 4; ++v) {
          const float4 b = Dx[((v * 2 + d) * XG + ww) * 64 + lane];
          a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
        }
        if (s == 2) dxs2 = a;
        else if (s == 1) dxs1 = a;
        else dxs0 = a;
      }
      __syncthreads();
    }
    // ---- input gradient: attention term + both directions' LSTM terms
    if (ww < XG) {
      Dx[((d * XG + ww) * 3 + (d ? 2 : 0)) * 64 + lane] = dxs0;
      Dx[((d * XG + ww) * 3 + 1) * 64 + lane] = dxs1;
      Dx[((d * XG + ww) * 3 + (d ? 0 : 2)) * 64 + lane] = dxs2;
    }
    __syncthreads();
    if (d == 0 && ww < XG && valid && 8 * ww + 4 * lhi < C) {
      const int q = ww;
      const float4 dy = *reinterpret_cast<const float4*>(dout + (size_t)node * C + 8 * q + 4 * lhi);
#pragma unroll
      for (int t = 0; t < 3; ++t) {
        const float4 p0 = Dx[((0 * XG + q) * 3 + t) * 64 + lane], p1 = Dx[((1 * XG + q) * 3 + t) * 64 + lane];
        float4 v;
        v.x = fmaf(a3[t], dy.x, p0.x + p1.x); v.y = fmaf(a3[t], dy.y, p0.y + p1.y);
        v.z = fmaf(a3[t], dy.z, p0.z + p1.z); v.w = fmaf(a3[t], dy.w, p0.w + p1.w);
        *reinterpret_cast<float4*>(dxs + (size_t)node * 3 * C + t * C + 8 * q + 4 * lhi) = v;
      }
    }
    // (no barrier needed here: the next tile's first barrier comes before anything above is overwritten -- the score partials
    // are rewritten, but they were last read before the first step's barriers)
  }
#undef JKU_LOAD_STEP
  float* pw = PART + (size_t)blockIdx.x * U::PG_FLOATS + (size_t)wave * U::PG_SLOTS * 64 + lane;
#pragma unroll
  for (int r = 0; r < 16; ++r) { pw[r * 64] = dW0[r]; pw[(16 + r) * 64] = dW1[r]; }
  // per-node sums of the attention gradients: folded over the 32 nodes of the lane half here (fixed butterfly order), so that the
  // finishing kernel adds one value per workgroup instead of 32
#pragma unroll
  for (int o = 16; o >= 1; o >>= 1) {
#pragma unroll
    for (int i = 0; i < 4; ++i) wacc[i] += __shfl_xor(wacc[i], o);
    bacc += __shfl_xor(bacc, o);
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) pw[(32 + i) * 64] = wacc[i];
  pw[36 * 64] = bacc;
}

// element (d, row = g H + j, col) of G [2][4H+1][C+2H+1] / attention entry `att` (0..H-1: d w_att[d H + att]; H: d b_att) out of
// the folded partials of the unit-split backward
template <int C>
__device__ __forceinline__ float jku_gather(const float* __restrict__ tmp, int P2, int d, int row, int col, int att) {
  using U = JkU<C>;
  constexpr int H = U::H, per = U::PG_FLOATS;
  float a = 0.f;
  if (att < 0) {
    const int g = row / H, j = row - g * H, ww = j >> 3, u = j & 7, lhi = u >> 2, r = 4 * g + (u & 3);
    const int m = col < C ? 1 : 0;                 // tile: 1 = the x part, 0 = the h part with the bias column at lane 31
    const int kin = col < C ? col : (col == C + H ? 31 : col - C);
    const int e = ((ww * 2 + d) * U::PG_SLOTS + m * 16 + r) * 64 + lhi * 32 + kin;
    for (int k = 0; k < P2; ++k) a += tmp[(size_t)k * per + e];
  } else if (att < H) {
    const int j = att, ww = j >> 3, u = j & 7, lhi = u >> 2;
    const int e = ((ww * 2 + d) * U::PG_SLOTS + 32 + (u & 3)) * 64 + lhi * 32;     // (already summed over the lane half)
    for (int k = 0; k < P2; ++k) a += tmp[(size_t)k * per + e];
  } else {
    for (int k = 0; k < P2; ++k) a += tmp[(size_t)k * per + 36 * 64];
  }
  return a;
}

template <int C>
__global__ void k_jku_finish_flat(const float* __restrict__ tmp, int P2, float* __restrict__ flat) {
  constexpr int H = JkU<C>::H;
  constexpr int per_dir = 4 * H * C + 4 * H * H + 8 * H, total = 2 * per_dir + 2 * H + 1;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  int d, row = 0, col = 0, att = -1;
  if (i < 2 * per_dir) {
    d = i / per_dir;
    int e = i - d * per_dir;
    if (e < 4 * H * C) { row = e / C; col = e % C; }
    else if ((e -= 4 * H * C) < 4 * H * H) { row = e / H; col = C + e % H; }
    else { e -= 4 * H * H; row = e % (4 * H); col = C + H; }
  } else {
    const int e = i - 2 * per_dir;
    if (e < 2 * H) { d = e / H; att = e % H; }
    else { d = 0; att = H; }
  }
  flat[i] = jku_gather<C>(tmp, P2, d, row, col, att);
}

// the same into G [2][4H+1][C+2H+1] (every element written: rows the kernels do not produce are zero)
template <int C>
__global__ void k_jku_finish_G(const float* __restrict__ tmp, int P2, float* __restrict__ G) {
  constexpr int H = JkU<C>::H, NG = 4 * H + 1, NI = C + 2 * H + 1;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= 2 * NG * NI) return;
  const int d = i / (NG * NI), e = i - d * NG * NI, row = e / NI, col = e - row * NI;
  float v = 0.f;
  if (row < 4 * H) {
    if (col <= C + H) v = jku_gather<C>(tmp, P2, d, row, col, -1);
  } else if (col > C + H) {
    v = jku_gather<C>(tmp, P2, d, 0, 0, col - (C + H + 1));
  } else if (col == C + H && d == 0) {
    v = jku_gather<C>(tmp, P2, 0, 0, 0, H);
  }
  G[i] = v;
}

// CGC_JK_UNITSPLIT=0: the one-wave-per-(tile, direction) kernels (A-B timing)
static const int g_unit_split = getenv("CGC_JK_UNITSPLIT") ? atoi(getenv("CGC_JK_UNITSPLIT")) : 1;

template <int C>
static int launch_fwd_us(const float* xs, int n, int npad, const JkWeights& w, float* out, float* HS, float* CS, hipStream_t st) {
  constexpr size_t lds = sizeof(float) * JkU<C>::FWD_TOTAL;
  static bool attr_set[CGC_MAX_DEVICES] = {};
  cgc_allow_lds(reinterpret_cast<const void*>(&k_jku_fwd<C>), (int)lds, attr_set);
  int grid = ceil_div(n, 32);
  static const int fgrid = getenv("CGC_JKU_FGRID") ? atoi(getenv("CGC_JKU_FGRID")) : 512;
  if (grid > fgrid) grid = fgrid;             // persistent over the 32-node tiles
  hipLaunchKernelGGL(k_jku_fwd<C>, dim3(grid), dim3(512), lds, st, xs, n, npad, w, out, HS, CS);
  CGC_RETURN_IF_LAUNCH_FAILED();
  return 0;
}

// flat != nullptr: gradients in parameter order (cgc_jk_unpack_param_grads' layout); else G [2][4H+1][C+2H+1]
template <int C>
static int launch_bwd_us(const float* xs, const float* dout, int n, int npad, const JkWeights& w, const float* HS, const float* CS,
                         float* dxs, float* G, float* ws, hipStream_t st, float* flat) {
  constexpr size_t lds = sizeof(float) * JkU<C>::BWD_TOTAL;
  static bool attr_set[CGC_MAX_DEVICES] = {};
  cgc_allow_lds(reinterpret_cast<const void*>(&k_jku_bwd<C>), (int)lds, attr_set);
  int grid = ceil_div(n, 32);
  if (grid > 256) grid = 256;                 // one workgroup per CU (LDS), persistent over the tiles
  hipLaunchKernelGGL(k_jku_bwd<C>, dim3(grid), dim3(512), lds, st, xs, dout, n, npad, w, HS, CS, dxs, ws);
  CGC_RETURN_IF_LAUNCH_FAILED();
  constexpr int per = JkU<C>::PG_FLOATS, H = JkU<C>::H;
  const int P = grid, P2 = ceil_div(P, 16);
  float* tmp = ws + (size_t)P * per;
  hipLaunchKernelGGL(k_jk_pg_fold, dim3(ceil_div(per, 256), P2, 1), dim3(256), 0, st, ws, P, P2, per, tmp);
  if (flat != nullptr) {
    constexpr int total = 2 * (4 * H * C + 4 * H * H + 8 * H) + 2 * H + 1;
    hipLaunchKernelGGL(k_jku_finish_flat<C>, dim3(ceil_div(total, 128)), dim3(128), 0, st, tmp, P2, flat);
  } else {
    constexpr int total = 2 * (4 * H + 1) * (C + 2 * H + 1);
    hipLaunchKernelGGL(k_jku_finish_G<C>, dim3(ceil_div(total, 128)), dim3(128), 0, st, tmp, P2, G);
  }
  CGC_RETURN_IF_LAUNCH_FAILED();
  return 0;
}

template <int C>
static int launch_fwd(const float* xs, int n, int npad, const JkWeights& w, float* out, float* HS, float* CS, hipStream_t st) {
  if (g_unit_split) return launch_fwd_us<C>(xs, n, npad, w, out, HS, CS, st);
  static bool attr_set[CGC_MAX_DEVICES] = {};
  cgc_allow_lds(reinterpret_cast<const void*>(&k_jk_fwd_mfma<C>), (int)JkM<C>::lds_bytes, attr_set);
  const int ntiles = ceil_div(n, 32);
  int grid = ceil_div(ntiles, 2);
  if (grid > 512) grid = 512;                 // 2 workgroups per CU (LDS), persistent over the 64-node tile pairs
  hipLaunchKernelGGL(k_jk_fwd_mfma<C>, dim3(grid), dim3(256), JkM<C>::lds_bytes, st, xs, n, npad, w, out, HS, CS);
  CGC_RETURN_IF_LAUNCH_FAILED();
  return 0;
}

int jk_mfma_fwd(const float* xs, int n, int npad, int C, const JkWeights& w, float* out, float* HS, float* CS, hipStream_t st) {
  if ((reinterpret_cast<uintptr_t>(xs) & 15u) || (reinterpret_cast<uintptr_t>(out) & 15u)) return CGC_EINVAL;
  switch (C) {
    case 4: return launch_fwd<4>(xs, n, npad, w, out, HS, CS, st);
    case 8: return launch_fwd<8>(xs, n, npad, w, out, HS, CS, st);
    case 12: return launch_fwd<12>(xs, n, npad, w, out, HS, CS, st);
    case 16: return launch_fwd<16>(xs, n, npad, w, out, HS, CS, st);
    case 20: return launch_fwd<20>(xs, n, npad, w, out, HS, CS, st);
    default: return CGC_EINVAL;
  }
}

template <int C, bool PG>
static int launch_bwd(const float* xs, const float* dout, int n, int npad, const JkWeights& w, const float* HS, const float* CS,
                      float* dxs, float* DGT, float* INT, float* G, float* ws, hipStream_t st, float* flat = nullptr) {
  if (PG && g_unit_split) return launch_bwd_us<C>(xs, dout, n, npad, w, HS, CS, dxs, G, ws, st, flat);
  size_t lds = sizeof(float) * (JkM<C>::TOTAL + JKB_TILES * 192) + sizeof(float4) * JKB_TILES * 2 * 3 * JkM<C>::XG * 64;
  if (PG) lds += sizeof(float) * JKB_TILES * 2 * 3 * 32 * 36;
  static bool attr_set[CGC_MAX_DEVICES] = {};
  cgc_allow_lds(reinterpret_cast<const void*>(&k_jk_bwd_mfma<C, PG>), (int)lds, attr_set);
  int grid = ceil_div(PG ? ceil_div(n, 32) : npad / 32, JKB_TILES);
  if (grid > 256) grid = 256;                 // one workgroup per CU
  hipLaunchKernelGGL((k_jk_bwd_mfma<C, PG>), dim3(grid), dim3(JKB_THREADS), lds, st, xs, dout, n, npad, w, HS, CS, dxs, DGT, INT, ws);
  CGC_RETURN_IF_LAUNCH_FAILED();
  if (PG) {
    constexpr int per = JkM<C>::PG_FLOATS, NG = 4 * JkM<C>::H + 1, NI = C + 2 * JkM<C>::H + 1;
    const int P = grid * JKB_TILES, P2 = ceil_div(P, 16);
    float* tmp = ws + (size_t)2 * P * per;
    hipLaunchKernelGGL(k_jk_pg_fold, dim3(ceil_div(per, 256), P2, 2), dim3(256), 0, st, ws, P, P2, per, tmp);
    if (flat != nullptr) {
      constexpr int H = JkM<C>::H, total = 2 * (4 * H * C + 4 * H * H + 8 * H) + 2 * H + 1;
      hipLaunchKernelGGL(k_jk_pg_finish_flat<C>, dim3(ceil_div(total, 256)), dim3(256), 0, st, tmp, P2, flat);
    } else {
      (void)hipMemsetAsync(G, 0, sizeof(float) * 2 * NG * NI, st);
      hipLaunchKernelGGL(k_jk_pg_finish<C>, dim3(ceil_div(per, 256), 2), dim3(256), 0, st, tmp, P2, G);
    }
    CGC_RETURN_IF_LAUNCH_FAILED();
  }
  return 0;
}

int jk_mfma_bwd(const float* xs, const float* dout, int n, int npad, int C, const JkWeights& w, const float* HS, const float* CS,
                float* dxs, float* DGT, float* INT, hipStream_t st) {
  if ((reinterpret_cast<uintptr_t>(xs) & 15u) || (reinterpret_cast<uintptr_t>(dout) & 15u) || (reinterpret_cast<uintptr_t>(dxs) & 15u) ||
      npad % (32 * JKB_TILES) != 0)
    return CGC_EINVAL;
  switch (C) {
    case 4: return launch_bwd<4, false>(xs, dout, n, npad, w, HS, CS, dxs, DGT, INT, nullptr, nullptr, st);
    case 8: return launch_bwd<8, false>(xs, dout, n, npad, w, HS, CS, dxs, DGT, INT, nullptr, nullptr, st);
    case 12: return launch_bwd<12, false>(xs, dout, n, npad, w, HS, CS, dxs, DGT, INT, nullptr, nullptr, st);
    case 16: return launch_bwd<16, false>(xs, dout, n, npad, w, HS, CS, dxs, DGT, INT, nullptr, nullptr, st);
    case 20: return launch_bwd<20, false>(xs, dout, n, npad, w, HS, CS, dxs, DGT, INT, nullptr, nullptr, st);
    default: return CGC_EINVAL;
  }
}

// workspace floats of the fused-parameter-gradient backward: partials of <= 512 waves per direction + their first fold
int64_t jk_mfma_bwd_ws_floats(int C) {
  const int64_t per = (8 * 16 + 16 + 1) * 64;
  (void)C;
  return 2 * 512 * per + 2 * 32 * per;
}

int jk_mfma_bwd_flat(const float* xs, const float* dout, int n, int npad, int C, const JkWeights& w, const float* HS, const float* CS,
                     float* dxs, float* flat, float* ws, hipStream_t st) {
  if ((reinterpret_cast<uintptr_t>(xs) & 15u) || (reinterpret_cast<uintptr_t>(dout) & 15u) || (reinterpret_cast<uintptr_t>(dxs) & 15u))
    return CGC_EINVAL;
  switch (C) {
    case 4: return launch_bwd<4, true>(xs, dout, n, npad, w, HS, CS, dxs, nullptr, nullptr, nullptr, ws, st, flat);
    case 8: return launch_bwd<8, true>(xs, dout, n, npad, w, HS, CS, dxs, nullptr, nullptr, nullptr, ws, st, flat);
    case 12: return launch_bwd<12, true>(xs, dout, n, npad, w, HS, CS, dxs, nullptr, nullptr, nullptr, ws, st, flat);
    case 16: return launch_bwd<16, true>(xs, dout, n, npad, w, HS, CS, dxs, nullptr, nullptr, nullptr, ws, st, flat);
    case 20: return launch_bwd<20, true>(xs, dout, n, npad, w, HS, CS, dxs, nullptr, nullptr, nullptr, ws, st, flat);
    default: return CGC_EINVAL;
  }
}

int jk_mfma_bwd_params(const float* xs, const float* dout, int n, int npad, int C, const JkWeights& w, const float* HS,
                       const float* CS, float* dxs, float* G, float* ws, hipStream_t st) {
  if ((reinterpret_cast<uintptr_t>(xs) & 15u) || (reinterpret_cast<uintptr_t>(dout) & 15u) || (reinterpret_cast<uintptr_t>(dxs) & 15u))
    return CGC_EINVAL;
  switch (C) {
    case 4: return launch_bwd<4, true>(xs, dout, n, npad, w, HS, CS, dxs, nullptr, nullptr, G, ws, st);
    case 8: return launch_bwd<8, true>(xs, dout, n, npad, w, HS, CS, dxs, nullptr, nullptr, G, ws, st);
    case 12: return launch_bwd<12, true>(xs, dout, n, npad, w, HS, CS, dxs, nullptr, nullptr, G, ws, st);
    case 16: return launch_bwd<16, true>(xs, dout, n, npad, w, HS, CS, dxs, nullptr, nullptr, G, ws, st);
    case 20: return launch_bwd<20, true>(xs, dout, n, npad, w, HS, CS, dxs, nullptr, nullptr, G, ws, st);
    default: return CGC_EINVAL;
  }
}
