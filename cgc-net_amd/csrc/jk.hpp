// Shared declarations of the DenseJK kernels (jk.hip: thread-per-direction VALU kernels; jk_mfma.hip: matrix-core kernels).
#pragma once
#include "common.hpp"

__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }

template <int C>
struct JkDims {
  static constexpr int H = 3 * C / 2;
  static constexpr int KIN = C + H;
  static constexpr size_t lds_bytes = sizeof(float4) * (2 * H * KIN + 2 * H) + sizeof(float) * (2 * H + 4);
};

struct JkWeights {            // PyTorch nn.LSTM layout, gate order i,f,g,o; [0] forward direction, [1] reverse
  const float* w_ih[2];       // [4H, C]
  const float* w_hh[2];       // [4H, H]
  const float* b_ih[2];       // [4H]
  const float* b_hh[2];       // [4H]
  const float* w_att;         // [2H]
  const float* b_att;         // [1]
};


// matrix-core forward (jk_mfma.hip); returns CGC_EINVAL for unsupported C
int jk_mfma_fwd(const float* xs, int n, int npad, int C, const JkWeights& w, float* out, float* HS, float* CS, hipStream_t st);
int jk_mfma_bwd(const float* xs, const float* dout, int n, int npad, int C, const JkWeights& w, const float* HS, const float* CS,
                float* dxs, float* DGT, float* INT, hipStream_t st);
// backward with the parameter gradients accumulated in-kernel: G [2][4H+1][C+2H+1] (same layout as DGT . INT^T)
int64_t jk_mfma_bwd_ws_floats(int C);
int jk_mfma_bwd_params(const float* xs, const float* dout, int n, int npad, int C, const JkWeights& w, const float* HS,
                       const float* CS, float* dxs, float* G, float* ws, hipStream_t st);
// the same with the gradients written straight into the flat parameter-order buffer of cgc_jk_unpack_param_grads
int jk_mfma_bwd_flat(const float* xs, const float* dout, int n, int npad, int C, const JkWeights& w, const float* HS, const float* CS,
                     float* dxs, float* flat, float* ws, hipStream_t st);
