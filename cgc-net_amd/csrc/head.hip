// Classification head + loss of SoftPoolingGcnEncoder (model/network.py:220-234, 286-289): cat(readouts) -> Linear -> activation ->
// Dropout -> Linear -> mean cross-entropy, and its backward, as ONE small kernel each.  The tensors are tiny ([B, 60..180] -> 50 ->
// 3 with B graphs): as separate operators the head was ~25 launches of a step (two GEMMs, bias / activation / dropout / softmax /
// nll kernels and their gradients); here one workgroup walks the phases with barriers in between.  Sums run in a fixed order.
#include "common.hpp"

struct HeadArgs {
  const float* x[3];       // the readouts, [B, D] each (nseg of them): the concatenation is never formed
  int nseg, B, D, H1, L, act;
  const float *W1, *b1, *W2, *b2;   // nn.Linear layouts: W1 [H1, nseg*D], W2 [L, H1]
  const long long* y;      // labels [B] (int64)
  float drop_p;
  unsigned long long seed;
};

__device__ __forceinline__ float head_x(const HeadArgs& a, int b, int k) { return a.x[k / a.D][(size_t)b * a.D + (k % a.D)]; }

// keep-scale of element i of the dropout mask: 0 or 1/(1-p); counter-based (splitmix64 of seed + i), independent of the launch shape
__device__ __forceinline__ float head_keep(unsigned long long seed, unsigned i, float p) {
  if (p <= 0.f) return 1.f;
  unsigned long long z = seed + 0x9E3779B97F4A7C15ull * (unsigned long long)(i + 1);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z ^= z >> 31;
  const float u = (float)(z >> 40) * (1.0f / 16777216.0f);      // [0, 1)
  return u < p ? 0.f : 1.f / (1.f - p);
}

// F.cross_entropy's labels (model/network.py:289 passes no ignore_index, so -100 is the ignored value): a label of -100 contributes
// neither loss nor gradient and the mean runs over the remaining samples.  torch raises a device-side assertion for any OTHER label
// outside [0, L); a kernel cannot raise, so such a label POISONS the step instead: head_valid_count returns NaN, which makes the loss
// and every gradient NaN (and it is never used as an index).  Silently dropping it would shrink the mean's denominator unnoticed.
__device__ __forceinline__ bool head_label_ok(const HeadArgs& a, int b) {
  const long long y = a.y[b];
  return y >= 0 && y < (long long)a.L;
}
__device__ __forceinline__ float head_valid_count(const HeadArgs& a) {
  int valid = 0;
  bool bad = false;
  for (int b = 0; b < a.B; ++b) {
    const bool ok = head_label_ok(a, b);
    valid += ok ? 1 : 0;
    bad = bad || (!ok && a.y[b] != -100);
  }
  return bad ? __builtin_nanf("") : (float)valid;
}

// z [B,H1] pre-activation, keep [B,H1] dropout scale, h [B,H1] = act(z) * keep, logits [B,L], loss [1]
__global__ __launch_bounds__(256) void k_head_fwd_simple(HeadArgs a, float* __restrict__ z, float* __restrict__ keep, float* h,
                                                  float* logits, float* __restrict__ loss, float* lse) {
  const int K = a.nseg * a.D;
  for (int i = threadIdx.x; i < a.B * a.H1; i += blockDim.x) {
    const int b = i / a.H1, j = i - b * a.H1;
    float s = a.b1 != nullptr ? a.b1[j] : 0.f;
    for (int k = 0; k < K; ++k) s = fmaf(a.W1[(size_t)j * K + k], head_x(a, b, k), s);
    const float kp = head_keep(a.seed, (unsigned)i, a.drop_p);
    z[i] = s;
    keep[i] = kp;
    h[i] = act_fwd(s, a.act) * kp;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < a.B * a.L; i += blockDim.x) {
    const int b = i / a.L, l = i - b * a.L;
    float s = a.b2 != nullptr ? a.b2[l] : 0.f;
    for (int j = 0; j < a.H1; ++j) s = fmaf(a.W2[(size_t)l * a.H1 + j], h[(size_t)b * a.H1 + j], s);
    logits[i] = s;
  }
  __syncthreads();
  if (a.y == nullptr) return;
  for (int b = threadIdx.x; b < a.B; b += blockDim.x) {          // per-sample -log softmax(logits)[y]
    float m = -INFINITY;
    for (int l = 0; l < a.L; ++l) m = fmaxf(m, logits[(size_t)b * a.L + l]);
    float s = 0.f;
    for (int l = 0; l < a.L; ++l) s += expf(logits[(size_t)b * a.L + l] - m);
    lse[b] = head_label_ok(a, b) ? m + logf(s) - logits[(size_t)b * a.L + (int)a.y[b]] : 0.f;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    float s = 0.f;
    for (int b = 0; b < a.B; ++b) s += lse[b];
    loss[0] = s / head_valid_count(a);
  }
}

// dloss: device scalar (gradient of the mean loss; NULL = 0); dlogits_ext [B,L] or NULL: a gradient that reaches the logits directly.
// Out: dW1 [H1, K], db1 [H1], dW2 [L, H1], db2 [L], dx[s] [B, D]; dl [B,L] and dz [B,H1] are scratch.
__global__ __launch_bounds__(256) void k_head_bwd_simple(HeadArgs a, const float* __restrict__ z, const float* __restrict__ keep,
                                                  const float* __restrict__ h, const float* __restrict__ logits, const float* __restrict__ dloss,
                                                  const float* __restrict__ dlogits_ext, float* dl, float* dz, float* __restrict__ dW1,
                                                  float* __restrict__ db1, float* __restrict__ dW2, float* __restrict__ db2, float* dx0,
                                                  float* dx1, float* dx2) {
  const int K = a.nseg * a.D;
  const float g = (dloss != nullptr && a.y != nullptr) ? dloss[0] / head_valid_count(a) : 0.f;
  for (int b = threadIdx.x; b < a.B; b += blockDim.x) {
    float m = -INFINITY;
    for (int l = 0; l < a.L; ++l) m = fmaxf(m, logits[(size_t)b * a.L + l]);
    float s = 0.f;
    for (int l = 0; l < a.L; ++l) s += expf(logits[(size_t)b * a.L + l] - m);
    const float inv = 1.f / s;
    for (int l = 0; l < a.L; ++l) {
      float v = 0.f;
      if (a.y != nullptr && head_label_ok(a, b)) v = g * (expf(logits[(size_t)b * a.L + l] - m) * inv - ((int)a.y[b] == l ? 1.f : 0.f));
      if (dlogits_ext != nullptr) v += dlogits_ext[(size_t)b * a.L + l];
      dl[(size_t)b * a.L + l] = v;
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < a.L * a.H1; i += blockDim.x) {                // dW2 = dl^T h
    const int l = i / a.H1, j = i - l * a.H1;
    float s = 0.f;
    for (int b = 0; b < a.B; ++b) s = fmaf(dl[(size_t)b * a.L + l], h[(size_t)b * a.H1 + j], s);
    dW2[i] = s;
  }
  for (int l = threadIdx.x; l < a.L; l += blockDim.x) {
    float s = 0.f;
    for (int b = 0; b < a.B; ++b) s += dl[(size_t)b * a.L + l];
    if (db2 != nullptr) db2[l] = s;
  }
  for (int i = threadIdx.x; i < a.B * a.H1; i += blockDim.x) {                // dz = (dl W2) * keep * act'(z)
    const int b = i / a.H1, j = i - b * a.H1;
    float s = 0.f;
    for (int l = 0; l < a.L; ++l) s = fmaf(dl[(size_t)b * a.L + l], a.W2[(size_t)l * a.H1 + j], s);
    dz[i] = s * keep[i] * act_bwd(z[i], a.act);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < a.H1 * K; i += blockDim.x) {                  // dW1 = dz^T x
    const int j = i / K, k = i - j * K;
    float s = 0.f;
    for (int b = 0; b < a.B; ++b) s = fmaf(dz[(size_t)b * a.H1 + j], head_x(a, b, k), s);
    dW1[i] = s;
  }
  for (int j = threadIdx.x; j < a.H1; j += blockDim.x) {
    float s = 0.f;
    for (int b = 0; b < a.B; ++b) s += dz[(size_t)b * a.H1 + j];
    if (db1 != nullptr) db1[j] = s;
  }
  for (int i = threadIdx.x; i < a.B * K; i += blockDim.x) {                   // dx = dz W1
    const int b = i / K, k = i - b * K;
    float s = 0.f;
    for (int j = 0; j < a.H1; ++j) s = fmaf(dz[(size_t)b * a.H1 + j], a.W1[(size_t)j * K + k], s);
    float* dst = k / a.D == 0 ? dx0 : k / a.D == 1 ? dx1 : dx2;
    dst[(size_t)b * a.D + (k % a.D)] = s;
  }
}


// ---- the same two kernels with the operands staged in LDS (used whenever they fit: always for the network's own sizes).  The simple
// versions above walk W1 with a stride of K floats between neighbouring threads and re-read x from global memory for every output:
// 72 + 75 us at 32 graphs (B x H1 x K = 32 x 50 x 180), more than the per-operator head they replaced.  Here W1 is transposed into
// LDS once ([k][j]: neighbouring threads = neighbouring words), x and dz live in LDS, 1024 threads share the items.
#define HEAD_THREADS 1024
static inline size_t head_fwd_lds_floats(int B, int K, int H1) { return (size_t)K * (H1 + 1) + (size_t)B * K + (size_t)B * H1; }
static inline size_t head_bwd_lds_floats(int B, int K, int H1, int L) { return (size_t)B * K + (size_t)B * H1 + (size_t)B * L; }

__global__ __launch_bounds__(HEAD_THREADS) void k_head_fwd(HeadArgs a, float* __restrict__ z, float* __restrict__ keep, float* __restrict__ h,
                                                           float* __restrict__ logits, float* __restrict__ loss, float* lse) {
  extern __shared__ float sm[];
  const int K = a.nseg * a.D, H1 = a.H1, ldw = H1 + 1;
  float* Wt = sm;                          // [K][H1 + 1]
  float* xs = Wt + (size_t)K * ldw;        // [B][K]
  float* hs = xs + (size_t)a.B * K;        // [B][H1]
  for (int i = threadIdx.x; i < H1 * K; i += blockDim.x) {
    const int j = i / K, k = i - j * K;
    Wt[k * ldw + j] = a.W1[i];
  }
  for (int i = threadIdx.x; i < a.B * K; i += blockDim.x) xs[i] = head_x(a, i / K, i % K);
  __syncthreads();
  for (int i = threadIdx.x; i < a.B * H1; i += blockDim.x) {
    const int b = i / H1, j = i - b * H1;
    float s = a.b1 != nullptr ? a.b1[j] : 0.f;
    for (int k = 0; k < K; ++k) s = fmaf(Wt[k * ldw + j], xs[b * K + k], s);
    const float kp = head_keep(a.seed, (unsigned)i, a.drop_p);
    const float hv = act_fwd(s, a.act) * kp;
    z[i] = s;
    keep[i] = kp;
    h[i] = hv;
    hs[i] = hv;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < a.B * a.L; i += blockDim.x) {
    const int b = i / a.L, l = i - b * a.L;
    float s = a.b2 != nullptr ? a.b2[l] : 0.f;
    for (int j = 0; j < H1; ++j) s = fmaf(a.W2[(size_t)l * H1 + j], hs[b * H1 + j], s);
    logits[i] = s;
    xs[i] = s;                             // (x is dead: its LDS holds the logits for the loss; B * L <= B * K)
  }
  __syncthreads();
  if (a.y == nullptr) return;
  for (int b = threadIdx.x; b < a.B; b += blockDim.x) {
    float m = -INFINITY;
    for (int l = 0; l < a.L; ++l) m = fmaxf(m, xs[b * a.L + l]);
    float s = 0.f;
    for (int l = 0; l < a.L; ++l) s += expf(xs[b * a.L + l] - m);
    const float v = head_label_ok(a, b) ? m + logf(s) - xs[b * a.L + (int)a.y[b]] : 0.f;
    lse[b] = v;
    hs[b] = v;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    float s = 0.f;
    for (int b = 0; b < a.B; ++b) s += hs[b];
    loss[0] = s / head_valid_count(a);
  }
}

__global__ __launch_bounds__(HEAD_THREADS) void k_head_bwd(HeadArgs a, const float* __restrict__ z, const float* __restrict__ keep,
                                                           const float* __restrict__ h, const float* __restrict__ logits,
                                                           const float* __restrict__ dloss, const float* __restrict__ dlogits_ext,
                                                           float* __restrict__ dW1, float* __restrict__ db1, float* __restrict__ dW2,
                                                           float* __restrict__ db2, float* dx0, float* dx1, float* dx2) {
  extern __shared__ float sm[];
  const int K = a.nseg * a.D, H1 = a.H1;
  float* xs = sm;                          // [B][K]
  float* dzs = xs + (size_t)a.B * K;       // [B][H1]
  float* dls = dzs + (size_t)a.B * H1;     // [B][L]
  const float g = (dloss != nullptr && a.y != nullptr) ? dloss[0] / head_valid_count(a) : 0.f;
  for (int i = threadIdx.x; i < a.B * K; i += blockDim.x) xs[i] = head_x(a, i / K, i % K);
  for (int b = threadIdx.x; b < a.B; b += blockDim.x) {
    float m = -INFINITY;
    for (int l = 0; l < a.L; ++l) m = fmaxf(m, logits[(size_t)b * a.L + l]);
    float s = 0.f;
    for (int l = 0; l < a.L; ++l) s += expf(logits[(size_t)b * a.L + l] - m);
    const float inv = 1.f / s;
    for (int l = 0; l < a.L; ++l) {
      float v = 0.f;
      if (a.y != nullptr && head_label_ok(a, b)) v = g * (expf(logits[(size_t)b * a.L + l] - m) * inv - ((int)a.y[b] == l ? 1.f : 0.f));
      if (dlogits_ext != nullptr) v += dlogits_ext[(size_t)b * a.L + l];
      dls[b * a.L + l] = v;
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < a.L * H1; i += blockDim.x) {                  // dW2 = dl^T h
    const int l = i / H1, j = i - l * H1;
    float s = 0.f;
    for (int b = 0; b < a.B; ++b) s = fmaf(dls[b * a.L + l], h[(size_t)b * H1 + j], s);
    dW2[i] = s;
  }
  for (int l = threadIdx.x; l < a.L; l += blockDim.x) {
    float s = 0.f;
    for (int b = 0; b < a.B; ++b) s += dls[b * a.L + l];
    if (db2 != nullptr) db2[l] = s;
  }
  for (int i = threadIdx.x; i < a.B * H1; i += blockDim.x) {                  // dz = (dl W2) * keep * act'(z)
    const int b = i / H1, j = i - b * H1;
    float s = 0.f;
    for (int l = 0; l < a.L; ++l) s = fmaf(dls[b * a.L + l], a.W2[(size_t)l * H1 + j], s);
    dzs[i] = s * keep[i] * act_bwd(z[i], a.act);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < H1 * K; i += blockDim.x) {                    // dW1 = dz^T x  (k fastest: x rows are read side by side)
    const int j = i / K, k = i - j * K;
    float s = 0.f;
    for (int b = 0; b < a.B; ++b) s = fmaf(dzs[b * H1 + j], xs[b * K + k], s);
    dW1[i] = s;
  }
  for (int j = threadIdx.x; j < H1; j += blockDim.x) {
    float s = 0.f;
    for (int b = 0; b < a.B; ++b) s += dzs[b * H1 + j];
    if (db1 != nullptr) db1[j] = s;
  }
  for (int i = threadIdx.x; i < a.B * K; i += blockDim.x) {                   // dx = dz W1  (k fastest: coalesced rows of W1)
    const int b = i / K, k = i - b * K;
    float s = 0.f;
    for (int j = 0; j < H1; ++j) s = fmaf(dzs[b * H1 + j], a.W1[(size_t)j * K + k], s);
    float* dst = k / a.D == 0 ? dx0 : k / a.D == 1 ? dx1 : dx2;
    dst[(size_t)b * a.D + (k % a.D)] = s;
  }
}

static int head_args(HeadArgs& a, const float* const* x, int nseg, int B, int D, int H1, int L, int act, const float* W1, const float* b1,
                     const float* W2, const float* b2, const int64_t* y, float drop_p, uint64_t seed) {
  if (nseg < 1 || nseg > 3 || B < 1 || D < 1 || H1 < 1 || L < 1 || drop_p < 0.f || drop_p >= 1.f) return CGC_EINVAL;
  for (int s = 0; s < 3; ++s) a.x[s] = s < nseg ? x[s] : nullptr;
  a.nseg = nseg; a.B = B; a.D = D; a.H1 = H1; a.L = L; a.act = act;
  a.W1 = W1; a.b1 = b1; a.W2 = W2; a.b2 = b2;
  a.y = reinterpret_cast<const long long*>(y);
  a.drop_p = drop_p;
  a.seed = seed;
  return 0;
}

// ws: B*H1*3 + B floats (z | keep | h | per-sample losses), kept for the backward
extern "C" int cgc_head_fwd(const float* const* x, int nseg, int B, int D, int H1, int L, int act, const float* W1, const float* b1,
                            const float* W2, const float* b2, const int64_t* y, float drop_p, uint64_t seed, float* ws, float* logits,
                            float* loss, cgc_stream_t stream) {
  HeadArgs a;
  const int rc = head_args(a, x, nseg, B, D, H1, L, act, W1, b1, W2, b2, y, drop_p, seed);
  if (rc != 0) return rc;
  const size_t m = (size_t)B * H1;
  const size_t lds = sizeof(float) * head_fwd_lds_floats(B, nseg * D, H1);
  if (lds <= 150 * 1024 && L <= nseg * D) {
    static bool attr[CGC_MAX_DEVICES] = {};
    cgc_allow_lds(reinterpret_cast<const void*>(&k_head_fwd), 150 * 1024, attr);
    hipLaunchKernelGGL(k_head_fwd, dim3(1), dim3(HEAD_THREADS), lds, as_stream(stream), a, ws, ws + m, ws + 2 * m, logits, loss, ws + 3 * m);
  } else {
    hipLaunchKernelGGL(k_head_fwd_simple, dim3(1), dim3(256), 0, as_stream(stream), a, ws, ws + m, ws + 2 * m, logits, loss, ws + 3 * m);
  }
  CGC_RETURN_IF_LAUNCH_FAILED();
  return 0;
}

// scratch: B*L + B*H1 floats.  grads: dW1 [H1, nseg*D] | db1 [H1] | dW2 [L, H1] | db2 [L] back to back; dx: nseg pointers [B, D]
extern "C" int cgc_head_bwd(const float* const* x, int nseg, int B, int D, int H1, int L, int act, const float* W1, const float* W2,
                            const int64_t* y, const float* ws, const float* logits, const float* dloss, const float* dlogits_ext,
                            float* scratch, float* grads, float* const* dx, cgc_stream_t stream) {
  HeadArgs a;
  const int rc = head_args(a, x, nseg, B, D, H1, L, act, W1, nullptr, W2, nullptr, y, 0.f, 0);
  if (rc != 0) return rc;
  const size_t m = (size_t)B * H1, K = (size_t)nseg * D;
  float* dW1 = grads;
  float* db1 = dW1 + (size_t)H1 * K;
  float* dW2 = db1 + H1;
  float* db2 = dW2 + (size_t)L * H1;
  const size_t lds = sizeof(float) * head_bwd_lds_floats(B, nseg * D, H1, L);
  if (lds <= 150 * 1024) {
    static bool attr[CGC_MAX_DEVICES] = {};
    cgc_allow_lds(reinterpret_cast<const void*>(&k_head_bwd), 150 * 1024, attr);
    hipLaunchKernelGGL(k_head_bwd, dim3(1), dim3(HEAD_THREADS), lds, as_stream(stream), a, ws, ws + m, ws + 2 * m, logits, dloss, dlogits_ext,
                       dW1, db1, dW2, db2, dx[0], nseg > 1 ? dx[1] : nullptr, nseg > 2 ? dx[2] : nullptr);
  } else {
    hipLaunchKernelGGL(k_head_bwd_simple, dim3(1), dim3(256), 0, as_stream(stream), a, ws, ws + m, ws + 2 * m, logits, dloss, dlogits_ext, scratch,
                       scratch + (size_t)B * L, dW1, db1, dW2, db2, dx[0], nseg > 1 ? dx[1] : nullptr, nseg > 2 ? dx[2] : nullptr);
  }
  CGC_RETURN_IF_LAUNCH_FAILED();
  return 0;
}
