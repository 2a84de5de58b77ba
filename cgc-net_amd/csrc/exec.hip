// Step sequencer: one level of SoftPoolingGcnEncoder.forward (model/network.py:258-285) and its backward as ONE host call each.
//
// Round 2 drove every kernel of a step from a Python autograd graph: 328 launches, ~12 us of interpreter / autograd / allocator
// time each -- 4.1 ms of host time per step, which is what bounded the step at 4 graphs per GPU (the strong-scaling shard of the
// reference's DataParallel batch, train.py:276-287) and at small cluster counts.  Here the schedule of a level -- which kernels,
// in which order, on which buffers -- is plain C++ over the library's own entry points (include/cgc_hip.h): a launch costs the
// ~3 us of hipLaunchKernel, activations live in two caller-provided arenas (bump allocation, no allocator calls), and the
// concatenations / gradient sums that autograd expressed with torch kernels are the two small kernels at the top of this file.
// The arithmetic is the per-operator path's (cgc-net_amd/ops.py), kernel for kernel: tests compare the two bit for bit.
//
// Reference expressions: GNN_Module.forward (model/network.py:109-125), _re_norm_adj (:183-191), DenseJK (:36-52), max readout
// (:264), _diff_pool (:194-208).
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "common.hpp"
#include "groups.hpp"

// ------------------------------------------------------------------------------------------------ layout kernels
struct CatArgs {
  const float* src[4];
  int ld[4], width[4], off[4];
};
__global__ __launch_bounds__(256) void k_cat_cols(float* __restrict__ dst, int ldd, long long rows, int total_w, CatArgs a, int nsrc,
                                                  int accumulate) {
  const long long total = rows * total_w;
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const long long r = i / total_w;
    const int c = (int)(i - r * total_w);
    int s = 0;
#pragma unroll
    for (int k = 1; k < 4; ++k)
      if (k < nsrc && c >= a.off[k]) s = k;
    const float v = a.src[s][r * a.ld[s] + (c - a.off[s])];
    float* p = dst + r * ldd + c;
    *p = accumulate ? *p + v : v;
  }
}

extern "C" int cgc_cat_cols(float* dst, int ldd, int rows, int nsrc, const float* const* srcs, const int* lds, const int* widths,
                            int accumulate, cgc_stream_t stream) {
  if (nsrc < 1 || nsrc > 4) return CGC_EINVAL;
  CatArgs a;
  int off = 0;
  for (int k = 0; k < 4; ++k) {
    a.src[k] = k < nsrc ? srcs[k] : nullptr;
    a.ld[k] = k < nsrc ? lds[k] : 0;
    a.width[k] = k < nsrc ? widths[k] : 0;
    a.off[k] = off;
    off += a.width[k];
  }
  if (rows <= 0 || off <= 0) return 0;
  const long long total = (long long)rows * off;
  const long long blocks = (total + 255) / 256;
  hipLaunchKernelGGL(k_cat_cols, dim3((unsigned)(blocks < 4096 ? blocks : 4096)), dim3(256), 0, as_stream(stream), dst, ldd,
                     (long long)rows, off, a, nsrc, accumulate);
  CGC_RETURN_IF_LAUNCH_FAILED();
  return 0;
}

__global__ __launch_bounds__(256) void k_transpose(const float* __restrict__ src, int lds, int rows, int cols, float* __restrict__ dst,
                                                   int ldd) {
  __shared__ float t[32][33];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;      // 32 x 8
  const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
  for (int j = ty; j < 32; j += 8) {
    const int r = r0 + j, c = c0 + tx;
    if (r < rows && c < cols) t[j][tx] = src[(size_t)r * lds + c];
  }
  __syncthreads();
  for (int j = ty; j < 32; j += 8) {
    const int c = c0 + j, r = r0 + tx;
    if (r < rows && c < cols) dst[(size_t)c * ldd + r] = t[tx][j];
  }
}

extern "C" int cgc_transpose(const float* src, int lds, int rows, int cols, float* dst, int ldd, cgc_stream_t stream) {
  if (rows <= 0 || cols <= 0) return 0;
  hipLaunchKernelGGL(k_transpose, dim3((unsigned)ceil_div(cols, 32), (unsigned)ceil_div(rows, 32)), dim3(256), 0, as_stream(stream), src,
                     lds, rows, cols, dst, ldd);
  CGC_RETURN_IF_LAUNCH_FAILED();
  return 0;
}

int reduce_batch_sum_rows(const float* ws, float* out, int parts, int rows, int width, int ldo, float beta, hipStream_t stream);   // gemm.hip

// ------------------------------------------------------------------------------------------------ host-side plumbing
namespace {

struct Arena {                    // bump allocation inside a caller-provided buffer; base == nullptr: size computation only
  float* base;
  size_t off = 0, high = 0;       // in floats
  explicit Arena(float* b) : base(b) {}
  float* f(size_t n) {
    off = (off + 63) & ~(size_t)63;                    // 256-byte alignment
    float* p = base + off;
    off += n;
    if (off > high) high = off;
    return p;
  }
  int* i(size_t n) { return reinterpret_cast<int*>(f(n)); }
  size_t mark() const { return off; }
  void release(size_t m) { off = m; }
};

struct Ctx {
  cgc_stream_t s;
  bool dry;                       // size computation: no launches
  Arena* scratch;
  float* gws;                     // slab workspace of the GEMM's tail split
  int64_t gws_floats;
  int gemm_mode = CGC_GEMM_EXACT; // cgc_level_desc.flags bit 1: CGC_GEMM_SPLIT_BF16 (gemm_split.hip), bit 2: CGC_GEMM_SPLIT_F16 (gemm_half.hip) for the products that qualify
};

static int fail_at(int rc, const char* what, int line) {      // CGC_EXEC_DEBUG=1: say which call of the schedule failed
  static const bool verbose = getenv("CGC_EXEC_DEBUG") != nullptr;
  if (verbose) fprintf(stderr, "cgc exec.hip:%d: rc %d from %s\n", line, rc, what);
  return rc;
}
#define CALL(expr)                                              \
  do {                                                          \
    if (!c.dry) {                                               \
      const int rc__ = (expr);                                  \
      if (rc__ != 0) return fail_at(rc__, #expr, __LINE__);     \
    }                                                           \
  } while (0)
#define TRY(expr)                                               \
  do {                                                          \
    const int rc__ = (expr);                                    \
    if (rc__ != 0) return fail_at(rc__, #expr, __LINE__);       \
  } while (0)

inline int up(int v, int m) { return (v + m - 1) / m * m; }
// row stride of a [rows, F] activation (ops._wide): wide rows on whole 128-byte lines, other rows 16-byte aligned
inline int wide_ld(int F) { return (F >= 256 && F % 32 != 0) ? up(F, 32) : (F > 32 && F % 4 != 0) ? up(F, 4) : F; }
inline int pad4_ld(int c) { return (c > 32 && c % 4 != 0) ? up(c, 4) : c; }

int gemm(const Ctx& c, int tA, int tB, int M, int N, int K, const float* A, int lda, const float* B, int ldb, float beta, float* C, int ldc,
         const float* bias = nullptr, int batch = 1, int64_t sA = 0, int64_t sB = 0, int64_t sC = 0, const int* gptr = nullptr,
         int ragged = 0, int max_ragged = 0) {
  if (c.dry) return 0;
  return cgc_gemm_f32_ws(tA, tB, M, N, K, 1.f, A, lda, B, ldb, beta, C, ldc, bias, batch, sA, sB, sC, gptr, ragged, max_ragged, c.gws,
                         c.gws_floats, c.gemm_mode, c.s);
}

int gemm_x1(const Ctx& c, int tA, int tB, int M, int N, int K, const float* A, int lda, const float* B, int ldb, float beta, float* C, int ldc,
            const float* bias, int batch, int64_t sA, int64_t sB, int64_t sC, const int* gptr, int ragged, int max_ragged, const float* xA,
            int xlda, int64_t xsA, const float* xB, int xldb, int64_t xsB, int xK) {
  if (c.dry) return 0;
  const float* pa[1] = {xA};
  const float* pb[1] = {xB};
  const int la[1] = {xlda}, lb[1] = {xldb}, kk[1] = {xK};
  const int64_t s1[1] = {xsA}, s2[1] = {xsB};
  return cgc_gemm_f32_cat_ws(tA, tB, M, N, K, 1.f, A, lda, B, ldb, beta, C, ldc, bias, batch, sA, sB, sC, gptr, ragged, max_ragged, 1, pa, la,
                             s1, pb, lb, s2, kk, c.gws, c.gws_floats, c.gemm_mode, c.s);
}

// how many row slices the tall-skinny "weight gradient" contraction out[Fa,Fb] = A[n,Fa]^T B[n,Fb] is cut into (ops._split_parts):
// fill whole rounds of resident workgroups, >= 512 rows per slice
int split_parts(int Fa, int Fb, int n) {
  const int tm = Fa <= 32 ? 32 : Fa <= 64 ? 64 : 128, tn = Fb <= 32 ? 32 : Fb <= 64 ? 64 : 128;
  const int tiles = ceil_div(Fa, tm) * ceil_div(Fb, tn);
  int best = 1;
  double best_score = -1.0;
  int hi = n / 512;
  if (hi > 128) hi = 128;
  if (hi < 1) hi = 1;
  for (int parts = 1; parts <= hi; ++parts) {
    const int blocks = tiles * parts, rounds = ceil_div(blocks, 512);
    const double fill = (double)blocks / ((double)rounds * 512.0);
    const double score = fill - 0.02 * rounds - (blocks < 256 ? 0.5 : 0.0);
    if (score > best_score + 1e-9) {
      best = parts;
      best_score = score;
    }
  }
  return best;
}

// out[Fa,Fb] = A[:n,:Fa]^T B[:n,:Fb], rows split over workgroups and combined deterministically (ops.gemm_tn_rows); ldo = row stride
// of out (0: Fb) -- a piece of a wider weight gradient is written in place
int gemm_tn_rows(const Ctx& c, const float* A, int lda, int Fa, const float* B, int ldb, int Fb, int n, float* out, int ldo = 0) {
  if (ldo <= 0) ldo = Fb;
  int parts = split_parts(Fa, Fb, n);
  if (parts == 1) return gemm(c, 1, 0, Fa, Fb, n, A, lda, B, ldb, 0.f, out, ldo);
  const int chunk = up(ceil_div(n, parts), 32);
  parts = ceil_div(n, chunk);
  const size_t m = c.scratch->mark();
  float* ws = c.scratch->f((size_t)parts * Fa * Fb);
  TRY(gemm(c, 1, 0, Fa, Fb, n, A, lda, B, ldb, 0.f, ws, Fb, nullptr, parts, 0, 0, (int64_t)Fa * Fb, nullptr, 3, chunk));
  CALL(reduce_batch_sum_rows(ws, out, parts, Fa, Fb, ldo, 0.f, as_stream(c.s)));
  c.scratch->release(m);
  return 0;
}

struct T3 {              // [b, r, c] with contiguous (possibly padded) rows, batches back to back
  float* p;
  int b, r, c, ld;
  int64_t bs() const { return (int64_t)r * ld; }
};

// C = op(A) op(B) (+ beta C) for strided batches (ops._bgemm): small outputs reduced over a long axis are cut into slices
int bgemm(const Ctx& c, const T3& A, const T3& B, const T3& C, int tA, int tB, float beta = 0.f) {
  const int batch = C.b, M = C.r, N = C.c, Kd = tA ? A.r : A.c;
  if (tA && !tB && M <= 128 && N <= 128 && Kd >= 512 && batch * 2 < 256 && C.ld == N) {
    int parts = Kd / 128;
    const int want = ceil_div(384, batch * (N > 64 ? 2 : 1));
    if (want < parts) parts = want;
    if (parts < 2) parts = 2;
    const int step = up(ceil_div(Kd, parts), 32);
    parts = ceil_div(Kd, step);
    const size_t m = c.scratch->mark();
    float* ws = c.scratch->f((size_t)batch * parts * M * N);
    TRY(gemm(c, 1, 0, M, N, Kd, A.p, A.ld, B.p, B.ld, 0.f, ws, N, nullptr, batch * parts, 0, 0, (int64_t)M * N, nullptr, 3, step));
    CALL(cgc_reduce_batched(ws, C.p, batch, parts, M * N, beta, c.s));
    c.scratch->release(m);
    return 0;
  }
  return gemm(c, tA, tB, M, N, Kd, A.p, A.ld, B.p, B.ld, beta, C.p, C.ld, nullptr, batch, A.bs(), B.bs(), C.bs());
}

int cat2(const Ctx& c, float* dst, int ldd, int rows, const float* a, int lda, int wa, const float* b, int ldb, int wb, int accumulate = 0) {
  const float* s[2] = {a, b};
  const int l[2] = {lda, ldb}, w[2] = {wa, wb};
  CALL(cgc_cat_cols(dst, ldd, rows, 2, s, l, w, accumulate, c.s));
  return 0;
}
int cat3(const Ctx& c, float* dst, int ldd, int rows, const float* a, int lda, int wa, const float* b, int ldb, int wb, const float* d, int ldd3,
         int wd) {
  const float* s[3] = {a, b, d};
  const int l[3] = {lda, ldb, ldd3}, w[3] = {wa, wb, wd};
  CALL(cgc_cat_cols(dst, ldd, rows, 3, s, l, w, 0, c.s));
  return 0;
}
int add1(const Ctx& c, float* dst, int ldd, int rows, const float* a, int lda, int wa) {      // dst[:, :wa] += a
  const float* s[1] = {a};
  const int l[1] = {lda}, w[1] = {wa};
  CALL(cgc_cat_cols(dst, ldd, rows, 1, s, l, w, 1, c.s));
  return 0;
}

// ------------------------------------------------------------------------------------------------ one SAGE layer
struct LayerP {                   // parameters of layer k of a block
  const float *W, *b, *gamma, *beta;
  float *rm, *rv;
  int64_t* nbt;
  float eps, mom;
};
struct LayerS {                   // what the forward keeps for the backward
  float *hn, *rinv, *mean, *istd;
};

inline LayerP layer_params(const cgc_level_desc& d, const cgc_block_params* p, int k, int slot) {
  LayerP r;
  r.W = p->W[k];
  r.b = d.has_bias ? p->b[k] : nullptr;
  r.gamma = d.has_bn ? p->gamma[k] : nullptr;
  r.beta = d.has_bn ? p->beta[k] : nullptr;
  r.rm = d.has_bn ? p->running_mean[k] : nullptr;
  r.rv = d.has_bn ? p->running_var[k] : nullptr;
  r.nbt = d.has_bn ? p->num_batches_tracked[k] : nullptr;
  r.eps = d.bn_eps[slot];
  r.mom = d.bn_momentum[slot];
  return r;
}

inline size_t stats_ws_floats(int n, int F) { return (size_t)cgc_stats_ws_floats(n, F); }      // slots of doubles (rowops.hip)

// y = BN(act(l2norm(agg W + b)))  (ops._SageProject.forward)
int layer_fwd(const Ctx& c, const cgc_level_desc& d, const LayerP& p, const LayerS& s, const float* agg, int lda, int n, int fin, int F,
              float* y, int ldy, float* y2 = nullptr, int ldy2 = 0) {
  const size_t m = c.scratch->mark();
  const int batch_stats = d.has_bn && !d.eval;       // inference: the running statistics normalise (below), nothing is accumulated
  float* ws = batch_stats ? c.scratch->f(stats_ws_floats(n, F)) : nullptr;
  int fused = 0;
  if ((F >= 256 || F <= 32) && fin <= 32 && !c.dry) {
    int rc;
    if (F <= 32)
      rc = cgc_sage_narrow_fwd(agg, lda, p.W, p.b, n, fin, F, 1, d.act, s.hn, s.rinv, batch_stats, ws, d.count, p.eps, p.mom, p.rm, p.rv, p.nbt,
                               s.mean, s.istd, c.s);
    else
      rc = cgc_sage_wide_fwd(agg, lda, p.W, p.b, n, fin, F, 1, d.act, s.hn, F, s.rinv, batch_stats, ws, d.count, p.eps, p.mom, p.rm, p.rv, p.nbt,
                             s.mean, s.istd, c.s);
    if (rc == 0) fused = 1;
    else if (rc != CGC_EINVAL) return rc;
  }
  if (!fused) {
    TRY(gemm(c, 0, 0, n, F, fin, agg, lda, p.W, F, 0.f, s.hn, F, p.b));
    if (batch_stats)
      CALL(cgc_l2norm_act_bn(s.hn, n, F, 1, d.act, s.hn, s.rinv, ws, d.count, p.eps, p.mom, p.rm, p.rv, p.nbt, s.mean, s.istd, c.s));
    else
      CALL(cgc_l2norm_act_stats(s.hn, n, F, 1, d.act, s.hn, s.rinv, nullptr, nullptr, c.s));
  }
  if (d.has_bn && d.eval) CALL(cgc_bn_running_stats(p.rm, p.rv, F, p.eps, s.mean, s.istd, c.s));
  CALL(cgc_bn_act_apply2(s.hn, n, F, d.act, d.has_bn ? s.mean : nullptr, s.istd, p.gamma, p.beta, y, ldy, y2, ldy2, c.s));
  c.scratch->release(m);
  return 0;
}

// backward of the same (ops._SageProject.backward): dagg (may be nullptr) [n, fin] with row stride ldd; dwdb = [dW (fin*F) | db (F)];
// sums = [d beta (F) | d gamma (F)]
int layer_bwd(const Ctx& c, const cgc_level_desc& d, const LayerP& p, const LayerS& s, const float* agg, int lda, int n, int fin, int F,
              const float* dy, int ldy, float* dagg, int ldd, float* dwdb, float* sums) {
  const size_t m = c.scratch->mark();
  const int mode = d.has_bn ? 2 : 0;
  const size_t slot_floats = (size_t)(cgc_stats_blocks(n, F) > 1 ? cgc_stats_blocks(n, F) : 1) * 2 * F;
  if (d.has_bn) {
    float* ws = c.scratch->f(slot_floats);
    CALL(cgc_bn_bwd_reduce(dy, ldy, s.hn, n, F, d.act, s.mean, s.istd, sums, ws, c.s));
  }
  float* db = d.has_bias ? dwdb + (size_t)fin * F : nullptr;
  if (F <= 32 && fin <= 32) {
    float* ws = c.scratch->f((size_t)cgc_sage_narrow_ws_floats(n, fin, F));
    CALL(cgc_sage_narrow_bwd_ld(dy, ldy, s.hn, s.rinv, n, F, d.act, 1, mode, s.mean, s.istd, p.gamma, d.has_bn ? sums : nullptr, d.count, agg,
                                lda, fin, p.W, dagg, ldd, dwdb, ws, c.s));
  } else {
    float* dh = c.scratch->f((size_t)n * F);
    float* ws = db ? c.scratch->f(slot_floats) : nullptr;
    CALL(cgc_bn_act_l2_bwd(dy, ldy, s.hn, s.rinv, n, F, d.act, 1, mode, s.mean, s.istd, p.gamma, d.has_bn ? sums : nullptr, d.count, dh, db, ws,
                           c.s));
    if (dagg) TRY(gemm(c, 0, 1, n, fin, F, dh, F, p.W, F, 0.f, dagg, ldd));
    TRY(gemm_tn_rows(c, agg, lda, fin, dh, F, F, n, dwdb));
  }
  c.scratch->release(m);
  return 0;
}

// The same two functions for the embedding and the assignment block's layer of one step TOGETHER, when both are narrow layers of one
// shape (hidden width -> hidden width from the same aggregation buffer): every kernel is launched once for both (groups.hpp).
// Identical arithmetic per layer; at 4 graphs per GPU these kernels are ~5 us of launch + drain each.
inline bool pairable(int fin_e, int F_e, int fin_p, int F_p) { return fin_e == fin_p && F_e == F_p && fin_e <= 32 && F_e <= 32; }

int layer_fwd_pair(const Ctx& c, const cgc_level_desc& d, const LayerP* p, const LayerS* s, const float* const* agg, int lda, int n, int fin, int F,
                   float* const* y, int ldy, float* const* y2, const int* ldy2) {
  const size_t m = c.scratch->mark();
  SnFwdPtrs g[2];
  SnFwdBn bn[2];
  BnApplyPtrs ap[2];
  const int batch_stats = d.has_bn && !d.eval;
  for (int i = 0; i < 2; ++i) {
    float* ws = batch_stats ? c.scratch->f(stats_ws_floats(n, F)) : nullptr;
    g[i] = SnFwdPtrs{agg[i], p[i].W, p[i].b, s[i].hn, s[i].rinv, ws};
    bn[i] = SnFwdBn{p[i].eps, p[i].mom, p[i].rm, p[i].rv, p[i].nbt, s[i].mean, s[i].istd};
    ap[i] = BnApplyPtrs{s[i].hn, d.has_bn ? s[i].mean : nullptr, s[i].istd, p[i].gamma, p[i].beta, y[i], y2[i], ldy2[i]};
  }
  CALL(sage_narrow_fwd_groups(g, bn, 2, lda, n, fin, F, 1, d.act, batch_stats, d.count, as_stream(c.s)));
  if (d.has_bn && d.eval)
    for (int i = 0; i < 2; ++i) CALL(cgc_bn_running_stats(p[i].rm, p[i].rv, F, p[i].eps, s[i].mean, s[i].istd, c.s));
  CALL(bn_act_apply_groups(ap, 2, n, F, d.act, ldy, as_stream(c.s)));
  c.scratch->release(m);
  return 0;
}

int layer_bwd_pair(const Ctx& c, const cgc_level_desc& d, const LayerP* p, const LayerS* s, const float* const* agg, int lda, int n, int fin, int F,
                   const float* const* dy, int ldy, float* const* dagg, int ldd, float* const* dwdb, float* const* sums) {
  const size_t m = c.scratch->mark();
  const int mode = d.has_bn ? 2 : 0;
  const size_t slot_floats = (size_t)(cgc_stats_blocks(n, F) > 1 ? cgc_stats_blocks(n, F) : 1) * 2 * F;
  if (d.has_bn) {
    BnRedPtrs r[2];
    for (int i = 0; i < 2; ++i) r[i] = BnRedPtrs{dy[i], s[i].hn, s[i].mean, s[i].istd, c.scratch->f(slot_floats)};
    CALL(bn_bwd_reduce_groups(r, sums, 2, ldy, n, F, d.act, as_stream(c.s)));
  }
  SnBwdPtrs g[2];
  for (int i = 0; i < 2; ++i)
    g[i] = SnBwdPtrs{dy[i], s[i].hn, s[i].rinv, s[i].mean, s[i].istd, p[i].gamma, d.has_bn ? sums[i] : nullptr, agg[i], p[i].W, dagg[i],
                     c.scratch->f((size_t)cgc_sage_narrow_ws_floats(n, fin, F))};
  CALL(sage_narrow_bwd_groups(g, dwdb, 2, ldy, n, F, d.act, 1, mode, d.count, lda, fin, ldd, as_stream(c.s)));
  c.scratch->release(m);
  return 0;
}

// ------------------------------------------------------------------------------------------------ a level
struct Level {
  const cgc_level_desc& d;
  int n, B, R, fin, H, E, AH, C, D, D3, wp, wt, ldp, ldC, ldP, ldW, ftot, npad_jk, seg_nmax;
  bool dense, pool, tall;
  // saved arena
  float *At, *An, *invd, *ge1, *agg0, *pair[2], *xcat, *aggk[2], *hp3, *cat_e, *HS, *CS, *jk_out, *x12, *S, *P;
  int* arg;
  LayerS L[6];
  // gradient layout
  cgc_level_grad_layout gl;

  explicit Level(const cgc_level_desc& dd) : d(dd) {
    n = d.n; B = d.B; R = d.rows_per_graph; fin = d.fin; H = d.H; E = d.E; AH = d.AH; C = d.C;
    dense = d.level >= 2;
    pool = C > 0;
    D3 = 2 * H + E;
    D = d.jk ? H : D3;
    wp = pool ? H + AH : H;
    // dense levels: the operands of the deferred adjacency gradient [pair 2 | pair 1 | x] live side by side in ONE buffer (row
    // stride wt), and so do the gradients [d agg 2 | d agg 1 | d agg 0] in the backward: that product needs no concatenation
    wt = 2 * wp + fin;
    ldp = dense ? wt : wp;
    ldC = pool ? wide_ld(C) : 0;
    ldP = dense ? pad4_ld(C) : ldC;
    ftot = 2 * AH + C;
    tall = pool && n >= 8 * C;                   // Linear over cat: transposed weight copy for tall products (ops._LinearCat)
    ldW = wide_ld(C);
    npad_jk = up(n > 1 ? n : 1, 1024);
    seg_nmax = dense ? R : d.npad;
    memset(&gl, 0, sizeof(gl));
  }

  void layout_saved(Arena& a) {
    At = An = invd = ge1 = nullptr;
    if (dense) {
      if (d.renorm) At = a.f((size_t)n * R);
      An = a.f((size_t)n * R);
      invd = a.f(n);
      ge1 = a.f(n);
    }
    agg0 = a.f((size_t)n * fin);
    xcat = dense ? a.f((size_t)n * wt) : nullptr;
    for (int k = 0; k < 2; ++k) {
      pair[k] = dense ? xcat + (1 - k) * wp : a.f((size_t)n * wp);
      aggk[k] = a.f((size_t)n * wp);
    }
    hp3 = pool ? a.f((size_t)n * ldC) : nullptr;
    for (int k = 0; k < 6; ++k) {
      const bool emb = k < 3;
      if (!emb && !pool) {
        L[k].hn = L[k].rinv = L[k].mean = L[k].istd = nullptr;
        continue;
      }
      const int F = width_out(k);
      L[k].hn = a.f((size_t)n * F);
      L[k].rinv = a.f(n);
      L[k].mean = a.f(F);
      L[k].istd = a.f(F);
    }
    cat_e = a.f((size_t)n * D3);
    HS = CS = jk_out = nullptr;
    if (d.jk) {
      const int Hh = 3 * H / 2;
      HS = a.f((size_t)6 * Hh * npad_jk);
      CS = a.f((size_t)6 * Hh * npad_jk);
      jk_out = a.f((size_t)n * H);
    }
    arg = a.i((size_t)B * D);
    x12 = S = P = nullptr;
    if (pool) {
      x12 = a.f((size_t)n * 2 * AH);
      S = a.f((size_t)n * ldC);
      P = a.f((size_t)n * ldP);
    }
  }
  int width_in(int k) const { return (k == 0 || k == 3) ? fin : (k < 3 ? H : AH); }
  int width_out(int k) const { return k < 2 ? H : k == 2 ? E : k < 5 ? AH : C; }
  const float* embed() const { return d.jk ? jk_out : cat_e; }

  void layout_grads() {
    int64_t o = 0;
    for (int k = 0; k < 6; ++k) {
      gl.W[k] = gl.b[k] = gl.bn_weight[k] = gl.bn_bias[k] = -1;
      if (k >= 3 && !pool) continue;
      const int fi = width_in(k), F = width_out(k);
      gl.W[k] = o;
      o += (int64_t)fi * F;
      if (d.has_bias) gl.b[k] = o;
      o += F;                                    // (reserved also without bias: [dW | db] is one output of the narrow backward kernel)
      if (d.has_bn) {
        gl.bn_bias[k] = o;
        gl.bn_weight[k] = o + F;
      }
      o += 2 * F;
      o = (o + 3) & ~(int64_t)3;
    }
    gl.lin_W = gl.lin_b = gl.jk = -1;
    if (pool) {
      gl.lin_W = o;
      o += (int64_t)C * ftot;
      gl.lin_b = o;
      o += C;
      o = (o + 3) & ~(int64_t)3;
    }
    if (d.jk) {
      gl.jk = o;
      o += cgc_jk_param_grad_floats(H);
      o = (o + 3) & ~(int64_t)3;
    }
    gl.total = o;
  }
};

// neighbour aggregation of a level and its transpose: level 1 on the CSR (ops._Aggregate), levels 2-3 A_norm @ h (ops._BMatmul)
int aggregate(const Ctx& c, const Level& L, const cgc_graph* g, const int* gptr, const float* h, int ldh, int w, float* out) {
  if (!L.dense) {
    CALL(cgc_spmm_graphs_ordered(g->rowptr, g->col, nullptr, g->val, nullptr, g->inv_d, h, out, L.n, w, w, gptr, L.B, L.d.nmax, 0, nullptr, c.s));
    return 0;
  }
  return bgemm(c, T3{L.An, L.B, L.R, L.R, L.R}, T3{const_cast<float*>(h), L.B, L.R, w, ldh}, T3{out, L.B, L.R, w, w}, 0, 0);
}
int aggregate_t(const Ctx& c, const Level& L, const cgc_graph* g, const int* gptr, const float* dy, int ldy, int w, float* dx) {
  if (!L.dense) {
    CALL(cgc_spmm_graphs_ordered(g->t_rowptr, g->t_col, nullptr, g->t_val, g->inv_d, nullptr, dy, dx, L.n, w, w, gptr, L.B, L.d.nmax, 0, nullptr,
                                 c.s));
    return 0;
  }
  return bgemm(c, T3{L.An, L.B, L.R, L.R, L.R}, T3{const_cast<float*>(dy), L.B, L.R, w, ldy}, T3{dx, L.B, L.R, w, w}, 1, 0);
}

int level_fwd(const Ctx& c, Level& L, const cgc_block_params* emb, const cgc_block_params* pl, const cgc_jk_params* jk, const cgc_graph* g,
              const int* gptr, const float* x_in, const float* A_in, float* readout, float* x_out, float* A_out) {
  const cgc_level_desc& d = L.d;
  const int n = L.n, H = L.H, AH = L.AH, wp = L.wp, C = L.C;
  if (L.dense)
    CALL(cgc_adj_prep_fwd(A_in, n, L.R, d.renorm ? d.renorm_p : -1.f, L.At, L.An, L.invd, L.ge1, c.s));
  TRY(aggregate(c, L, g, gptr, x_in, L.fin, L.fin, L.agg0));
  if (L.dense) {      // the level's input next to the pair buffers (third operand block of the deferred adjacency gradient)
    const float* s1[1] = {x_in};
    const int l1[1] = {L.fin}, w1[1] = {L.fin};
    CALL(cgc_cat_cols(L.xcat + 2 * wp, L.wt, n, 1, s1, l1, w1, 0, c.s));
  }
  // the two blocks layer by layer: both read the SAME aggregation (network.run_blocks_paired)
  for (int k = 0; k < 3; ++k) {
    const float* ain = k == 0 ? L.agg0 : L.aggk[k - 1];
    const int lda = k == 0 ? L.fin : wp;
    // a layer's output goes where the next aggregation reads it ([he | hp] side by side) AND into its slot of the block's
    // concatenation (cat[x1, x2, x3] of the embedding block, [hp1 | hp2] of the assignment block): no concatenation kernels
    if (k < 2 && L.pool && pairable(L.width_in(k), H, L.width_in(3 + k), AH)) {
      const LayerP pp[2] = {layer_params(d, emb, k, k), layer_params(d, pl, k, 3 + k)};
      const LayerS ss[2] = {L.L[k], L.L[3 + k]};
      const float* ag[2] = {ain, k == 0 ? ain : ain + H};
      float* yy[2] = {L.pair[k], L.pair[k] + H};
      float* y2[2] = {L.cat_e + k * H, L.x12 + k * AH};
      const int l2[2] = {L.D3, 2 * AH};
      TRY(layer_fwd_pair(c, d, pp, ss, ag, lda, n, L.width_in(k), H, yy, L.ldp, y2, l2));
      if (k < 2) TRY(aggregate(c, L, g, gptr, L.pair[k], L.ldp, wp, L.aggk[k]));
      continue;
    }
    if (k < 2)
      TRY(layer_fwd(c, d, layer_params(d, emb, k, k), L.L[k], ain, lda, n, L.width_in(k), L.width_out(k), L.pair[k], L.ldp, L.cat_e + k * H,
                    L.D3));
    else
      TRY(layer_fwd(c, d, layer_params(d, emb, k, k), L.L[k], ain, lda, n, L.width_in(k), L.width_out(k), L.cat_e + 2 * H, L.D3));
    if (L.pool) {
      const float* ainp = k == 0 ? ain : ain + H;
      if (k < 2)
        TRY(layer_fwd(c, d, layer_params(d, pl, k, 3 + k), L.L[3 + k], ainp, lda, n, L.width_in(3 + k), L.width_out(3 + k), L.pair[k] + H, L.ldp,
                      L.x12 + k * AH, 2 * AH));
      else
        TRY(layer_fwd(c, d, layer_params(d, pl, k, 3 + k), L.L[3 + k], ainp, lda, n, L.width_in(3 + k), L.width_out(3 + k), L.hp3, L.ldC));
    }
    if (k < 2) TRY(aggregate(c, L, g, gptr, L.pair[k], L.ldp, wp, L.aggk[k]));
  }
  if (d.jk) CALL(cgc_jk_lstm_fwd(L.cat_e, n, L.npad_jk, H, jk->lstm, jk->w_att, jk->b_att, L.jk_out, L.HS, L.CS, c.s));
  CALL(cgc_segment_max_fwd(L.embed(), gptr, L.B, L.D, L.seg_nmax, readout, L.arg, c.s));
  if (!L.pool) return 0;
  // assignment matrix: softmax(Linear(cat[hp1, hp2, hp3]))  (model/network.py:118-124, 200) -- the cat is never formed for the wide piece
  {
    const bool wide_main = C > 2 * AH;           // the widest piece is the main operand, the other rides along as an extra K segment
    const float* xm = wide_main ? L.hp3 : L.x12;
    const float* xe = wide_main ? L.x12 : L.hp3;
    const int ldm = wide_main ? L.ldC : 2 * AH, lde = wide_main ? 2 * AH : L.ldC;
    const int wm = wide_main ? C : 2 * AH, we = wide_main ? 2 * AH : C;
    const int om = wide_main ? 2 * AH : 0, oe = wide_main ? 0 : 2 * AH;
    if (L.tall) {
      const size_t m = c.scratch->mark();
      float* wt = c.scratch->f((size_t)L.ftot * L.ldW);
      CALL(cgc_transpose(pl->lin_W, L.ftot, C, L.ftot, wt, L.ldW, c.s));
      TRY(gemm_x1(c, 0, 0, n, C, wm, xm, ldm, wt + (size_t)om * L.ldW, L.ldW, 0.f, L.S, L.ldC, pl->lin_b, 1, 0, 0, 0, nullptr, 0, 0, xe, lde, 0,
                  wt + (size_t)oe * L.ldW, L.ldW, 0, we));
      c.scratch->release(m);
    } else {
      TRY(gemm_x1(c, 0, 1, n, C, wm, xm, ldm, pl->lin_W + om, L.ftot, 0.f, L.S, L.ldC, pl->lin_b, 1, 0, 0, 0, nullptr, 0, 0, xe, lde, 0,
                  pl->lin_W + oe, L.ftot, 0, we));
    }
  }
  CALL(cgc_softmax_fwd(L.S, n, C, L.ldC, L.S, c.s));
  // _diff_pool (model/network.py:194-208): X' = S^T X, A' = S^T (A S)
  if (!L.dense) {
    CALL(cgc_spmm_graphs_ordered(g->rowptr, g->col, nullptr, g->val, nullptr, nullptr, L.S, L.P, n, C, L.ldC, gptr, L.B, d.nmax, 1 | (g->spatial ? 4 : 0), g->gorder, c.s));
    TRY(gemm(c, 1, 0, C, L.D, 0, L.S, L.ldC, L.embed(), L.D, 0.f, x_out, L.D, nullptr, L.B, 0, 0, (int64_t)C * L.D, gptr, 2, d.nmax));
    TRY(gemm(c, 1, 0, C, C, 0, L.S, L.ldC, L.P, L.ldC, 0.f, A_out, C, nullptr, L.B, 0, 0, (int64_t)C * C, gptr, 2, d.nmax));
  } else {
    const T3 s3{L.S, L.B, L.R, C, L.ldC}, e3{const_cast<float*>(L.embed()), L.B, L.R, L.D, L.D}, p3{L.P, L.B, L.R, C, L.ldP};
    const T3 a3{const_cast<float*>(d.renorm ? L.At : A_in), L.B, L.R, L.R, L.R};
    TRY(bgemm(c, s3, e3, T3{x_out, L.B, C, L.D, L.D}, 1, 0));
    TRY(bgemm(c, a3, s3, p3, 0, 0));
    TRY(bgemm(c, s3, p3, T3{A_out, L.B, C, C, C}, 1, 0));
  }
  return 0;
}

// second half of a level's backward: DenseJK, then the blocks layer by layer from the last to the first, the transposed
// aggregations in between, and (levels 2-3) the gradients of the level's inputs.  d_embed [n, D] = gradient of the level's node
// embedding (readout + pooled features); dx12 [n, 2 AH] / dagg1 [n, H + AH] = what the assignment tail sends to hp1|hp2 and to the
// aggregation feeding the third layers (its assignment half filled in, nullptr without an assignment block); gAt = gradient reaching the re-normalised adjacency directly (or nullptr).
int level_bwd_blocks(const Ctx& c, Level& L, const cgc_block_params* emb, const cgc_block_params* pl, const cgc_jk_params* jk,
                     const cgc_graph* g, const int* gptr, const float* x_in, const float* A_in, float* d_embed, const float* dx12,
                     float* dagg1, const float* gAt, float* grads, float* d_x_in, float* d_A_in) {
  const cgc_level_desc& d = L.d;
  Arena& sc = *c.scratch;
  const int n = L.n, H = L.H, AH = L.AH, wp = L.wp, B = L.B, R = L.R, D3 = L.D3, fin = L.fin;
  auto sums = [&](int k) { return grads + L.gl.W[k] + (int64_t)L.width_in(k) * L.width_out(k) + L.width_out(k); };
  float* d_cat = d_embed;                                   // [n, 2H + E]: gradient of cat[x1, x2, x3] of the embedding block
  if (d.jk) {
    d_cat = sc.f((size_t)n * D3);
    const size_t m = sc.mark();
    float* ws = sc.f((size_t)cgc_jk_bwd_ws_floats(H));
    CALL(cgc_jk_lstm_bwd_flat(L.cat_e, d_embed, n, L.npad_jk, H, jk->lstm, jk->w_att, jk->b_att, L.HS, L.CS, d_cat, grads + L.gl.jk, ws, c.s));
    sc.release(m);
  }
  // ---- layer 3 -> gradient of the aggregation that fed it: [d agg_e3 | d agg_p3]
  float* dagg[2];                                           // dagg[k]: gradient of aggk[k] = A [pair k]
  const int ldg = L.ldp;                                    // row stride of the d agg blocks (dense levels: one [n, wt] buffer)
  float* const gcat = dagg1 != nullptr ? dagg1 : sc.f((size_t)n * ldg);
  dagg[1] = gcat;
  TRY(layer_bwd(c, d, layer_params(d, emb, 2, 2), L.L[2], L.aggk[1], wp, n, H, L.E, d_cat + 2 * H, D3, dagg[1], ldg, grads + L.gl.W[2], sums(2)));
  float* dpair = nullptr;
  for (int k = 1; k >= 0; --k) {
    // gradient of pair[k] = [he_{k+1} | hp_{k+1}]: through the aggregation, plus what the concatenations (embedding cat, x12) send
    dpair = sc.f((size_t)n * wp);
    TRY(aggregate_t(c, L, g, gptr, dagg[k], ldg, wp, dpair));
    if (L.pool) TRY(cat2(c, dpair, wp, n, d_cat + k * H, D3, H, dx12 + k * AH, 2 * AH, AH, 1));
    else TRY(add1(c, dpair, wp, n, d_cat + k * H, D3, H));
    // layer k+1 of both blocks (slot k)
    const bool first = k == 0;
    const bool need_in = !first || L.dense;                 // the level-1 input features carry no gradient
    const float* ain = first ? L.agg0 : L.aggk[k - 1];
    const int lda = first ? fin : wp;
    const int fi_e = L.width_in(k), fi_p = L.width_in(3 + k);
    // where the input gradients of this pair of layers go (row stride ldd): second layers -> the two halves of d aggk[0]; first
    // layers (dense levels only) -> the embedding block's into the d agg_0 block, the assignment block's into a buffer of the same
    // stride, added afterwards (both read the same aggregation A x)
    float *de = nullptr, *dp = nullptr;
    int ldd = ldg;
    if (need_in && !first) {
      de = dagg[0] = L.dense ? gcat + wp : sc.f((size_t)n * wp);
      dp = de + H;
    } else if (need_in) {
      de = gcat + 2 * wp;
      dp = L.pool ? sc.f((size_t)n * ldg) : nullptr;
    }
    if (L.pool && pairable(fi_e, H, fi_p, AH)) {
      const LayerP pp[2] = {layer_params(d, emb, k, k), layer_params(d, pl, k, 3 + k)};
      const LayerS ss[2] = {L.L[k], L.L[3 + k]};
      const float* ag[2] = {ain, first ? ain : ain + H};
      const float* dyy[2] = {dpair, dpair + H};
      float* dg[2] = {de, dp};
      float* dw[2] = {grads + L.gl.W[k], grads + L.gl.W[3 + k]};
      float* sm[2] = {sums(k), sums(3 + k)};
      TRY(layer_bwd_pair(c, d, pp, ss, ag, lda, n, fi_e, H, dyy, wp, dg, ldd, dw, sm));
    } else {
      TRY(layer_bwd(c, d, layer_params(d, emb, k, k), L.L[k], ain, lda, n, fi_e, H, dpair, wp, de, ldd, grads + L.gl.W[k], sums(k)));
      if (L.pool)
        TRY(layer_bwd(c, d, layer_params(d, pl, k, 3 + k), L.L[3 + k], first ? ain : ain + H, lda, n, fi_p, AH, dpair + H, wp, dp, ldd,
                      grads + L.gl.W[3 + k], sums(3 + k)));
    }
    if (first && L.dense) {
      if (L.pool) TRY(add1(c, de, ldg, n, dp, ldg, fin));
      TRY(aggregate_t(c, L, g, gptr, de, ldg, fin, d_x_in));
      // deferred, batched gradient of the row-normalised adjacency (ops.SharedGrad): every aggregation A h_i contributed
      // (d agg_i, h_i); ONE product [d agg_2 | d agg_1 | d agg_0] [h_2 | h_1 | h_0]^T writes the [B, C, C] gradient once.  Both
      // operands were laid out side by side when they were produced: no concatenation here.
      // N x N gradient matrices, then one pass over four of them (the per-operator path's schedule).  (Rounds 4-5 kept an opt-in route
      // that formed d A as ONE product of thin operands; its row terms carried row-coherent rounding that two reference fixtures
      // amplified past 1e-4 -- removed in round 6)
      float* dAn = sc.f((size_t)n * R);
      TRY(bgemm(c, T3{gcat, B, R, L.wt, L.wt}, T3{L.xcat, B, R, L.wt, L.wt}, T3{dAn, B, R, R, R}, 0, 1));
      CALL(cgc_adj_prep_bwd(A_in, L.An, L.invd, L.ge1, dAn, gAt, n, R, d.renorm ? d.renorm_p : -1.f, d_A_in, c.s));
    }
  }
  return 0;
}

int level_bwd(const Ctx& c, Level& L, const cgc_block_params* emb, const cgc_block_params* pl, const cgc_jk_params* jk, const cgc_graph* g,
              const int* gptr, const float* x_in, const float* A_in, const float* d_readout, const float* d_xo, const float* d_ao, float* grads,
              float* d_x_in, float* d_A_in) {
  const cgc_level_desc& d = L.d;
  Arena& sc = *c.scratch;
  const int n = L.n, H = L.H, AH = L.AH, wp = L.wp, C = L.C, D = L.D, B = L.B, R = L.R;
  float* d_embed = sc.f((size_t)n * D);
  CALL(cgc_segment_max_bwd_full(d_readout, L.arg, gptr, B, D, L.seg_nmax, d_embed, c.s));
  float *dx12 = nullptr, *gAt = nullptr;
  if (L.pool) {
    const size_t m0 = sc.mark();
    float* ds = sc.f((size_t)n * L.ldC);
    if (!L.dense) {
      float* dp = sc.f((size_t)n * L.ldC);
      TRY(gemm(c, 0, 0, 0, C, C, L.S, L.ldC, d_ao, C, 0.f, dp, L.ldC, nullptr, B, 0, (int64_t)C * C, 0, gptr, 1, d.nmax));      // dP = S dA'
      CALL(cgc_spmm_graphs_ordered(g->t_rowptr, g->t_col, nullptr, g->t_val, nullptr, nullptr, dp, ds, n, C, L.ldC, gptr, B, d.nmax, 2 | (g->spatial ? 4 : 0), g->gorder,
                                   c.s));                                                                                          // dS = A^T dP
      TRY(gemm_x1(c, 0, 1, 0, C, C, L.P, L.ldC, d_ao, C, 1.f, ds, L.ldC, nullptr, B, 0, (int64_t)C * C, 0, gptr, 1, d.nmax, L.embed(), D, 0, d_xo,
                  D, (int64_t)C * D, D));                                                                  // + P dA'^T + X dX'^T
      TRY(gemm(c, 0, 0, 0, D, C, L.S, L.ldC, d_xo, D, 1.f, d_embed, D, nullptr, B, 0, (int64_t)C * D, 0, gptr, 1, d.nmax));     // dX += S dX'
    } else {
      // (allocated below ds / de on purpose: it outlives them -- see the release further down)
      gAt = sc.f((size_t)n * R);
      float* dP = sc.f((size_t)n * L.ldP);
      const T3 s3{L.S, B, R, C, L.ldC}, e3{const_cast<float*>(L.embed()), B, R, D, D}, p3{L.P, B, R, C, L.ldP};
      const T3 a3{const_cast<float*>(d.renorm ? L.At : A_in), B, R, R, R};
      const T3 dao{const_cast<float*>(d_ao), B, C, C, C}, dxo{const_cast<float*>(d_xo), B, C, D, D};
      const T3 ds3{ds, B, R, C, L.ldC}, dP3{dP, B, R, C, L.ldP};
      TRY(bgemm(c, p3, dao, ds3, 0, 1));                      // dS  = P dA'^T
      TRY(bgemm(c, s3, dao, dP3, 0, 0));                      // dP  = S dA'
      TRY(bgemm(c, dP3, s3, T3{gAt, B, R, R, R}, 0, 1));   // d(A~) = dP S^T   (the gradient that reaches the re-normalised adjacency directly)
      TRY(bgemm(c, a3, dP3, ds3, 1, 0, 1.f));                 // dS += A~^T dP
      TRY(bgemm(c, e3, dxo, ds3, 0, 1, 1.f));                 // dS += X dX'^T
      TRY(bgemm(c, s3, dxo, T3{d_embed, B, R, D, D}, 0, 0, 1.f));   // dX += S dX'
    }
    // Linear over cat + softmax backward (ops._LinearCat.backward)
    float* dz = sc.f((size_t)n * L.ldC);
    {
      const size_t m = sc.mark();
      float* ws = sc.f((size_t)(cgc_stats_blocks(n, C) > 1 ? cgc_stats_blocks(n, C) : 1) * 2 * C);
      CALL(cgc_softmax_bwd(L.S, ds, n, C, L.ldC, dz, grads + L.gl.lin_b, ws, c.s));
      sc.release(m);
    }
    // (ds and, at level 1, dp are dead from here on; dz stays.  The arena is a stack: what must survive was allocated first.)
    dx12 = sc.f((size_t)n * 2 * AH);
    float* dy3 = sc.f((size_t)n * L.ldC);
    TRY(gemm(c, 0, 0, n, 2 * AH, C, dz, L.ldC, pl->lin_W, L.ftot, 0.f, dx12, 2 * AH));
    TRY(gemm(c, 0, 0, n, C, C, dz, L.ldC, pl->lin_W + 2 * AH, L.ftot, 0.f, dy3, L.ldC));
    {      // the Linear's weight gradient, piece by piece straight into its columns of the flat buffer (row stride ftot)
      float* dW = grads + L.gl.lin_W;
      TRY(gemm_tn_rows(c, dz, L.ldC, C, L.x12, 2 * AH, 2 * AH, n, dW, L.ftot));
      TRY(gemm_tn_rows(c, dz, L.ldC, C, L.hp3, L.ldC, C, n, dW + 2 * AH, L.ftot));
    }
    // third layer of the assignment block, from d hp3
    (void)m0;
    // gradient of the aggregation feeding the third layers, [embedding part | assignment part]; on dense levels the first block of
    // [d agg 2 | d agg 1 | d agg 0] (row stride ldp)
    float* dagg1 = sc.f((size_t)n * L.ldp);
    TRY(layer_bwd(c, d, layer_params(d, pl, 2, 5), L.L[5], L.aggk[1] + H, wp, n, AH, C, dy3, L.ldC, dagg1 + H, L.ldp, grads + L.gl.W[5],
                  grads + L.gl.W[5] + (int64_t)AH * C + C));
    // keep: dagg1, dx12, gAt.  (dy3, dz are not needed any more but sit below dagg1 on the stack; they are simply left there.)
    return level_bwd_blocks(c, L, emb, pl, jk, g, gptr, x_in, A_in, d_embed, dx12, dagg1, gAt, grads, d_x_in, d_A_in);
  }
  return level_bwd_blocks(c, L, emb, pl, jk, g, gptr, x_in, A_in, d_embed, nullptr, nullptr, nullptr, grads, d_x_in, d_A_in);
}

}  // namespace

// ------------------------------------------------------------------------------------------------ C ABI
extern "C" int cgc_level_supported(const cgc_level_desc* d) {
  if (d == nullptr || d->level < 1 || d->level > 3 || d->B < 1 || d->B > 65535 || d->n < 1) return 0;
  if (d->fin < 1 || d->H < 1 || d->E < 1 || d->act < 0 || d->act > 3) return 0;
  if (d->level >= 2 && (d->rows_per_graph < 1 || (long long)d->B * d->rows_per_graph != d->n)) return 0;
  if (d->level == 1 && (d->nmax < 1 || d->npad < d->nmax)) return 0;
  if (d->C > 0 && (d->AH < 1 || d->H + d->AH > 256 || (2 * d->AH) % 4 != 0)) return 0;
  if (d->C == 0 && d->level == 1) return 0;
  if ((d->flags & ~6) || (d->flags & 6) == 6) return 0;                   // bit 0 is reserved (ABI 4), bits 1 and 2 exclude each other, bits above 2 are unassigned
  if (d->jk && (d->E != d->H || !cgc_jk_matrix_core(d->H))) return 0;   // (other channel counts: staged parameter gradients, per-operator path)
  return 1;
}

extern "C" int64_t cgc_level_saved_floats(const cgc_level_desc* d) {
  if (!cgc_level_supported(d)) return -1;
  Level L(*d);
  Arena a(nullptr);
  L.layout_saved(a);
  return (int64_t)a.high + 64;
}

extern "C" int cgc_level_grad_layout_of(const cgc_level_desc* d, cgc_level_grad_layout* out) {
  if (!cgc_level_supported(d) || out == nullptr) return CGC_EINVAL;
  Level L(*d);
  L.layout_grads();
  *out = L.gl;
  return 0;
}

extern "C" int64_t cgc_level_scratch_floats(const cgc_level_desc* d) {
  if (!cgc_level_supported(d)) return -1;
  Level L(*d);
  Arena sv(nullptr);
  L.layout_saved(sv);
  L.layout_grads();
  size_t high = 0;
  for (int pass = 0; pass < (d->eval ? 1 : 2); ++pass) {       // (an inference descriptor never runs a backward)
    Arena sc(nullptr);
    Ctx c{nullptr, true, &sc, nullptr, 0};
    sc.f((size_t)cgc_gemm_ws_floats());
    cgc_block_params bp;
    memset(&bp, 0, sizeof(bp));
    cgc_jk_params jp;
    memset(&jp, 0, sizeof(jp));
    cgc_graph gr;
    memset(&gr, 0, sizeof(gr));
    const int rc = pass == 0 ? level_fwd(c, L, &bp, &bp, &jp, &gr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr)
                             : level_bwd(c, L, &bp, &bp, &jp, &gr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr);
    if (rc != 0) return -1;
    if (sc.high > high) high = sc.high;
  }
  return (int64_t)high + 64;
}

extern "C" int cgc_level_fwd(const cgc_level_desc* d, const cgc_block_params* emb, const cgc_block_params* pool, const cgc_jk_params* jk,
                             const cgc_graph* g, const int* gptr, const float* x_in, const float* A_in, float* saved, float* scratch,
                             float* readout, float* x_out, float* A_out, const float** assign_out, int* assign_ld, cgc_stream_t stream) {
  if (!cgc_level_supported(d) || saved == nullptr || scratch == nullptr) return CGC_EINVAL;
  if (!aligned16(saved) || !aligned16(scratch)) return CGC_EINVAL;
  Level L(*d);
  Arena sv(saved), sc(scratch);
  L.layout_saved(sv);
  L.layout_grads();
  Ctx c{stream, false, &sc, nullptr, cgc_gemm_ws_floats()};
  c.gws = sc.f((size_t)c.gws_floats);
  c.gemm_mode = (d->flags & 2) ? CGC_GEMM_SPLIT_BF16 : (d->flags & 4) ? CGC_GEMM_SPLIT_F16 : CGC_GEMM_EXACT;
  const int rc = level_fwd(c, L, emb, pool, jk, g, gptr, x_in, A_in, readout, x_out, A_out);
  if (assign_out != nullptr) *assign_out = L.S;
  if (assign_ld != nullptr) *assign_ld = L.ldC;
  return rc;
}

extern "C" int cgc_level_bwd(const cgc_level_desc* d, const cgc_block_params* emb, const cgc_block_params* pool, const cgc_jk_params* jk,
                             const cgc_graph* g, const int* gptr, const float* x_in, const float* A_in, const float* saved, float* scratch,
                             const float* d_readout, const float* d_x_out, const float* d_A_out, float* grads, float* d_x_in, float* d_A_in,
                             cgc_stream_t stream) {
  if (!cgc_level_supported(d) || saved == nullptr || scratch == nullptr || grads == nullptr || d->eval) return CGC_EINVAL;
  if (!aligned16(saved) || !aligned16(scratch) || !aligned16(grads)) return CGC_EINVAL;
  Level L(*d);
  Arena sv(const_cast<float*>(saved)), sc(scratch);
  L.layout_saved(sv);
  L.layout_grads();
  Ctx c{stream, false, &sc, nullptr, cgc_gemm_ws_floats()};
  c.gws = sc.f((size_t)c.gws_floats);
  c.gemm_mode = (d->flags & 2) ? CGC_GEMM_SPLIT_BF16 : (d->flags & 4) ? CGC_GEMM_SPLIT_F16 : CGC_GEMM_EXACT;
  return level_bwd(c, L, emb, pool, jk, g, gptr, x_in, A_in, d_readout, d_x_out, d_A_out, grads, d_x_in, d_A_in);
}
