// Launch groups: up to two layers of EQUAL shape -- the embedding and the assignment block of a level run layer by layer in
// lockstep (network.run_blocks_paired / csrc/exec.hip) -- share one launch of each narrow-layer kernel; blockIdx.y picks the
// pointer set.  At 4 graphs per GPU those kernels are 5 us of pure launch + drain each: halving their number is what the pairing buys.
#pragma once
#include "common.hpp"

struct SnFwdPtrs {           // cgc_sage_narrow_fwd
  const float *agg, *W, *bias;
  float *hn, *rinv, *ws;
};
struct SnFwdBn {             // BatchNorm side of a forward layer
  float eps, momentum;
  float *running_mean, *running_var;
  int64_t* nbt;
  float *mean, *istd;
};
struct SnBwdPtrs {           // cgc_sage_narrow_bwd
  const float *dy, *hn, *rinv, *mean, *istd, *gamma, *sums, *agg, *W;
  float *dagg, *ws;
};
struct StatsFinPtrs {        // slots in, BatchNorm vectors out
  const float* ws;
  float *running_mean, *running_var, *mean, *istd;
  long long* nbt;
  float eps, momentum;
};
struct BnApplyPtrs {         // cgc_bn_act_apply2
  const float *hn, *mean, *istd, *gamma, *beta;
  float *y, *y2;
  int ldy2;
};
struct BnRedPtrs {           // cgc_bn_bwd_reduce
  const float *dy, *hn, *mean, *istd;
  float* ws;
};

int sage_narrow_fwd_groups(const SnFwdPtrs* g, const SnFwdBn* bn, int ng, int lda, int n, int K, int F, int normalize, int act, int stats,
                           double count, hipStream_t st);
int sage_narrow_bwd_groups(const SnBwdPtrs* g, float* const* dwdb, int ng, int ldy, int n, int F, int act, int normalize, int mode, double count,
                           int lda, int fin, int ldd, hipStream_t st);
int bn_act_apply_groups(const BnApplyPtrs* g, int ng, int n, int F, int act, int ldy, hipStream_t stream);
int bn_bwd_reduce_groups(const BnRedPtrs* g, float* const* sums, int ng, int ldy, int n, int F, int act, hipStream_t stream);
int launch_stats_finalize_groups(const StatsFinPtrs* g, int ng, int slots, int F, double count, hipStream_t stream);
int launch_reduce_slots_f32_pair(const float* ws0, float* out0, const float* ws1, float* out1, int ng, int slots, int width, hipStream_t stream);
