// The optimiser of the reference's training step (common/utils.py:119-121: torch.optim.Adam, lr 1e-3, weight decay 1e-4) as ONE launch
// over all parameter tensors.
//
// The step sequencer leaves the gradients of a whole level in one flat buffer (cgc_level_grad_layout) and the head's in another, so a
// parameter's gradient is (buffer, offset) -- static for the life of the model.  The caller keeps a device table of segments
// {parameter, exp_avg, exp_avg_sq, gradient buffer index, offset, length} and a table (segment, chunk) per workgroup, both built once;
// a step passes the four buffer addresses and the scalars.  No per-step lists, no per-tensor metadata upload, no step-counter kernel:
// the host cost of an update is one call.
//
// Arithmetic: Adam with L2 weight decay folded into the gradient (Kingma & Ba; torch.optim.Adam's definition), evaluated with the
// same mixed precision as torch's fused kernel -- hyper-parameters are doubles, state and parameters float, bias corrections from a
// double pow rounded to float -- so that switching optimisers does not change a training run (tests/test_native_gpu.py checks
// the parameters and both moments bit for bit).
#include "common.hpp"

struct AdamSeg {          // == cgc_adam_seg
  float* p;
  float* m;
  float* v;
  long long off;          // first element of this parameter's gradient inside its buffer
  long long n;
  int slot;               // which of the gradient buffers
  int reserved;
};

__global__ __launch_bounds__(256) void k_adam_segments(const AdamSeg* __restrict__ segs, const int2* __restrict__ blocks,
                                                       const float* g0, const float* g1, const float* g2, const float* g3, double lr,
                                                       double beta1, double beta2, double wd, double eps, float step, float grad_mul) {
  const int2 bc = blocks[blockIdx.x];
  const AdamSeg sg = segs[bc.x];
  const float* gb = sg.slot == 0 ? g0 : (sg.slot == 1 ? g1 : (sg.slot == 2 ? g2 : g3));
  const float* g = gb + sg.off;
  const float bc1 = (float)(1.0 - pow(beta1, (double)step));
  const float bc2s = (float)sqrt(1.0 - pow(beta2, (double)step));
  const float step_size = (float)(lr / (double)bc1);
  const long long base = (long long)bc.y * 1024 + threadIdx.x;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const long long i = base + k * 256;
    if (i < sg.n) {
      float param = sg.p[i];
      float grad = g[i];
      if (grad_mul != 1.f) grad *= grad_mul;
      if (wd != 0.0) grad = (float)((double)grad + (double)param * wd);
      const float m = (float)(beta1 * (double)sg.m[i] + (1.0 - beta1) * (double)grad);
      const float v = (float)(beta2 * (double)sg.v[i] + (1.0 - beta2) * (double)grad * (double)grad);
      const float denom = (float)((double)(sqrtf(v) / bc2s) + eps);
      param -= step_size * m / denom;
      sg.p[i] = param;
      sg.m[i] = m;
      sg.v[i] = v;
    }
  }
}

extern "C" int cgc_adam_step(const void* segs, const void* blocks, int nblocks, const float* const* grad_buffers, double lr, double beta1,
                             double beta2, double weight_decay, double eps, float step, float grad_mul, cgc_stream_t stream) {
  if (nblocks <= 0) return 0;
  if (segs == nullptr || blocks == nullptr || grad_buffers == nullptr || !(step >= 1.f)) return CGC_EINVAL;
  hipLaunchKernelGGL(k_adam_segments, dim3((unsigned)nblocks), dim3(256), 0, as_stream(stream), static_cast<const AdamSeg*>(segs),
                     static_cast<const int2*>(blocks), grad_buffers[0], grad_buffers[1], grad_buffers[2], grad_buffers[3], lr, beta1, beta2,
                     weight_decay, eps, step, grad_mul);
  CGC_RETURN_IF_LAUNCH_FAILED();
  return 0;
}
