// Measurement hook: HIP events around selected launches (the dominant 128 x 128 GEMM, the wide SpMM), recorded on the stream the
// kernel is launched on, wherever the launch comes from (per-operator calls or the step sequencer).  Off unless an observer is
// attached (one per process); product calls never depend on it.  bench.py's `roofline` objects are computed from these records.
#include <atomic>

#include "common.hpp"

struct CgcTimingRec {
  int tag, dims[7];
  hipEvent_t e0, e1;
};
struct CgcTiming {
  int cap;
  std::atomic<int> n;
  CgcTimingRec* r;
};
static std::atomic<CgcTiming*> g_timing{nullptr};

int cgc_timing_begin(int tag, int d0, int d1, int d2, int d3, int d4, int d5, int d6, hipStream_t stream) {
  CgcTiming* t = g_timing.load(std::memory_order_acquire);
  if (t == nullptr) return -1;
  const int i = t->n.fetch_add(1);
  if (i >= t->cap) {
    t->n.store(t->cap);
    return -1;
  }
  CgcTimingRec& r = t->r[i];
  r.tag = tag;
  r.dims[0] = d0; r.dims[1] = d1; r.dims[2] = d2; r.dims[3] = d3; r.dims[4] = d4; r.dims[5] = d5; r.dims[6] = d6;
  (void)hipEventRecord(r.e0, stream);
  return i;
}
void cgc_timing_end(int idx, hipStream_t stream) {
  CgcTiming* t = g_timing.load(std::memory_order_acquire);
  if (t == nullptr || idx < 0) return;
  (void)hipEventRecord(t->r[idx].e1, stream);
}

extern "C" void* cgc_timing_create(int max_records) {
  if (max_records <= 0) return nullptr;
  CgcTiming* t = new CgcTiming;
  t->cap = max_records;
  t->n.store(0);
  t->r = new CgcTimingRec[max_records];
  for (int i = 0; i < max_records; ++i) {
    if (hipEventCreate(&t->r[i].e0) != hipSuccess || hipEventCreate(&t->r[i].e1) != hipSuccess) return nullptr;
  }
  return t;
}
extern "C" int cgc_timing_attach(void* h) {
  g_timing.store(static_cast<CgcTiming*>(h), std::memory_order_release);
  return 0;
}
extern "C" int cgc_timing_count(void* h) {
  CgcTiming* t = static_cast<CgcTiming*>(h);
  const int n = t->n.load();
  return n < t->cap ? n : t->cap;
}
extern "C" int cgc_timing_read(void* h, int i, int* tag_and_dims /*[8]*/, float* ms) {
  CgcTiming* t = static_cast<CgcTiming*>(h);
  if (i < 0 || i >= cgc_timing_count(h)) return CGC_EINVAL;
  tag_and_dims[0] = t->r[i].tag;
  for (int k = 0; k < 7; ++k) tag_and_dims[1 + k] = t->r[i].dims[k];
  const hipError_t e = hipEventElapsedTime(ms, t->r[i].e0, t->r[i].e1);
  return e == hipSuccess ? 0 : (int)e;
}
extern "C" int cgc_timing_destroy(void* h) {
  CgcTiming* t = static_cast<CgcTiming*>(h);
  if (t == nullptr) return 0;
  if (g_timing.load() == t) g_timing.store(nullptr);
  for (int i = 0; i < t->cap; ++i) {
    (void)hipEventDestroy(t->r[i].e0);
    (void)hipEventDestroy(t->r[i].e1);
  }
  delete[] t->r;
  delete t;
  return 0;
}
