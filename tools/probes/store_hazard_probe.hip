// Reproducer: a 128-bit raw buffer store whose row offset sits in the instruction's SGPR offset leaves gfx950 (MI355X, ROCm 7.2
// hipcc) with the FIRST WORD of its data replaced by what a later vector instruction wrote to that data register -- here the 0 / 1 of a
// v_cndmask that the compiler places six instructions behind the store (LLVM's hazard table asks for 2 wait states between such a
// store and a VALU write of its data).  With the same
// offset carried in the lane's VGPR (scalar offset 0) the output is exact.  The kernel is the LDS-patch variant of
// k_sage_wide_cols (cgc-net_amd/csrc/sagewide.hip) as it stood when the effect was found:
// hn = (agg W + b) * rinv for [n, 20] x [20, 1140], each wave parking its 32 x 32 accumulator tiles in LDS and storing them as
// 16-byte pieces.  The instruction sequence in question (hipcc 7.2, -O3):
//     buffer_store_dwordx4 v[38:41], v37, s[92:95], s15 offen     <- data v38..v41, row offset in s15
//     v_subrev_u32_e32 v37, 24, v65
//     v_cmp_gt_i32_e64 s[0:1], s80, v37
//     s_add_i32 s49, s9, s79
//     s_add_i32 s50, s10, s79
//     v_cndmask_b32_e64 v37, 0, 1, s[0:1]
//     v_cndmask_b32_e32 v38, v37, v160, vcc                        <- writes v38 = 0 / 1: that value reaches memory
// Measured on MI355X (profiles/r05_store_hazard_probe.txt): 55 k wrong first words per run of 65.8 M elements (one piece in ~300)
// with the SGPR form, none with the VGPR form.   build + run: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o /tmp/store_hazard tools/probes/store_hazard_probe.hip && /tmp/store_hazard
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include <type_traits>

#define CGC_ACT_IDENTITY 0
#define CGC_ACT_RELU 1
#define CGC_ACT_ELU 2
#define CGC_ACT_LEAKYRELU 3
__device__ __forceinline__ float act_fwd(float x, int act) {
  switch (act) {
    case CGC_ACT_RELU: return x > 0.f ? x : 0.f;
    case CGC_ACT_ELU: return x > 0.f ? x : expf(x) - 1.f;
    case CGC_ACT_LEAKYRELU: return x > 0.f ? x : 0.01f * x;
    default: return x;
  }
}
#define SWC_TILES 3
#define SWC_PITCH 36
#define CGC_SWC_WAVES 3
typedef float floatx16 __attribute__((ext_vector_type(16)));
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

template <int KS, int ACT, bool SGPR_OFF>
#ifndef CGC_SWC_WAVES
#define CGC_SWC_WAVES 3
#endif
#ifndef CGC_SWC_AUX
#define CGC_SWC_AUX 0
#endif
__global__ __launch_bounds__(256, CGC_SWC_WAVES) void k_sage_wide_cols(const float* __restrict__ agg, int lda, const float* __restrict__ W,
                                                           const float* __restrict__ bias, int n, int K, int F,
                                                           const float* __restrict__ rinv, float* __restrict__ hn, int ldh,
                                                           float* __restrict__ ws, int row_tiles, int chunks, int ngroups) {
  constexpr int NT = SWC_TILES;
  __shared__ __attribute__((aligned(16))) float park[4][32 * SWC_PITCH];      // one 32 x 32 patch per wave (rows padded to 36 floats)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int l31 = lane & 31, lhi = lane >> 5;
  const int prow = lane >> 3, pc4 = (lane & 7) * 4;                           // the lane's place when the patch is read back row-wise
  float* pk = park[wave] + 4 * lhi * SWC_PITCH + l31;           // (pk and pq alias: no __restrict__)
  const float* pq = park[wave] + prow * SWC_PITCH + pc4;
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
      for (int t = 0; t < NT; ++t) acc[t][s % 16] += av[s] * bw[t][s];
#endif
    // This tile's row factors and the next tile's A fragments are requested behind this tile's MFMAs (they have the chains' ~2000
    // cycles to arrive) and waited for in front of its stores (vmcnt counts loads and stores in issue order: see k_sage_wide_fwd8)
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
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const bool colok = c_base + t * 32 + l31 < F;
      float a1 = 0.f, a2 = 0.f;
      // scaled values: into the statistics from the accumulator layout (lane = column: no cross-lane work) and into the wave's
      // LDS patch, from which they leave as 16-byte pieces of whole 128-byte row segments (8 lanes per row, 8 rows per
      // instruction: a dword store per accumulator register is 16 store instructions per tile instead of 4)
      if (full) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float v = acc[t][r] * rin[r];
#ifndef CGC_SWC_NOSTATS
          const float d = act_fwd(v, ACT) - shift[t];
          a1 += d;
          a2 = fmaf(d, d, a2);
#endif
          pk[((r & 3) + 8 * (r >> 2)) * SWC_PITCH] = v;
        }
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float v = acc[t][r] * rin[r];
          const bool ok = row0 + (r & 3) + 8 * (r >> 2) + 4 * lhi < n;
          const float d = ok ? act_fwd(v, ACT) - shift[t] : 0.f;
          a1 += d;
          a2 = fmaf(d, d, a2);
          pk[((r & 3) + 8 * (r >> 2)) * SWC_PITCH] = v;
        }
      }
      if (colok) { s1[t] += a1; s2[t] += a2; }
      __builtin_amdgcn_wave_barrier();           // (LDS operations of one wave execute in order: no hardware barrier needed)
      const int col4 = c_base + t * 32 + pc4;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float4 q = *reinterpret_cast<const float4*>(pq + 8 * j * SWC_PITCH);
        const bool ok = col4 < F && (full || row0 + 8 * j + prow < n);
        if (SGPR_OFF)      // the row offset in the instruction's scalar offset (an SGPR), the lane offset loop-invariant
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((ext_vector_type(4))) unsigned, q), rsrc_hn,
                                                 ok ? (unsigned)(prow * ldh + col4) * 4u : 0x80000000u,
                                                 (unsigned)(row0 + 8 * j) * (unsigned)ldh * 4u, 0);
        else               // the whole offset in the lane's VGPR, scalar offset 0
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((ext_vector_type(4))) unsigned, q), rsrc_hn,
                                                 ok ? (unsigned)((row0 + 8 * j + prow) * ldh + col4) * 4u : 0x80000000u, 0, 0);
      }
      __builtin_amdgcn_wave_barrier();
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



#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

int main() {
  const int n = 57696, K = 20, F = 1140, ldh = F;
  std::vector<float> agg((size_t)n * K), W((size_t)K * F), b(F), ri(n);
  unsigned s = 12345u;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (float)((s >> 8) & 0xffff) / 65536.f - 0.5f; };
  for (auto& v : agg) v = rnd();
  for (auto& v : W) v = rnd();
  for (auto& v : b) v = rnd();
  for (auto& v : ri) v = 1.f + 0.5f * rnd();
  std::vector<float> ref((size_t)n * F);
  for (int i = 0; i < n; ++i)
    for (int f = 0; f < F; ++f) {
      float a = 0.f;
      for (int k = 0; k < K; ++k) a = fmaf(agg[(size_t)i * K + k], W[(size_t)k * F + f], a);
      ref[(size_t)i * F + f] = (a + b[f]) * ri[i];
    }
  float *dagg, *dW, *db, *dri, *dhn;
  CHECK(hipMalloc(&dagg, agg.size() * 4)); CHECK(hipMalloc(&dW, W.size() * 4)); CHECK(hipMalloc(&db, b.size() * 4));
  CHECK(hipMalloc(&dri, ri.size() * 4)); CHECK(hipMalloc(&dhn, ref.size() * 4));
  CHECK(hipMemcpy(dagg, agg.data(), agg.size() * 4, hipMemcpyHostToDevice)); CHECK(hipMemcpy(dW, W.data(), W.size() * 4, hipMemcpyHostToDevice));
  CHECK(hipMemcpy(db, b.data(), b.size() * 4, hipMemcpyHostToDevice)); CHECK(hipMemcpy(dri, ri.data(), ri.size() * 4, hipMemcpyHostToDevice));
  const int row_tiles = (n + 31) / 32, chunks = 256, ngroups = ((F + 31) / 32 + 2) / 3;
  std::vector<float> out(ref.size());
  for (int form = 1; form >= 0; --form) {
    long long bad = 0, bad_first_word = 0, ones = 0;
    const int reps = 20;
    for (int r = 0; r < reps; ++r) {
      CHECK(hipMemset(dhn, 0, ref.size() * 4));
      if (form) hipLaunchKernelGGL((k_sage_wide_cols<11, CGC_ACT_RELU, true>), dim3(chunks * ((ngroups + 3) / 4)), dim3(256), 0, 0, dagg, K, dW, db, n, K, F, dri, dhn, ldh, (float*)nullptr, row_tiles, chunks, ngroups);
      else hipLaunchKernelGGL((k_sage_wide_cols<11, CGC_ACT_RELU, false>), dim3(chunks * ((ngroups + 3) / 4)), dim3(256), 0, 0, dagg, K, dW, db, n, K, F, dri, dhn, ldh, (float*)nullptr, row_tiles, chunks, ngroups);
      CHECK(hipDeviceSynchronize());
      CHECK(hipMemcpy(out.data(), dhn, out.size() * 4, hipMemcpyDeviceToHost));
      for (size_t i = 0; i < out.size(); ++i) {
        const float d = fabsf(out[i] - ref[i]);
        if (!(d <= 1e-4f)) {
          ++bad;
          if ((i % F) % 4 == 0) ++bad_first_word;
          unsigned u; memcpy(&u, &out[i], 4);
          if (u == 1u || u == 0u) ++ones;
        }
      }
    }
    printf("%s: %lld wrong elements in %d runs of %zu (%lld of them the first word of a 16-byte piece, %lld hold the integer 0 or 1)\n",
           form ? "row offset in the SGPR offset   " : "whole offset in the lane's VGPR", bad, reps, out.size(), bad_first_word, ones);
  }
  return 0;
}
