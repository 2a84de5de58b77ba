// What an fp32 product costs on the bf16 matrix cores of gfx950 (measurement only: nothing in the library uses this).
//
// v_mfma_f32_32x32x2_f32 runs at 64 flop / cycle / SIMD, v_mfma_f32_32x32x16_bf16 at 1024.  An fp32 value is the exact sum of three
// bf16 values (hi = bf16(x), mid = bf16(x - hi), lo = bf16(x - hi - mid): 8 + 8 + 8 mantissa bits, both subtractions exact), and a
// bf16 x bf16 product is exact in fp32, so   a * b = sum over the nine pairs (a_p * b_q)   with every partial product exact and
// the additions done by the fp32 accumulator of the matrix core -- the same kind of rounding an fp32 MFMA chain has.  Nine bf16
// MFMAs per k = 16 replace eight fp32 MFMAs of k = 2: 288 against 512 matrix-pipe cycles.  Dropping the three pairs below 2^-24
// (mid*lo, lo*mid, lo*lo) leaves six: 192 cycles.
//
// The probe: C[M,N] = A[M,K] * B[N,K]^T  (both operands K-contiguous, the 'NT' product of the step: M = 58761, N = K = 1140, rows
// padded with zeros to a whole k-tile), 128 x 128 tiles, 256 threads (2 x 2 waves of 64 x 64), double-buffered LDS, two register
// stages, one barrier per k-tile, in three forms:
//   k_gemm_split<TERMS, 32>   operands fetched as fp32, split in registers, kept in LDS as three bf16 planes; one workgroup per CU
//   k_gemm_split<TERMS, 16>   the same with 16-wide k-tiles: two workgroups per CU (one's conversions beside the other's MFMAs)
//   k_gemm_pre<TERMS>         operands split by a pass of their own (6 bytes per element in memory), the k loop only copies and multiplies
// TERMS = 9 / 6 / 3 / 1 pairs.  Reports the launch time, the fp32-equivalent TFLOP/s and the error of sampled outputs against a
// float64 dot product, next to a plain fp32 fma chain on the host and an fp32-MFMA kernel of the same (naive) structure -- NOT the
// library's kernel, which does this product in 1160-1180 us.   Results: profiles/r04_bf16_split_probe.txt, DESIGN.md section 8.
//   hipcc --offload-arch=gfx950 -O3 -mllvm -unroll-threshold=100000 tools/bf16x9_probe.hip -o tools/bf16x9_probe.bin ;  tools/bf16x9_probe.bin [M N K [extra row pad]]
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <type_traits>
#include <vector>

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

#define BM 128
#define BN 128
#define BK 32

typedef float float2v __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

// x = hi + mid + lo exactly (up to the last bf16 rounding of lo); two values per v_cvt_pk_bf16_f32
__device__ __forceinline__ void split4(const float4 v, bf16x4& hi, bf16x4& mid, bf16x4& lo) {
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    float2v x;
    x[0] = i ? v.z : v.x;
    x[1] = i ? v.w : v.y;
    const bf16x2 h = __builtin_convertvector(x, bf16x2);
    float2v r1 = x - __builtin_convertvector(h, float2v);
    const bf16x2 m = __builtin_convertvector(r1, bf16x2);
    float2v r2 = r1 - __builtin_convertvector(m, float2v);
    const bf16x2 l = __builtin_convertvector(r2, bf16x2);
    hi[2 * i] = h[0]; hi[2 * i + 1] = h[1];
    mid[2 * i] = m[0]; mid[2 * i + 1] = m[1];
    lo[2 * i] = l[0]; lo[2 * i + 1] = l[1];
  }
}

// BK_ = 32: one workgroup per CU (120 KB of LDS); BK_ = 16: two (74 KB each), so that one workgroup's conversions and LDS traffic
// run in the shadow of the other's MFMAs
template <int TERMS, int BK_>
__global__ __launch_bounds__(256, BK_ == 16 ? 2 : 1) void k_gemm_split(const float* __restrict__ A, const float* __restrict__ B,
                                                                       float* __restrict__ C, int M, int N, int K, int ld, int tiles_n) {
  constexpr int ROW = BK_ * 2 + 16;              // bytes per LDS row of a plane (conflict-free 16-byte reads of 16 lanes: 80 / 48)
  constexpr int PLANE = 128 * ROW, OPER = 3 * PLANE, STAGE = 2 * OPER;
  constexpr int UPR = BK_ / 4;                   // float4 units per tile row
  constexpr int NL = 128 * UPR / 256;            // float4 loads per thread and operand
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int tile = blockIdx.x;
  const int tm = tile / tiles_n, tn = tile - tm * tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int l31 = lane & 31, lhi = lane >> 5;
  const float* pa[NL];
  const float* pb[NL];
  int lrow[NL];
  const int kq = (threadIdx.x % UPR) * 4;
#pragma unroll
  for (int i = 0; i < NL; ++i) {
    const int r = threadIdx.x / UPR + (256 / UPR) * i;
    lrow[i] = r;
    const int ra = m0 + r < M ? m0 + r : M - 1, rb = n0 + r < N ? n0 + r : N - 1;
    pa[i] = A + (size_t)ra * ld + kq;
    pb[i] = B + (size_t)rb * ld + kq;
  }
  float4 rA[2][NL], rB[2][NL];                   // two register sets: tile t travels in set t & 1
  auto fetch = [&](int kt, float4 (&ra)[NL], float4 (&rb)[NL]) {
    const int k0 = kt * BK_;
#pragma unroll
    for (int i = 0; i < NL; ++i) {               // (rows are padded with zeros to a whole k-tile: no k guard)
      ra[i] = *reinterpret_cast<const float4*>(pa[i] + k0);
      rb[i] = *reinterpret_cast<const float4*>(pb[i] + k0);
    }
  };
  auto stage = [&](int buf, const float4 (&ra)[NL], const float4 (&rb)[NL]) {
    unsigned char* base = lds + buf * STAGE;
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      bf16x4 h, m, l;
      split4(ra[i], h, m, l);
      unsigned char* p = base + lrow[i] * ROW + kq * 2;
      *reinterpret_cast<bf16x4*>(p) = h;
      *reinterpret_cast<bf16x4*>(p + PLANE) = m;
      *reinterpret_cast<bf16x4*>(p + 2 * PLANE) = l;
      split4(rb[i], h, m, l);
      p += OPER;
      *reinterpret_cast<bf16x4*>(p) = h;
      *reinterpret_cast<bf16x4*>(p + PLANE) = m;
      *reinterpret_cast<bf16x4*>(p + 2 * PLANE) = l;
    }
  };
  floatx16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int nk = (K + BK_ - 1) / BK_;
  fetch(0, rA[0], rB[0]);
  if (nk > 1) fetch(1, rA[1], rB[1]);
  stage(0, rA[0], rB[0]);
  if (nk > 2) fetch(2, rA[0], rB[0]);
  __syncthreads();
  // which pairs (plane of A, plane of B), most significant first
  constexpr int PA[9] = {0, 0, 1, 0, 1, 2, 1, 2, 2};
  constexpr int PB[9] = {0, 1, 0, 2, 1, 0, 2, 1, 2};
  auto tile_step = [&](int kt, auto cur_c) {
    constexpr int cur = decltype(cur_c)::value;
    const unsigned char* as = lds + cur * STAGE + (wm * 64 + l31) * ROW + lhi * 16;
    const unsigned char* bs = lds + cur * STAGE + OPER + (wn * 64 + l31) * ROW + lhi * 16;
#pragma unroll
    for (int ks = 0; ks < BK_ / 16; ++ks) {
      bf16x8 af[2][3], bf[2][3];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int p = 0; p < 3; ++p) {
          if (p < 2 || TERMS > 3) {
            af[i][p] = *reinterpret_cast<const bf16x8*>(as + i * 32 * ROW + p * PLANE + ks * 32);
            bf[i][p] = *reinterpret_cast<const bf16x8*>(bs + i * 32 * ROW + p * PLANE + ks * 32);
          }
        }
#pragma unroll
      for (int t = 0; t < TERMS; ++t)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][PA[t]], bf[j][PB[t]], acc[i][j], 0, 0, 0);
      if (ks == 0) {
        if (kt + 1 < nk) stage(cur ^ 1, rA[cur ^ 1], rB[cur ^ 1]);    // tile kt+1: registers -> split -> the other LDS buffer
        if (kt + 3 < nk) fetch(kt + 3, rA[cur ^ 1], rB[cur ^ 1]);     // and its register set takes tile kt+3
      }
    }
    __syncthreads();
  };
  for (int kt = 0; kt < nk; kt += 2) {
    tile_step(kt, std::integral_constant<int, 0>());
    if (kt + 1 < nk) tile_step(kt + 1, std::integral_constant<int, 1>());
  }
  // C: lane (l31, lhi) of accumulator (i, j) holds column n = l31, rows (r & 3) + 8 (r >> 2) + 4 lhi
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int col = n0 + wn * 64 + j * 32 + l31;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
        if (row < M && col < N) C[(size_t)row * N + col] = acc[i][j][r];
      }
    }
}

// The same product with the operands split BEFORE the product (three bf16 planes per operand in memory, 6 bytes per element instead
// of 4): the k loop then moves 16-byte pieces from memory to LDS and issues MFMAs, nothing else.
__global__ __launch_bounds__(256) void k_presplit(const float* __restrict__ X, long long n4, __bf16* __restrict__ P, long long plane) {
  for (long long u = blockIdx.x * 256LL + threadIdx.x; u < n4; u += (long long)gridDim.x * 256) {
    bf16x4 h, m, l;
    split4(reinterpret_cast<const float4*>(X)[u], h, m, l);
    *reinterpret_cast<bf16x4*>(P + 4 * u) = h;
    *reinterpret_cast<bf16x4*>(P + plane + 4 * u) = m;
    *reinterpret_cast<bf16x4*>(P + 2 * plane + 4 * u) = l;
  }
}

template <int TERMS, int NW>      // NW = workgroups per CU the launch bounds ask for (LDS: 120 KB per workgroup at BK = 32 -> 1)
__global__ __launch_bounds__(256, NW) void k_gemm_pre(const __bf16* __restrict__ A, long long planeA, const __bf16* __restrict__ B,
                                                      long long planeB, float* __restrict__ C, int M, int N, int K, int ld, int tiles_n) {
  constexpr int NP = TERMS > 3 ? 3 : TERMS > 1 ? 2 : 1;     // planes in use
  constexpr int ROW = BK * 2 + 16, PLANE = 128 * ROW, OPER = 3 * PLANE, STAGE = 2 * OPER;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int tile = blockIdx.x;
  const int tm = tile / tiles_n, tn = tile - tm * tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int l31 = lane & 31, lhi = lane >> 5;
  // 16-byte units: 4 per tile row and plane, 2 per thread
  const __bf16* pa[2];
  const __bf16* pb[2];
  int loff[2];
  const int kq = (threadIdx.x & 3) * 8;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int r = (threadIdx.x >> 2) + 64 * i;
    loff[i] = r * ROW + kq * 2;
    const int ra = m0 + r < M ? m0 + r : M - 1, rb = n0 + r < N ? n0 + r : N - 1;
    pa[i] = A + (size_t)ra * ld + kq;
    pb[i] = B + (size_t)rb * ld + kq;
  }
  bf16x8 rA[2][NP][2], rB[2][NP][2];
  auto fetch = [&](int kt, bf16x8 (&ra)[NP][2], bf16x8 (&rb)[NP][2]) {
#pragma unroll
    for (int p = 0; p < NP; ++p)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        ra[p][i] = *reinterpret_cast<const bf16x8*>(pa[i] + p * planeA + kt * BK);
        rb[p][i] = *reinterpret_cast<const bf16x8*>(pb[i] + p * planeB + kt * BK);
      }
  };
  auto stage = [&](int buf, const bf16x8 (&ra)[NP][2], const bf16x8 (&rb)[NP][2]) {
    unsigned char* base = lds + buf * STAGE;
#pragma unroll
    for (int p = 0; p < NP; ++p)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        *reinterpret_cast<bf16x8*>(base + p * PLANE + loff[i]) = ra[p][i];
        *reinterpret_cast<bf16x8*>(base + OPER + p * PLANE + loff[i]) = rb[p][i];
      }
  };
  floatx16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  const int nk = (K + BK - 1) / BK;
  fetch(0, rA[0], rB[0]);
  if (nk > 1) fetch(1, rA[1], rB[1]);
  stage(0, rA[0], rB[0]);
  if (nk > 2) fetch(2, rA[0], rB[0]);
  __syncthreads();
  constexpr int PA[9] = {0, 0, 1, 0, 1, 2, 1, 2, 2};
  constexpr int PB[9] = {0, 1, 0, 2, 1, 0, 2, 1, 2};
  auto tile_step = [&](int kt, auto cur_c) {
    constexpr int cur = decltype(cur_c)::value;
    const unsigned char* as = lds + cur * STAGE + (wm * 64 + l31) * ROW + lhi * 16;
    const unsigned char* bs = lds + cur * STAGE + OPER + (wn * 64 + l31) * ROW + lhi * 16;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 af[2][NP], bf[2][NP];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int p = 0; p < NP; ++p) {
          af[i][p] = *reinterpret_cast<const bf16x8*>(as + i * 32 * ROW + p * PLANE + ks * 32);
          bf[i][p] = *reinterpret_cast<const bf16x8*>(bs + i * 32 * ROW + p * PLANE + ks * 32);
        }
#pragma unroll
      for (int t = 0; t < TERMS; ++t)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][PA[t]], bf[j][PB[t]], acc[i][j], 0, 0, 0);
      if (ks == 0) {
        if (kt + 1 < nk) stage(cur ^ 1, rA[cur ^ 1], rB[cur ^ 1]);
        if (kt + 3 < nk) fetch(kt + 3, rA[cur ^ 1], rB[cur ^ 1]);
      }
    }
    __syncthreads();
  };
  for (int kt = 0; kt < nk; kt += 2) {
    tile_step(kt, std::integral_constant<int, 0>());
    if (kt + 1 < nk) tile_step(kt + 1, std::integral_constant<int, 1>());
  }
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int col = n0 + wn * 64 + j * 32 + l31;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
        if (row < M && col < N) C[(size_t)row * N + col] = acc[i][j][r];
      }
    }
}

// The in-register split again, with the instruction order placed by hand (tools/pipe_overlap_probe.hip: behind a bf16 MFMA up to
// ~6 plain vector instructions of the SAME wave cost ~1 cycle each, while a second wave's vector work does not overlap at all, and
// packed fp32 instructions cost more than two plain ones).  A tile's split is cut into 64 micro-steps of 2-4 instructions (per float4:
// convert / expand / subtract / convert / expand / subtract / convert / three 8-byte LDS writes) that are dealt out behind the tile's
// MFMAs, the fragment reads of the second k step behind the first MFMAs, and the global loads behind the last; a scheduling fence
// after every MFMA keeps the compiler from regrouping them.
template <int TERMS>
__global__ __launch_bounds__(256, 1) void k_gemm_split_placed(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ C, int M,
                                                              int N, int K, int ld, int tiles_n) {
  constexpr int ROW = BK * 2 + 16, PLANE = 128 * ROW, OPER = 3 * PLANE, STAGE = 2 * OPER;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int tile = blockIdx.x;
  const int tm = tile / tiles_n, tn = tile - tm * tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int l31 = lane & 31, lhi = lane >> 5;
  const float* pg[8];                       // units 0-3: A rows, 4-7: B rows
  int loff[8];
  const int kq = (threadIdx.x & 7) * 4;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = (threadIdx.x >> 3) + 32 * i;
    const int ra = m0 + r < M ? m0 + r : M - 1, rb = n0 + r < N ? n0 + r : N - 1;
    pg[i] = A + (size_t)ra * ld + kq;
    pg[4 + i] = B + (size_t)rb * ld + kq;
    loff[i] = r * ROW + kq * 2;
    loff[4 + i] = OPER + r * ROW + kq * 2;
  }
  float4 rx[2][8];                          // two register sets: tile t travels in set t & 1
  floatx16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  const int nk = (K + BK - 1) / BK;
  // prologue: tile 0 through the plain path
#pragma unroll
  for (int u = 0; u < 8; ++u) rx[0][u] = *reinterpret_cast<const float4*>(pg[u]);
  if (nk > 1) {
#pragma unroll
    for (int u = 0; u < 8; ++u) rx[1][u] = *reinterpret_cast<const float4*>(pg[u] + BK);
  }
#pragma unroll
  for (int u = 0; u < 8; ++u) {
    bf16x4 h, m, l;
    split4(rx[0][u], h, m, l);
    unsigned char* p = lds + loff[u];
    *reinterpret_cast<bf16x4*>(p) = h;
    *reinterpret_cast<bf16x4*>(p + PLANE) = m;
    *reinterpret_cast<bf16x4*>(p + 2 * PLANE) = l;
  }
  if (nk > 2) {
#pragma unroll
    for (int u = 0; u < 8; ++u) rx[0][u] = *reinterpret_cast<const float4*>(pg[u] + 2 * BK);
  }
  __syncthreads();
  constexpr int PA[9] = {0, 0, 1, 0, 1, 2, 1, 2, 2};
  constexpr int PB[9] = {0, 1, 0, 2, 1, 0, 2, 1, 2};
  constexpr int NM = TERMS * 8;             // MFMAs per tile and wave
  auto tile_step = [&](int kt, auto cur_c, auto last_c) {
    constexpr int cur = decltype(cur_c)::value;
    constexpr bool has_next = !decltype(last_c)::value;      // (compile-time: no branch behind the MFMAs)
    const unsigned char* as = lds + cur * STAGE + (wm * 64 + l31) * ROW + lhi * 16;
    const unsigned char* bs = lds + cur * STAGE + OPER + (wn * 64 + l31) * ROW + lhi * 16;
    unsigned char* const wbase = lds + (cur ^ 1) * STAGE;
    float4 (&src)[8] = rx[cur ^ 1];         // tile kt+1
    bf16x8 fr[2][12];                       // fragments of the two k steps: [A i=0..1][plane], then B
    // split state of the unit in flight
    float xs[8][4], r1[8][4], r2[8][4];
    unsigned hp[8][2], mp[8][2], lp[8][2];
    auto frag_read = [&](int ks, int q) {   // q = 0..11
      const int i = (q % 6) / 3, pl = q % 3;
      fr[ks][q] = *reinterpret_cast<const bf16x8*>((q < 6 ? as : bs) + i * 32 * ROW + pl * PLANE + ks * 32);
    };
    auto cvt2 = [&](float a, float b) -> unsigned {
      float2v t;
      t[0] = a;
      t[1] = b;
      return __builtin_bit_cast(unsigned, __builtin_convertvector(t, bf16x2));
    };
    auto micro = [&](int sidx) {
      const int u = sidx >> 3, st = sidx & 7;
      if (st == 0) {
        xs[u][0] = src[u].x; xs[u][1] = src[u].y; xs[u][2] = src[u].z; xs[u][3] = src[u].w;
        hp[u][0] = cvt2(xs[u][0], xs[u][1]);
        hp[u][1] = cvt2(xs[u][2], xs[u][3]);
      } else if (st == 1 || st == 2) {       // expand + subtract, two values per step
        const int h = st - 1;
        const float e0 = __builtin_bit_cast(float, hp[u][h] << 16), e1 = __builtin_bit_cast(float, hp[u][h] & 0xffff0000u);
        r1[u][2 * h] = xs[u][2 * h] - e0;
        r1[u][2 * h + 1] = xs[u][2 * h + 1] - e1;
        asm volatile("" : "+v"(r1[u][2 * h]), "+v"(r1[u][2 * h + 1]));
      } else if (st == 3) {
        mp[u][0] = cvt2(r1[u][0], r1[u][1]);
        mp[u][1] = cvt2(r1[u][2], r1[u][3]);
      } else if (st == 4 || st == 5) {
        const int h = st - 4;
        const float e0 = __builtin_bit_cast(float, mp[u][h] << 16), e1 = __builtin_bit_cast(float, mp[u][h] & 0xffff0000u);
        r2[u][2 * h] = r1[u][2 * h] - e0;
        r2[u][2 * h + 1] = r1[u][2 * h + 1] - e1;
        asm volatile("" : "+v"(r2[u][2 * h]), "+v"(r2[u][2 * h + 1]));
      } else if (st == 6) {
        lp[u][0] = cvt2(r2[u][0], r2[u][1]);
        lp[u][1] = cvt2(r2[u][2], r2[u][3]);
      } else {
        unsigned char* p = wbase + loff[u];
        *reinterpret_cast<uint2*>(p) = make_uint2(hp[u][0], hp[u][1]);
        *reinterpret_cast<uint2*>(p + PLANE) = make_uint2(mp[u][0], mp[u][1]);
        *reinterpret_cast<uint2*>(p + 2 * PLANE) = make_uint2(lp[u][0], lp[u][1]);
      }
    };
#pragma unroll
    for (int q = 0; q < 12; ++q) frag_read(0, q);
    __builtin_amdgcn_sched_barrier(0);
#pragma clang loop unroll(full)
    for (int m = 0; m < NM; ++m) {
      const int ks = m / (NM / 2), mm = m % (NM / 2);
      const int t = mm / 4, ij = mm % 4, i = ij >> 1, j = ij & 1;
      acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[ks][i * 3 + PA[t]], fr[ks][6 + j * 3 + PB[t]], acc[i][j], 0, 0, 0);
      if (m < 12) frag_read(1, m);                              // the second k step's fragments, one read behind each of the first MFMAs
      if constexpr (has_next) {
#pragma unroll
        for (int sidx = (m * 64) / NM; sidx < ((m + 1) * 64) / NM; ++sidx) micro(sidx);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if (kt + 3 < nk) {
#pragma unroll
      for (int u = 0; u < 8; ++u) src[u] = *reinterpret_cast<const float4*>(pg[u] + (kt + 3) * BK);
    }
    __syncthreads();
  };
  typedef std::integral_constant<int, 0> C0;
  typedef std::integral_constant<int, 1> C1;
  int kt = 0;
  for (; kt + 2 < nk; kt += 2) {
    tile_step(kt, C0(), std::false_type());
    tile_step(kt + 1, C1(), std::false_type());
  }
  if (kt + 1 < nk) {
    tile_step(kt, C0(), std::false_type());
    tile_step(kt + 1, C1(), std::true_type());
  } else if (kt < nk) {
    tile_step(kt, C0(), std::true_type());
  }
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int col = n0 + wn * 64 + j * 32 + l31;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
        if (row < M && col < N) C[(size_t)row * N + col] = acc[i][j][r];
      }
    }
}

// The placed split on a 256 x 128 tile: 512 threads (4 x 2 waves of 64 x 64), 16-wide k-tiles (110 KB of LDS double-buffered, one
// workgroup per CU), four register sets (a tile is requested four k-tiles before it is split), the three loads of a k-tile dealt
// out behind MFMAs like everything else, no branch inside a k-tile.  A 256 x 128 tile asks L2 for 3.7 GB per launch instead of 4.9.
template <int TERMS>
__global__ __launch_bounds__(512, 1) void k_gemm_split_big(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ C, int M,
                                                           int N, int K, int ld, int tiles_n) {
  constexpr int BKb = 16, ROW = BKb * 2 + 16;
  constexpr int PLA = 256 * ROW, PLB = 128 * ROW, STAGE = 3 * PLA + 3 * PLB;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int tile = blockIdx.x;
  const int tm = tile / tiles_n, tn = tile - tm * tiles_n;
  const int m0 = tm * 256, n0 = tn * 128;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int l31 = lane & 31, lhi = lane >> 5;
  const float* pg[3];                       // units 0-1: A rows, 2: a B row
  int loff[3];
  const int kq = (threadIdx.x & 3) * 4;
  {
    const int r = threadIdx.x >> 2;         // 0..127
    const int ra0 = m0 + r < M ? m0 + r : M - 1, ra1 = m0 + 128 + r < M ? m0 + 128 + r : M - 1, rb = n0 + r < N ? n0 + r : N - 1;
    pg[0] = A + (size_t)ra0 * ld + kq;
    pg[1] = A + (size_t)ra1 * ld + kq;
    pg[2] = B + (size_t)rb * ld + kq;
    loff[0] = r * ROW + kq * 2;
    loff[1] = (128 + r) * ROW + kq * 2;
    loff[2] = 3 * PLA + r * ROW + kq * 2;
  }
  float4 rx[4][3];                          // tile t travels in set t & 3
  floatx16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  const int nk = (K + BKb - 1) / BKb;
  auto koff = [&](int t) { return (t < nk ? t : nk - 1) * BKb; };      // past the end: the last tile again (never used)
  float4 first[3];
#pragma unroll
  for (int u = 0; u < 3; ++u) first[u] = *reinterpret_cast<const float4*>(pg[u]);
#pragma unroll
  for (int t = 1; t <= 4; ++t)
#pragma unroll
    for (int u = 0; u < 3; ++u) rx[t & 3][u] = *reinterpret_cast<const float4*>(pg[u] + koff(t));
#pragma unroll
  for (int u = 0; u < 3; ++u) {
    bf16x4 h, m, l;
    split4(first[u], h, m, l);
    unsigned char* p = lds + loff[u];
    const int pl = u < 2 ? PLA : PLB;
    *reinterpret_cast<bf16x4*>(p) = h;
    *reinterpret_cast<bf16x4*>(p + pl) = m;
    *reinterpret_cast<bf16x4*>(p + 2 * pl) = l;
  }
  __syncthreads();
  constexpr int PA[9] = {0, 0, 1, 0, 1, 2, 1, 2, 2};
  constexpr int PB[9] = {0, 1, 0, 2, 1, 0, 2, 1, 2};
  constexpr int NM = TERMS * 4;             // MFMAs per k-tile and wave
  auto tile_step = [&](int kt, auto cur_c, auto set_c) {
    constexpr int cur = decltype(cur_c)::value;
    constexpr int set = decltype(set_c)::value;                    // the set of tile kt + 1
    const unsigned char* as = lds + cur * STAGE + (wm * 64 + l31) * ROW + lhi * 16;
    const unsigned char* bs = lds + cur * STAGE + 3 * PLA + (wn * 64 + l31) * ROW + lhi * 16;
    unsigned char* const wbase = lds + (cur ^ 1) * STAGE;
    float4 (&src)[3] = rx[set];
    bf16x8 fr[12];
    float xs[3][4], r1[3][4], r2[3][4];
    unsigned hp[3][2], mp[3][2], lp[3][2];
    auto cvt2 = [&](float a, float b) -> unsigned {
      float2v t;
      t[0] = a;
      t[1] = b;
      return __builtin_bit_cast(unsigned, __builtin_convertvector(t, bf16x2));
    };
    auto micro = [&](int sidx) {
      const int u = sidx >> 3, st = sidx & 7;
      if (st == 0) {
        xs[u][0] = src[u].x; xs[u][1] = src[u].y; xs[u][2] = src[u].z; xs[u][3] = src[u].w;
        hp[u][0] = cvt2(xs[u][0], xs[u][1]);
        hp[u][1] = cvt2(xs[u][2], xs[u][3]);
      } else if (st == 1 || st == 2) {
        const int h = st - 1;
        const float e0 = __builtin_bit_cast(float, hp[u][h] << 16), e1 = __builtin_bit_cast(float, hp[u][h] & 0xffff0000u);
        r1[u][2 * h] = xs[u][2 * h] - e0;
        r1[u][2 * h + 1] = xs[u][2 * h + 1] - e1;
        asm volatile("" : "+v"(r1[u][2 * h]), "+v"(r1[u][2 * h + 1]));
      } else if (st == 3) {
        mp[u][0] = cvt2(r1[u][0], r1[u][1]);
        mp[u][1] = cvt2(r1[u][2], r1[u][3]);
      } else if (st == 4 || st == 5) {
        const int h = st - 4;
        const float e0 = __builtin_bit_cast(float, mp[u][h] << 16), e1 = __builtin_bit_cast(float, mp[u][h] & 0xffff0000u);
        r2[u][2 * h] = r1[u][2 * h] - e0;
        r2[u][2 * h + 1] = r1[u][2 * h + 1] - e1;
        asm volatile("" : "+v"(r2[u][2 * h]), "+v"(r2[u][2 * h + 1]));
      } else if (st == 6) {
        lp[u][0] = cvt2(r2[u][0], r2[u][1]);
        lp[u][1] = cvt2(r2[u][2], r2[u][3]);
      } else {
        unsigned char* p = wbase + loff[u];
        const int pl = u < 2 ? PLA : PLB;
        *reinterpret_cast<uint2*>(p) = make_uint2(hp[u][0], hp[u][1]);
        *reinterpret_cast<uint2*>(p + pl) = make_uint2(mp[u][0], mp[u][1]);
        *reinterpret_cast<uint2*>(p + 2 * pl) = make_uint2(lp[u][0], lp[u][1]);
      }
    };
#pragma unroll
    for (int q = 0; q < 12; ++q) {       // (issued in the order the MFMAs want them instead: 876 vs 847 us -- no gain)
      const int i = (q % 6) / 3, pl = q % 3;
      fr[q] = *reinterpret_cast<const bf16x8*>((q < 6 ? as + pl * PLA : bs + pl * PLB) + i * 32 * ROW);
    }
    __builtin_amdgcn_sched_barrier(0);
    const int knext = koff(kt + 5);
#pragma clang loop unroll(full)
    for (int m = 0; m < NM; ++m) {
      const int t = m / 4, ij = m % 4, i = ij >> 1, j = ij & 1;
      acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[i * 3 + PA[t]], fr[6 + j * 3 + PB[t]], acc[i][j], 0, 0, 0);
#pragma unroll
      for (int sidx = (m * 24) / NM; sidx < ((m + 1) * 24) / NM; ++sidx) micro(sidx);
      // the set's registers are free once its three units have been picked up (micro-steps 0, 8, 16): tile kt+5 goes there
      if (m >= NM - 3) src[m - (NM - 3)] = *reinterpret_cast<const float4*>(pg[m - (NM - 3)] + knext);
      __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();
  };
  typedef std::integral_constant<int, 0> I0;
  typedef std::integral_constant<int, 1> I1;
  typedef std::integral_constant<int, 2> I2;
  typedef std::integral_constant<int, 3> I3;
  for (int kt = 0; kt < nk; kt += 4) {
    tile_step(kt, I0(), I1());
    if (kt + 1 < nk) tile_step(kt + 1, I1(), I2());
    if (kt + 2 < nk) tile_step(kt + 2, I0(), I3());
    if (kt + 3 < nk) tile_step(kt + 3, I1(), I0());
  }
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int col = n0 + wn * 64 + j * 32 + l31;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
        if (row < M && col < N) C[(size_t)row * N + col] = acc[i][j][r];
      }
    }
}

// the fp32 matrix-core chain over the same tiling, as the yardstick of both time and error (32x32x2, operands as fp32 in LDS)
__global__ __launch_bounds__(256, 2) void k_gemm_f32ref(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ C, int M,
                                                        int N, int K, int ld, int tiles_n) {
  __shared__ __attribute__((aligned(16))) float sa[2][128 * 36], sb[2][128 * 36];
  const int tile = blockIdx.x;
  const int tm = tile / tiles_n, tn = tile - tm * tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int l31 = lane & 31, lhi = lane >> 5;
  const int kq = (threadIdx.x & 7) * 4;
  const float* pa[4];
  const float* pb[4];
  int lrow[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = (threadIdx.x >> 3) + 32 * i;
    lrow[i] = r;
    const int ra = m0 + r < M ? m0 + r : M - 1, rb = n0 + r < N ? n0 + r : N - 1;
    pa[i] = A + (size_t)ra * ld + kq;
    pb[i] = B + (size_t)rb * ld + kq;
  }
  float4 ra[4], rb[4];
  auto fetch = [&](int kt) {
    const int k0 = kt * BK;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      ra[i] = *reinterpret_cast<const float4*>(pa[i] + k0);
      rb[i] = *reinterpret_cast<const float4*>(pb[i] + k0);
    }
  };
  auto stage = [&](int buf) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      *reinterpret_cast<float4*>(&sa[buf][lrow[i] * 36 + kq]) = ra[i];
      *reinterpret_cast<float4*>(&sb[buf][lrow[i] * 36 + kq]) = rb[i];
    }
  };
  floatx16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  const int nk = (K + BK - 1) / BK;
  fetch(0);
  stage(0);
  if (nk > 1) fetch(1);
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
      float4 av[2], bv[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        av[i] = *reinterpret_cast<const float4*>(&sa[cur][(wm * 64 + i * 32 + l31) * 36 + kb * 8 + lhi * 4]);
        bv[i] = *reinterpret_cast<const float4*>(&sb[cur][(wn * 64 + i * 32 + l31) * 36 + kb * 8 + lhi * 4]);
      }
      const float* a0 = reinterpret_cast<const float*>(av);
      const float* b0 = reinterpret_cast<const float*>(bv);
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[i * 4 + t], b0[j * 4 + t], acc[i][j], 0, 0, 0);
      if (kb == 1 && kt + 1 < nk) stage(cur ^ 1);
    }
    if (kt + 2 < nk) fetch(kt + 2);
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int col = n0 + wn * 64 + j * 32 + l31;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
        if (row < M && col < N) C[(size_t)row * N + col] = acc[i][j][r];
      }
    }
}

#define CK(x)                                                                         \
  do {                                                                                \
    hipError_t e_ = (x);                                                              \
    if (e_ != hipSuccess) {                                                           \
      printf("%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_));                \
      exit(1);                                                                        \
    }                                                                                 \
  } while (0)

struct Sample {
  int m, n;
  double ref, mag;
};

template <class Launch>
static void run(const char* name, Launch launch, const float* dA, const float* dB, float* dC, int M, int N, int K,
                const std::vector<Sample>& samples, double cycles_note) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  CK(hipMemset(dC, 0, sizeof(float) * (size_t)M * N));
  for (int i = 0; i < 30; ++i) launch();                    // the clock ramps over milliseconds
  CK(hipDeviceSynchronize());
  const int reps = 40;
  CK(hipEventRecord(e0));
  for (int i = 0; i < reps; ++i) launch();
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  CK(hipGetLastError());
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  ms /= reps;
  double worst = 0.0, sum2 = 0.0;
  for (const Sample& s : samples) {
    float v;
    CK(hipMemcpy(&v, dC + (size_t)s.m * N + s.n, sizeof(float), hipMemcpyDeviceToHost));
    const double e = std::fabs((double)v - s.ref) / s.mag;
    worst = e > worst ? e : worst;
    sum2 += e * e;
  }
  printf("%-36s %8.1f us  %6.1f TFLOP/s (fp32-equivalent)   error / (sum |a||b|): max %.2e  rms %.2e   (%g matrix-pipe cycles per 32x32x16 block)\n",
         name, ms * 1e3, 2.0 * M * N * K / ms / 1e9, worst, std::sqrt(sum2 / samples.size()), cycles_note);
}

int main(int argc, char** argv) {
  const int M = argc > 1 ? atoi(argv[1]) : 58761, N = argc > 2 ? atoi(argv[2]) : 1140, K = argc > 3 ? atoi(argv[3]) : 1140;
  if (K % 4) {
    printf("K must be a multiple of 4\n");
    return 1;
  }
  const int ld = (K + BK - 1) / BK * BK + (argc > 4 ? atoi(argv[4]) : 0);   // rows padded with zeros to a whole k-tile (+4: a row stride of 4608 bytes
                                                                         // piles the 128 rows of a tile onto few L2 channels: 1350 -> 1950 us for the fp32 kernel)
  std::vector<float> hA((size_t)M * ld, 0.f), hB((size_t)N * ld, 0.f);
  uint64_t s = 0x9E3779B97F4A7C15ull;
  auto rnd = [&]() {                                        // roughly normal, all 24 mantissa bits in use
    float acc = 0.f;
    for (int i = 0; i < 4; ++i) {
      s = s * 6364136223846793005ull + 1442695040888963407ull;
      acc += (float)((s >> 40) & 0xFFFFFF) / 16777216.0f - 0.5f;
    }
    return acc * 1.7320508f;
  };
  for (int m = 0; m < M; ++m)
    for (int k = 0; k < K; ++k) hA[(size_t)m * ld + k] = rnd();
  for (int n = 0; n < N; ++n)
    for (int k = 0; k < K; ++k) hB[(size_t)n * ld + k] = rnd() * 0.05f;
  float *dA, *dB, *dC;
  CK(hipMalloc(&dA, sizeof(float) * hA.size()));
  CK(hipMalloc(&dB, sizeof(float) * hB.size()));
  CK(hipMalloc(&dC, sizeof(float) * (size_t)M * N));
  CK(hipMemcpy(dA, hA.data(), sizeof(float) * hA.size(), hipMemcpyHostToDevice));
  CK(hipMemcpy(dB, hB.data(), sizeof(float) * hB.size(), hipMemcpyHostToDevice));
  std::vector<Sample> samples;
  double fma_worst = 0.0, fma_sum2 = 0.0;
  for (int i = 0; i < 4096; ++i) {
    s = s * 6364136223846793005ull + 1442695040888963407ull;
    Sample sm;
    sm.m = (int)((s >> 33) % (uint64_t)M);
    sm.n = (int)((s >> 13) % (uint64_t)N);
    if (i < 8) {
      sm.m = i & 1 ? M - 1 - i : i;                         // corners and edges too
      sm.n = i & 2 ? N - 1 - i : i;
    }
    double ref = 0.0, mag = 0.0;
    float f = 0.f;
    for (int k = 0; k < K; ++k) {
      const float a = hA[(size_t)sm.m * ld + k], b = hB[(size_t)sm.n * ld + k];
      ref += (double)a * (double)b;
      mag += std::fabs((double)a * (double)b);
      f = std::fmaf(a, b, f);
    }
    sm.ref = ref;
    sm.mag = mag;
    samples.push_back(sm);
    const double e = std::fabs((double)f - ref) / mag;
    fma_worst = e > fma_worst ? e : fma_worst;
    fma_sum2 += e * e;
  }
  printf("C[%d, %d] = A[%d, %d] * B[%d, %d]^T, %zu sampled outputs against float64\n", M, N, M, K, N, K, samples.size());
  printf("%-36s %8s  %6s                                   error / (sum |a||b|): max %.2e  rms %.2e\n", "host fmaf chain (k ascending)", "", "",
         fma_worst, std::sqrt(fma_sum2 / samples.size()));
  const int tiles_n = (N + BN - 1) / BN, tiles = ((M + BM - 1) / BM) * tiles_n;
  auto lds_bytes = [](int bk) { return (size_t)2 * 2 * 3 * 128 * (bk * 2 + 16); };
#define SPLIT_RUN(T_, BK__, name_, cyc_)                                                                                          \
  do {                                                                                                                            \
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gemm_split<T_, BK__>), hipFuncAttributeMaxDynamicSharedMemorySize,    \
                           (int)lds_bytes(BK__)));                                                                                \
    run(name_, [&] { hipLaunchKernelGGL((k_gemm_split<T_, BK__>), dim3(tiles), dim3(256), lds_bytes(BK__), 0, dA, dB, dC, M, N, K, ld, tiles_n); }, \
        dA, dB, dC, M, N, K, samples, cyc_);                                                                                      \
  } while (0)
  run("fp32 MFMA 32x32x2 (same tiling)", [&] { hipLaunchKernelGGL(k_gemm_f32ref, dim3(tiles), dim3(256), 0, 0, dA, dB, dC, M, N, K, ld, tiles_n); }, dA, dB,
      dC, M, N, K, samples, 512);
  SPLIT_RUN(9, 32, "bf16 x 9 (all pairs), 1 wg/CU", 288);
  SPLIT_RUN(6, 32, "bf16 x 6 (pairs >= 2^-24), 1 wg/CU", 192);
  SPLIT_RUN(1, 32, "bf16 x 1 (plain bf16), 1 wg/CU", 32);
  SPLIT_RUN(9, 16, "bf16 x 9 (all pairs), 2 wg/CU", 288);
  SPLIT_RUN(6, 16, "bf16 x 6 (pairs >= 2^-24), 2 wg/CU", 192);
  SPLIT_RUN(3, 16, "bf16 x 3 (pairs >= 2^-16), 2 wg/CU", 96);
  SPLIT_RUN(1, 16, "bf16 x 1 (plain bf16), 2 wg/CU", 32);
#define PLACED_RUN(T_, name_, cyc_)                                                                                               \
  do {                                                                                                                            \
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gemm_split_placed<T_>), hipFuncAttributeMaxDynamicSharedMemorySize,   \
                           (int)lds_bytes(32)));                                                                                  \
    run(name_, [&] { hipLaunchKernelGGL((k_gemm_split_placed<T_>), dim3(tiles), dim3(256), lds_bytes(32), 0, dA, dB, dC, M, N, K, ld, tiles_n); }, \
        dA, dB, dC, M, N, K, samples, cyc_);                                                                                      \
  } while (0)
  PLACED_RUN(9, "bf16 x 9, split placed by hand", 288);
  PLACED_RUN(6, "bf16 x 6, split placed by hand", 192);
  PLACED_RUN(3, "bf16 x 3, split placed by hand", 96);
  {
    const int tiles_big = ((M + 255) / 256) * tiles_n;
    const size_t lds_big = (size_t)2 * (3 * 256 + 3 * 128) * 48;
#define BIG_RUN(T_, name_, cyc_)                                                                                                  \
  do {                                                                                                                            \
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gemm_split_big<T_>), hipFuncAttributeMaxDynamicSharedMemorySize,      \
                           (int)lds_big));                                                                                        \
    run(name_, [&] { hipLaunchKernelGGL((k_gemm_split_big<T_>), dim3(tiles_big), dim3(512), lds_big, 0, dA, dB, dC, M, N, K, ld, tiles_n); }, \
        dA, dB, dC, M, N, K, samples, cyc_);                                                                                      \
  } while (0)
    BIG_RUN(9, "bf16 x 9, placed, 256 x 128 tile", 288);
    BIG_RUN(6, "bf16 x 6, placed, 256 x 128 tile", 192);
    BIG_RUN(3, "bf16 x 3, placed, 256 x 128 tile", 96);
  }
  // operands split ahead of the product
  __bf16 *pA, *pB;
  const long long planeA = (long long)M * ld, planeB = (long long)N * ld;
  CK(hipMalloc(&pA, sizeof(__bf16) * 3 * planeA));
  CK(hipMalloc(&pB, sizeof(__bf16) * 3 * planeB));
  {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 3; ++rep) {
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL(k_presplit, dim3(4096), dim3(256), 0, 0, dA, planeA / 4, pA, planeA);
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
    }
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    hipLaunchKernelGGL(k_presplit, dim3(1024), dim3(256), 0, 0, dB, planeB / 4, pB, planeB);
    CK(hipDeviceSynchronize());
    printf("splitting A ahead of the product (%.0f MB fp32 -> %.0f MB of bf16 planes): %.1f us\n", planeA * 4e-6, planeA * 6e-6, ms * 1e3);
  }
  const size_t lds_pre = (size_t)2 * 2 * 3 * 128 * (BK * 2 + 16);
#define PRE_RUN(T_, name_, cyc_)                                                                                                  \
  do {                                                                                                                            \
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gemm_pre<T_, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_pre)); \
    run(name_, [&] { hipLaunchKernelGGL((k_gemm_pre<T_, 1>), dim3(tiles), dim3(256), lds_pre, 0, pA, planeA, pB, planeB, dC, M, N, K, ld, tiles_n); }, \
        dA, dB, dC, M, N, K, samples, cyc_);                                                                                      \
  } while (0)
  PRE_RUN(9, "pre-split planes, bf16 x 9", 288);
  PRE_RUN(6, "pre-split planes, bf16 x 6", 192);
  PRE_RUN(3, "pre-split planes, bf16 x 3", 96);
  PRE_RUN(1, "pre-split planes, bf16 x 1", 32);
  return 0;
}
