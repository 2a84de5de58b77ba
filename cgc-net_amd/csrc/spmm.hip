// Neighbour aggregation on the CSR:  out[i,:] = post[i] * sum_{k in row i} w_k * pre[col[k]] * x[col[k],:]
//
// Replaces torch.matmul(adj, x) of DenseSAGEConv at level 1 (model/network.py:114-116; narrow widths 16..64) and the
// inner product A*S of (S^T A) S (model/network.py:207; width = cluster count, "K4").  HBM-bound gather kernel:
//   * a row group of `lpr` lanes owns one row x one column tile; every lane holds VEC (=4 -> 16-byte) consecutive columns,
//     so each neighbour row is fetched as one contiguous, fully coalesced segment (1 KiB per wave for wide rows);
//   * the row's (col, weight) pairs are read ONCE by the group's lanes (coalesced) and staged in registers; they reach the
//     other lanes through wavefront shuffles -- no per-edge scalar re-reads, no atomics (the row owner writes its result);
//   * gathers are issued four at a time before the first FMA so that ~4 KiB per wave is in flight;
//   * wide rows are cut into column tiles and the block index is remapped so that one XCD (private 4 MiB L2) works on a
//     contiguous range of rows (= a few whole graphs): the ~9x re-read of neighbour rows is then served by that L2.
#include "common.hpp"

template <int VEC>
__global__ __launch_bounds__(256) void k_spmm(const int* __restrict__ rowptr, const int* __restrict__ col, const int* __restrict__ perm,
                                              const float* __restrict__ val, const float* __restrict__ pre,
                                              const float* __restrict__ post, const float* __restrict__ x, float* __restrict__ out,
                                              int n, int W, int lpr, int n_ctiles, int rows_per_block, int blocks_per_ct,
                                              int n_chunks) {
  // XCD-contiguous virtual block id (blocks are dealt round-robin to the 8 XCDs; speed only, never correctness)
  const int nb = gridDim.x, b = blockIdx.x;
  const int vb = (nb % 8 == 0) ? (b % 8) * (nb / 8) + b / 8 : b;
  const int per_chunk = blocks_per_ct * n_ctiles;
  const int chunk = vb / per_chunk;
  if (chunk >= n_chunks) return;
  const int rem = vb - chunk * per_chunk;
  const int ct = rem / blocks_per_ct, rb = rem - ct * blocks_per_ct;
  const int row0 = (chunk * blocks_per_ct + rb) * rows_per_block;
  const int row_end = min(row0 + rows_per_block, n);

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int sl = lane & (lpr - 1), sub = lane / lpr, rpw = 64 / lpr;
  const int c0 = (ct * lpr + sl) * VEC;
  const bool colok = c0 < W;

  for (int rbase = row0 + wave * rpw; rbase < row_end; rbase += 4 * rpw) {
    const int r = rbase + sub;
    const bool valid = r < row_end;
    int s = 0, e = 0;
    if (valid) { s = rowptr[r]; e = rowptr[r + 1]; }
    float acc[VEC];
#pragma unroll
    for (int v = 0; v < VEC; ++v) acc[v] = 0.f;
    for (int k0 = s; k0 < e; k0 += lpr) {
      // stage up to lpr (col, weight) pairs of this row in the group's lanes
      const int kk = k0 + sl;
      int myc = 0;
      float myw = 0.f;
      if (kk < e) {
        myc = col[kk];
        float w = val != nullptr ? val[perm != nullptr ? perm[kk] : kk] : 1.f;
        if (pre != nullptr) w *= pre[myc];
        myw = w;
      }
      const int cnt = min(lpr, e - k0);
      for (int t = 0; t < cnt; t += 4) {
        Vec<VEC> xv[4];
        float ww[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int tt = t + u;
          const int cc = __shfl(myc, tt & (lpr - 1), lpr);
          ww[u] = __shfl(myw, tt & (lpr - 1), lpr);
          if (tt < cnt && colok) {
            xv[u].load(x + (size_t)cc * W + c0);
          } else {
            ww[u] = 0.f;
#pragma unroll
            for (int v = 0; v < VEC; ++v) xv[u].v[v] = 0.f;
          }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int v = 0; v < VEC; ++v) acc[v] = fmaf(ww[u], xv[u].v[v], acc[v]);
      }
    }
    if (valid && colok) {
      const float ps = post != nullptr ? post[r] : 1.f;
      Vec<VEC> o;
#pragma unroll
      for (int v = 0; v < VEC; ++v) o.v[v] = acc[v] * ps;
      o.store(out + (size_t)r * W + c0);
    }
  }
}

extern "C" int cgc_spmm(const int* rowptr, const int* col, const int* perm, const float* val, const float* pre, const float* post,
                        const float* x, float* out, int n, int width, cgc_stream_t stream) {
  if (n <= 0 || width <= 0) return 0;
  const bool vec = (width % 4 == 0) && aligned16(x) && aligned16(out);
  const int chunks = vec ? width / 4 : width;        // per-lane column chunks in a row
  const int lpr = pick_lpr(chunks);
  const int n_ctiles = ceil_div(chunks, lpr);         // > 1 only for rows wider than 64 chunks (lpr == 64)
  const int rpw = 64 / lpr;
  // narrow rows: 4 passes of (4 waves x rpw rows); wide rows: 4 rows per wave
  const int rows_per_block = 4 * rpw * 4;
  // rows handled by one XCD-contiguous "chunk": about one graph (2048 rows) so that its neighbour rows stay in that L2
  const int chunk_rows = 2048;
  const int blocks_per_ct = chunk_rows / rows_per_block;
  const int n_chunks = ceil_div(n, chunk_rows);
  int nb = n_chunks * blocks_per_ct * n_ctiles;
  nb = ceil_div(nb, 8) * 8;
  dim3 grid(nb), block(CGC_BLOCK);
  if (vec)
    hipLaunchKernelGGL(k_spmm<4>, grid, block, 0, as_stream(stream), rowptr, col, perm, val, pre, post, x, out, n, width, lpr,
                       n_ctiles, rows_per_block, blocks_per_ct, n_chunks);
  else
    hipLaunchKernelGGL(k_spmm<1>, grid, block, 0, as_stream(stream), rowptr, col, perm, val, pre, post, x, out, n, width, lpr,
                       n_ctiles, rows_per_block, blocks_per_ct, n_chunks);
  CGC_RETURN_IF_LAUNCH_FAILED();
  return 0;
}
