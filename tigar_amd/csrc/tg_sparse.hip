// CSR kernels on the application side of the extraction path:
//   * explicit transpose M^T (reference FORM_MT option, tIGAr/common.py:84) -- deterministic
//   * SpMV  y = A x      : K p in the Krylov solve, u = M U prolongation (common.py:1259),
//                          M^T b through the explicit transpose (common.py:97-109)
//   * MatZeroRowsColumns (common.py:1200)
//
// SpMV design (HBM-stream-bound, 12 B/nnz): "CSR-stream".  Rows are packed into row
// blocks of at most TG_SPMV_CAP non-zeros; a workgroup streams its block's val[]/col[]
// with 16-byte loads that ignore row boundaries (fully coalesced; every load of a block is issued
// before its first dependent gather), multiplies by gathered
// x (L2-resident), parks the products in LDS and then reduces row segments out of LDS.
// No atomics: results are bit-reproducible.  Row blocks are laid out so that each XCD's
// L2 sees one contiguous range of rows (x-gather locality).
#include "tg_common.h"
#include <algorithm>

typedef double tg_d2 __attribute__((ext_vector_type(2)));
typedef int tg_i4 __attribute__((ext_vector_type(4)));

#define TG_SPMV_CAP 4096  // products (doubles) staged in LDS per workgroup

// ----------------------------------------------------------------------------------------
// plan
// ----------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_max_row_nnz(const int64_t *rowptr, int64_t nrows, int *out) {
  __shared__ int lds[4];
  int m = 0;
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < nrows; i += stride) {
    const int64_t l = rowptr[i + 1] - rowptr[i];
    m = max(m, (int)min(l, (int64_t)0x7fffffff));
  }
  for (int o = 32; o > 0; o >>= 1) m = max(m, __shfl_down(m, o, 64));
  if ((threadIdx.x & 63) == 0) lds[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) atomicMax(out, max(max(lds[0], lds[1]), max(lds[2], lds[3])));
}

// rowblocks[b] = first row r with rowptr[r] >= b * quantum   (b = 0..nblocks), last = nrows
__global__ void k_build_rowblocks(const int64_t *rowptr, int64_t nrows, int64_t quantum, int64_t nblocks,
                                  int32_t *rowblocks) {
  int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (b > nblocks) return;
  if (b == nblocks) {
    rowblocks[b] = (int32_t)nrows;
    return;
  }
  const int64_t target = b * quantum;
  int64_t lo = 0, hi = nrows;
  while (lo < hi) {
    const int64_t mid = (lo + hi) >> 1;
    if (rowptr[mid] < target)
      lo = mid + 1;
    else
      hi = mid;
  }
  rowblocks[b] = (int32_t)lo;
}

int tg_spmv_plan(tg_csr_s *a) {
  if (a->spmv_mode) return 0;
  TG_REQUIRE_CANONICAL(a);
  TG_REQUIRE(a->nrows < 0x7fffffffll, "too many rows for the SpMV plan");
  int *dmax = (int *)g_tg.scratch;
  TG_CHECK_HIP(hipMemsetAsync(dmax, 0, sizeof(int), g_tg.stream));
  if (a->nrows > 0) {
    hipLaunchKernelGGL(k_max_row_nnz, dim3(tg_grid_1d(a->nrows, 256)), dim3(256), 0, g_tg.stream, a->rowptr, a->nrows,
                       dmax);
    TG_LAUNCH_CHECK();
  }
  int hmax = 0;
  TG_CHECK_HIP(hipMemcpyAsync(&hmax, dmax, sizeof(int), hipMemcpyDeviceToHost, g_tg.stream));
  TG_CHECK_HIP(hipStreamSynchronize(g_tg.stream));
  a->max_row_nnz = hmax;
  const char *ecap = getenv("TIGAR_SPMV_CAP");
  int cap = ecap ? atoi(ecap) : TG_SPMV_CAP;
  if (cap != 1024 && cap != 2048 && cap != 4096 && cap != 8192) cap = TG_SPMV_CAP;
  while (cap < 8192 && hmax > cap / 2) cap *= 2;
  a->spmv_cap = cap;
  if (hmax <= cap / 2) {
    // a block holds fewer than quantum + hmax entries and the kernel starts at the entry index
    // rounded DOWN to a multiple of 4 (aligned 16-byte loads): keep 3 entries of slack so that
    // [n0 & ~3, n1) never exceeds the cap (without it a nearly full block dropped its last
    // entries: 159 of 17 M rows of M wrong at 128^3 p=2, found by the full-size unity test)
    const int64_t quantum = cap - hmax - 3;
    a->nblocks = a->nnz / quantum + 1;
    TG_TRY(tg_dmalloc(&a->rowblocks, a->nblocks + 1));
    hipLaunchKernelGGL(k_build_rowblocks, dim3((unsigned)tg_cdiv(a->nblocks + 1, 256)), dim3(256), 0, g_tg.stream,
                       a->rowptr, a->nrows, quantum, a->nblocks, a->rowblocks);
    TG_LAUNCH_CHECK();
    a->spmv_mode = 1;
  } else {
    a->spmv_mode = 2;  // wave per row
  }
  return 0;
}

// ----------------------------------------------------------------------------------------
// kernels
// ----------------------------------------------------------------------------------------
// DOT: additionally accumulates sum_r y[r] * dvec[r] into dot_partial[blockIdx.x]
template <bool DOT, int CAP, bool NT>
__global__ void __launch_bounds__(256)
    k_spmv_stream(const int64_t *__restrict__ rowptr, const int32_t *__restrict__ col, const double *__restrict__ val,
                  const double *__restrict__ x, double *__restrict__ y, const int32_t *__restrict__ rowblocks,
                  int64_t nblocks, const double *__restrict__ dvec, double *__restrict__ dot_partial) {
  __shared__ double prod[CAP];
  __shared__ double red4[4];
  const int tid = threadIdx.x;
  const int64_t L = tg_xcd_block(blockIdx.x, nblocks);
  double dsum = 0.0;
  if (L < nblocks) {
    const int64_t r0 = rowblocks[L], r1 = rowblocks[L + 1];
    const int64_t n0 = r1 > r0 ? rowptr[r0] : 0, n1 = r1 > r0 ? rowptr[r1] : 0;
    if (r1 > r0 && n1 <= n0) {
      // a block of EMPTY rows (an empty matrix: one block): nothing to stream -- the clamped loads below would read entry
      // n1 - 1 < n0, which for n0 = 0 lies in front of the arrays, and gather x at whatever column they find there
      for (int64_t r = r0 + tid; r < r1; r += 256) y[r] = 0.0;
    } else if (r1 > r0) {
      const int64_t q0 = n0 & ~3ll;
      // All streaming loads of the block are issued before the first dependent gather, and all
      // gathers before the first LDS store: two memory latencies per block instead of two per
      // 1024-entry slice.  (clamped addresses keep out-of-range slices legal and branch-free)
      constexpr int SL = CAP / 1024;
      tg_d2 v01[SL], v23[SL];
      tg_i4 cc[SL];
      const int64_t tlast = (n1 - 1) & ~3ll;
#pragma unroll
      for (int k = 0; k < SL; k++) {
        int64_t t = q0 + 4 * tid + 1024 * k;
        t = t > tlast ? tlast : t;
        v01[k] = NT ? __builtin_nontemporal_load(reinterpret_cast<const tg_d2 *>(val + t))
                    : *reinterpret_cast<const tg_d2 *>(val + t);
        v23[k] = NT ? __builtin_nontemporal_load(reinterpret_cast<const tg_d2 *>(val + t + 2))
                    : *reinterpret_cast<const tg_d2 *>(val + t + 2);
        cc[k] = NT ? __builtin_nontemporal_load(reinterpret_cast<const tg_i4 *>(col + t))
                   : *reinterpret_cast<const tg_i4 *>(col + t);
      }
      double xv[SL][4];
#pragma unroll
      for (int k = 0; k < SL; k++) {
        int64_t t = q0 + 4 * tid + 1024 * k;
        t = t > tlast ? tlast : t;
        // entry t is always a real entry of the matrix; entries past n1 may lie in the padding
        // of the arrays, so their (garbage) column is replaced before it is dereferenced
        const int c0 = cc[k].x;
        xv[k][0] = x[c0];
        xv[k][1] = x[(t + 1 < n1) ? cc[k].y : c0];
        xv[k][2] = x[(t + 2 < n1) ? cc[k].z : c0];
        xv[k][3] = x[(t + 3 < n1) ? cc[k].w : c0];
      }
#pragma unroll
      for (int k = 0; k < SL; k++) {
        const int64_t t = q0 + 4 * tid + 1024 * k;
        if (t <= tlast) {
          const int64_t o = t - n0;
          if (t >= n0) prod[o] = v01[k].x * xv[k][0];
          if (t + 1 >= n0 && t + 1 < n1) prod[o + 1] = v01[k].y * xv[k][1];
          if (t + 2 >= n0 && t + 2 < n1) prod[o + 2] = v23[k].x * xv[k][2];
          if (t + 3 >= n0 && t + 3 < n1) prod[o + 3] = v23[k].y * xv[k][3];
        }
      }
      __syncthreads();
      const int nr = (int)(r1 - r0);
      // lanes per row: largest power of two <= 256/nr, clipped to [1,64]
      int G = 1;
      while (G < 64 && G * 2 * nr <= 256) G <<= 1;
      const int rows_per_pass = 256 / G;
      const int sub = tid & (G - 1);
      const int rgrp = tid / G;
      for (int base = 0; base < nr; base += rows_per_pass) {
        const int rr = base + rgrp;
        double s = 0.0;
        if (rr < nr) {
          const int64_t a = rowptr[r0 + rr] - n0, b = rowptr[r0 + rr + 1] - n0;
          for (int64_t q = a + sub; q < b; q += G) s += prod[q];
        }
        for (int o = G >> 1; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
        if (rr < nr && sub == 0) {
          y[r0 + rr] = s;
          if (DOT) dsum += s * dvec[r0 + rr];
        }
      }
    }
  }
  if (DOT) {
    dsum = tg_block_sum256(dsum, red4);
    if (tid == 0) dot_partial[blockIdx.x] = dsum;
  }
}

// Default stream kernel: lane-major entry mapping (lane i of pass k owns entry n0 + 256 k + i).  The
// gather x[col] of a wave then covers 64 CONSECUTIVE entries -- for K = M^T A M about 9 runs of 2p+1
// consecutive columns, i.e. a dozen cache lines -- instead of 64 entries that are 4 apart (the quad
// mapping of k_spmv_stream with its 16-byte loads: ~37 runs per gather instruction).  A third of
// the SpMV time is the gather, and this mapping is 3.5-5 % faster at 128^3..256^3 p=3
// although it needs 32 instead of 12 load instructions per 1024 entries.  Same products in the same
// LDS slots, same reduction: y is bit-identical to k_spmv_stream.
template <int CAP>
__global__ void __launch_bounds__(256)
    k_spmv_lane(const int64_t *__restrict__ rowptr, const int32_t *__restrict__ col, const double *__restrict__ val,
                const double *__restrict__ x, double *__restrict__ y, const int32_t *__restrict__ rowblocks,
                int64_t nblocks) {
  __shared__ double prod[CAP];
  const int tid = threadIdx.x;
  const int64_t L = tg_xcd_block(blockIdx.x, nblocks);
  if (L >= nblocks) return;
  const int64_t r0 = rowblocks[L], r1 = rowblocks[L + 1];
  if (r1 <= r0) return;
  const int64_t n0 = rowptr[r0], n1 = rowptr[r1];
  if (n1 <= n0) {                                  // a block of empty rows: see k_spmv_stream
    for (int64_t r = r0 + tid; r < r1; r += 256) y[r] = 0.0;
    return;
  }
  constexpr int SL = CAP / 256;
  double v[SL];
  int32_t c[SL];
#pragma unroll
  for (int k = 0; k < SL; k++) {
    int64_t t = n0 + tid + 256 * k;
    t = t < n1 ? t : n1 - 1;
    v[k] = val[t];
    c[k] = col[t];
  }
  double xv[SL];
#pragma unroll
  for (int k = 0; k < SL; k++) xv[k] = x[c[k]];
#pragma unroll
  for (int k = 0; k < SL; k++) {
    const int64_t t = n0 + tid + 256 * k;
    if (t < n1) prod[t - n0] = v[k] * xv[k];
  }
  __syncthreads();
  const int nr = (int)(r1 - r0);
  int G = 1;
  while (G < 64 && G * 2 * nr <= 256) G <<= 1;
  const int rows_per_pass = 256 / G;
  const int sub = tid & (G - 1);
  const int rgrp = tid / G;
  for (int base = 0; base < nr; base += rows_per_pass) {
    const int rr = base + rgrp;
    double s = 0.0;
    if (rr < nr) {
      const int64_t a = rowptr[r0 + rr] - n0, b = rowptr[r0 + rr + 1] - n0;
      for (int64_t q = a + sub; q < b; q += G) s += prod[q];
    }
    for (int o = G >> 1; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if (rr < nr && sub == 0) y[r0 + rr] = s;
  }
}

// generic fallback: one wave per row (rows longer than TG_SPMV_CAP/2)
template <bool DOT>
__global__ void __launch_bounds__(256)
    k_spmv_vector(const int64_t *__restrict__ rowptr, const int32_t *__restrict__ col, const double *__restrict__ val,
                  const double *__restrict__ x, double *__restrict__ y, int64_t nrows, const double *__restrict__ dvec,
                  double *__restrict__ dot_partial) {
  __shared__ double red4[4];
  const int lane = threadIdx.x & 63;
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  double dsum = 0.0;
  for (int64_t r = wave; r < nrows; r += nwaves) {
    double s = 0.0;
    for (int64_t q = rowptr[r] + lane; q < rowptr[r + 1]; q += 64) s += val[q] * x[col[q]];
    s = tg_wave_sum(s);
    if (lane == 0) {
      y[r] = s;
      if (DOT) dsum += s * dvec[r];
    }
  }
  if (DOT) {
    dsum = tg_block_sum256(dsum, red4);
    if (threadIdx.x == 0) dot_partial[blockIdx.x] = dsum;
  }
}

// y = A x with x addressed by the matrix's (global) column indices: x_shifted[col], valid for
// cmin <= col <= cmax.
int tg_spmv_raw(tg_csr_s *a, const double *x_shifted, int64_t cmin, int64_t cmax, double *y) {
  double *dot_partial = nullptr;
  const double *dvec = nullptr;
  if (a->sell_state == 1 && a->sell) return a->nrows ? tg_sell_spmv(a, x_shifted, cmin, cmax, y) : 0;
  TG_TRY(tg_spmv_plan(a));
  if (a->nrows == 0) return 0;
  if (a->spmv_mode == 1) {
    const unsigned grid = (unsigned)(((a->nblocks + 7) / 8) * 8);  // tg_xcd_block needs a multiple of 8
    static int nt = -1;
    if (nt < 0) nt = getenv("TIGAR_SPMV_NT") ? atoi(getenv("TIGAR_SPMV_NT")) : 0;
    TG_REQUIRE(!dot_partial, "fused dot partials are not used any more");
    static int lane_variant = getenv("TIGAR_SPMV_LANE") ? atoi(getenv("TIGAR_SPMV_LANE")) : 1;   // 0: quad mapping
    if (lane_variant) {
#define TG_SPMV_LANE(CAP)                                                                                          \
  hipLaunchKernelGGL((k_spmv_lane<CAP>), dim3(grid), dim3(256), 0, g_tg.stream, a->rowptr, a->col, a->val, x_shifted, \
                     y, a->rowblocks, a->nblocks)
      if (a->spmv_cap == 1024) TG_SPMV_LANE(1024);
      else if (a->spmv_cap == 2048) TG_SPMV_LANE(2048);
      else if (a->spmv_cap == 4096) TG_SPMV_LANE(4096);
      else TG_SPMV_LANE(8192);
#undef TG_SPMV_LANE
      TG_LAUNCH_CHECK();
      return 0;
    }
#define TG_SPMV_LAUNCH(CAP, NTF)                                                                                  \
  hipLaunchKernelGGL((k_spmv_stream<false, CAP, NTF>), dim3(grid), dim3(256), 0, g_tg.stream, a->rowptr, a->col, \
                     a->val, x_shifted, y, a->rowblocks, a->nblocks, (const double *)nullptr, (double *)nullptr)
    if (nt) {
      if (a->spmv_cap == 1024) TG_SPMV_LAUNCH(1024, true);
      else if (a->spmv_cap == 2048) TG_SPMV_LAUNCH(2048, true);
      else if (a->spmv_cap == 4096) TG_SPMV_LAUNCH(4096, true);
      else TG_SPMV_LAUNCH(8192, true);
    } else {
      if (a->spmv_cap == 1024) TG_SPMV_LAUNCH(1024, false);
      else if (a->spmv_cap == 2048) TG_SPMV_LAUNCH(2048, false);
      else if (a->spmv_cap == 4096) TG_SPMV_LAUNCH(4096, false);
      else TG_SPMV_LAUNCH(8192, false);
    }
#undef TG_SPMV_LAUNCH
  } else {
    const unsigned grid = (unsigned)std::min<int64_t>(tg_cdiv(a->nrows, 4), (int64_t)g_tg.num_cu * 8);
    if (dot_partial)
      hipLaunchKernelGGL((k_spmv_vector<true>), dim3(grid), dim3(256), 0, g_tg.stream, a->rowptr, a->col, a->val,
                         x_shifted, y, a->nrows, dvec, dot_partial);
    else
      hipLaunchKernelGGL((k_spmv_vector<false>), dim3(grid), dim3(256), 0, g_tg.stream, a->rowptr, a->col, a->val,
                         x_shifted, y, a->nrows, (const double *)nullptr, (double *)nullptr);
  }
  TG_LAUNCH_CHECK();
  return 0;
}

// number of dot partials tg_spmv_raw writes for this matrix
int64_t tg_spmv_num_partials(tg_csr_s *a) {
  if (a->spmv_mode == 1) return ((a->nblocks + 7) / 8) * 8;
  return std::min<int64_t>(tg_cdiv(a->nrows, 4), (int64_t)g_tg.num_cu * 8);
}

extern "C" int tg_spmv(tg_csr_t a, tg_vec_t x, tg_vec_t y) {
  TG_REQUIRE_INIT();
  TG_REQUIRE(a && x && y, "null argument to tg_spmv");
  TG_REQUIRE(x->n == a->ncols, "tg_spmv: x has %lld entries, matrix has %lld columns", (long long)x->n,
             (long long)a->ncols);
  TG_REQUIRE(y->n == a->nrows, "tg_spmv: y has %lld entries, matrix has %lld rows", (long long)y->n,
             (long long)a->nrows);
  return tg_spmv_raw(a, x->d, 0, a->ncols - 1, y->d);
}

extern "C" int tg_spmv_offset(tg_csr_t a, tg_vec_t x, int64_t x_col0, tg_vec_t y) {
  TG_REQUIRE_INIT();
  TG_REQUIRE(a && x && y, "null argument to tg_spmv_offset");
  TG_REQUIRE(x_col0 >= 0 && x_col0 + x->n <= a->ncols, "tg_spmv_offset: x range outside the matrix columns");
  TG_REQUIRE(y->n == a->nrows, "tg_spmv_offset: y has %lld entries, matrix has %lld rows", (long long)y->n,
             (long long)a->nrows);
  return tg_spmv_raw(a, x->d - x_col0, x_col0, x_col0 + x->n - 1, y->d);
}

extern "C" int tg_spmv_t(tg_csr_t mt, tg_vec_t b, tg_vec_t y) { return tg_spmv(mt, b, y); }

extern "C" int tg_spmm_host(tg_csr_t a, const double *X, int k, double *Y) {
  TG_REQUIRE_INIT();
  TG_REQUIRE(a && X && Y && k >= 1, "bad arguments to tg_spmm_host");
  tg_vec_t x = nullptr, y = nullptr;
  TG_TRY(tg_vec_create(a->ncols, &x));
  int rc = tg_vec_create(a->nrows, &y);
  for (int j = 0; j < k && !rc; j++) {
    rc = tg_vec_upload(x, X + (int64_t)j * a->ncols, a->ncols);
    if (!rc) rc = tg_spmv(a, x, y);
    if (!rc) rc = tg_vec_download(y, Y + (int64_t)j * a->nrows, a->nrows);
  }
  tg_vec_destroy(x);
  tg_vec_destroy(y);
  return rc;
}

// ----------------------------------------------------------------------------------------
// transpose
// ----------------------------------------------------------------------------------------
__global__ void k_tr_count(const int32_t *__restrict__ col, int64_t nnz, unsigned long long *cnt) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < nnz; i += stride) atomicAdd(&cnt[col[i]], 1ull);
}

// one wave per source row; lanes stride over its entries
__global__ void __launch_bounds__(256)
    k_tr_fill(const int64_t *__restrict__ rowptr, const int32_t *__restrict__ col, const double *__restrict__ val,
              int64_t nrows, int64_t row_base, unsigned long long *cursor, int32_t *__restrict__ colT,
              double *__restrict__ valT) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t r = wave; r < nrows; r += nwaves) {
    for (int64_t q = rowptr[r] + lane; q < rowptr[r + 1]; q += 64) {
      const unsigned long long pos = atomicAdd(&cursor[col[q]], 1ull);
      colT[pos] = (int32_t)(r + row_base);
      valT[pos] = val[q];
    }
  }
}

// sorts each row segment by column index: block per row, bitonic network in LDS
template <int CAPACITY>
__global__ void __launch_bounds__(256)
    k_sort_rows(const int64_t *__restrict__ rowptr, int32_t *__restrict__ col, double *__restrict__ val,
                int64_t nrows, int min_len, int max_len) {
  __shared__ int32_t skey[CAPACITY];
  __shared__ double sval[CAPACITY];
  for (int64_t r = blockIdx.x; r < nrows; r += gridDim.x) {
    const int64_t a = rowptr[r];
    const int len = (int)(rowptr[r + 1] - a);
    if (len < min_len || len > max_len || len < 2) continue;  // uniform per block
    int n2 = 1;
    while (n2 < len) n2 <<= 1;
    for (int i = threadIdx.x; i < n2; i += 256) {
      skey[i] = (i < len) ? col[a + i] : 0x7fffffff;
      sval[i] = (i < len) ? val[a + i] : 0.0;
    }
    __syncthreads();
    for (int k = 2; k <= n2; k <<= 1) {
      for (int j = k >> 1; j > 0; j >>= 1) {
        for (int i = threadIdx.x; i < n2; i += 256) {
          const int ixj = i ^ j;
          if (ixj > i) {
            const bool up = (i & k) == 0;
            const int32_t ki = skey[i], kj = skey[ixj];
            if ((ki > kj) == up) {
              skey[i] = kj;
              skey[ixj] = ki;
              const double t = sval[i];
              sval[i] = sval[ixj];
              sval[ixj] = t;
            }
          }
        }
        __syncthreads();
      }
    }
    for (int i = threadIdx.x; i < len; i += 256) {
      col[a + i] = skey[i];
      val[a + i] = sval[i];
    }
    __syncthreads();
  }
}

// sorts the rows of `m` by column (values follow); used after atomically-ordered fills
int tg_csr_sort_rows(tg_csr_s *m) {
  if (m->nrows == 0 || m->nnz == 0) return 0;
  int *dmax = (int *)g_tg.scratch;
  TG_CHECK_HIP(hipMemsetAsync(dmax, 0, sizeof(int), g_tg.stream));
  hipLaunchKernelGGL(k_max_row_nnz, dim3(tg_grid_1d(m->nrows, 256)), dim3(256), 0, g_tg.stream, m->rowptr, m->nrows,
                     dmax);
  int hmax = 0;
  TG_CHECK_HIP(hipMemcpyAsync(&hmax, dmax, sizeof(int), hipMemcpyDeviceToHost, g_tg.stream));
  TG_CHECK_HIP(hipStreamSynchronize(g_tg.stream));
  const unsigned grid = (unsigned)std::min<int64_t>(m->nrows, (int64_t)g_tg.num_cu * 64);
  hipLaunchKernelGGL((k_sort_rows<512>), dim3(grid), dim3(256), 0, g_tg.stream, m->rowptr, m->col, m->val, m->nrows, 2,
                     512);
  if (hmax > 512)
    hipLaunchKernelGGL((k_sort_rows<2048>), dim3(grid), dim3(256), 0, g_tg.stream, m->rowptr, m->col, m->val, m->nrows,
                       513, 2048);
  if (hmax > 2048)
    hipLaunchKernelGGL((k_sort_rows<8192>), dim3(grid), dim3(256), 0, g_tg.stream, m->rowptr, m->col, m->val, m->nrows,
                       2049, 8192);
  TG_LAUNCH_CHECK();
  if (hmax > 8192) {
    // rare: very long rows -> host fallback
    std::vector<int64_t> rp((size_t)m->nrows + 1);
    std::vector<int32_t> c((size_t)m->nnz);
    std::vector<double> v((size_t)m->nnz);
    TG_TRY(tg_csr_download(m, rp.data(), c.data(), v.data()));
    std::vector<std::pair<int32_t, double>> tmp;
    for (int64_t r = 0; r < m->nrows; r++) {
      const int64_t len = rp[r + 1] - rp[r];
      if (len <= 8192) continue;
      tmp.resize((size_t)len);
      for (int64_t i = 0; i < len; i++) tmp[i] = {c[rp[r] + i], v[rp[r] + i]};
      std::sort(tmp.begin(), tmp.end(), [](auto &x, auto &y) { return x.first < y.first; });
      for (int64_t i = 0; i < len; i++) {
        c[rp[r] + i] = tmp[i].first;
        v[rp[r] + i] = tmp[i].second;
      }
    }
    TG_CHECK_HIP(hipMemcpy(m->col, c.data(), (size_t)m->nnz * sizeof(int32_t), hipMemcpyHostToDevice));
    TG_CHECK_HIP(hipMemcpy(m->val, v.data(), (size_t)m->nnz * sizeof(double), hipMemcpyHostToDevice));
  }
  return 0;
}

// transpose of a row block whose local rows start at global row `row_base`; the result has
// m->ncols rows and its column indices are global row numbers of the source.
int tg_csr_transpose_block(tg_csr_s *m, int64_t row_base, int64_t out_ncols, tg_csr_s **out) {
  TG_REQUIRE_CANONICAL(m);
  tg_csr_s *t = nullptr;
  TG_TRY(tg_csr_alloc(m->ncols, out_ncols, m->nnz, &t));
  TG_CHECK_HIP(hipMemsetAsync(t->rowptr, 0, (size_t)(m->ncols + 1) * sizeof(int64_t), g_tg.stream));
  if (m->nnz > 0) {
    hipLaunchKernelGGL(k_tr_count, dim3(tg_grid_1d(m->nnz, 256)), dim3(256), 0, g_tg.stream, m->col, m->nnz,
                       (unsigned long long *)t->rowptr);
    TG_LAUNCH_CHECK();
  }
  int64_t total = 0;
  if (tg_exclusive_scan_i64(t->rowptr, m->ncols, &total) || total != m->nnz) {
    if (total != m->nnz) tg_set_error("transpose: nnz mismatch after scan (%lld vs %lld)", (long long)total,
                                      (long long)m->nnz);
    tg_csr_destroy(t);
    return 1;
  }
  if (m->nnz > 0) {
    unsigned long long *cursor = nullptr;
    TG_TRY(tg_dmalloc(&cursor, m->ncols + 1));
    TG_CHECK_HIP(hipMemcpyAsync(cursor, t->rowptr, (size_t)(m->ncols + 1) * sizeof(int64_t), hipMemcpyDeviceToDevice,
                                g_tg.stream));
    const unsigned grid = (unsigned)std::min<int64_t>(tg_cdiv(m->nrows, 4), (int64_t)g_tg.num_cu * 16);
    hipLaunchKernelGGL(k_tr_fill, dim3(grid), dim3(256), 0, g_tg.stream, m->rowptr, m->col, m->val, m->nrows, row_base,
                       cursor, t->col, t->val);
    TG_LAUNCH_CHECK();
    int rc = tg_csr_sort_rows(t);
    hipStreamSynchronize(g_tg.stream);
    tg_dfree(cursor);
    if (rc) {
      tg_csr_destroy(t);
      return rc;
    }
  }
  TG_CHECK_HIP(hipStreamSynchronize(g_tg.stream));
  *out = t;
  return 0;
}

// ----------------------------------------------------------------------------------------
// field blocks of a mixed-space matrix (rows and columns field-major): cut one out, put nF x nF of them together
// ----------------------------------------------------------------------------------------
__global__ void k_add_i64(int64_t *__restrict__ p, int64_t n, int64_t add) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) p[i] += add;
}

// first position q in [a, e) with col[q] >= c (columns ascending)
__device__ __forceinline__ int64_t tg_lower_bound_col(const int32_t *__restrict__ col, int64_t a, int64_t e, int64_t c) {
  while (a < e) {
    const int64_t m = (a + e) >> 1;
    if ((int64_t)col[m] < c) a = m + 1;
    else e = m;
  }
  return a;
}

__global__ void k_block_count(const int64_t *__restrict__ rowptr, const int32_t *__restrict__ col, int64_t r0, int64_t n,
                              int64_t c0, int64_t c1, int64_t *__restrict__ out_len, int64_t *__restrict__ first) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    const int64_t a = rowptr[r0 + i], e = rowptr[r0 + i + 1];
    const int64_t lo = tg_lower_bound_col(col, a, e, c0), hi = tg_lower_bound_col(col, lo, e, c1);
    out_len[i] = hi - lo;
    first[i] = lo;
  }
}

// one wave per row: copies the row's segment, columns shifted
__global__ void __launch_bounds__(256)
    k_block_fill(const int32_t *__restrict__ col, const double *__restrict__ val, const int64_t *__restrict__ first,
                 const int64_t *__restrict__ orowptr, int64_t n, int32_t cshift, int32_t *__restrict__ ocol,
                 double *__restrict__ oval) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t r = wave; r < n; r += nwaves) {
    const int64_t o = orowptr[r], len = orowptr[r + 1] - o, src = first[r];
    for (int64_t q = lane; q < len; q += 64) {
      ocol[o + q] = col[src + q] + cshift;
      oval[o + q] = val[src + q];
    }
  }
}

// rows [r0, r1) of a, entries with c0 <= column < c1, columns renumbered from 0: the (i, j) field block of a matrix
// on a mixed space whose dofs are numbered field after field
extern "C" int tg_csr_block(tg_csr_t a, int64_t r0, int64_t r1, int64_t c0, int64_t c1, tg_csr_t *out) {
  TG_REQUIRE_INIT();
  TG_REQUIRE(a && out && r0 >= 0 && r1 >= r0 && r1 <= a->nrows && c0 >= 0 && c1 >= c0 && c1 <= a->ncols,
             "bad arguments to tg_csr_block");
  TG_REQUIRE_CANONICAL(a);
  const int64_t n = r1 - r0;
  int64_t *len = nullptr, *first = nullptr;
  int rc = tg_dmalloc(&len, n + 1) || tg_dmalloc(&first, std::max<int64_t>(n, 1));
  tg_csr_s *m = nullptr;
  int64_t total = 0;
  if (!rc && n > 0) {
    hipLaunchKernelGGL(k_block_count, dim3(tg_grid_1d(n, 256)), dim3(256), 0, g_tg.stream, a->rowptr, a->col, r0, n, c0, c1,
                       len, first);
    if (hipGetLastError() != hipSuccess) rc = 1;
  }
  if (!rc) rc = tg_exclusive_scan_i64(len, n, &total);
  if (!rc) rc = tg_csr_alloc(n, c1 - c0, total, &m);
  if (!rc) {
    if (hipMemcpyAsync(m->rowptr, len, (size_t)(n + 1) * sizeof(int64_t), hipMemcpyDeviceToDevice, g_tg.stream) != hipSuccess)
      rc = 1;
    if (!rc && total > 0) {
      const unsigned grid = (unsigned)std::min<int64_t>(tg_cdiv(n, 4), (int64_t)g_tg.num_cu * 16);
      hipLaunchKernelGGL(k_block_fill, dim3(grid), dim3(256), 0, g_tg.stream, a->col, a->val, first, m->rowptr, n,
                         (int32_t)(-c0), m->col, m->val);
      if (hipGetLastError() != hipSuccess) rc = 1;
    }
  }
  if (hipStreamSynchronize(g_tg.stream) != hipSuccess) rc = 1;
  tg_dfree(len);
  tg_dfree(first);
  if (rc) {
    if (m) tg_csr_destroy(m);
    tg_set_error("tg_csr_block failed");
    return 1;
  }
  *out = m;
  return 0;
}

// ---- a copy without the columns whose mask byte is 0 (shape kept): the operand of a product split by columns
__global__ void __launch_bounds__(256)
    k_select_count(const int64_t *__restrict__ rowptr, const int32_t *__restrict__ col, const uint8_t *__restrict__ keep, int64_t n,
                   int64_t *__restrict__ out_len) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t r = wave; r < n; r += nwaves) {
    const int64_t a = rowptr[r], e = rowptr[r + 1];
    int cnt = 0;
    for (int64_t q = a + lane; q < e; q += 64) cnt += keep[col[q]] ? 1 : 0;
    for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o, 64);
    if (lane == 0) out_len[r] = cnt;
  }
}

__global__ void __launch_bounds__(256)
    k_select_fill(const int64_t *__restrict__ rowptr, const int32_t *__restrict__ col, const double *__restrict__ val,
                  const uint8_t *__restrict__ keep, const int64_t *__restrict__ orowptr, int64_t n, int32_t *__restrict__ ocol,
                  double *__restrict__ oval) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t r = wave; r < n; r += nwaves) {
    const int64_t a = rowptr[r], e = rowptr[r + 1];
    int64_t o = orowptr[r];
    for (int64_t q0 = a; q0 < e; q0 += 64) {          // (uniform trip count: the ballot sees every lane)
      const int64_t q = q0 + lane;
      const bool k = q < e && keep[col[q]] != 0;
      const unsigned long long b = __ballot(k);
      if (k) {
        const int pos = __popcll(b & ((1ull << lane) - 1ull));
        ocol[o + pos] = col[q];
        oval[o + pos] = val[q];
      }
      o += __popcll(b);
    }
  }
}

extern "C" int tg_csr_select_columns(tg_csr_t a, const uint8_t *keep_host, tg_csr_t *out) {
  TG_REQUIRE_INIT();
  TG_REQUIRE(a && keep_host && out, "bad arguments to tg_csr_select_columns");
  TG_REQUIRE_CANONICAL(a);
  const int64_t n = a->nrows;
  int64_t *len = nullptr;
  uint8_t *keep = nullptr;
  int rc = tg_dmalloc(&len, n + 1) || tg_dmalloc(&keep, std::max<int64_t>(a->ncols, 1));
  tg_csr_s *m = nullptr;
  int64_t total = 0;
  if (!rc && a->ncols > 0 &&
      hipMemcpyAsync(keep, keep_host, (size_t)a->ncols, hipMemcpyHostToDevice, g_tg.stream) != hipSuccess)
    rc = 1;
  const unsigned grid = (unsigned)std::max<int64_t>(1, std::min<int64_t>(tg_cdiv(n, 4), (int64_t)g_tg.num_cu * 16));
  if (!rc && n > 0) {
    hipLaunchKernelGGL(k_select_count, dim3(grid), dim3(256), 0, g_tg.stream, a->rowptr, a->col, keep, n, len);
    if (hipGetLastError() != hipSuccess) rc = 1;
  }
  if (!rc) rc = tg_exclusive_scan_i64(len, n, &total);
  if (!rc) rc = tg_csr_alloc(n, a->ncols, total, &m);
  if (!rc) {
    if (hipMemcpyAsync(m->rowptr, len, (size_t)(n + 1) * sizeof(int64_t), hipMemcpyDeviceToDevice, g_tg.stream) != hipSuccess)
      rc = 1;
    if (!rc && total > 0) {
      hipLaunchKernelGGL(k_select_fill, dim3(grid), dim3(256), 0, g_tg.stream, a->rowptr, a->col, a->val, keep, m->rowptr, n,
                         m->col, m->val);
      if (hipGetLastError() != hipSuccess) rc = 1;
    }
  }
  if (hipStreamSynchronize(g_tg.stream) != hipSuccess) rc = 1;
  tg_dfree(len);
  tg_dfree(keep);
  if (rc) {
    if (m) tg_csr_destroy(m);
    tg_set_error("tg_csr_select_columns failed");
    return 1;
  }
  *out = m;
  return 0;
}

// ---- A = D + R for a cell-local FE space with hand-added couplings (demos/kl-shell-svk/reef-knot.py:455-467: T-spline
// K plus contact terms): D = the entries inside the diagonal b x b cell blocks -- accepted only when every row holds all b of
// them (dense blocks, what dolfin assembles on a mesh of disconnected cells) --, R = everything else, both with A's shape.
// side 0: count / copy the in-block entries, side 1: the others.
__global__ void __launch_bounds__(256)
    k_cells_count(const int64_t *__restrict__ rowptr, const int32_t *__restrict__ col, int b, int64_t n, int64_t *__restrict__ len_d,
                  int64_t *__restrict__ len_r, int *__restrict__ bad) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t r = wave; r < n; r += nwaves) {
    const int64_t a = rowptr[r], e = rowptr[r + 1];
    const int64_t c0 = (r / b) * b;
    int cnt = 0;
    for (int64_t q = a + lane; q < e; q += 64) cnt += (col[q] >= c0 && col[q] < c0 + b) ? 1 : 0;
    for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o, 64);
    if (lane == 0) {
      len_d[r] = cnt;
      len_r[r] = (e - a) - cnt;
      if (cnt != b) atomicOr(bad, 1);
    }
  }
}

__global__ void __launch_bounds__(256)
    k_cells_fill(const int64_t *__restrict__ rowptr, const int32_t *__restrict__ col, const double *__restrict__ val, int b, int side,
                 const int64_t *__restrict__ orowptr, int64_t n, int32_t *__restrict__ ocol, double *__restrict__ oval) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t r = wave; r < n; r += nwaves) {
    const int64_t a = rowptr[r], e = rowptr[r + 1];
    const int64_t c0 = (r / b) * b;
    int64_t o = orowptr[r];
    for (int64_t q0 = a; q0 < e; q0 += 64) {          // (uniform trip count: the ballot sees every lane)
      const int64_t q = q0 + lane;
      const bool in = q < e && col[q] >= c0 && col[q] < c0 + b;
      const bool k = q < e && (side == 0 ? in : !in);
      const unsigned long long m = __ballot(k);
      if (k) {
        const int pos = __popcll(m & ((1ull << lane) - 1ull));
        ocol[o + pos] = col[q];
        oval[o + pos] = val[q];
      }
      o += __popcll(m);
    }
  }
}

extern "C" int tg_csr_split_cells(tg_csr_t a, int b, tg_csr_t *d_out, tg_csr_t *r_out) {
  TG_REQUIRE_INIT();
  TG_REQUIRE(a && d_out && r_out && b >= 1, "bad arguments to tg_csr_split_cells");
  TG_REQUIRE_CANONICAL(a);
  const int64_t n = a->nrows;
  if (n == 0 || a->ncols != n || n % b) return 100;
  int64_t *len_d = nullptr, *len_r = nullptr;
  int *bad = nullptr;
  tg_csr_s *md = nullptr, *mr = nullptr;
  int rc = tg_dmalloc(&len_d, n + 1) || tg_dmalloc(&len_r, n + 1) || tg_dmalloc(&bad, 4);
  const unsigned grid = (unsigned)std::max<int64_t>(1, std::min<int64_t>(tg_cdiv(n, 4), (int64_t)g_tg.num_cu * 16));
  int h_bad = 0;
  if (!rc) {
    hipMemsetAsync(bad, 0, sizeof(int), g_tg.stream);
    hipLaunchKernelGGL(k_cells_count, dim3(grid), dim3(256), 0, g_tg.stream, a->rowptr, a->col, b, n, len_d, len_r, bad);
    if (hipGetLastError() != hipSuccess) rc = 1;
    if (!rc && hipMemcpyAsync(&h_bad, bad, sizeof(int), hipMemcpyDeviceToHost, g_tg.stream) != hipSuccess) rc = 1;
  }
  int64_t tot_d = 0, tot_r = 0;
  if (!rc) rc = tg_exclusive_scan_i64(len_d, n, &tot_d);      // (synchronises: h_bad is valid afterwards)
  if (!rc && h_bad) {
    tg_dfree(len_d);
    tg_dfree(len_r);
    tg_dfree(bad);
    return 100;                                                // some row lacks entries of its own cell block
  }
  if (!rc) rc = tg_exclusive_scan_i64(len_r, n, &tot_r);
  if (!rc) rc = tg_csr_alloc(n, n, tot_d, &md);
  if (!rc) rc = tg_csr_alloc(n, n, tot_r, &mr);
  if (!rc) {
    if (hipMemcpyAsync(md->rowptr, len_d, (size_t)(n + 1) * sizeof(int64_t), hipMemcpyDeviceToDevice, g_tg.stream) != hipSuccess ||
        hipMemcpyAsync(mr->rowptr, len_r, (size_t)(n + 1) * sizeof(int64_t), hipMemcpyDeviceToDevice, g_tg.stream) != hipSuccess)
      rc = 1;
    if (!rc && tot_d > 0)
      hipLaunchKernelGGL(k_cells_fill, dim3(grid), dim3(256), 0, g_tg.stream, a->rowptr, a->col, a->val, b, 0, md->rowptr, n, md->col,
                         md->val);
    if (!rc && tot_r > 0)
      hipLaunchKernelGGL(k_cells_fill, dim3(grid), dim3(256), 0, g_tg.stream, a->rowptr, a->col, a->val, b, 1, mr->rowptr, n, mr->col,
                         mr->val);
    if (hipGetLastError() != hipSuccess) rc = 1;
  }
  if (hipStreamSynchronize(g_tg.stream) != hipSuccess) rc = 1;
  tg_dfree(len_d);
  tg_dfree(len_r);
  tg_dfree(bad);
  if (rc) {
    if (md) tg_csr_destroy(md);
    if (mr) tg_csr_destroy(mr);
    tg_set_error("tg_csr_split_cells failed");
    return 1;
  }
  md->nnz = tot_d;
  mr->nnz = tot_r;
  *d_out = md;
  *r_out = mr;
  return 0;
}

struct tg_merge_args {
  const int64_t *rowptr[16];     // [i * nf + j] (one block row at a time: nf entries used)
  const int32_t *col[16];
  const double *val[16];
  int nf;
  int64_t n;                     // rows of the blocks of this block row
  int64_t coff[16];              // first column of block column j in the result
};

__global__ void k_merge_count(tg_merge_args A, int64_t *__restrict__ out_len) {
  int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; r < A.n; r += stride) {
    int64_t s = 0;
    for (int j = 0; j < A.nf; j++) s += A.rowptr[j][r + 1] - A.rowptr[j][r];
    out_len[r] = s;
  }
}

__global__ void __launch_bounds__(256)
    k_merge_fill(tg_merge_args A, const int64_t *__restrict__ orowptr, int32_t *__restrict__ ocol,
                 double *__restrict__ oval) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t r = wave; r < A.n; r += nwaves) {
    int64_t o = orowptr[r];
    for (int j = 0; j < A.nf; j++) {
      const int64_t a = A.rowptr[j][r], len = A.rowptr[j][r + 1] - a;
      const int32_t shift = (int32_t)A.coff[j];
      for (int64_t q = lane; q < len; q += 64) {
        ocol[o + q] = A.col[j][a + q] + shift;
        oval[o + q] = A.val[j][a + q];
      }
      o += len;
    }
  }
}

// blocks[i * nf + j]: the matrix with field-major rows and columns whose (i, j) block is blocks[i*nf+j]; the blocks of a block
// row share their row count, those of a block column their column count (fields on different bases: the counts differ
// from field to field)
extern "C" int tg_csr_from_blocks(int nf, const tg_csr_t *blocks, tg_csr_t *out) {
  TG_REQUIRE_INIT();
  TG_REQUIRE(nf >= 1 && nf <= 16 && blocks && out, "bad arguments to tg_csr_from_blocks");
  int64_t nnz = 0, rows_total = 0, cols_total = 0;
  int64_t rown[16], coff[17];
  coff[0] = 0;
  for (int q = 0; q < nf; q++) {
    TG_REQUIRE(blocks[q * nf + q], "tg_csr_from_blocks: null block");
    rown[q] = blocks[q * nf + q]->nrows;
    coff[q + 1] = coff[q] + blocks[q * nf + q]->ncols;
    rows_total += rown[q];
  }
  cols_total = coff[nf];
  for (int q = 0; q < nf * nf; q++) {
    TG_REQUIRE(blocks[q] && blocks[q]->nrows == rown[q / nf] && blocks[q]->ncols == coff[q % nf + 1] - coff[q % nf],
               "tg_csr_from_blocks: block %d has another shape than its block row / column", q);
    TG_REQUIRE_CANONICAL(blocks[q]);
    nnz += blocks[q]->nnz;
  }
  TG_REQUIRE(cols_total < 0x7fffffffll, "tg_csr_from_blocks: more than 2^31 columns");
  tg_csr_s *m = nullptr;
  TG_TRY(tg_csr_alloc(rows_total, cols_total, nnz, &m));
  int rc = 0;
  int64_t at = 0, row_at = 0;
  for (int i = 0; i < nf && !rc; i++) {
    tg_merge_args A;
    memset(&A, 0, sizeof(A));
    const int64_t n = rown[i];
    A.nf = nf;
    A.n = n;
    for (int j = 0; j < nf; j++) A.coff[j] = coff[j];
    int64_t block_row_nnz = 0;
    for (int j = 0; j < nf; j++) {
      A.rowptr[j] = blocks[i * nf + j]->rowptr;
      A.col[j] = blocks[i * nf + j]->col;
      A.val[j] = blocks[i * nf + j]->val;
      block_row_nnz += blocks[i * nf + j]->nnz;
    }
    int64_t *orp = m->rowptr + row_at;
    row_at += n;
    if (n > 0) {
      hipLaunchKernelGGL(k_merge_count, dim3(tg_grid_1d(n, 256)), dim3(256), 0, g_tg.stream, A, orp);
      if (hipGetLastError() != hipSuccess) rc = 1;
    }
    int64_t total = 0;
    // (the scan writes n + 1 entries: the last one is the first of the next block row, rewritten there)
    if (!rc) rc = tg_exclusive_scan_i64(orp, n, &total);
    if (!rc && total != block_row_nnz) {
      tg_set_error("tg_csr_from_blocks: block row %d holds %lld entries, its blocks %lld", i, (long long)total,
                   (long long)block_row_nnz);
      rc = 1;
    }
    if (!rc && n > 0 && total > 0) {
      // row starts of this block row are relative to its first entry: the fill adds `at` through the base pointers
      const unsigned grid = (unsigned)std::min<int64_t>(tg_cdiv(n, 4), (int64_t)g_tg.num_cu * 16);
      hipLaunchKernelGGL(k_merge_fill, dim3(grid), dim3(256), 0, g_tg.stream, A, orp, m->col + at, m->val + at);
      if (hipGetLastError() != hipSuccess) rc = 1;
    }
    if (!rc && n > 0 && at > 0) {
      // make the row pointers of this block row global
      hipLaunchKernelGGL(k_add_i64, dim3(tg_grid_1d(n + 1, 256)), dim3(256), 0, g_tg.stream, orp, n + 1, at);
      if (hipGetLastError() != hipSuccess) rc = 1;
    }
    at += total;
    if (!rc && n == 0 && hipMemcpyAsync(orp, &at, sizeof(int64_t), hipMemcpyHostToDevice, g_tg.stream) != hipSuccess) rc = 1;
  }
  if (hipStreamSynchronize(g_tg.stream) != hipSuccess) rc = 1;
  if (rc) {
    tg_csr_destroy(m);
    return 1;
  }
  *out = m;
  return 0;
}

// ----------------------------------------------------------------------------------------
// C = A + B on the union of the two patterns (MatAXPY with DIFFERENT_NONZERO_PATTERN [ext]); rows in ascending column
// order.  One thread per row walks both rows (B is the small operand where this is used: the product of the entries of
// an FE matrix that lie outside the element-coupling pattern).
__global__ void k_csr_add_count(const int64_t *__restrict__ arp, const int32_t *__restrict__ ac,
                                const int64_t *__restrict__ brp, const int32_t *__restrict__ bc, int64_t n,
                                int64_t *__restrict__ crp) {
  int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; r < n; r += stride) {
    int64_t i = arp[r], j = brp[r];
    const int64_t ie = arp[r + 1], je = brp[r + 1];
    int64_t o = ie - i;                           // (the common case: nothing to add in this row)
    while (j < je) {                              // columns of B that A does not have
      const int32_t cb = bc[j];
      while (i < ie && ac[i] < cb) i++;
      if (!(i < ie && ac[i] == cb)) o++;
      j++;
    }
    crp[r] = o;
  }
}

// rows without an entry of B: copied by a wave (coalesced) instead of the thread walk
__global__ void __launch_bounds__(256)
    k_csr_add_copy(const int64_t *__restrict__ arp, const int32_t *__restrict__ ac, const double *__restrict__ av,
                   const int64_t *__restrict__ brp, int64_t n, const int64_t *__restrict__ crp, int32_t *__restrict__ cc,
                   double *__restrict__ cv) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t r = wave; r < n; r += nwaves) {
    if (brp[r + 1] != brp[r]) continue;
    const int64_t a = arp[r], len = arp[r + 1] - a, o = crp[r];
    for (int64_t q = lane; q < len; q += 64) {
      cc[o + q] = ac[a + q];
      cv[o + q] = av[a + q];
    }
  }
}

__global__ void k_csr_add_mixed(const int64_t *__restrict__ arp, const int32_t *__restrict__ ac,
                                const double *__restrict__ av, const int64_t *__restrict__ brp,
                                const int32_t *__restrict__ bc, const double *__restrict__ bv, int64_t n,
                                const int64_t *__restrict__ crp, int32_t *__restrict__ cc, double *__restrict__ cv) {
  int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; r < n; r += stride) {
    int64_t i = arp[r], j = brp[r];
    const int64_t ie = arp[r + 1], je = brp[r + 1];
    if (j == je) continue;                        // (copied by k_csr_add_copy)
    int64_t o = crp[r];
    while (i < ie || j < je) {
      const int32_t ca = i < ie ? ac[i] : 0x7fffffff, cb = j < je ? bc[j] : 0x7fffffff;
      if (ca == cb) {
        cc[o] = ca;
        cv[o] = av[i] + bv[j];
      } else if (ca < cb) {
        cc[o] = ca;
        cv[o] = av[i];
      } else {
        cc[o] = cb;
        cv[o] = bv[j];
      }
      o++;
      if (ca <= cb) i++;
      if (cb <= ca) j++;
    }
  }
}

extern "C" int tg_csr_add(tg_csr_t a, tg_csr_t b, tg_csr_t *out) {
  TG_REQUIRE_INIT();
  TG_REQUIRE(a && b && out, "null argument to tg_csr_add");
  TG_REQUIRE(a->nrows == b->nrows && a->ncols == b->ncols, "tg_csr_add: shapes differ");
  TG_REQUIRE_CANONICAL(a);
  TG_REQUIRE_CANONICAL(b);
  const int64_t n = a->nrows;
  int64_t *len = nullptr;
  TG_TRY(tg_dmalloc(&len, n + 1));
  int rc = 0;
  tg_csr_s *m = nullptr;
  if (n > 0) {
    hipLaunchKernelGGL(k_csr_add_count, dim3(tg_grid_1d(n, 256)), dim3(256), 0, g_tg.stream, a->rowptr, a->col, b->rowptr,
                       b->col, n, len);
    if (hipGetLastError() != hipSuccess) rc = 1;
  }
  int64_t total = 0;
  if (!rc) rc = tg_exclusive_scan_i64(len, n, &total);
  if (!rc) rc = tg_csr_alloc(n, a->ncols, total, &m);
  if (!rc) {
    if (hipMemcpyAsync(m->rowptr, len, (size_t)(n + 1) * sizeof(int64_t), hipMemcpyDeviceToDevice, g_tg.stream) != hipSuccess)
      rc = 1;
    if (!rc && n > 0 && total > 0) {
      const unsigned wg = (unsigned)std::min<int64_t>(tg_cdiv(n, 4), (int64_t)g_tg.num_cu * 16);
      hipLaunchKernelGGL(k_csr_add_copy, dim3(wg), dim3(256), 0, g_tg.stream, a->rowptr, a->col, a->val, b->rowptr, n,
                         m->rowptr, m->col, m->val);
      hipLaunchKernelGGL(k_csr_add_mixed, dim3(tg_grid_1d(n, 256)), dim3(256), 0, g_tg.stream, a->rowptr, a->col, a->val,
                         b->rowptr, b->col, b->val, n, m->rowptr, m->col, m->val);
      if (hipGetLastError() != hipSuccess) rc = 1;
    }
  }
  if (hipStreamSynchronize(g_tg.stream) != hipSuccess) rc = 1;
  tg_dfree(len);
  if (rc) {
    if (m) tg_csr_destroy(m);
    tg_set_error("tg_csr_add failed");
    return 1;
  }
  *out = m;
  return 0;
}

// ----------------------------------------------------------------------------------------
// IGA dof permutation (tIGAr/common.py:407-433, 1583-1665)
// ----------------------------------------------------------------------------------------
__global__ void k_relabel_cols(const int32_t *__restrict__ col, const int32_t *__restrict__ new_of_old, int64_t nnz,
                               int32_t *__restrict__ out) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < nnz; i += stride) out[i] = new_of_old[col[i]];
}

// Copy of m with column c renamed new_of_old[c] (a permutation of 0..ncols-1, host array) and the rows put back
// into ascending column order: MatPermute with the identity on the rows, as applyPermutation uses it.
extern "C" int tg_csr_permute_columns(tg_csr_t m, const int32_t *new_of_old, tg_csr_t *out) {
  TG_REQUIRE_INIT();
  TG_REQUIRE(m && new_of_old && out, "null argument to tg_csr_permute_columns");
  TG_REQUIRE_CANONICAL(m);
  {
    std::vector<char> seen((size_t)m->ncols, 0);
    for (int64_t c = 0; c < m->ncols; c++) {
      const int64_t t = new_of_old[c];
      TG_REQUIRE(t >= 0 && t < m->ncols && !seen[(size_t)t], "tg_csr_permute_columns: not a permutation (entry %lld)",
                 (long long)c);
      seen[(size_t)t] = 1;
    }
  }
  tg_csr_s *t = nullptr;
  TG_TRY(tg_csr_alloc(m->nrows, m->ncols, m->nnz, &t));
  int32_t *map = nullptr;
  int rc = tg_dmalloc(&map, std::max<int64_t>(m->ncols, 1));
  if (!rc && m->ncols > 0 &&
      hipMemcpyAsync(map, new_of_old, (size_t)m->ncols * sizeof(int32_t), hipMemcpyHostToDevice, g_tg.stream) != hipSuccess)
    rc = 1;
  if (!rc && hipMemcpyAsync(t->rowptr, m->rowptr, (size_t)(m->nrows + 1) * sizeof(int64_t), hipMemcpyDeviceToDevice,
                            g_tg.stream) != hipSuccess)
    rc = 1;
  if (!rc && m->nnz > 0) {
    if (hipMemcpyAsync(t->val, m->val, (size_t)m->nnz * sizeof(double), hipMemcpyDeviceToDevice, g_tg.stream) != hipSuccess)
      rc = 1;
    hipLaunchKernelGGL(k_relabel_cols, dim3(tg_grid_1d(m->nnz, 256)), dim3(256), 0, g_tg.stream, m->col, map, m->nnz,
                       t->col);
    if (!rc && hipGetLastError() != hipSuccess) rc = 1;
    if (!rc) rc = tg_csr_sort_rows(t);
  }
  hipStreamSynchronize(g_tg.stream);
  tg_dfree(map);
  if (rc) {
    tg_set_error("tg_csr_permute_columns failed");
    tg_csr_destroy(t);
    return 1;
  }
  *out = t;
  return 0;
}

// One thread per row of mt (= an IGA dof; its columns are the FE rows of its support): counts of the owners of those
// FE rows in the thread's own line of `cnt`, then the most frequent owner, the lowest rank on a tie like
// scipy.stats.mode in the reference.  Serial per row, no atomics.
__global__ void k_partition_mode(const int64_t *__restrict__ rowptr, const int32_t *__restrict__ col, int64_t nrows,
                                 const int32_t *__restrict__ fe_owner, int world, int32_t *__restrict__ cnt,
                                 int32_t *__restrict__ owner) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < nrows; i += stride) {
    int32_t *c = cnt + i * world;
    for (int r = 0; r < world; r++) c[r] = 0;
    for (int64_t q = rowptr[i]; q < rowptr[i + 1]; q++) c[fe_owner[col[q]]]++;
    int best = 0;
    for (int r = 1; r < world; r++)
      if (c[r] > c[best]) best = r;
    owner[i] = best;
  }
}

extern "C" int tg_partition_mode(tg_csr_t mt, const int32_t *fe_owner, int world, int32_t *owner_out) {
  TG_REQUIRE_INIT();
  TG_REQUIRE(mt && fe_owner && owner_out && world >= 1, "bad arguments to tg_partition_mode");
  for (int64_t j = 0; j < mt->ncols; j++)
    TG_REQUIRE(fe_owner[j] >= 0 && fe_owner[j] < world, "tg_partition_mode: FE row %lld has owner %d of %d ranks",
               (long long)j, (int)fe_owner[j], world);
  if (mt->nrows == 0) return 0;
  TG_REQUIRE(mt->nrows * (int64_t)world < (1ll << 32), "tg_partition_mode: %lld dofs x %d ranks is beyond the count table",
             (long long)mt->nrows, world);
  int32_t *own_fe = nullptr, *cnt = nullptr, *own = nullptr;
  int rc = tg_dmalloc(&own_fe, std::max<int64_t>(mt->ncols, 1));
  if (!rc) rc = tg_dmalloc(&cnt, mt->nrows * (int64_t)world);
  if (!rc) rc = tg_dmalloc(&own, mt->nrows);
  if (!rc && mt->ncols > 0 &&
      hipMemcpyAsync(own_fe, fe_owner, (size_t)mt->ncols * sizeof(int32_t), hipMemcpyHostToDevice, g_tg.stream) != hipSuccess)
    rc = 1;
  if (!rc) {
    hipLaunchKernelGGL(k_partition_mode, dim3(tg_grid_1d(mt->nrows, 256)), dim3(256), 0, g_tg.stream, mt->rowptr, mt->col,
                       mt->nrows, own_fe, world, cnt, own);
    if (hipGetLastError() != hipSuccess ||
        hipMemcpyAsync(owner_out, own, (size_t)mt->nrows * sizeof(int32_t), hipMemcpyDeviceToHost, g_tg.stream) != hipSuccess)
      rc = 1;
  }
  if (hipStreamSynchronize(g_tg.stream) != hipSuccess) rc = 1;
  if (own_fe) tg_dfree(own_fe);
  if (cnt) tg_dfree(cnt);
  if (own) tg_dfree(own);
  if (rc) tg_set_error("tg_partition_mode failed");
  return rc;
}

extern "C" int tg_csr_transpose(tg_csr_t m, tg_csr_t *out) {
  TG_REQUIRE_INIT();
  TG_REQUIRE(m && out, "null argument to tg_csr_transpose");
  return tg_csr_transpose_block(m, 0, m->nrows, out);
}

// ----------------------------------------------------------------------------------------
// MatZeroRowsColumns
// ----------------------------------------------------------------------------------------
__global__ void k_mark(uint8_t *mask, int64_t n, const int32_t *dofs, int64_t nd) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < nd) {
    const int32_t d = dofs[i];
    if (d >= 0 && d < n) mask[d] = 1;
  }
}

__global__ void __launch_bounds__(256)
    k_zero_rows_cols(const int64_t *__restrict__ rowptr, const int32_t *__restrict__ col, double *__restrict__ val,
                     int64_t nrows, int64_t row0, const uint8_t *__restrict__ mask, double diag) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t r = wave; r < nrows; r += nwaves) {
    const int64_t g = r + row0;
    const bool mr = mask[g] != 0;
    for (int64_t q = rowptr[r] + lane; q < rowptr[r + 1]; q += 64) {
      const int32_t c = col[q];
      if (mr || mask[c]) val[q] = (mr && c == g) ? diag : 0.0;
    }
  }
}

// builds the byte mask over all `ncols` global dofs from a host index list (device pointer out)
int tg_build_dof_mask(const int32_t *dofs, int64_t n, int64_t ndofs_total, uint8_t **mask_out) {
  uint8_t *mask = nullptr;
  TG_TRY(tg_dmalloc(&mask, ndofs_total));
  TG_CHECK_HIP(hipMemsetAsync(mask, 0, (size_t)(ndofs_total > 0 ? ndofs_total : 1), g_tg.stream));
  if (n > 0) {
    int32_t *d = nullptr;
    TG_TRY(tg_dmalloc(&d, n));
    TG_CHECK_HIP(hipMemcpyAsync(d, dofs, (size_t)n * sizeof(int32_t), hipMemcpyHostToDevice, g_tg.stream));
    hipLaunchKernelGGL(k_mark, dim3((unsigned)tg_cdiv(n, 256)), dim3(256), 0, g_tg.stream, mask, ndofs_total, d, n);
    TG_LAUNCH_CHECK();
    TG_CHECK_HIP(hipStreamSynchronize(g_tg.stream));
    tg_dfree(d);
  }
  *mask_out = mask;
  return 0;
}

extern "C" int tg_zero_rows_cols(tg_csr_t k, int64_t row0, const int32_t *dofs, int64_t n, double diag) {
  TG_REQUIRE_CANONICAL(k);
  TG_REQUIRE_INIT();
  TG_REQUIRE(k, "null matrix");
  if (n <= 0 || k->nrows == 0) return 0;
  tg_dfree(k->diag_cache);      // values change: a recorded diagonal is stale
  k->diag_cache = nullptr;
  k->diag_rows = 0;
  k->sym_verified = 0;          // (and so is what was found out about the symmetry of the old values)
  uint8_t *mask = nullptr;
  TG_TRY(tg_build_dof_mask(dofs, n, k->ncols, &mask));
  const unsigned grid = (unsigned)std::min<int64_t>(tg_cdiv(k->nrows, 4), (int64_t)g_tg.num_cu * 16);
  hipLaunchKernelGGL(k_zero_rows_cols, dim3(grid), dim3(256), 0, g_tg.stream, k->rowptr, k->col, k->val, k->nrows, row0,
                     mask, diag);
  TG_LAUNCH_CHECK();
  TG_CHECK_HIP(hipStreamSynchronize(g_tg.stream));
  tg_dfree(mask);
  return 0;
}

// ---- out = a X + b Y diag(colscale) for two matrices on ONE pattern (tangent matrices of
// semilinear problems: K + M diag(g'(u)); PETSc MatAXPY with SAME_NONZERO_PATTERN + MatDiagonalScale)
__global__ void k_csr_combine(int64_t nnz, const int32_t *__restrict__ colx, const int32_t *__restrict__ coly,
                              const double *__restrict__ vx, const double *__restrict__ vy, double a, double b,
                              const double *__restrict__ cs, int32_t *__restrict__ colo, double *__restrict__ vo,
                              int *__restrict__ mismatch) {
  int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  bool bad = false;
  for (; q < nnz; q += stride) {
    const int32_t c = colx[q];
    bad |= (c != coly[q]);
    colo[q] = c;
    vo[q] = a * vx[q] + b * vy[q] * (cs ? cs[c] : 1.0);
  }
  if (bad) atomicOr(mismatch, 1);
}

__global__ void k_rowptr_equal(int64_t n, const int64_t *__restrict__ a, const int64_t *__restrict__ b, int64_t *__restrict__ o,
                               int *__restrict__ mismatch) {
  int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  bool bad = false;
  for (; q < n; q += stride) {
    bad |= (a[q] != b[q]);
    o[q] = a[q];
  }
  if (bad) atomicOr(mismatch, 1);
}

extern "C" int tg_csr_combine(double a, tg_csr_t X, double b, tg_csr_t Y, tg_vec_t colscale, tg_csr_t *out) {
  TG_REQUIRE_CANONICAL(X);
  TG_REQUIRE_CANONICAL(Y);
  TG_REQUIRE_INIT();
  TG_REQUIRE(X && Y && out, "bad arguments to tg_csr_combine");
  TG_REQUIRE(X->nrows == Y->nrows && X->ncols == Y->ncols && X->nnz == Y->nnz, "tg_csr_combine: the operands differ in shape or nnz");
  TG_REQUIRE(!colscale || colscale->n == X->ncols, "tg_csr_combine: column scaling has the wrong length");
  tg_csr_s *m = nullptr;
  TG_TRY(tg_csr_alloc(X->nrows, X->ncols, X->nnz, &m));
  int *flag = (int *)g_tg.scratch;
  hipMemsetAsync(flag, 0, sizeof(int), g_tg.stream);
  hipLaunchKernelGGL(k_rowptr_equal, dim3(tg_grid_1d(X->nrows + 1, 256)), dim3(256), 0, g_tg.stream, X->nrows + 1, X->rowptr,
                     Y->rowptr, m->rowptr, flag);
  if (X->nnz > 0)
    hipLaunchKernelGGL(k_csr_combine, dim3(tg_grid_1d(X->nnz, 256)), dim3(256), 0, g_tg.stream, X->nnz, X->col, Y->col, X->val,
                       Y->val, a, b, colscale ? colscale->d : (const double *)nullptr, m->col, m->val, flag);
  int h = 0;
  hipMemcpyAsync(&h, flag, sizeof(int), hipMemcpyDeviceToHost, g_tg.stream);
  if (hipStreamSynchronize(g_tg.stream) != hipSuccess || hipGetLastError() != hipSuccess) {
    tg_csr_destroy(m);
    tg_set_error("tg_csr_combine: kernel failed");
    return 1;
  }
  if (h) {
    tg_csr_destroy(m);
    tg_set_error("tg_csr_combine: the operands do not share one sparsity pattern");
    return 2;
  }
  *out = m;
  return 0;
}

// ----------------------------------------------------------------------------------------
// out row r = a row rows[r]: any selection / reordering of rows (with repetitions), columns untouched.  Used to bring
// the row blocks of a multi-field product from field-major order into the plane-interleaved order of the
// distributed numbering (tigar_amd/dist.py: FieldSlabPath).
__global__ void k_gather_len(const int64_t *__restrict__ arp, const int64_t *__restrict__ rows, int64_t n,
                             int64_t *__restrict__ len) {
  // (tg_grid_1d caps the grid: grid-stride loop.  Without it the rows beyond 8 x CUs x 256 = 524 288 kept whatever the
  //  buffer held -- found in round 5 by the first multi-field product with more local rows than that)
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < n; r += stride) len[r] = arp[rows[r] + 1] - arp[rows[r]];
}
__global__ void __launch_bounds__(256)
    k_gather_copy(const int64_t *__restrict__ arp, const int32_t *__restrict__ ac, const double *__restrict__ av,
                  const int64_t *__restrict__ rows, int64_t n, const int64_t *__restrict__ orp, int32_t *__restrict__ oc,
                  double *__restrict__ ov) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t r = wave; r < n; r += nwaves) {
    const int64_t a = arp[rows[r]], len = arp[rows[r] + 1] - a, o = orp[r];
    for (int64_t q = lane; q < len; q += 64) {
      oc[o + q] = ac[a + q];
      ov[o + q] = av[a + q];
    }
  }
}

extern "C" int tg_csr_gather_rows(tg_csr_t a, const int64_t *rows, int64_t n, tg_csr_t *out) {
  TG_REQUIRE_INIT();
  TG_REQUIRE(a && out && n >= 0 && (rows || n == 0), "bad arguments to tg_csr_gather_rows");
  TG_REQUIRE_CANONICAL(a);
  for (int64_t r = 0; r < n; r++)
    TG_REQUIRE(rows[r] >= 0 && rows[r] < a->nrows, "tg_csr_gather_rows: row %lld out of range", (long long)rows[r]);
  int64_t *d_rows = nullptr, *len = nullptr;
  tg_csr_s *m = nullptr;
  int rc = tg_dmalloc(&d_rows, std::max<int64_t>(n, 1));
  if (!rc) rc = tg_dmalloc(&len, n + 1);
  if (!rc && n > 0 &&
      hipMemcpyAsync(d_rows, rows, (size_t)n * sizeof(int64_t), hipMemcpyHostToDevice, g_tg.stream) != hipSuccess)
    rc = 1;
  int64_t total = 0;
  if (!rc && n > 0) {
    hipLaunchKernelGGL(k_gather_len, dim3(tg_grid_1d(n, 256)), dim3(256), 0, g_tg.stream, a->rowptr, d_rows, n, len);
    if (hipGetLastError() != hipSuccess) rc = 1;
  }
  if (!rc) rc = tg_exclusive_scan_i64(len, n, &total);
  if (!rc) rc = tg_csr_alloc(n, a->ncols, total, &m);
  if (!rc) {
    if (hipMemcpyAsync(m->rowptr, len, (size_t)(n + 1) * sizeof(int64_t), hipMemcpyDeviceToDevice, g_tg.stream) != hipSuccess)
      rc = 1;
    if (!rc && n > 0 && total > 0) {
      const unsigned grid = (unsigned)std::min<int64_t>(tg_cdiv(n, 4), (int64_t)g_tg.num_cu * 32);
      hipLaunchKernelGGL(k_gather_copy, dim3(grid), dim3(256), 0, g_tg.stream, a->rowptr, a->col, a->val, d_rows, n,
                         m->rowptr, m->col, m->val);
      if (hipGetLastError() != hipSuccess) rc = 1;
    }
  }
  const hipError_t sync_rc = hipStreamSynchronize(g_tg.stream);     // (`rows` is the caller's host array)
  if (sync_rc != hipSuccess) rc = 1;
  tg_dfree(d_rows);
  tg_dfree(len);
  if (rc) {
    if (m) tg_csr_destroy(m);
    tg_set_error("tg_csr_gather_rows failed (%s; %lld rows of %lld, %lld entries)", hipGetErrorString(sync_rc), (long long)n,
                 (long long)a->nrows, (long long)total);
    return 1;
  }
  m->nnz = total;
  *out = m;
  return 0;
}
