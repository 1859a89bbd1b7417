// Extraction-operator build: M[row, col] = N_col(x_row), written directly as CSR.
//
// Device twin of the reference's only native routine, basisFuncsInner
// (tIGAr/BSplines.py:73-120), of BSpline1.getKnotSpan/getNodes (:285-319),
// BSpline.getNodesAndEvals (:450-503) and the row loop + eps filter of generateM
// (tIGAr/common.py:1554-1571).
//
// Layout: wave-parallel.  One lane per candidate entry (i,j,k) of a row; a wave covers
// floor(64 / C) rows at once (C = prod(p_k+1) candidates per row); survivors of the
// |v| > eps filter are ranked with a ballot + popcount and stored straight to their CSR
// slot, so every wave writes one contiguous segment of col[] / val[].  The kernel is
// HBM-write-bound: 12 B per nnz + 8 B per row.
#include "tg_common.h"
#include <algorithm>

// ----------------------------------------------------------------------------------------
// 1-D evaluation.  All floating-point operations are the reference's, in the reference's
// order, with contraction disabled (no FMA) so values are bit-identical to the CPU path.
// ----------------------------------------------------------------------------------------
struct tg_dir_dev {
  int32_t p, nknots, mult_first, mult_last, ncp;
  const double *ghost;  // device
};

__device__ __forceinline__ int tg_knot_span(const tg_dir_dev &D, double u) {
  // numpy.searchsorted(knots, u) - 1 (side='left'), then the clamps of
  // tIGAr/BSplines.py:302-307
  const double *knots = D.ghost + (D.p + 1);
  int lo = 0, hi = D.nknots;
  while (lo < hi) {
    int mid = (lo + hi) >> 1;
    if (knots[mid] < u)
      lo = mid + 1;
    else
      hi = mid;
  }
  int span = lo - 1;
  const int nspans = D.nknots - 1;
  const int cl = D.mult_first - 1;
  const int ch = nspans - (D.mult_last - 1) - 1;
  if (span < cl) span = cl;
  if (span > ch) span = ch;
  return span;
}

// N[0..p] = the p+1 non-zero basis functions at u in `span` (Cox-de Boor, A2.2 of
// Piegl-Tiller exactly as coded at tIGAr/BSplines.py:102-119; only the last column of
// the reference's ndu table is kept, the arithmetic is identical).
__device__ __forceinline__ void tg_basis_funcs(const tg_dir_dev &D, int span, double u, double *N) {
#pragma clang fp contract(off)
  const int p = D.p;
  const int nG = p + 1;
  const int i = span + 1;
  const double *U = D.ghost;
  double left[TG_MAX_DEGREE + 1], right[TG_MAX_DEGREE + 1];
  N[0] = 1.0;
  for (int j = 1; j <= p; j++) {
    left[j] = __dsub_rn(u, U[i - j + nG]);
    right[j] = __dsub_rn(U[i + j - 1 + nG], u);
    double saved = 0.0;
    for (int r = 0; r < j; r++) {
      const double den = __dadd_rn(right[r + 1], left[j - r]);
      const double temp = __ddiv_rn(N[r], den);
      N[r] = __dadd_rn(saved, __dmul_rn(right[r + 1], temp));
      saved = __dmul_rn(left[j - r], temp);
    }
    N[j] = saved;
  }
}

__device__ __forceinline__ int tg_pymod(int v, int m) {
  int r = v % m;
  return r < 0 ? r + m : r;
}

// idx/val: [n x (p+1)].  sorted != 0: entries of a node sorted by basis index (needed so
// that tensor candidates come out in CSR column order, also under periodic wrap).
__global__ void k_eval_1d(tg_dir_dev D, const double *__restrict__ u, int64_t n, int32_t *__restrict__ span_out,
                          int32_t *__restrict__ idx, double *__restrict__ val, int sorted) {
  int64_t a = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (a >= n) return;
  const double x = u[a];
  const int span = tg_knot_span(D, x);
  double N[TG_MAX_DEGREE + 1];
  int id[TG_MAX_DEGREE + 1];
  tg_basis_funcs(D, span, x, N);
  const int p = D.p;
  for (int r = 0; r <= p; r++) id[r] = tg_pymod(span - p + r, D.ncp);
  if (sorted) {
    for (int r = 1; r <= p; r++) {  // insertion sort, p+1 <= 9
      int ki = id[r];
      double kv = N[r];
      int q = r - 1;
      while (q >= 0 && id[q] > ki) {
        id[q + 1] = id[q];
        N[q + 1] = N[q];
        q--;
      }
      id[q + 1] = ki;
      N[q + 1] = kv;
    }
  }
  if (span_out) span_out[a] = span;
  for (int r = 0; r <= p; r++) {
    idx[a * (p + 1) + r] = id[r];
    val[a * (p + 1) + r] = N[r];
  }
}

struct tg_dir_tables {
  double *ghost = nullptr, *nodes = nullptr, *val = nullptr;
  int32_t *idx = nullptr;
  void free_all() {
    tg_dfree(ghost);
    tg_dfree(nodes);
    tg_dfree(val);
    tg_dfree(idx);
  }
};

static int tg_check_dir(const tg_dir_t &d) {
  TG_REQUIRE(d.p >= 1 && d.p <= TG_MAX_DEGREE, "spline degree %d outside [1,%d]", d.p, TG_MAX_DEGREE);
  TG_REQUIRE(d.nknots >= 2 && d.ghost && d.ncp >= 1, "bad knot data");
  return 0;
}

// uploads one direction and evaluates its 1-D table for `n` coordinates (host pointer)
static int tg_build_dir_table(const tg_dir_t &d, const double *coords, int64_t n, int64_t stride_elems,
                              bool coords_on_device, tg_dir_tables *T, int32_t *span_dev, int sorted) {
  TG_TRY(tg_check_dir(d));
  const int64_t nghost = (int64_t)d.nknots + 2 * (d.p + 1);
  TG_TRY(tg_dmalloc(&T->ghost, nghost));
  TG_CHECK_HIP(hipMemcpyAsync(T->ghost, d.ghost, nghost * sizeof(double), hipMemcpyHostToDevice, g_tg.stream));
  const double *dev_coords = coords;
  if (!coords_on_device) {
    TG_TRY(tg_dmalloc(&T->nodes, n));
    TG_CHECK_HIP(hipMemcpyAsync(T->nodes, coords, n * sizeof(double), hipMemcpyHostToDevice, g_tg.stream));
    dev_coords = T->nodes;
  }
  (void)stride_elems;
  TG_TRY(tg_dmalloc(&T->idx, n * (d.p + 1)));
  TG_TRY(tg_dmalloc(&T->val, n * (d.p + 1)));
  tg_dir_dev D{d.p, d.nknots, d.mult_first, d.mult_last, d.ncp, T->ghost};
  hipLaunchKernelGGL(k_eval_1d, dim3((unsigned)tg_cdiv(n, 256)), dim3(256), 0, g_tg.stream, D, dev_coords, n, span_dev,
                     T->idx, T->val, sorted);
  TG_LAUNCH_CHECK();
  return 0;
}

extern "C" int tg_eval_basis_1d(const tg_dir_t *dir, const double *u, int64_t n, int32_t *span, int32_t *idx,
                                double *val) {
  TG_REQUIRE_INIT();
  TG_REQUIRE(dir && u && n >= 0 && idx && val, "bad arguments to tg_eval_basis_1d");
  if (n == 0) return 0;
  tg_dir_tables T;
  int32_t *span_dev = nullptr;
  TG_TRY(tg_dmalloc(&span_dev, n));
  int rc = tg_build_dir_table(*dir, u, n, 1, false, &T, span_dev, 0);
  if (!rc) {
    const int pp1 = dir->p + 1;
    hipMemcpyAsync(idx, T.idx, n * pp1 * sizeof(int32_t), hipMemcpyDeviceToHost, g_tg.stream);
    hipMemcpyAsync(val, T.val, n * pp1 * sizeof(double), hipMemcpyDeviceToHost, g_tg.stream);
    if (span) hipMemcpyAsync(span, span_dev, n * sizeof(int32_t), hipMemcpyDeviceToHost, g_tg.stream);
    if (hipStreamSynchronize(g_tg.stream) != hipSuccess) {
      tg_set_error("tg_eval_basis_1d: stream sync failed");
      rc = 1;
    }
  }
  T.free_all();
  tg_dfree(span_dev);
  return rc;
}

// ----------------------------------------------------------------------------------------
// Tensor / point-cloud CSR build
// ----------------------------------------------------------------------------------------
struct tg_extract_params {
  int d;
  int pp1[3];            // p_k + 1
  int C;                 // candidates per row = prod pp1
  int rpw;               // rows per wave (C <= 64), else 1
  int iters;             // row groups per wave
  int64_t n[3];          // nodes per direction (tensor mode)
  int64_t cstride[3];    // column strides 1, ncp0, ncp0*ncp1
  const int32_t *idx[3];
  const double *val[3];
  int64_t col_offset;
  double eps;
  int64_t row0;          // first global row of the output block
  int64_t nrows;         // rows in the output block
  // tensor mode: pencils (fixed b,c) p0 .. p0+npencils, chunks of rows along direction 0
  int64_t pencil0, npencils;
  int32_t chunks_per_pencil;
  int32_t rows_per_block;
};

// value of candidate (i,j,k): (Nu_i * Nv_j) * Nw_k, left to right (tIGAr/BSplines.py:473,500)
template <int D>
__device__ __forceinline__ double tg_cand_value(const tg_extract_params &P, const int64_t *t, int i, int j, int k) {
  double v = P.val[0][t[0] * P.pp1[0] + i];
  if (D > 1) v = v * P.val[1][t[1] * P.pp1[1] + j];
  if (D > 2) v = v * P.val[2][t[2] * P.pp1[2] + k];
  return v;
}

template <int D>
__device__ __forceinline__ int64_t tg_cand_col(const tg_extract_params &P, const int64_t *t, int i, int j, int k) {
  int64_t c = P.idx[0][t[0] * P.pp1[0] + i];
  if (D > 1) c += P.cstride[1] * P.idx[1][t[1] * P.pp1[1] + j];
  if (D > 2) c += P.cstride[2] * P.idx[2][t[2] * P.pp1[2] + k];
  return c + P.col_offset;
}

// Per-block context: TENSOR: block = (pencil, chunk) -- one pencil = fixed (b,c), rows run
// along direction 0; POINTS: rows are consecutive.  Computed once per block (two integer
// divisions), so the per-lane row mapping below is add/compare only.
struct tg_blk_ctx {
  int64_t a0;      // first direction-0 index (TENSOR) / first row (POINTS) of the block
  int64_t b, c;    // pencil coordinates
  int64_t grow0;   // global row of a = 0 in this pencil
};

template <int D, bool POINTS>
__device__ __forceinline__ tg_blk_ctx tg_block_ctx(const tg_extract_params &P, int64_t blk) {
  tg_blk_ctx X;
  if (POINTS) {
    X.a0 = blk * P.rows_per_block;
    X.b = X.c = 0;
    X.grow0 = 0;
  } else {
    const int64_t pencil = P.pencil0 + blk / P.chunks_per_pencil;
    const int chunk = (int)(blk % P.chunks_per_pencil);
    X.a0 = (int64_t)chunk * P.rows_per_block;
    X.b = (D == 2) ? pencil : ((D == 3) ? pencil % P.n[1] : 0);
    X.c = (D == 3) ? pencil / P.n[1] : 0;
    X.grow0 = P.n[0] * pencil;
  }
  return X;
}

// row_in_block -> local row + table indices; false if there is no such row
template <int D, bool POINTS>
__device__ __forceinline__ bool tg_lane_row(const tg_extract_params &P, const tg_blk_ctx &X, int row_in_block,
                                            int64_t *lrow, int64_t *t) {
  const int64_t a = X.a0 + row_in_block;
  if (POINTS) {
    if (a >= P.nrows) return false;
    *lrow = a;
    t[0] = t[1] = t[2] = a;
    return true;
  } else {
    if (a >= P.n[0]) return false;
    const int64_t lr = X.grow0 + a - P.row0;
    if (lr < 0 || lr >= P.nrows) return false;
    *lrow = lr;
    t[0] = a;
    t[1] = X.b;
    t[2] = X.c;
    return true;
  }
}

// pass 1: per-row survivor counts -> rowptr[lrow] (scanned afterwards); thread per row
template <int D, bool POINTS>
__global__ void __launch_bounds__(256) k_extract_count(tg_extract_params P, int64_t *__restrict__ rowptr) {
  const tg_blk_ctx X = tg_block_ctx<D, POINTS>(P, blockIdx.x);
  const int nk = (D > 2) ? P.pp1[2] : 1, nj = (D > 1) ? P.pp1[1] : 1;
  // TENSOR mode: the factors of the directions 1 and 2 belong to the pencil, not to the row
  __shared__ double v1r[TG_MAX_DEGREE + 1], v2r[TG_MAX_DEGREE + 1];
  if (!POINTS) {
    if (threadIdx.x <= TG_MAX_DEGREE) {
      const int q = threadIdx.x;
      v1r[q] = (D > 1 && q < nj) ? P.val[1][X.b * P.pp1[1] + q] : 1.0;
      v2r[q] = (D > 2 && q < nk) ? P.val[2][X.c * P.pp1[2] + q] : 1.0;
    }
    __syncthreads();
  }
  for (int q = threadIdx.x; q < P.rows_per_block; q += 256) {
    int64_t lrow, t[3];
    if (!tg_lane_row<D, POINTS>(P, X, q, &lrow, t)) continue;
    int cnt = 0;
    if (!POINTS) {
      for (int i = 0; i < P.pp1[0]; i++) {
        const double v0 = P.val[0][t[0] * P.pp1[0] + i];
        for (int k = 0; k < nk; k++)
          for (int j = 0; j < nj; j++) {
            double v = v0;
            if (D > 1) v = v * v1r[j];
            if (D > 2) v = v * v2r[k];
            cnt += (fabs(v) > P.eps) ? 1 : 0;
          }
      }
      rowptr[lrow] = cnt;
      continue;
    }
    for (int k = 0; k < nk; k++)
      for (int j = 0; j < nj; j++)
        for (int i = 0; i < P.pp1[0]; i++) cnt += (fabs(tg_cand_value<D>(P, t, i, j, k)) > P.eps) ? 1 : 0;
    rowptr[lrow] = cnt;
  }
}

// pass 2: lane-per-candidate fill
template <int D, bool POINTS>
__global__ void __launch_bounds__(256)
    k_extract_fill(tg_extract_params P, const int64_t *__restrict__ rowptr, int32_t *__restrict__ col,
                   double *__restrict__ val) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const tg_blk_ctx X = tg_block_ctx<D, POINTS>(P, blockIdx.x);
  if (P.C <= 64) {
    const int rsub = lane / P.C;
    const int c = lane - rsub * P.C;
    const bool lane_used = rsub < P.rpw;
    const int i = c % P.pp1[0];
    const int jk = c / P.pp1[0];
    const int j = (D > 1) ? jk % P.pp1[1] : 0;
    const int k = (D > 2) ? jk / P.pp1[1] : 0;
    const unsigned long long rowmask_base = (P.C == 64) ? ~0ull : ((1ull << P.C) - 1ull);
    // TENSOR mode: a block walks ONE pencil (b, c fixed), so the lane's factors of the directions 1 and 2 and its column
    // part are the same for every row: loaded once (the product order (v0 * v1) * v2 of the reference is kept)
    double v1 = 1.0, v2 = 1.0;
    int64_t cjk = P.col_offset;
    if (!POINTS && lane_used) {
      if (D > 1) {
        v1 = P.val[1][X.b * P.pp1[1] + j];
        cjk += P.cstride[1] * P.idx[1][X.b * P.pp1[1] + j];
      }
      if (D > 2) {
        v2 = P.val[2][X.c * P.pp1[2] + k];
        cjk += P.cstride[2] * P.idx[2][X.c * P.pp1[2] + k];
      }
    }
    // four row groups at a time: their table values, row starts and 1-D column parts are requested before the first
    // ballot (the loads of a group are independent of the other groups'; one at a time the kernel waits for each)
    constexpr int U = 4;
    for (int it0 = 0; it0 < P.iters; it0 += U) {
      int64_t lrow[U], t0[U], rs[U];
      double v[U];
      int32_t c0[U];
      bool has[U];
#pragma unroll
      for (int u = 0; u < U; u++) {
        const int group = (it0 + u) * 4 + w;
        int64_t t[3] = {0, 0, 0};
        lrow[u] = 0;
        has[u] = (it0 + u < P.iters) && lane_used && tg_lane_row<D, POINTS>(P, X, group * P.rpw + rsub, &lrow[u], t);
        t0[u] = t[0];
        v[u] = 0.0;
        c0[u] = 0;
        rs[u] = 0;
        if (has[u]) {
          if (POINTS) {
            v[u] = tg_cand_value<D>(P, t, i, j, k);
            c0[u] = (int32_t)tg_cand_col<D>(P, t, i, j, k);
          } else {
            v[u] = P.val[0][t[0] * P.pp1[0] + i];
            c0[u] = P.idx[0][t[0] * P.pp1[0] + i];
          }
          rs[u] = rowptr[lrow[u]];
        }
      }
#pragma unroll
      for (int u = 0; u < U; u++) {
        double vv = v[u];
        if (!POINTS) {
          if (D > 1) vv = vv * v1;
          if (D > 2) vv = vv * v2;
        }
        const bool keep = has[u] && fabs(vv) > P.eps;
        const unsigned long long m = __ballot(keep);
        if (keep) {
          const unsigned long long rowmask = rowmask_base << (rsub * P.C);
          const unsigned long long below = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
          const int rank = __popcll(m & rowmask & below);
          const int64_t pos = rs[u] + rank;
          col[pos] = POINTS ? c0[u] : (int32_t)(cjk + c0[u]);
          val[pos] = vv;
        }
      }
    }
  } else {
    // C > 64: one row per wave, ceil(C/64) passes with a running offset
    for (int it = 0; it < P.iters; it++) {
      const int group = it * 4 + w;
      int64_t lrow = 0, t[3] = {0, 0, 0};
      const bool has = tg_lane_row<D, POINTS>(P, X, group, &lrow, t);  // wave-uniform
      if (!has) continue;
      int64_t base = rowptr[lrow];
      for (int c0 = 0; c0 < P.C; c0 += 64) {
        const int c = c0 + lane;
        double v = 0.0;
        bool keep = false;
        int i = 0, j = 0, k = 0;
        if (c < P.C) {
          i = c % P.pp1[0];
          const int jk = c / P.pp1[0];
          j = (D > 1) ? jk % P.pp1[1] : 0;
          k = (D > 2) ? jk / P.pp1[1] : 0;
          v = tg_cand_value<D>(P, t, i, j, k);
          keep = fabs(v) > P.eps;
        }
        const unsigned long long m = __ballot(keep);
        if (keep) {
          const unsigned long long below = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
          const int64_t pos = base + __popcll(m & below);
          col[pos] = (int32_t)tg_cand_col<D>(P, t, i, j, k);
          val[pos] = v;
        }
        base += __popcll(m);
      }
    }
  }
}

static void tg_fill_common(tg_extract_params &P, int d, const tg_dir_t *dirs, int32_t col_offset, double eps);

// Matrix-free application y[row] = sum_c M[row, c] * x[c] of the same operator: the candidates of
// a row are evaluated exactly as in k_extract_fill (same products, same filter) and contracted
// with x on the fly, so M never has to exist in memory (prolongation u = M U of
// tIGAr/common.py:1259 at sizes where M does not fit).  Lane per candidate, wave-level reduction.
template <int D>
__global__ void __launch_bounds__(256)
    k_extract_apply(tg_extract_params P, const double *__restrict__ x, double *__restrict__ y) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const tg_blk_ctx X = tg_block_ctx<D, false>(P, blockIdx.x);
  const int nwave_rows = P.rows_per_block / 4;      // rows handled by each wave of the block
  for (int q = 0; q < nwave_rows; q++) {
    int64_t lrow = 0, t[3] = {0, 0, 0};
    const bool has = tg_lane_row<D, false>(P, X, w * nwave_rows + q, &lrow, t);   // wave-uniform
    if (!has) continue;
    double s = 0.0;
    for (int c = lane; c < P.C; c += 64) {
      const int i = c % P.pp1[0];
      const int jk = c / P.pp1[0];
      const int j = (D > 1) ? jk % P.pp1[1] : 0;
      const int k = (D > 2) ? jk / P.pp1[1] : 0;
      const double v = tg_cand_value<D>(P, t, i, j, k);
      if (fabs(v) > P.eps) s += v * x[tg_cand_col<D>(P, t, i, j, k)];
    }
    s = tg_wave_sum(s);
    if (lane == 0) y[lrow] = s;
  }
}

extern "C" int tg_extract_apply_tensor(int d, const tg_dir_t *dirs, int32_t col_offset, double eps, int64_t row0,
                                       int64_t row1, tg_vec_t x, int64_t x_col0, tg_vec_t y) {
  TG_REQUIRE_INIT();
  TG_REQUIRE(d >= 1 && d <= 3 && dirs && x && y, "bad arguments to tg_extract_apply_tensor");
  int64_t total = 1;
  for (int k = 0; k < d; k++) {
    TG_TRY(tg_check_dir(dirs[k]));
    TG_REQUIRE(dirs[k].nnodes >= 1 && dirs[k].nodes, "direction %d has no nodes", k);
    total *= dirs[k].nnodes;
  }
  TG_REQUIRE(row0 >= 0 && row1 >= row0 && row1 <= total && y->n == row1 - row0, "row range / y size mismatch");
  tg_extract_params P;
  tg_fill_common(P, d, dirs, col_offset, eps);
  tg_dir_tables T[3];
  int rc = 0;
  for (int k = 0; k < d && !rc; k++) {
    rc = tg_build_dir_table(dirs[k], dirs[k].nodes, dirs[k].nnodes, 1, false, &T[k], nullptr, 1);
    P.n[k] = dirs[k].nnodes;
    P.idx[k] = T[k].idx;
    P.val[k] = T[k].val;
  }
  for (int k = d; k < 3; k++) P.n[k] = 1;
  if (!rc && row1 > row0) {
    P.row0 = row0;
    P.nrows = row1 - row0;
    P.rows_per_block = 32;
    const int64_t n0 = P.n[0];
    P.pencil0 = row0 / n0;
    P.npencils = (row1 - 1) / n0 + 1 - P.pencil0;
    P.chunks_per_pencil = (int32_t)tg_cdiv(n0, P.rows_per_block);
    const int64_t nblocks = P.npencils * P.chunks_per_pencil;
    const double *xs = x->d - x_col0;      // x holds the columns [x_col0, x_col0 + size(x))
    if (nblocks >= (1ll << 31)) {
      tg_set_error("tg_extract_apply_tensor: grid too large");
      rc = 2;
    } else if (d == 1)
      hipLaunchKernelGGL((k_extract_apply<1>), dim3((unsigned)nblocks), dim3(256), 0, g_tg.stream, P, xs, y->d);
    else if (d == 2)
      hipLaunchKernelGGL((k_extract_apply<2>), dim3((unsigned)nblocks), dim3(256), 0, g_tg.stream, P, xs, y->d);
    else
      hipLaunchKernelGGL((k_extract_apply<3>), dim3((unsigned)nblocks), dim3(256), 0, g_tg.stream, P, xs, y->d);
    if (!rc && hipGetLastError() != hipSuccess) {
      tg_set_error("tg_extract_apply_tensor launch failed");
      rc = 1;
    }
  }
  hipStreamSynchronize(g_tg.stream);
  for (int k = 0; k < d; k++) T[k].free_all();
  return rc;
}

template <int D, bool POINTS>
static int tg_run_extract(tg_extract_params &P, int64_t nblocks, int64_t ncols, tg_csr_t *out) {
  int64_t *rowptr = nullptr;
  TG_TRY(tg_dmalloc(&rowptr, P.nrows + 1));
  hipMemsetAsync(rowptr, 0, (size_t)(P.nrows + 1) * sizeof(int64_t), g_tg.stream);
  if (nblocks > 0) {
    hipLaunchKernelGGL((k_extract_count<D, POINTS>), dim3((unsigned)nblocks), dim3(256), 0, g_tg.stream, P, rowptr);
    TG_LAUNCH_CHECK();
  }
  int64_t nnz = 0;
  if (tg_exclusive_scan_i64(rowptr, P.nrows, &nnz)) {
    tg_dfree(rowptr);
    return 1;
  }
  tg_csr_s *m = new tg_csr_s();
  m->nrows = P.nrows;
  m->ncols = ncols;
  m->nnz = nnz;
  m->rowptr = rowptr;
  if (tg_dmalloc(&m->col, nnz + TG_CSR_PAD) || tg_dmalloc(&m->val, nnz + TG_CSR_PAD)) {
    tg_csr_destroy(m);
    return 1;
  }
  if (nblocks > 0 && nnz > 0) {
    hipLaunchKernelGGL((k_extract_fill<D, POINTS>), dim3((unsigned)nblocks), dim3(256), 0, g_tg.stream, P, rowptr,
                       m->col, m->val);
    TG_LAUNCH_CHECK();
  }
  TG_CHECK_HIP(hipStreamSynchronize(g_tg.stream));
  *out = m;
  return 0;
}

static void tg_fill_common(tg_extract_params &P, int d, const tg_dir_t *dirs, int32_t col_offset, double eps) {
  memset(&P, 0, sizeof(P));
  P.d = d;
  P.C = 1;
  for (int k = 0; k < 3; k++) {
    P.pp1[k] = (k < d) ? dirs[k].p + 1 : 1;
    P.C *= P.pp1[k];
  }
  P.cstride[0] = 1;
  P.cstride[1] = (d > 1) ? dirs[0].ncp : 0;
  P.cstride[2] = (d > 2) ? (int64_t)dirs[0].ncp * dirs[1].ncp : 0;
  P.col_offset = col_offset;
  P.eps = eps;
  P.rpw = (P.C <= 64) ? 64 / P.C : 1;
  // rows per block: the per-block set-up (pencil coordinates, the lane's factors of the directions 1 and 2) is paid once
  // per block and the count pass works a thread per row, so more rows per block help -- up to 16 row groups per wave
  // (cfg2, TIGAR_EXTRACT_KRON=0: fill 1.34 / 1.25 / 1.93 ms and count 0.28 / 0.23 / 0.25 ms for 8 / 16 / 32 groups: with
  // 32 a pencil of 257 rows is two blocks of 256 + 1 rows); not more than a short first direction has
  const int64_t n0 = dirs[0].nnodes > 0 ? dirs[0].nnodes : 256;
  int iters = (int)tg_cdiv(tg_cdiv(n0, 4 * P.rpw), 4) * 4;           // multiples of the unroll of the fill kernel
  if (getenv("TIGAR_EXTRACT_ITERS")) iters = std::max(4, atoi(getenv("TIGAR_EXTRACT_ITERS")) / 4 * 4);   // (experiments)
  P.iters = std::max(4, std::min(iters, 16));
  P.rows_per_block = 4 * P.rpw * P.iters;
}

extern "C" int tg_extract_csr_tensor(int d, const tg_dir_t *dirs, int32_t col_offset, int64_t ncols, double eps,
                                     int64_t row0, int64_t row1, tg_csr_t *out) {
  TG_REQUIRE_INIT();
  TG_REQUIRE(d >= 1 && d <= 3 && dirs && out, "bad arguments to tg_extract_csr_tensor");
  int64_t total = 1;
  for (int k = 0; k < d; k++) {
    TG_TRY(tg_check_dir(dirs[k]));
    TG_REQUIRE(dirs[k].nnodes >= 1 && dirs[k].nodes, "direction %d has no nodes", k);
    total *= dirs[k].nnodes;
  }
  TG_REQUIRE(row0 >= 0 && row1 >= row0 && row1 <= total, "row range [%lld,%lld) outside [0,%lld)", (long long)row0,
             (long long)row1, (long long)total);
  tg_extract_params P;
  tg_fill_common(P, d, dirs, col_offset, eps);
  tg_dir_tables T[3];
  int rc = 0;
  for (int k = 0; k < d && !rc; k++) {
    rc = tg_build_dir_table(dirs[k], dirs[k].nodes, dirs[k].nnodes, 1, false, &T[k], nullptr, 1);
    P.n[k] = dirs[k].nnodes;
    P.idx[k] = T[k].idx;
    P.val[k] = T[k].val;
  }
  for (int k = d; k < 3; k++) P.n[k] = 1;
  // ---- the filter abs(v) > eps acts on PRODUCTS of 1-D values (tIGAr/common.py:1569).  When the 1-D tables prove that it
  // acts factor by factor -- every 1-D value is <= 1 in magnitude, so a product with a factor <= eps is dropped, and the
  // product of the smallest kept factors of the directions is still > eps, so nothing else is -- the row lengths are
  // products of 1-D counts and the row starts follow in closed form: no count pass, no scan, and the entries can be
  // written entry by entry instead of candidate by candidate (tg_kron3_csr: the same values in the same order, bit for
  // bit; 5.1 against 2.1 TB/s at cfg2).  The tables are tiny (nodes of ONE direction): checked on the host.  Point
  // clouds, filters that do cut into non-zero products (p = 1 with perturbed nodes), periodic wraps that put a row's
  // columns out of order, and TIGAR_EXTRACT_SEPARABLE=0 keep the count / fill kernels below.
  if (!rc && row1 > row0 && !(getenv("TIGAR_EXTRACT_SEPARABLE") && atoi(getenv("TIGAR_EXTRACT_SEPARABLE")) == 0) && eps >= 0.0) {
    std::vector<std::vector<int32_t>> hrp(d), hcl(d);
    std::vector<std::vector<double>> hvl(d);
    bool separable = true;
    double minprod = 1.0;
    for (int k = 0; k < d && separable; k++) {
      const int64_t nk = dirs[k].nnodes;
      const int pp1 = dirs[k].p + 1;
      std::vector<int32_t> idx((size_t)(nk * pp1));
      std::vector<double> val((size_t)(nk * pp1));
      if (hipMemcpyAsync(idx.data(), T[k].idx, idx.size() * sizeof(int32_t), hipMemcpyDeviceToHost, g_tg.stream) != hipSuccess ||
          hipMemcpyAsync(val.data(), T[k].val, val.size() * sizeof(double), hipMemcpyDeviceToHost, g_tg.stream) != hipSuccess ||
          hipStreamSynchronize(g_tg.stream) != hipSuccess) {
        separable = false;
        break;
      }
      double mink = 1.7e308;
      hrp[k].assign((size_t)nk + 1, 0);
      for (int64_t t = 0; t < nk && separable; t++) {
        int32_t last = -1;
        for (int q = 0; q < pp1; q++) {
          const double a = fabs(val[(size_t)(t * pp1 + q)]);
          if (!(a <= 1.0 + 1e-12)) separable = false;            // (also NaN)
          if (a > eps) {
            const int32_t c = idx[(size_t)(t * pp1 + q)];
            if (c <= last) separable = false;                     // wrapped / repeated column: not in CSR order
            last = c;
            mink = std::min(mink, a);
            hcl[k].push_back(c);
            hvl[k].push_back(val[(size_t)(t * pp1 + q)]);
          }
        }
        hrp[k][(size_t)t + 1] = (int32_t)hcl[k].size();
      }
      if (hcl[k].empty()) separable = false;
      minprod *= mink;
    }
    if (separable && minprod > eps) {
      tg_kron_dir_t kd[3];
      int64_t cdim[3] = {1, 1, 1};
      for (int k = 0; k < d; k++) {
        kd[k].n = dirs[k].nnodes;
        kd[k].rowptr = hrp[k].data();
        kd[k].col = hcl[k].data();
        kd[k].val = hvl[k].data();
        cdim[k] = dirs[k].ncp;
      }
      for (int k = 0; k < d; k++) T[k].free_all();
      return tg_kron3_csr(d, kd, cdim, row0, row1, col_offset, ncols, out);
    }
  }
  if (!rc) {
    P.row0 = row0;
    P.nrows = row1 - row0;
    const int64_t n0 = P.n[0];
    P.pencil0 = row0 / n0;
    const int64_t pencil1 = (row1 > row0) ? (row1 - 1) / n0 + 1 : P.pencil0;
    P.npencils = pencil1 - P.pencil0;
    P.chunks_per_pencil = (int32_t)tg_cdiv(n0, P.rows_per_block);
    const int64_t nblocks = P.npencils * P.chunks_per_pencil;
    if (nblocks >= (1ll << 31)) {
      tg_set_error("tg_extract_csr_tensor: grid too large (%lld blocks)", (long long)nblocks);
      rc = 2;
    } else if (d == 1)
      rc = tg_run_extract<1, false>(P, nblocks, ncols, out);
    else if (d == 2)
      rc = tg_run_extract<2, false>(P, nblocks, ncols, out);
    else
      rc = tg_run_extract<3, false>(P, nblocks, ncols, out);
  }
  hipStreamSynchronize(g_tg.stream);
  for (int k = 0; k < d; k++) T[k].free_all();
  return rc;
}

extern "C" int tg_extract_csr_points(int d, const tg_dir_t *dirs, int32_t col_offset, int64_t ncols, double eps,
                                     const double *x, int64_t nrows, tg_csr_t *out) {
  TG_REQUIRE_INIT();
  TG_REQUIRE(d >= 1 && d <= 3 && dirs && out && nrows >= 0 && (x || nrows == 0),
             "bad arguments to tg_extract_csr_points");
  tg_extract_params P;
  tg_fill_common(P, d, dirs, col_offset, eps);
  tg_dir_tables T[3];
  int rc = 0;
  // de-interleave coordinates on the host side of the ABI: x[row*d + k] -> per-direction arrays
  std::vector<double> xk((size_t)(nrows > 0 ? nrows : 1));
  for (int k = 0; k < d && !rc; k++) {
    for (int64_t r = 0; r < nrows; r++) xk[r] = x[r * d + k];
    rc = tg_build_dir_table(dirs[k], xk.data(), nrows, 1, false, &T[k], nullptr, 1);
    if (!rc && hipStreamSynchronize(g_tg.stream) != hipSuccess) rc = 1;  // xk is reused
    P.idx[k] = T[k].idx;
    P.val[k] = T[k].val;
    P.n[k] = nrows;
  }
  if (!rc) {
    P.row0 = 0;
    P.nrows = nrows;
    const int64_t nblocks = tg_cdiv(nrows, P.rows_per_block);
    if (d == 1)
      rc = tg_run_extract<1, true>(P, nblocks, ncols, out);
    else if (d == 2)
      rc = tg_run_extract<2, true>(P, nblocks, ncols, out);
    else
      rc = tg_run_extract<3, true>(P, nblocks, ncols, out);
  }
  hipStreamSynchronize(g_tg.stream);
  for (int k = 0; k < d; k++) T[k].free_all();
  return rc;
}

// ----------------------------------------------------------------------------------------
// Transposed extraction: rows of M^T (one per spline basis function) written directly, as the
// Kronecker product of the transposed 1-D evaluation tables with the same (Nu*Nv)*Nw
// arithmetic and the same |v| > eps filter -- so M^T is bit-identical to transposing M, at
// the cost of one more streaming write instead of a scatter + per-row sort.
// ----------------------------------------------------------------------------------------
int tg_kron_build_rect(int d, int nterms, const tg_kron_dir_t *dirs, const int64_t *cdim, int64_t row0, int64_t row1,
                       int filter, double eps, int64_t col_offset, int64_t ncols_total, tg_csr_t *out);

extern "C" int tg_extract_csr_tensor_t(int d, const tg_dir_t *dirs, int64_t fe_row_offset, int64_t fe_rows_total,
                                       double eps, int64_t dof0, int64_t dof1, tg_csr_t *out) {
  TG_REQUIRE_INIT();
  TG_REQUIRE(d >= 1 && d <= 3 && dirs && out, "bad arguments to tg_extract_csr_tensor_t");
  std::vector<int32_t> rp[3], cl[3];
  std::vector<double> vl[3];
  tg_kron_dir_t kd[3];
  int64_t cdim[3] = {1, 1, 1};
  for (int k = 0; k < d; k++) {
    TG_TRY(tg_check_dir(dirs[k]));
    const int64_t n = dirs[k].nnodes;
    const int pp1 = dirs[k].p + 1;
    const int ncp = dirs[k].ncp;
    TG_REQUIRE(n >= 1 && dirs[k].nodes, "direction %d has no nodes", k);
    std::vector<int32_t> idx((size_t)n * pp1);
    std::vector<double> val((size_t)n * pp1);
    TG_TRY(tg_eval_basis_1d(&dirs[k], dirs[k].nodes, n, nullptr, idx.data(), val.data()));
    // transpose the 1-D table on the host (a few thousand entries); exact zeros cannot
    // survive the product filter and are dropped here
    rp[k].assign((size_t)ncp + 1, 0);
    for (int64_t a = 0; a < n; a++)
      for (int r = 0; r < pp1; r++)
        if (val[a * pp1 + r] != 0.0) rp[k][(size_t)idx[a * pp1 + r] + 1]++;
    for (int i = 0; i < ncp; i++) rp[k][i + 1] += rp[k][i];
    cl[k].resize((size_t)rp[k][ncp]);
    vl[k].resize((size_t)rp[k][ncp]);
    std::vector<int32_t> cur(rp[k].begin(), rp[k].end() - 1);
    for (int64_t a = 0; a < n; a++)       // ascending node index => sorted columns
      for (int r = 0; r < pp1; r++)
        if (val[a * pp1 + r] != 0.0) {
          const int i = idx[a * pp1 + r];
          cl[k][cur[i]] = (int32_t)a;
          vl[k][cur[i]] = val[a * pp1 + r];
          cur[i]++;
        }
    kd[k].n = ncp;
    kd[k].rowptr = rp[k].data();
    kd[k].col = cl[k].data();
    kd[k].val = vl[k].data();
    cdim[k] = n;
  }
  return tg_kron_build_rect(d, 1, kd, cdim, dof0, dof1, 1, eps, fe_row_offset, fe_rows_total, out);
}

// ----------------------------------------------------------------------------------------
// vstack (multi-field M = one row block per field, tIGAr/common.py:1546-1573)
// ----------------------------------------------------------------------------------------
__global__ void k_copy_rowptr_shift(int64_t *dst, const int64_t *src, int64_t n, int64_t shift) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) dst[i] = src[i] + shift;
}

extern "C" int tg_csr_vstack(int nblocks, const tg_csr_t *blocks, tg_csr_t *out) {
  TG_REQUIRE_INIT();
  TG_REQUIRE(nblocks >= 1 && blocks && out, "bad arguments to tg_csr_vstack");
  int64_t nrows = 0, nnz = 0, ncols = blocks[0]->ncols;
  int nloose = 0;
  for (int b = 0; b < nblocks; b++) {
    TG_REQUIRE(blocks[b] && blocks[b]->ncols == ncols, "vstack: column count mismatch");
    nrows += blocks[b]->nrows;
    nnz += blocks[b]->nnz;
    nloose += blocks[b]->rowcnt ? 1 : 0;
  }
  TG_REQUIRE(nloose == 0 || nloose == nblocks, "vstack: loose-row and canonical blocks cannot be mixed");
  for (int b = 0; b < nblocks; b++) TG_REQUIRE(!blocks[b]->rowptr_val, "vstack: views cannot be stacked");
  tg_csr_s *m = nullptr;
  TG_TRY(tg_csr_alloc(nrows, ncols, nnz, &m));
  if (nloose && tg_dmalloc(&m->rowcnt, nrows)) {
    tg_csr_destroy(m);
    return 1;
  }
  int64_t r = 0, z = 0;
  for (int b = 0; b < nblocks; b++) {
    const tg_csr_s *s = blocks[b];
    // (loose rows: the blocks' entry arrays are concatenated as they are, row starts shift with them)
    hipLaunchKernelGGL(k_copy_rowptr_shift, dim3(tg_grid_1d(s->nrows + 1, 256)), dim3(256), 0, g_tg.stream,
                       m->rowptr + r, s->rowptr, s->nrows + 1, z);
    if (nloose && s->nrows)
      hipMemcpyAsync(m->rowcnt + r, s->rowcnt, (size_t)s->nrows * sizeof(int32_t), hipMemcpyDeviceToDevice, g_tg.stream);
    if (s->nnz) {
      hipMemcpyAsync(m->col + z, s->col, (size_t)s->nnz * sizeof(int32_t), hipMemcpyDeviceToDevice, g_tg.stream);
      hipMemcpyAsync(m->val + z, s->val, (size_t)s->nnz * sizeof(double), hipMemcpyDeviceToDevice, g_tg.stream);
    }
    r += s->nrows;
    z += s->nnz;
  }
  if (nloose) {   // end marker of the last block = entries in use
    const int64_t used = z;
    hipMemcpyAsync(m->rowptr + nrows, &used, sizeof(int64_t), hipMemcpyHostToDevice, g_tg.stream);
  }
  TG_LAUNCH_CHECK();
  TG_CHECK_HIP(hipStreamSynchronize(g_tg.stream));
  *out = m;
  return 0;
}

// Row tables of a stacked matrix WITHOUT copying entries: row r of block b becomes the loose row
// (first-block-relative offset of its entries in col / in val, length).  The intermediate stage
// results of neighbouring sub-slabs are stacked like this for the last PtAP stage; only 20 B per row
// are written instead of 12 B per entry.
__global__ void k_view_rows(const int64_t *__restrict__ rp, const int32_t *__restrict__ rc, int64_t n, int64_t dcol,
                            int64_t dval, int64_t *__restrict__ orp, int64_t *__restrict__ orv, int32_t *__restrict__ orc) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    const int64_t a = rp[i];
    orp[i] = a + dcol;
    orv[i] = a + dval;
    orc[i] = rc ? rc[i] : (int32_t)(rp[i + 1] - a);
  }
}

extern "C" int tg_csr_vstack_view(int nblocks, const tg_csr_t *blocks, tg_csr_t *out) {
  TG_REQUIRE_INIT();
  TG_REQUIRE(nblocks >= 1 && blocks && out, "bad arguments to tg_csr_vstack_view");
  int64_t nrows = 0, nnz = 0, ncols = blocks[0]->ncols;
  for (int b = 0; b < nblocks; b++) {
    TG_REQUIRE(blocks[b] && blocks[b]->ncols == ncols, "vstack view: column count mismatch");
    TG_REQUIRE(!blocks[b]->rowptr_val, "vstack view: a view cannot be stacked again");
    nrows += blocks[b]->nrows;
    nnz += blocks[b]->nnz;
  }
  tg_csr_s *m = new tg_csr_s();
  m->nrows = nrows;
  m->ncols = ncols;
  m->nnz = nnz;
  m->view = true;
  m->col = blocks[0]->col;
  m->val = blocks[0]->val;
  if (tg_dmalloc(&m->rowptr, nrows + 1) || tg_dmalloc(&m->rowptr_val, nrows + 1) || tg_dmalloc(&m->rowcnt, nrows)) {
    tg_csr_destroy(m);
    return 1;
  }
  int64_t r = 0;
  for (int b = 0; b < nblocks; b++) {
    const tg_csr_s *s = blocks[b];
    const int64_t dcol = s->col - m->col, dval = s->val - m->val;     // element offsets (may be negative)
    if (s->nrows)
      hipLaunchKernelGGL(k_view_rows, dim3(tg_grid_1d(s->nrows, 256)), dim3(256), 0, g_tg.stream, s->rowptr,
                         (const int32_t *)s->rowcnt, s->nrows, dcol, dval, m->rowptr + r, m->rowptr_val + r, m->rowcnt + r);
    r += s->nrows;
  }
  TG_LAUNCH_CHECK();
  TG_CHECK_HIP(hipStreamSynchronize(g_tg.stream));
  *out = m;
  return 0;
}

// ----------------------------------------------------------------------------------------
// incremental vstack
// ----------------------------------------------------------------------------------------
extern "C" int tg_csr_builder_create(int64_t nrows_total, int64_t ncols, int64_t nnz_capacity,
                                     tg_csr_builder_t *out) {
  TG_REQUIRE_INIT();
  TG_REQUIRE(nrows_total >= 0 && ncols >= 0 && nnz_capacity >= 0 && out, "bad arguments to tg_csr_builder_create");
  tg_csr_builder_s *b = new tg_csr_builder_s();
  if (tg_csr_alloc(nrows_total, ncols, nnz_capacity, &b->m)) {
    delete b;
    return 1;
  }
  b->cap = nnz_capacity;
  b->m->nnz = 0;
  TG_CHECK_HIP(hipMemsetAsync(b->m->rowptr, 0, sizeof(int64_t), g_tg.stream));
  *out = b;
  return 0;
}

int tg_csr_builder_reserve(tg_csr_builder_s *b, int64_t nrows, int64_t nnz) {
  TG_REQUIRE(b && b->m, "null builder");
  TG_REQUIRE(b->rows_done + nrows <= b->m->nrows, "builder: more rows appended than announced");
  if (b->nnz_done + nnz > b->cap) {
    // grow: new arrays, copy what is there
    const int64_t ncap = std::max<int64_t>(b->nnz_done + nnz, b->cap + b->cap / 4 + 1024);
    int32_t *ncol = nullptr;
    double *nval = nullptr;
    TG_TRY(tg_dmalloc(&ncol, ncap + TG_CSR_PAD));
    if (tg_dmalloc(&nval, ncap + TG_CSR_PAD)) {
      tg_dfree(ncol);
      return 1;
    }
    if (b->nnz_done) {
      hipMemcpyAsync(ncol, b->m->col, (size_t)b->nnz_done * sizeof(int32_t), hipMemcpyDeviceToDevice, g_tg.stream);
      hipMemcpyAsync(nval, b->m->val, (size_t)b->nnz_done * sizeof(double), hipMemcpyDeviceToDevice, g_tg.stream);
    }
    hipStreamSynchronize(g_tg.stream);
    tg_dfree(b->m->col);
    tg_dfree(b->m->val);
    b->m->col = ncol;
    b->m->val = nval;
    b->cap = ncap;
  }
  return 0;
}

extern "C" int tg_csr_builder_append(tg_csr_builder_t b, tg_csr_t blk) {
  TG_REQUIRE_INIT();
  TG_REQUIRE(b && b->m && blk, "null argument to tg_csr_builder_append");
  TG_REQUIRE_CANONICAL(blk);
  TG_REQUIRE(blk->ncols == b->m->ncols, "builder: column count mismatch");
  TG_TRY(tg_csr_builder_reserve(b, blk->nrows, blk->nnz));
  if (blk->nrows) {
    // rowptr entries 1..nrows of the block, shifted (entry 0 of the builder's range is already set)
    hipLaunchKernelGGL(k_copy_rowptr_shift, dim3(tg_grid_1d(blk->nrows, 256)), dim3(256), 0, g_tg.stream,
                       b->m->rowptr + b->rows_done + 1, blk->rowptr + 1, blk->nrows, b->nnz_done);
  }
  if (blk->nnz) {
    hipMemcpyAsync(b->m->col + b->nnz_done, blk->col, (size_t)blk->nnz * sizeof(int32_t), hipMemcpyDeviceToDevice,
                   g_tg.stream);
    hipMemcpyAsync(b->m->val + b->nnz_done, blk->val, (size_t)blk->nnz * sizeof(double), hipMemcpyDeviceToDevice,
                   g_tg.stream);
  }
  TG_LAUNCH_CHECK();
  b->rows_done += blk->nrows;
  b->nnz_done += blk->nnz;
  return 0;
}

// abandons a builder that was not finished (its matrix is released)
extern "C" int tg_csr_builder_destroy(tg_csr_builder_t b) {
  if (!b) return 0;
  if (b->m) tg_csr_destroy(b->m);
  delete b;
  return 0;
}

extern "C" int tg_csr_builder_finish(tg_csr_builder_t b, tg_csr_t *out) {
  TG_REQUIRE_INIT();
  TG_REQUIRE(b && b->m && out, "null argument to tg_csr_builder_finish");
  TG_REQUIRE(b->rows_done == b->m->nrows, "builder: %lld of %lld rows appended", (long long)b->rows_done,
             (long long)b->m->nrows);
  b->m->nnz = b->nnz_done;
  TG_CHECK_HIP(hipStreamSynchronize(g_tg.stream));
  *out = b->m;
  delete b;
  return 0;
}

// ----------------------------------------------------------------------------------------
// triplet fallback for arbitrary scalar-basis plug-ins (host sort; the Python loop that
// feeds it dominates).  INSERT semantics: the last (row,col) write wins.
// ----------------------------------------------------------------------------------------
extern "C" int tg_csr_from_triplets(int64_t nrows, int64_t ncols, int64_t nt, const int64_t *rows,
                                    const int32_t *cols, const double *vals, double eps, tg_csr_t *out) {
  TG_REQUIRE_INIT();
  TG_REQUIRE(nrows >= 0 && ncols >= 0 && nt >= 0 && out, "bad arguments to tg_csr_from_triplets");
  std::vector<int64_t> order;
  order.reserve((size_t)nt);
  for (int64_t t = 0; t < nt; t++) {
    TG_REQUIRE(rows[t] >= 0 && rows[t] < nrows && cols[t] >= 0 && cols[t] < ncols, "triplet %lld out of range",
               (long long)t);
    if (fabs(vals[t]) > eps) order.push_back(t);
  }
  std::stable_sort(order.begin(), order.end(), [&](int64_t a, int64_t b) {
    if (rows[a] != rows[b]) return rows[a] < rows[b];
    return cols[a] < cols[b];
  });
  std::vector<int64_t> rowptr((size_t)nrows + 1, 0);
  std::vector<int32_t> c;
  std::vector<double> v;
  c.reserve(order.size());
  v.reserve(order.size());
  for (size_t q = 0; q < order.size(); q++) {
    const int64_t t = order[q];
    const bool dup = q + 1 < order.size() && rows[order[q + 1]] == rows[t] && cols[order[q + 1]] == cols[t];
    if (dup) continue;  // a later INSERT overwrites this one
    c.push_back(cols[t]);
    v.push_back(vals[t]);
    rowptr[(size_t)rows[t] + 1]++;
  }
  for (int64_t r = 0; r < nrows; r++) rowptr[(size_t)r + 1] += rowptr[(size_t)r];
  return tg_csr_from_host(nrows, ncols, rowptr.data(), c.data(), v.data(), out);
}
