// Extraction rows of a spline given by element-wise Bezier extraction operators (Rhino T-splines,
// tIGAr/RhinoTSplines.py:37-137): on Bezier element e the basis functions are the rows of C_e applied to the
// tensor Bernstein basis of the element, N_a(xi) = sum_b C_e[a][b] B_b(xi), with global function indices
// nodes_e[a].  The reference evaluates this per FE node in Python loops (RhinoTSplineScalarBasisFuncs :37-60,
// getNodesAndEvals :122-137, then the generateM row loop of tIGAr/common.py:1554-1571); here one thread owns one
// FE row (element, local Lagrange node), evaluates the (q+1)^2 Bernstein values from a per-node table and walks
// the element's functions: count pass, scan, fill pass -- CSR written directly, abs(v) > eps filter as in
// generateM.  Columns of an element are handed over sorted (host), so rows come out in column order.
#include "tg_common.h"
#include <vector>

struct tg_bez_args {
  int64_t nrows;                 // nel * nloc
  int nloc, nbern;               // FE nodes per element, Bernstein functions per element ((q+1)^2)
  const double *bern;            // [nel][nloc][nbern] Bernstein values at the FE nodes of every element (the local
                                 // coordinate of a node depends on the element through rounding, as in the reference)
  const int64_t *eoff;           // [nel+1] offset of element e in `nodes` (functions a) and, times nbern, in `coef`
  const int32_t *nodes;          // global function indices, ascending within an element
  const double *coef;            // C_e[a][b], rows in the order of `nodes`
  double eps;
  int32_t col_offset;
};

template <bool FILL>
__global__ void __launch_bounds__(256)
    k_bezier_rows(tg_bez_args A, int64_t *__restrict__ rowptr, int32_t *__restrict__ col, double *__restrict__ val) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= A.nrows) return;
  const int64_t e = r / A.nloc;                 // element of this FE row (rows are element-major)
  const double *B = A.bern + r * A.nbern;
  const int64_t a0 = A.eoff[e], a1 = A.eoff[e + 1];
  int64_t pos = FILL ? rowptr[r] : 0;
  int64_t cnt = 0;
  for (int64_t a = a0; a < a1; a++) {
    const double *C = A.coef + a * A.nbern;
    double s = 0.0;
    for (int b = 0; b < A.nbern; b++) s += C[b] * B[b];     // same order as the reference's accumulation (:55-58)
    if (fabs(s) > A.eps) {
      if (FILL) {
        col[pos] = A.nodes[a] + A.col_offset;
        val[pos] = s;
        pos++;
      }
      cnt++;
    }
  }
  if (!FILL) rowptr[r] = cnt;
}

extern "C" int tg_extract_csr_bezier(int64_t nel, int nloc, int nbern, const double *bern, const int64_t *eoff,
                                     const int32_t *nodes, const double *coef, int32_t col_offset, int64_t ncols,
                                     double eps, tg_csr_t *out) {
  TG_REQUIRE_INIT();
  TG_REQUIRE(nel >= 0 && nloc >= 1 && nbern >= 1 && bern && eoff && nodes && coef && out && ncols >= 0,
             "bad arguments to tg_extract_csr_bezier");
  const int64_t nfun = eoff[nel], nrows = nel * nloc;
  for (int64_t e = 0; e < nel; e++) {
    TG_REQUIRE(eoff[e + 1] >= eoff[e], "tg_extract_csr_bezier: element offsets must not decrease");
    for (int64_t a = eoff[e]; a < eoff[e + 1]; a++) {
      TG_REQUIRE(nodes[a] >= 0 && nodes[a] + (int64_t)col_offset < ncols, "tg_extract_csr_bezier: function index out of range");
      TG_REQUIRE(a == eoff[e] || nodes[a] > nodes[a - 1], "tg_extract_csr_bezier: functions of element %lld are not in "
                 "ascending order", (long long)e);
    }
  }
  double *d_bern = nullptr, *d_coef = nullptr;
  int64_t *d_eoff = nullptr, *rowptr = nullptr;
  int32_t *d_nodes = nullptr;
  tg_csr_s *m = nullptr;
  int rc = tg_dmalloc(&d_bern, nrows * nbern) || tg_dmalloc(&d_coef, nfun * nbern) || tg_dmalloc(&d_eoff, nel + 1) ||
           tg_dmalloc(&d_nodes, nfun) || tg_dmalloc(&rowptr, nrows + 1);
  if (!rc) {
    if (nrows) hipMemcpyAsync(d_bern, bern, (size_t)(nrows * nbern) * sizeof(double), hipMemcpyHostToDevice, g_tg.stream);
    if (nfun) {
      hipMemcpyAsync(d_coef, coef, (size_t)(nfun * nbern) * sizeof(double), hipMemcpyHostToDevice, g_tg.stream);
      hipMemcpyAsync(d_nodes, nodes, (size_t)nfun * sizeof(int32_t), hipMemcpyHostToDevice, g_tg.stream);
    }
    hipMemcpyAsync(d_eoff, eoff, (size_t)(nel + 1) * sizeof(int64_t), hipMemcpyHostToDevice, g_tg.stream);
    tg_bez_args A;
    A.nrows = nrows;
    A.nloc = nloc;
    A.nbern = nbern;
    A.bern = d_bern;
    A.eoff = d_eoff;
    A.nodes = d_nodes;
    A.coef = d_coef;
    A.eps = eps;
    A.col_offset = col_offset;
    const unsigned grid = (unsigned)std::max<int64_t>(1, tg_cdiv(nrows, 256));
    hipLaunchKernelGGL((k_bezier_rows<false>), dim3(grid), dim3(256), 0, g_tg.stream, A, rowptr, (int32_t *)nullptr,
                       (double *)nullptr);
    int64_t nnz = 0;
    rc = tg_exclusive_scan_i64(rowptr, nrows, &nnz);
    if (!rc) rc = tg_csr_alloc(nrows, ncols, nnz, &m);
    if (!rc) {
      tg_dfree(m->rowptr);
      m->rowptr = rowptr;
      rowptr = nullptr;
      hipLaunchKernelGGL((k_bezier_rows<true>), dim3(grid), dim3(256), 0, g_tg.stream, A, m->rowptr, m->col, m->val);
      if (hipGetLastError() != hipSuccess || hipStreamSynchronize(g_tg.stream) != hipSuccess) {
        tg_set_error("tg_extract_csr_bezier: kernel failed");
        rc = 1;
      }
    }
  }
  tg_dfree(d_bern);
  tg_dfree(d_coef);
  tg_dfree(d_eoff);
  tg_dfree(d_nodes);
  tg_dfree(rowptr);
  if (rc) {
    if (m) tg_csr_destroy(m);
    return rc;
  }
  *out = m;
  return 0;
}
