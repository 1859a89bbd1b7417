/*
 * TEST INFRASTRUCTURE / CPU BASELINE ONLY -- never imported by the product path (tigar_amd/).
 *
 * Plain C + OpenMP restatement of the heavy loops of tIGAr's extraction hot path, the same CSR
 * algorithms PETSc AIJ runs on the CPU [ext]:
 *   tgo_generate_M   generateM on the tensor node grid: one row per FE node, value (Nu*Nv)*Nw,
 *                    entries with abs(v) > eps, columns sorted        tIGAr/common.py:1516-1578
 *   tgo_ptap         K = M^T A M by two Gustavson products (A*M, then M^T*(AM)) + MatZeroRowsColumns
 *                                                                       tIGAr/common.py:1176-1204
 *   tgo_spmv_t       y = M^T b                                          tIGAr/common.py:97-109
 *   tgo_cg_jacobi    PETSc KSPCG + PCJACOBI semantics (zero initial guess, preconditioned residual
 *                    norm, rtol/atol/maxit)                             tIGAr/common.py:1236-1263
 * It is pinned against the numpy oracle (oracle/tigar_oracle.py, itself pinned bit-exactly against
 * the reference's own getNodesAndEvals through the golden fixtures) by tests/test_oracle_c.py: M
 * bit-exact, K to 1e-13, CG iteration counts equal.  a-8...a-12 remain "parity unpinned" against
 * PETSc itself (absent).  Two-call protocol: a NULL output array means "count only".
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#else
static int omp_get_max_threads(void) { return 1; }
static int omp_get_thread_num(void) { return 0; }
#endif

int tgo_num_threads(void) { return omp_get_max_threads(); }
void tgo_set_threads(int n) {
#ifdef _OPENMP
  if (n >= 1) omp_set_num_threads(n);
#else
  (void)n;
#endif
}

/* ---- generateM on the tensor node grid.  idx/val: per direction k a table [n[k]][p1[k]] of the 1-D
 * nodes and values (BSpline1.getNodes / basisFuncs at the FE nodes of that direction). */
int64_t tgo_generate_M(int d, const int64_t *n, const int32_t *p1, const int32_t *const *idx, const double *const *val,
                       const int64_t *ncp, double eps, int64_t *rowptr, int32_t *col, double *v) {
  int64_t n0 = n[0], n1 = d > 1 ? n[1] : 1, n2 = d > 2 ? n[2] : 1;
  int q0 = p1[0], q1 = d > 1 ? p1[1] : 1, q2 = d > 2 ? p1[2] : 1;
  int64_t nrows = n0 * n1 * n2;
  const int cand = q0 * q1 * q2;
  /* pass 1: row lengths */
#pragma omp parallel for schedule(static)
  for (int64_t r = 0; r < nrows; r++) {
    int64_t a = r % n0, b = (r / n0) % n1, c = r / (n0 * n1);
    int cnt = 0;
    for (int i = 0; i < q0; i++)
      for (int j = 0; j < q1; j++)
        for (int k = 0; k < q2; k++) {
          double x = val[0][a * q0 + i];
          if (d > 1) x = x * val[1][b * q1 + j];
          if (d > 2) x = x * val[2][c * q2 + k];
          if (fabs(x) > eps) cnt++;
        }
    rowptr[r + 1] = cnt;
  }
  rowptr[0] = 0;
  for (int64_t r = 0; r < nrows; r++) rowptr[r + 1] += rowptr[r];
  if (!col || !v) return rowptr[nrows];
  /* pass 2: entries in the reference's order (i outer, k innermost), then sorted by column as PETSc
   * stores them */
#pragma omp parallel
  {
    int32_t *tc = (int32_t *)malloc(sizeof(int32_t) * (size_t)cand);
    double *tv = (double *)malloc(sizeof(double) * (size_t)cand);
#pragma omp for schedule(static)
    for (int64_t r = 0; r < nrows; r++) {
      int64_t a = r % n0, b = (r / n0) % n1, c = r / (n0 * n1);
      int cnt = 0;
      for (int i = 0; i < q0; i++)
        for (int j = 0; j < q1; j++)
          for (int k = 0; k < q2; k++) {
            double x = val[0][a * q0 + i];
            int64_t cc = idx[0][a * q0 + i];
            if (d > 1) {
              x = x * val[1][b * q1 + j];
              cc += ncp[0] * idx[1][b * q1 + j];
            }
            if (d > 2) {
              x = x * val[2][c * q2 + k];
              cc += ncp[0] * ncp[1] * idx[2][c * q2 + k];
            }
            if (fabs(x) > eps) {
              tc[cnt] = (int32_t)cc;
              tv[cnt] = x;
              cnt++;
            }
          }
      /* insertion sort by column (<= (p+1)^d entries); a repeated column keeps the LAST insert */
      for (int s = 1; s < cnt; s++) {
        int32_t kc = tc[s];
        double kv = tv[s];
        int t = s - 1;
        while (t >= 0 && tc[t] > kc) {
          tc[t + 1] = tc[t];
          tv[t + 1] = tv[t];
          t--;
        }
        tc[t + 1] = kc;
        tv[t + 1] = kv;
      }
      int64_t o = rowptr[r];
      for (int s = 0; s < cnt; s++) {
        col[o + s] = tc[s];
        v[o + s] = tv[s];
      }
    }
    free(tc);
    free(tv);
  }
  return rowptr[nrows];
}

/* ---- CSR transpose (counting sort; rows of the result come out with sorted columns) */
static void csr_transpose(int64_t nr, int64_t nc, const int64_t *rp, const int32_t *ci, const double *cv, int64_t *trp,
                          int32_t *tci, double *tcv) {
  memset(trp, 0, sizeof(int64_t) * (size_t)(nc + 1));
  for (int64_t q = 0; q < rp[nr]; q++) trp[ci[q] + 1]++;
  for (int64_t c = 0; c < nc; c++) trp[c + 1] += trp[c];
  int64_t *cur = (int64_t *)malloc(sizeof(int64_t) * (size_t)(nc + 1));
  memcpy(cur, trp, sizeof(int64_t) * (size_t)(nc + 1));
  for (int64_t r = 0; r < nr; r++)
    for (int64_t q = rp[r]; q < rp[r + 1]; q++) {
      int64_t o = cur[ci[q]]++;
      tci[o] = (int32_t)r;
      tcv[o] = cv[q];
    }
  free(cur);
}

static int cmp_i32(const void *a, const void *b) {
  int32_t x = *(const int32_t *)a, y = *(const int32_t *)b;
  return (x > y) - (x < y);
}

/* ---- C = X * Y (Gustavson, row-wise, dense accumulator + marker per thread), columns sorted.
 * Two passes; returns nnz(C).  crp must hold nrx+1 entries. */
static int64_t spgemm_off(int64_t nrx, int64_t ncy, const int64_t *xrp, const int32_t *xci, const double *xcv,
                          const int64_t *yrp, const int32_t *yci, const double *ycv, int64_t yoff, int64_t *crp,
                          int32_t **cci_out, double **ccv_out);
static int64_t spgemm(int64_t nrx, int64_t ncy, const int64_t *xrp, const int32_t *xci, const double *xcv, const int64_t *yrp,
                      const int32_t *yci, const double *ycv, int64_t *crp, int32_t **cci_out, double **ccv_out) {
  return spgemm_off(nrx, ncy, xrp, xci, xcv, yrp, yci, ycv, 0, crp, cci_out, ccv_out);
}
/* rows of X index the rows of Y shifted by yoff (Y holds the rows yoff, yoff+1, ... of a larger matrix); the row
 * pointer of X may be a slice of a larger one (entries are addressed by xrp[r] .. xrp[r+1] as they stand) */
static int64_t spgemm_off(int64_t nrx, int64_t ncy, const int64_t *xrp, const int32_t *xci, const double *xcv,
                          const int64_t *yrp, const int32_t *yci, const double *ycv, int64_t yoff, int64_t *crp,
                          int32_t **cci_out, double **ccv_out) {
#pragma omp parallel
  {
    int64_t *mark = (int64_t *)malloc(sizeof(int64_t) * (size_t)ncy);
    for (int64_t c = 0; c < ncy; c++) mark[c] = -1;
#pragma omp for schedule(dynamic, 256)
    for (int64_t r = 0; r < nrx; r++) {
      int64_t cnt = 0;
      for (int64_t q = xrp[r]; q < xrp[r + 1]; q++) {
        int64_t k = (int64_t)xci[q] - yoff;
        for (int64_t t = yrp[k]; t < yrp[k + 1]; t++)
          if (mark[yci[t]] != r) {
            mark[yci[t]] = r;
            cnt++;
          }
      }
      crp[r + 1] = cnt;
    }
    free(mark);
  }
  crp[0] = 0;
  for (int64_t r = 0; r < nrx; r++) crp[r + 1] += crp[r];
  int64_t nnz = crp[nrx];
  int32_t *cci = (int32_t *)malloc(sizeof(int32_t) * (size_t)(nnz > 0 ? nnz : 1));
  double *ccv = (double *)malloc(sizeof(double) * (size_t)(nnz > 0 ? nnz : 1));
#pragma omp parallel
  {
    int64_t *mark = (int64_t *)malloc(sizeof(int64_t) * (size_t)ncy);
    double *acc = (double *)malloc(sizeof(double) * (size_t)ncy);
    for (int64_t c = 0; c < ncy; c++) mark[c] = -1;
#pragma omp for schedule(dynamic, 256)
    for (int64_t r = 0; r < nrx; r++) {
      int64_t o = crp[r], cnt = 0;
      for (int64_t q = xrp[r]; q < xrp[r + 1]; q++) {
        int64_t k = (int64_t)xci[q] - yoff;
        double xv = xcv[q];
        for (int64_t t = yrp[k]; t < yrp[k + 1]; t++) {
          int32_t c = yci[t];
          if (mark[c] != r) {
            mark[c] = r;
            acc[c] = xv * ycv[t];
            cci[o + cnt++] = c;
          } else
            acc[c] += xv * ycv[t];
        }
      }
      qsort(cci + o, (size_t)cnt, sizeof(int32_t), cmp_i32);
      for (int64_t s = 0; s < cnt; s++) ccv[o + s] = acc[cci[o + s]];
    }
    free(mark);
    free(acc);
  }
  *cci_out = cci;
  *ccv_out = ccv;
  return nnz;
}

/* ---- K = M^T A M, then rows and columns of zero_dofs zeroed with diag on the diagonal.
 * M: nfe x ncp, A: nfe x nfe.  Call once with kcol == NULL to get nnz(K) (the product is kept in
 * a static cache until the second call fetches it). */
static int64_t *g_krp = NULL;
static int32_t *g_kci = NULL;
static double *g_kcv = NULL;
static int64_t g_knr = 0;

int64_t tgo_ptap(int64_t nfe, int64_t ncp, const int64_t *mrp, const int32_t *mci, const double *mcv, const int64_t *arp,
                 const int32_t *aci, const double *acv, const int32_t *zero_dofs, int64_t nzero, double diag, int64_t *krp,
                 int32_t *kcol, double *kval) {
  if (!kcol || !kval) {
    free(g_krp);
    free(g_kci);
    free(g_kcv);
    /* AM = A * M */
    int64_t *amrp = (int64_t *)malloc(sizeof(int64_t) * (size_t)(nfe + 1));
    int32_t *amci;
    double *amcv;
    spgemm(nfe, ncp, arp, aci, acv, mrp, mci, mcv, amrp, &amci, &amcv);
    /* M^T */
    int64_t mnnz = mrp[nfe];
    int64_t *trp = (int64_t *)malloc(sizeof(int64_t) * (size_t)(ncp + 1));
    int32_t *tci = (int32_t *)malloc(sizeof(int32_t) * (size_t)(mnnz > 0 ? mnnz : 1));
    double *tcv = (double *)malloc(sizeof(double) * (size_t)(mnnz > 0 ? mnnz : 1));
    csr_transpose(nfe, ncp, mrp, mci, mcv, trp, tci, tcv);
    g_krp = (int64_t *)malloc(sizeof(int64_t) * (size_t)(ncp + 1));
    spgemm(ncp, ncp, trp, tci, tcv, amrp, amci, amcv, g_krp, &g_kci, &g_kcv);
    g_knr = ncp;
    free(amrp);
    free(amci);
    free(amcv);
    free(trp);
    free(tci);
    free(tcv);
    /* MatZeroRowsColumns: entries stay in the pattern, values zeroed, diag on the diagonal */
    if (nzero > 0) {
      unsigned char *mask = (unsigned char *)calloc((size_t)ncp, 1);
      for (int64_t z = 0; z < nzero; z++) mask[zero_dofs[z]] = 1;
#pragma omp parallel for schedule(static)
      for (int64_t r = 0; r < ncp; r++)
        for (int64_t q = g_krp[r]; q < g_krp[r + 1]; q++) {
          int32_t c = g_kci[q];
          if (mask[r] || mask[c]) g_kcv[q] = (mask[r] && c == r) ? diag : 0.0;
        }
      free(mask);
    }
    if (krp) memcpy(krp, g_krp, sizeof(int64_t) * (size_t)(ncp + 1));
    return g_krp[ncp];
  }
  if (!g_krp || g_knr != ncp) return -1;
  memcpy(krp, g_krp, sizeof(int64_t) * (size_t)(ncp + 1));
  memcpy(kcol, g_kci, sizeof(int32_t) * (size_t)g_krp[ncp]);
  memcpy(kval, g_kcv, sizeof(double) * (size_t)g_krp[ncp]);
  free(g_krp);
  free(g_kci);
  free(g_kcv);
  g_krp = NULL;
  g_kci = NULL;
  g_kcv = NULL;
  return 0;
}

/* ---- the same product with the intermediate A*M bounded: rows of K in blocks; for a block the FE rows its M^T rows
 * reference form a range [f0, f1) (contiguous z-slabs on a tensor patch), (A M) is computed for that range only, then
 * M^T[block] * (A M)[f0:f1].  Blocks are cut so that an upper bound of the entries of (A M)[f0:f1] stays below
 * max_am_entries.  Same two-call protocol and result as tgo_ptap (rows recomputed in overlapping ranges cost time, not
 * accuracy: every row of K is produced once, in one block). */
int64_t tgo_ptap_blocked(int64_t nfe, int64_t ncp, const int64_t *mrp, const int32_t *mci, const double *mcv,
                         const int64_t *arp, const int32_t *aci, const double *acv, const int32_t *zero_dofs, int64_t nzero,
                         double diag, int64_t max_am_entries, int64_t *krp, int32_t *kcol, double *kval) {
  if (kcol && kval) {
    if (!g_krp || g_knr != ncp) return -1;
    memcpy(krp, g_krp, sizeof(int64_t) * (size_t)(ncp + 1));
    memcpy(kcol, g_kci, sizeof(int32_t) * (size_t)g_krp[ncp]);
    memcpy(kval, g_kcv, sizeof(double) * (size_t)g_krp[ncp]);
    free(g_krp);
    free(g_kci);
    free(g_kcv);
    g_krp = NULL;
    g_kci = NULL;
    g_kcv = NULL;
    return 0;
  }
  free(g_krp);
  free(g_kci);
  free(g_kcv);
  int64_t mnnz = mrp[nfe];
  int64_t *trp = (int64_t *)malloc(sizeof(int64_t) * (size_t)(ncp + 1));
  int32_t *tci = (int32_t *)malloc(sizeof(int32_t) * (size_t)(mnnz > 0 ? mnnz : 1));
  double *tcv = (double *)malloc(sizeof(double) * (size_t)(mnnz > 0 ? mnnz : 1));
  csr_transpose(nfe, ncp, mrp, mci, mcv, trp, tci, tcv);
  /* entries per row of A*M, estimated from the exact count of a sample of rows (a bound from the row lengths alone
   * overestimates a tensor patch by (p+1)^d: the supports of neighbouring nodes coincide) */
  double avg_am = 1.0;
  {
    const int64_t nsamp = nfe < 2000 ? nfe : 2000;
    int64_t *mark = (int64_t *)malloc(sizeof(int64_t) * (size_t)ncp);
    for (int64_t c = 0; c < ncp; c++) mark[c] = -1;
    int64_t total = 0;
    for (int64_t sidx = 0; sidx < nsamp; sidx++) {
      const int64_t f = (nfe * sidx) / nsamp;
      for (int64_t q = arp[f]; q < arp[f + 1]; q++) {
        const int32_t k = aci[q];
        for (int64_t t = mrp[k]; t < mrp[k + 1]; t++)
          if (mark[mci[t]] != f) {
            mark[mci[t]] = f;
            total++;
          }
      }
    }
    free(mark);
    avg_am = nsamp > 0 ? 1.15 * (double)total / (double)nsamp + 1.0 : 1.0;
  }
  int64_t *ub = (int64_t *)malloc(sizeof(int64_t) * (size_t)(nfe + 1));
  for (int64_t f = 0; f <= nfe; f++) ub[f] = (int64_t)(avg_am * (double)f);
  /* FE range referenced by every row of M^T (columns are sorted) */
  g_krp = (int64_t *)malloc(sizeof(int64_t) * (size_t)(ncp + 1));
  g_krp[0] = 0;
  int64_t cap = 1 << 20, used = 0;
  g_kci = (int32_t *)malloc(sizeof(int32_t) * (size_t)cap);
  g_kcv = (double *)malloc(sizeof(double) * (size_t)cap);
  int64_t r0 = 0;
  while (r0 < ncp) {
    int64_t f0 = nfe, f1 = 0, r1 = r0;
    while (r1 < ncp) {
      int64_t lo = f0, hi = f1;
      if (trp[r1 + 1] > trp[r1]) {
        if (tci[trp[r1]] < lo) lo = tci[trp[r1]];
        if ((int64_t)tci[trp[r1 + 1] - 1] + 1 > hi) hi = (int64_t)tci[trp[r1 + 1] - 1] + 1;
      }
      if (r1 > r0 && hi > lo && ub[hi] - ub[lo] > max_am_entries) break;
      f0 = lo;
      f1 = hi;
      r1++;
    }
    if (f1 > f0) {
      int64_t nb = f1 - f0;
      int64_t *amrp = (int64_t *)malloc(sizeof(int64_t) * (size_t)(nb + 1));
      int32_t *amci;
      double *amcv;
      spgemm(nb, ncp, arp + f0, aci, acv, mrp, mci, mcv, amrp, &amci, &amcv);
      int64_t *brp = (int64_t *)malloc(sizeof(int64_t) * (size_t)(r1 - r0 + 1));
      int32_t *bci;
      double *bcv;
      int64_t bnnz = spgemm_off(r1 - r0, ncp, trp + r0, tci, tcv, amrp, amci, amcv, f0, brp, &bci, &bcv);
      if (used + bnnz > cap) {
        while (used + bnnz > cap) cap *= 2;
        g_kci = (int32_t *)realloc(g_kci, sizeof(int32_t) * (size_t)cap);
        g_kcv = (double *)realloc(g_kcv, sizeof(double) * (size_t)cap);
      }
      memcpy(g_kci + used, bci, sizeof(int32_t) * (size_t)bnnz);
      memcpy(g_kcv + used, bcv, sizeof(double) * (size_t)bnnz);
      for (int64_t r = r0; r < r1; r++) g_krp[r + 1] = used + brp[r - r0 + 1];
      used += bnnz;
      free(amrp);
      free(amci);
      free(amcv);
      free(brp);
      free(bci);
      free(bcv);
    } else
      for (int64_t r = r0; r < r1; r++) g_krp[r + 1] = used;
    r0 = r1;
  }
  g_knr = ncp;
  free(trp);
  free(tci);
  free(tcv);
  free(ub);
  if (nzero > 0) {
    unsigned char *mask = (unsigned char *)calloc((size_t)ncp, 1);
    for (int64_t z = 0; z < nzero; z++) mask[zero_dofs[z]] = 1;
#pragma omp parallel for schedule(static)
    for (int64_t r = 0; r < ncp; r++)
      for (int64_t q = g_krp[r]; q < g_krp[r + 1]; q++) {
        int32_t c = g_kci[q];
        if (mask[r] || mask[c]) g_kcv[q] = (mask[r] && c == r) ? diag : 0.0;
      }
    free(mask);
  }
  if (krp) memcpy(krp, g_krp, sizeof(int64_t) * (size_t)(ncp + 1));
  return g_krp[ncp];
}

void tgo_spmv(int64_t nr, const int64_t *rp, const int32_t *ci, const double *cv, const double *x, double *y) {
#pragma omp parallel for schedule(static)
  for (int64_t r = 0; r < nr; r++) {
    double s = 0.0;
    for (int64_t q = rp[r]; q < rp[r + 1]; q++) s += cv[q] * x[ci[q]];
    y[r] = s;
  }
}

/* y = M^T b (scatter-add form, as MatMultTranspose does it), y zeroed at zero_dofs */
void tgo_spmv_t(int64_t nr, int64_t nc, const int64_t *rp, const int32_t *ci, const double *cv, const double *b,
                const int32_t *zero_dofs, int64_t nzero, double *y) {
  memset(y, 0, sizeof(double) * (size_t)nc);
  for (int64_t r = 0; r < nr; r++) {
    double br = b[r];
    for (int64_t q = rp[r]; q < rp[r + 1]; q++) y[ci[q]] += cv[q] * br;
  }
  for (int64_t z = 0; z < nzero; z++) y[zero_dofs[z]] = 0.0;
}

static double dotp(int64_t n, const double *a, const double *b) {
  double s = 0.0;
#pragma omp parallel for reduction(+ : s) schedule(static)
  for (int64_t i = 0; i < n; i++) s += a[i] * b[i];
  return s;
}

/* Jacobi-preconditioned CG with PETSc's defaults: zero initial guess, convergence on the
 * PRECONDITIONED residual norm ||B r|| <= max(rtol ||B b||, atol).  Returns the iteration count
 * (negative: not converged); *resid = last preconditioned residual norm. */
int tgo_cg_jacobi(int64_t n, const int64_t *rp, const int32_t *ci, const double *cv, const double *b, double rtol, double atol,
                  int maxit, double *x, double *resid) {
  double *dinv = (double *)malloc(sizeof(double) * (size_t)n), *r = (double *)malloc(sizeof(double) * (size_t)n);
  double *z = (double *)malloc(sizeof(double) * (size_t)n), *p = (double *)malloc(sizeof(double) * (size_t)n);
  double *w = (double *)malloc(sizeof(double) * (size_t)n);
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < n; i++) {
    double dd = 0.0;
    for (int64_t q = rp[i]; q < rp[i + 1]; q++)
      if (ci[q] == i) dd = cv[q];
    dinv[i] = dd != 0.0 ? 1.0 / dd : 1.0;
    x[i] = 0.0;
    r[i] = b[i];
    z[i] = dinv[i] * b[i];
    p[i] = z[i];
  }
  double rz = dotp(n, r, z);
  double znorm = sqrt(dotp(n, z, z));
  const double tol = fmax(rtol * znorm, atol);
  int it = 0, status = -1;
  if (znorm <= tol) status = 0;
  while (status < 0 && it < maxit) {
    tgo_spmv(n, rp, ci, cv, p, w);
    double pw = dotp(n, p, w);
    if (pw == 0.0) break;
    double alpha = rz / pw;
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; i++) {
      x[i] += alpha * p[i];
      r[i] -= alpha * w[i];
      z[i] = dinv[i] * r[i];
    }
    it++;
    znorm = sqrt(dotp(n, z, z));
    if (znorm <= tol) {
      status = 0;
      break;
    }
    double rz2 = dotp(n, r, z);
    double beta = rz2 / rz;
    rz = rz2;
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; i++) p[i] = z[i] + beta * p[i];
  }
  *resid = znorm;
  free(dinv);
  free(r);
  free(z);
  free(p);
  free(w);
  return status == 0 ? it : -it - 1;
}
