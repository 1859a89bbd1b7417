// Synthetic FE-side INPUT generator (not on the timed path):
// A = sum_t (x)_k F[t][k], a Kronecker-sum of 1-D CSR factors on the tensor FE node grid,
// e.g. the exact Q_p Laplace stiffness K1xM1xM1 + M1xK1xM1 + M1xM1xK1 that dolfin's
// assemble() would hand to extractMatrix (tIGAr/common.py:1206-1220).  Rows are z-slab
// ranges [row0,row1) in lexicographic node order (direction 0 fastest); columns global.
#include "tg_common.h"
// The next tg_kron_sum_csr call writes row pointer and columns only and leaves the values UNWRITTEN (one-shot switch, for
// callers inside the library that store every value themselves -- the sum-factorised mapped assembly: a third of the
// pattern kernel's bytes and time for nothing otherwise, 0.11 of 0.17 s per step at 256^3 p = 3)
static bool g_k3_skip_val = false;
void tg_kron_pattern_only(void) { g_k3_skip_val = true; }
#include <algorithm>
#include <chrono>
static double tg_now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define TG_TRACE(msg) do { if (getenv("TIGAR_TRACE")) fprintf(stderr, "[trace] %s %.3f ms\n", msg, (tg_now() - _t0) * 1e3); } while (0)

#define TG_KRON_MAX_TERMS 9

struct tg_kron_params {
  int d, nterms;
  int64_t n[3];
  const int32_t *rowptr[3];
  const int32_t *col[3];
  const double *val[3];  // term-major: val[k][t*nnz1d[k] + q]
  int64_t nnz1d[3];
  int64_t row0, nrows;
  int64_t pencil0, npencils;
  int filter;          // 1: keep only |v| > eps (single term), as generateM does
  double eps;
  int64_t col_offset;
  int64_t cstride[3];  // column strides of the product index
};

__device__ __forceinline__ void tg_kron_pencil(const tg_kron_params &P, int64_t pencil, int64_t *b, int64_t *c) {
  if (P.d == 1) {
    *b = 0;
    *c = 0;
  } else if (P.d == 2) {
    *b = pencil;
    *c = 0;
  } else {
    *b = pencil % P.n[1];
    *c = pencil / P.n[1];
  }
}

__global__ void __launch_bounds__(256) k_kron_count(tg_kron_params P, int64_t *__restrict__ rowptr) {
  const int64_t pencil = P.pencil0 + blockIdx.x;
  int64_t b, c;
  tg_kron_pencil(P, pencil, &b, &c);
  int64_t lyz = 1;
  if (P.d > 1) lyz *= P.rowptr[1][b + 1] - P.rowptr[1][b];
  if (P.d > 2) lyz *= P.rowptr[2][c + 1] - P.rowptr[2][c];
  for (int64_t a = threadIdx.x; a < P.n[0]; a += 256) {
    const int64_t lr = P.n[0] * pencil + a - P.row0;
    if (lr < 0 || lr >= P.nrows) continue;
    if (!P.filter) {
      rowptr[lr] = (int64_t)(P.rowptr[0][a + 1] - P.rowptr[0][a]) * lyz;
    } else {
      const int y0 = (P.d > 1) ? P.rowptr[1][b] : 0, y1 = (P.d > 1) ? P.rowptr[1][b + 1] : 1;
      const int z0 = (P.d > 2) ? P.rowptr[2][c] : 0, z1 = (P.d > 2) ? P.rowptr[2][c + 1] : 1;
      int cnt = 0;
      for (int k = z0; k < z1; k++)
        for (int j = y0; j < y1; j++)
          for (int i = P.rowptr[0][a]; i < P.rowptr[0][a + 1]; i++) {
            double v = P.val[0][i];
            if (P.d > 1) v *= P.val[1][j];
            if (P.d > 2) v *= P.val[2][k];
            cnt += (fabs(v) > P.eps) ? 1 : 0;
          }
      rowptr[lr] = cnt;
    }
  }
}

__global__ void __launch_bounds__(256)
    k_kron_fill(tg_kron_params P, const int64_t *__restrict__ rowptr, int32_t *__restrict__ col,
                double *__restrict__ val) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int64_t pencil = P.pencil0 + blockIdx.x;
  int64_t b, c;
  tg_kron_pencil(P, pencil, &b, &c);
  const int y0 = (P.d > 1) ? P.rowptr[1][b] : 0;
  const int ly = (P.d > 1) ? P.rowptr[1][b + 1] - y0 : 1;
  const int z0 = (P.d > 2) ? P.rowptr[2][c] : 0;
  const int lz = (P.d > 2) ? P.rowptr[2][c + 1] - z0 : 1;
  for (int64_t a = w; a < P.n[0]; a += 4) {
    const int64_t lr = P.n[0] * pencil + a - P.row0;
    if (lr < 0 || lr >= P.nrows) continue;  // wave-uniform
    const int x0 = P.rowptr[0][a];
    const int lx = P.rowptr[0][a + 1] - x0;
    const int L = lx * ly * lz;
    int64_t out0 = rowptr[lr];
    // e -> (i, j, k): L < 2^20, so a float reciprocal with a +0.5 bias divides exactly
    const float rlx = 1.0f / (float)lx, rly = 1.0f / (float)ly;
    for (int e0 = 0; e0 < L; e0 += 64) {
      const int e = e0 + lane;
      const bool in = e < L;
      const int ee = in ? e : 0;
      const int jk = (int)(((float)ee + 0.5f) * rlx);
      const int i = ee - jk * lx;
      const int k = (int)(((float)jk + 0.5f) * rly);
      const int j = jk - k * ly;
      int64_t cc = P.col[0][x0 + i];
      if (P.d > 1) cc += P.cstride[1] * (int64_t)P.col[1][y0 + j];
      if (P.d > 2) cc += P.cstride[2] * (int64_t)P.col[2][z0 + k];
      double s = 0.0;
      for (int t = 0; t < P.nterms; t++) {
        double v = P.val[0][t * P.nnz1d[0] + x0 + i];   // (x*y)*z, left to right
        if (P.d > 1) v *= P.val[1][t * P.nnz1d[1] + y0 + j];
        if (P.d > 2) v *= P.val[2][t * P.nnz1d[2] + z0 + k];
        s += v;
      }
      if (!P.filter) {
        if (in) {
          col[out0 + e] = (int32_t)(cc + P.col_offset);
          val[out0 + e] = s;
        }
      } else {
        const bool keep = in && fabs(s) > P.eps;
        const unsigned long long m = __ballot(keep);
        if (keep) {
          const unsigned long long below = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
          const int64_t pos = out0 + __popcll(m & below);
          col[pos] = (int32_t)(cc + P.col_offset);
          val[pos] = s;
        }
        out0 += __popcll(m);
      }
    }
  }
}

// Unfiltered fill, streaming form: the rows of one pencil (fixed b,c) are consecutive in the
// output and their lengths are lx(a)*ly*lz, so the pencil's entries form ONE contiguous range.
// The workgroup walks that range 256 entries at a time (fully coalesced 2-KB / 1-KB stores);
// each thread finds its row by bisection over the row offsets kept in LDS.
#define TG_KRON_MAXN0 4096
#define TG_KRON_MAXJK 128       // (y,z) column combinations per pencil row staged in LDS
__global__ void __launch_bounds__(256)
    k_kron_fill_stream(tg_kron_params P, const int64_t *__restrict__ rowptr, int32_t *__restrict__ col,
                       double *__restrict__ val) {
  __shared__ int roff[TG_KRON_MAXN0 + 1];
  const int tid = threadIdx.x;
  const int64_t pencil = P.pencil0 + blockIdx.x;
  int64_t b, c;
  tg_kron_pencil(P, pencil, &b, &c);
  const int y0 = (P.d > 1) ? P.rowptr[1][b] : 0;
  const int ly = (P.d > 1) ? P.rowptr[1][b + 1] - y0 : 1;
  const int z0 = (P.d > 2) ? P.rowptr[2][c] : 0;
  const int lz = (P.d > 2) ? P.rowptr[2][c + 1] - z0 : 1;
  const int lyz = ly * lz;
  const int n0 = (int)P.n[0];
  for (int a = tid; a <= n0; a += 256) roff[a] = P.rowptr[0][a] * lyz;
  __syncthreads();
  // rows of this pencil that belong to the output block
  const int64_t g0 = P.n[0] * pencil;
  const int a_lo = (int)max((int64_t)0, P.row0 - g0);
  const int a_hi = (int)min((int64_t)n0, P.row0 + P.nrows - g0);
  if (a_hi <= a_lo) return;
  const int64_t base = rowptr[g0 + a_lo - P.row0] - roff[a_lo];   // output position of local offset 0
  const float rly = 1.0f / (float)ly;
  // everything that depends only on the (y,z) part of an entry is the same for the whole pencil:
  // column base and, per term, the product of the y and z factors -> LDS once per workgroup
  __shared__ double wjk[TG_KRON_MAX_TERMS][TG_KRON_MAXJK];
  __shared__ int64_t cjk[TG_KRON_MAXJK];
  const bool staged = lyz <= TG_KRON_MAXJK;
  if (staged) {
    for (int jk = tid; jk < lyz; jk += 256) {
      const int k = jk / ly, j = jk - k * ly;
      int64_t cc = 0;
      if (P.d > 1) cc += P.cstride[1] * (int64_t)P.col[1][y0 + j];
      if (P.d > 2) cc += P.cstride[2] * (int64_t)P.col[2][z0 + k];
      cjk[jk] = cc + P.col_offset;
      for (int q = 0; q < P.nterms; q++) {
        double v = 1.0;
        if (P.d > 1) v *= P.val[1][q * P.nnz1d[1] + y0 + j];
        if (P.d > 2) v *= P.val[2][q * P.nnz1d[2] + z0 + k];
        wjk[q][jk] = v;
      }
    }
    __syncthreads();
  }
  // row a with roff[a] <= t < roff[a+1]: bisection for the thread's first entry, then the row only
  // moves forward (t advances by 256 per pass)
  int a = a_lo;
  {
    const int t0 = roff[a_lo] + tid;
    int lo = a_lo, hi = a_hi;
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if (roff[mid] <= t0)
        lo = mid;
      else
        hi = mid;
    }
    a = lo;
  }
  if (staged && P.nterms <= 4) {
    // four entries per thread and pass: the look-ups of an entry (row offsets, 1-D tables: L1/LDS
    // latencies in series) are independent of the other three, so they overlap
    const int tend = roff[a_hi];
    const int nt = P.nterms;
    for (int tb = roff[a_lo] + tid; tb < tend; tb += 1024) {
      int aa[4], x0v[4], iv[4], jkv[4];
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const int t = min(tb + 256 * u, tend - 1);
        while (a + 1 < a_hi && roff[a + 1] <= t) a++;
        aa[u] = a;
        x0v[u] = P.rowptr[0][a];
      }
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const int t = min(tb + 256 * u, tend - 1);
        const int lx = P.rowptr[0][aa[u] + 1] - x0v[u];
        const int e = t - roff[aa[u]];
        jkv[u] = (int)(((float)e + 0.5f) / (float)lx);
        iv[u] = e - jkv[u] * lx;
      }
      double xv[4][4];
      int32_t cx[4];
#pragma unroll
      for (int u = 0; u < 4; u++) {
        cx[u] = P.col[0][x0v[u] + iv[u]];
#pragma unroll
        for (int q = 0; q < 4; q++) xv[u][q] = q < nt ? P.val[0][q * P.nnz1d[0] + x0v[u] + iv[u]] : 0.0;
      }
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const int t = tb + 256 * u;
        if (t >= tend) break;
        double sum = 0.0;
#pragma unroll
        for (int q = 0; q < 4; q++)
          if (q < nt) sum += xv[u][q] * wjk[q][jkv[u]];
        col[base + t] = (int32_t)(cx[u] + cjk[jkv[u]]);
        val[base + t] = sum;
      }
    }
    return;
  }
  for (int t = roff[a_lo] + tid; t < roff[a_hi]; t += 256) {
    while (a + 1 < a_hi && roff[a + 1] <= t) a++;
    const int x0 = P.rowptr[0][a];
    const int lx = P.rowptr[0][a + 1] - x0;
    const int e = t - roff[a];
    const int jk = (int)(((float)e + 0.5f) / (float)lx);
    const int i = e - jk * lx;
    if (staged) {
      // value = sum_q x_q[i] * (y_q[j] z_q[k]) : the y-z product is multiplied last, exactly as below
      // ((x*y)*z differs from x*(y*z) in the last bit, so the staged path is only used for values
      // that do not enter a bit-exact comparison: the Kronecker-SUM inputs; see host code)
      double sum = 0.0;
      for (int q = 0; q < P.nterms; q++) sum += P.val[0][q * P.nnz1d[0] + x0 + i] * wjk[q][jk];
      col[base + t] = (int32_t)(P.col[0][x0 + i] + cjk[jk]);
      val[base + t] = sum;
      continue;
    }
    const int k = (int)(((float)jk + 0.5f) * rly);
    const int j = jk - k * ly;
    int64_t cc = P.col[0][x0 + i];
    if (P.d > 1) cc += P.cstride[1] * (int64_t)P.col[1][y0 + j];
    if (P.d > 2) cc += P.cstride[2] * (int64_t)P.col[2][z0 + k];
    double sum = 0.0;
    for (int q = 0; q < P.nterms; q++) {
      double v = P.val[0][q * P.nnz1d[0] + x0 + i];
      if (P.d > 1) v *= P.val[1][q * P.nnz1d[1] + y0 + j];
      if (P.d > 2) v *= P.val[2][q * P.nnz1d[2] + z0 + k];
      sum += v;
    }
    col[base + t] = (int32_t)(cc + P.col_offset);
    val[base + t] = sum;
  }
}

int tg_kron_build(int d, int nterms, const tg_kron_dir_t *dirs, int64_t row0, int64_t row1, int filter, double eps,
                  int64_t col_offset, int64_t ncols_total, tg_csr_t *out);

static int tg_kron3_build(int d, int nterms, const tg_kron_dir_t *dirs, const int64_t *cdim, int64_t row0, int64_t row1,
                          int64_t col_offset, int64_t ncols_total, tg_csr_t *out);

extern "C" int tg_kron_sum_csr(int d, int nterms, const tg_kron_dir_t *dirs, int64_t row0, int64_t row1,
                               tg_csr_t *out) {
  // streamed fill with closed-form row starts (k_kron3_fill_sum) whenever the rows are short enough for its LDS tables
  if (d >= 1 && d <= 3 && dirs && nterms >= 1 && nterms <= 9 && !getenv("TIGAR_KRON_LEGACY")) {
    int64_t cdim[3] = {1, 1, 1}, total = 1;
    int mx[3] = {1, 1, 1};
    bool ok = true;
    for (int k = 0; k < d && ok; k++) {
      ok = dirs[k].n >= 1 && dirs[k].rowptr && dirs[k].col && dirs[k].val;
      if (!ok) break;
      cdim[k] = dirs[k].n;
      total *= dirs[k].n;
      for (int64_t r = 0; r < dirs[k].n; r++) mx[k] = std::max(mx[k], dirs[k].rowptr[r + 1] - dirs[k].rowptr[r]);
    }
    if (ok && mx[1] * mx[2] <= 256) return tg_kron3_build(d, nterms, dirs, cdim, row0, row1, 0, total, out);
  }
  return tg_kron_build(d, nterms, dirs, row0, row1, 0, 0.0, 0, -1, out);
}

// column index of the product = sum_k col_k * stride_k with strides 1, ncols_1d[0], ...; the 1-D
// factors may be rectangular: ncols_1d is passed through tg_kron_dir_t.n of a SECOND array
// when needed (here: stride_k = number of 1-D columns, given by `cdim`).
int tg_kron_build_rect(int d, int nterms, const tg_kron_dir_t *dirs, const int64_t *cdim, int64_t row0, int64_t row1,
                       int filter, double eps, int64_t col_offset, int64_t ncols_total, tg_csr_t *out);

extern "C" int tg_kron_csr_rect(int d, int nterms, const tg_kron_dir_t *dirs, const int64_t *cdim, int64_t row0,
                                int64_t row1, int filter, double eps, int64_t col_offset, int64_t ncols_total,
                                tg_csr_t *out) {
  return tg_kron_build_rect(d, nterms, dirs, cdim, row0, row1, filter, eps, col_offset, ncols_total, out);
}

int tg_kron_build(int d, int nterms, const tg_kron_dir_t *dirs, int64_t row0, int64_t row1, int filter, double eps,
                  int64_t col_offset, int64_t ncols_total, tg_csr_t *out) {
  return tg_kron_build_rect(d, nterms, dirs, nullptr, row0, row1, filter, eps, col_offset, ncols_total, out);
}

int tg_kron_build_rect(int d, int nterms, const tg_kron_dir_t *dirs, const int64_t *cdim, int64_t row0, int64_t row1,
                       int filter, double eps, int64_t col_offset, int64_t ncols_total, tg_csr_t *out) {
  TG_REQUIRE_INIT();
  g_k3_skip_val = false;                 // (this builder always writes values)
  const double _t0 = tg_now();
  TG_REQUIRE(d >= 1 && d <= 3 && nterms >= 1 && nterms <= TG_KRON_MAX_TERMS && dirs && out,
             "bad arguments to tg_kron_sum_csr");
  tg_kron_params P;
  memset(&P, 0, sizeof(P));
  P.d = d;
  P.nterms = nterms;
  P.filter = filter;
  P.eps = eps;
  P.col_offset = col_offset;
  TG_REQUIRE(!filter || nterms == 1, "the eps filter applies to single-term products");
  int64_t total = 1, ctotal = 1;
  void *dev[9] = {nullptr};
  int rc = 0;
  for (int k = 0; k < 3; k++) P.n[k] = 1;
  for (int k = 0; k < d && !rc; k++) {
    const tg_kron_dir_t &D = dirs[k];
    TG_REQUIRE(D.n >= 1 && D.rowptr && D.col && D.val, "bad 1-D factor %d", k);
    const int64_t nnz1 = D.rowptr[D.n];
    P.n[k] = D.n;
    P.nnz1d[k] = nnz1;
    total *= D.n;
    P.cstride[k] = ctotal;
    ctotal *= cdim ? cdim[k] : D.n;
    int32_t *rp = nullptr, *cl = nullptr;
    double *vl = nullptr;
    rc = tg_dmalloc(&rp, D.n + 1) || tg_dmalloc(&cl, nnz1) || tg_dmalloc(&vl, nnz1 * nterms);
    dev[3 * k] = rp;
    dev[3 * k + 1] = cl;
    dev[3 * k + 2] = vl;
    if (rc) break;
    hipMemcpyAsync(rp, D.rowptr, (D.n + 1) * sizeof(int32_t), hipMemcpyHostToDevice, g_tg.stream);
    hipMemcpyAsync(cl, D.col, nnz1 * sizeof(int32_t), hipMemcpyHostToDevice, g_tg.stream);
    hipMemcpyAsync(vl, D.val, nnz1 * nterms * sizeof(double), hipMemcpyHostToDevice, g_tg.stream);
    P.rowptr[k] = rp;
    P.col[k] = cl;
    P.val[k] = vl;
  }
  if (!rc && !(row0 >= 0 && row1 >= row0 && row1 <= total)) {
    tg_set_error("tg_kron_sum_csr: row range outside [0,%lld)", (long long)total);
    rc = 2;
  }
  tg_csr_s *m = nullptr;
  if (!rc) {
    P.row0 = row0;
    P.nrows = row1 - row0;
    P.pencil0 = row0 / P.n[0];
    const int64_t pencil1 = (row1 > row0) ? (row1 - 1) / P.n[0] + 1 : P.pencil0;
    P.npencils = pencil1 - P.pencil0;
    int64_t *rowptr = nullptr;
    rc = tg_dmalloc(&rowptr, P.nrows + 1);
    if (!rc) {
      hipMemsetAsync(rowptr, 0, (size_t)(P.nrows + 1) * sizeof(int64_t), g_tg.stream);
      if (P.npencils > 0)
        hipLaunchKernelGGL(k_kron_count, dim3((unsigned)P.npencils), dim3(256), 0, g_tg.stream, P, rowptr);
      int64_t nnz = 0;
      TG_TRACE("count launched");
      rc = tg_exclusive_scan_i64(rowptr, P.nrows, &nnz);
      TG_TRACE("scan done");
      if (!rc) {
        m = new tg_csr_s();
        m->nrows = P.nrows;
        m->ncols = ncols_total >= 0 ? ncols_total : ctotal;
        m->nnz = nnz;
        m->rowptr = rowptr;
        rc = tg_dmalloc(&m->col, nnz + TG_CSR_PAD) || tg_dmalloc(&m->val, nnz + TG_CSR_PAD);
        TG_TRACE("malloc done");
        if (!rc && P.npencils > 0 && nnz > 0) {
          int64_t lyzmax = 1;
          for (int k = 1; k < d; k++) {
            int mx = 1;
            for (int64_t i = 0; i < dirs[k].n; i++) mx = std::max(mx, dirs[k].rowptr[i + 1] - dirs[k].rowptr[i]);
            lyzmax *= mx;
          }
          const bool stream_ok = !filter && P.n[0] <= TG_KRON_MAXN0 && P.nnz1d[0] * lyzmax < (1ll << 30) &&
                                 !getenv("TIGAR_KRON_ROWWISE");
          if (stream_ok)
            hipLaunchKernelGGL(k_kron_fill_stream, dim3((unsigned)P.npencils), dim3(256), 0, g_tg.stream, P, rowptr,
                               m->col, m->val);
          else
            hipLaunchKernelGGL(k_kron_fill, dim3((unsigned)P.npencils), dim3(256), 0, g_tg.stream, P, rowptr, m->col,
                               m->val);
          if (hipGetLastError() != hipSuccess) {
            tg_set_error("kron fill launch failed");
            rc = 1;
          }
        }
      } else
        tg_dfree(rowptr);
    }
  }
  // (no synchronisation: the host tables were consumed before the scan returned its total, and the
  // device copies go back to the pool, whose re-use is ordered behind the fill kernel -- the caller may
  // overlap the fill with work on the other stream)
  for (int i = 0; i < 9; i++) tg_dfree(dev[i]);
  TG_TRACE("freed");
  if (rc) {
    if (m) tg_csr_destroy(m);
    return rc;
  }
  *out = m;
  return 0;
}

// ----------------------------------------------------------------------------------------
// One 1-D sparse factor applied along one direction of a tensor-indexed vector:
//   out[lo, I, hi] = sum_t F[I, c_t] * in[lo, c_t - col_shift, hi]
// with lo running over the product of the dimensions below direction k (fastest) and hi over those
// above.  M^T b and M U of a Kronecker-structured extraction operator are three such passes each
// (sum factorisation: (p+1) or ~3p+1 terms per entry and direction instead of (p+1)^d / (3p+1)^d, and
// neither M nor M^T is ever formed).
__global__ void __launch_bounds__(256)
    k_tensor_apply_1d(const int32_t *__restrict__ rp, const int32_t *__restrict__ ci, const double *__restrict__ fv,
                      int64_t n_lo, int64_t nin_k, int64_t nout_k, int64_t n_hi, int64_t col_shift,
                      const double *__restrict__ in, double *__restrict__ out) {
  const int64_t total = n_lo * nout_k * n_hi;
  int64_t o = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const bool small = total < 0xffffffffll;       // (32-bit divisions where the index allows: the 64-bit ones dominate otherwise)
  for (; o < total; o += stride) {
    int64_t lo, I, hi;
    if (small) {
      const uint32_t o32 = (uint32_t)o, r32 = o32 / (uint32_t)n_lo;
      lo = o32 - r32 * (uint32_t)n_lo;
      const uint32_t h32 = r32 / (uint32_t)nout_k;
      I = r32 - h32 * (uint32_t)nout_k;
      hi = h32;
    } else {
      lo = o % n_lo;
      const int64_t r = o / n_lo;
      I = r % nout_k;
      hi = r / nout_k;
    }
    const double *src = in + lo + n_lo * nin_k * hi;
    double acc = 0.0;
    for (int t = rp[I]; t < rp[I + 1]; t++) acc += fv[t] * src[n_lo * ((int64_t)ci[t] - col_shift)];
    out[o] = acc;
  }
}

// The same along the FASTEST direction (n_lo = 1): there the entries an output needs are neighbours in memory and the
// factor rows of neighbouring outputs differ, so the kernel above spends its time on per-lane table and gather loads
// (2.1 ms for the 452 M -> 153 M pass of M^T b at cfg3, 34 % of the HBM peak).  Here a WAVE takes a whole line: the line
// goes to LDS with coalesced loads, the factor sits in LDS in padded column-major form (term j of output I at [j][I]: the
// lanes of a wave read consecutive words), and an output is T LDS reads and T FMAs in the factor's own term order.
#define TG_APPLY_LINES_NT 1024       // sixteen waves share one copy of the factor
#define TG_APPLY_LINES_NL 13         // entries of a line per lane: lines of up to 64 * 13 = 832 entries
__global__ void __launch_bounds__(TG_APPLY_LINES_NT)
    k_tensor_apply_lines(const int32_t *__restrict__ rp, const int32_t *__restrict__ ci, const double *__restrict__ fv,
                         int nin_k, int nout_k, int64_t n_hi, int64_t col_shift, int maxt, const double *__restrict__ in,
                         double *__restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) char smem_al[];
  double *ell_v = reinterpret_cast<double *>(smem_al);                       // [maxt][nout_k]
  double *buf = ell_v + (size_t)maxt * nout_k;                               // [waves][nin_k]
  int32_t *ell_c = reinterpret_cast<int32_t *>(buf + (size_t)(TG_APPLY_LINES_NT / 64) * nin_k);   // [maxt][nout_k]; -1: no term
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  for (int I = tid; I < nout_k; I += TG_APPLY_LINES_NT) {
    const int t0 = rp[I], n = rp[I + 1] - t0;
    for (int j = 0; j < maxt; j++) {
      ell_c[j * nout_k + I] = j < n ? (int32_t)(ci[t0 + j] - col_shift) : -1;
      ell_v[j * nout_k + I] = j < n ? fv[t0 + j] : 0.0;
    }
  }
  __syncthreads();
  double *line = buf + (size_t)w * nin_k;
  const int64_t nw = (int64_t)gridDim.x * (TG_APPLY_LINES_NT / 64);
  int64_t hi = (int64_t)blockIdx.x * (TG_APPLY_LINES_NT / 64) + w;
  // the next line travels in registers while the current one is worked on (unconditional loads, index clamped: a
  // branch around a load would make the compiler wait for all loads at the next use)
  double nx[TG_APPLY_LINES_NL];
  {
    const double *src = in + (int64_t)nin_k * min(hi, n_hi - 1);
#pragma unroll
    for (int q = 0; q < TG_APPLY_LINES_NL; q++) nx[q] = src[min(lane + 64 * q, nin_k - 1)];
  }
  for (; hi < n_hi; hi += nw) {
#pragma unroll
    for (int q = 0; q < TG_APPLY_LINES_NL; q++)
      if (lane + 64 * q < nin_k) line[lane + 64 * q] = nx[q];
    // (the line is private to the wave, whose LDS operations execute in order: a compiler fence is all it takes)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    {
      const double *src = in + (int64_t)nin_k * min(hi + nw, n_hi - 1);
#pragma unroll
      for (int q = 0; q < TG_APPLY_LINES_NL; q++) nx[q] = src[min(lane + 64 * q, nin_k - 1)];
    }
    double *dst = out + (int64_t)nout_k * hi;
    for (int I0 = lane; I0 < nout_k; I0 += 256) {                            // four outputs per lane at a time
      double acc[4] = {0.0, 0.0, 0.0, 0.0};
      for (int j = 0; j < maxt; j++) {
        int c[4];
        double v[4], x[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
          const int I = min(I0 + 64 * u, nout_k - 1);
          c[u] = ell_c[j * nout_k + I];
          v[u] = ell_v[j * nout_k + I];
        }
#pragma unroll
        for (int u = 0; u < 4; u++) x[u] = line[max(c[u], 0)];
#pragma unroll
        for (int u = 0; u < 4; u++)
          if (c[u] >= 0) acc[u] += v[u] * x[u];
      }
#pragma unroll
      for (int u = 0; u < 4; u++)
        if (I0 + 64 * u < nout_k) dst[I0 + 64 * u] = acc[u];
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
}

extern "C" int tg_tensor_apply_1d(int d, const int64_t *dims_in, int k, int64_t nout_k, const int32_t *rowptr,
                                  const int32_t *col, const double *val, int64_t col_shift, tg_vec_t in, tg_vec_t out) {
  TG_REQUIRE_INIT();
  TG_REQUIRE(d >= 1 && d <= 3 && dims_in && k >= 0 && k < d && nout_k >= 0 && rowptr && in && out,
             "bad arguments to tg_tensor_apply_1d");
  int64_t n_lo = 1, n_hi = 1, nin = 1;
  for (int j = 0; j < d; j++) {
    TG_REQUIRE(dims_in[j] >= 0, "negative dimension");
    nin *= dims_in[j];
    if (j < k) n_lo *= dims_in[j];
    if (j > k) n_hi *= dims_in[j];
  }
  const int64_t nin_k = dims_in[k];
  TG_REQUIRE(in->n == nin, "tg_tensor_apply_1d: input has %lld entries, dimensions give %lld", (long long)in->n, (long long)nin);
  TG_REQUIRE(out->n == n_lo * nout_k * n_hi, "tg_tensor_apply_1d: output has %lld entries, expected %lld", (long long)out->n,
             (long long)(n_lo * nout_k * n_hi));
  const int64_t nnz1 = rowptr[nout_k];
  for (int64_t t = 0; t < nnz1; t++)
    TG_REQUIRE(col[t] - col_shift >= 0 && col[t] - col_shift < nin_k, "tg_tensor_apply_1d: factor column %d outside the input range", (int)col[t]);
  if (out->n == 0) return 0;
  int32_t *drp = nullptr, *dci = nullptr;
  double *dfv = nullptr;
  int rc = tg_dmalloc(&drp, nout_k + 1) || tg_dmalloc(&dci, nnz1) || tg_dmalloc(&dfv, nnz1);
  if (!rc) {
    // (stream-ordered uploads through the pinned ring: no wait, the tables are released to the pool in stream order)
    rc = tg_h2d_staged(drp, rowptr, (size_t)(nout_k + 1) * sizeof(int32_t));
    if (!rc && nnz1) rc = tg_h2d_staged(dci, col, (size_t)nnz1 * sizeof(int32_t)) || tg_h2d_staged(dfv, val, (size_t)nnz1 * sizeof(double));
  }
  int maxt = 0;
  for (int64_t i = 0; i < nout_k; i++) maxt = std::max(maxt, rowptr[i + 1] - rowptr[i]);
  // the line kernel: fastest direction, lines and table fit in LDS, enough lines to fill the chip (TIGAR_APPLY_LINES=0: off)
  const size_t lines_lds = (size_t)maxt * nout_k * 12 + (size_t)(TG_APPLY_LINES_NT / 64) * nin_k * 8 + 16;
  const bool lines_on = !(getenv("TIGAR_APPLY_LINES") && atoi(getenv("TIGAR_APPLY_LINES")) == 0);
  bool use_lines = lines_on && n_lo == 1 && maxt >= 1 && lines_lds <= 150 * 1024 && n_hi >= 16 * (int64_t)g_tg.num_cu &&
                   nin_k <= 64 * TG_APPLY_LINES_NL && nout_k < (1 << 24);
  if (use_lines && hipFuncSetAttribute((const void *)k_tensor_apply_lines, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lines_lds) != hipSuccess) {
    (void)hipGetLastError();
    use_lines = false;
  }
  if (!rc && use_lines) {
    const int64_t blocks = std::min<int64_t>(tg_cdiv(n_hi, TG_APPLY_LINES_NT / 64), (int64_t)g_tg.num_cu);
    hipLaunchKernelGGL(k_tensor_apply_lines, dim3((unsigned)blocks), dim3(TG_APPLY_LINES_NT), lines_lds, g_tg.stream, drp, dci, dfv,
                       (int)nin_k, (int)nout_k, n_hi, col_shift, maxt, in->d, out->d);
    if (hipGetLastError() != hipSuccess) {
      tg_set_error("k_tensor_apply_lines failed to launch");
      rc = 1;
    }
  } else if (!rc) {
    const int64_t blocks = std::min<int64_t>(tg_cdiv(out->n, 256), (int64_t)g_tg.num_cu * 32);
    hipLaunchKernelGGL(k_tensor_apply_1d, dim3((unsigned)blocks), dim3(256), 0, g_tg.stream, drp, dci, dfv, n_lo, nin_k, nout_k,
                       n_hi, col_shift, in->d, out->d);
    if (hipGetLastError() != hipSuccess) {
      tg_set_error("k_tensor_apply_1d failed to launch");
      rc = 1;
    }
  }
  tg_dfree(drp);
  tg_dfree(dci);
  tg_dfree(dfv);
  return rc;
}

// ----------------------------------------------------------------------------------------
// generateM for a tensor-product B-spline whose filter drops nothing but exact zeros (checked by the caller on the
// 1-D tables): M = M_2 (x) M_1 (x) M_0 entry by entry, values (v0*v1)*v2 in the reference's order
// (tIGAr/BSplines.py:450-503), and equally M^T from the transposed 1-D factors.  "Pencil walk": the rows of a pencil
// (fixed (b, c), a running) are consecutive in the output and their entries contiguous, row a holding n0(a) * n12
// entries in (k, j, i) order.  A lane owns one (j, k) combination of one pencil -- its value v1*v2-free prefix (the
// reference multiplies left to right, so v1 and v2 are applied per entry), its column offset and its rank among the
// pencil's combinations -- walks a, and writes the n0(a) consecutive (col, val) pairs of its combination; the lanes
// of a pencil cover a row contiguously and consecutive rows follow each other: pure streaming stores.  Row starts
// follow in closed form from three 1-D prefix sums: no count pass, no scan.
struct tg_kron3_args {
  int d;
  int64_t n[3];                  // rows per direction
  const int32_t *rp[3];          // 1-D CSR row pointers (device)
  const int32_t *ci[3];
  const double *cv[3];
  const int64_t *ps[3];          // exclusive prefix sums of the 1-D row lengths (n[k] + 1 entries)
  int64_t cstride[3];
  int64_t col_offset;
  int64_t row0, nrows;           // output rows [row0, row0 + nrows)
  int64_t out0;                  // entry index of row row0 in the closed form (subtracted)
  int slot;                      // max n1 * max n2
  int L;
  int64_t npencils;
  int nterms;                    // Kronecker SUM: values are term-major, cv[k][t * nnz1d[k] + q]
  int64_t nnz1d[3];
  int skip_val;                  // pattern only: the caller overwrites every value (tg_kron_pattern_only)
};


__device__ __forceinline__ int64_t tg_kron3_rowstart(const tg_kron3_args &A, int64_t a, int64_t b, int64_t c) {
  // entries before row (a, b, c): separable prefix sums of n0 * n1 * n2
  const int64_t t0 = A.ps[0][A.n[0]];
  if (A.d == 1) return A.ps[0][a];
  const int64_t n1 = A.ps[1][b + 1] - A.ps[1][b];
  if (A.d == 2) return t0 * A.ps[1][b] + n1 * A.ps[0][a];
  const int64_t t1 = A.ps[1][A.n[1]];
  const int64_t n2 = A.ps[2][c + 1] - A.ps[2][c];
  return t0 * t1 * A.ps[2][c] + n2 * (t0 * A.ps[1][b] + n1 * A.ps[0][a]);
}

__global__ void __launch_bounds__(64) k_kron3_rowptr(tg_kron3_args A, int64_t *__restrict__ rowptr) {
  const int64_t lr = (int64_t)blockIdx.x * 64 + threadIdx.x;
  if (lr > A.nrows) return;
  const int64_t r = A.row0 + lr;
  const int64_t total = A.n[0] * A.n[1] * A.n[2];
  if (r >= total) {            // (only lr == nrows at the very end of the matrix)
    rowptr[lr] = A.ps[0][A.n[0]] * (A.d > 1 ? A.ps[1][A.n[1]] : 1) * (A.d > 2 ? A.ps[2][A.n[2]] : 1) - A.out0;
    return;
  }
  const int64_t a = r % A.n[0], bc = r / A.n[0];
  const int64_t b = A.d > 1 ? bc % A.n[1] : 0, c = A.d > 2 ? bc / A.n[1] : 0;
  rowptr[lr] = tg_kron3_rowstart(A, a, b, c) - A.out0;
}

// One workgroup per pencil: the pencil's entries form one contiguous range of n12 * nnz0 entries (rows a = 0..n0-1
// follow each other, row a holding n0(a) * n12 entries in (k, j, i) order).  Thread t of a pass owns entry t of that
// range -- fully coalesced 2-KB / 1-KB stores; its row is rowof[t / n12] (a look-up in the 1-D entry -> row table),
// its (k, j) combination and i follow by two small divisions; the (j, k) parts (v1, v2, column offset) sit in LDS.
#define TG_KRON3_MAXJK 2048     // (j, k) combinations of a row: 24 bytes of LDS each, sized per launch
__global__ void __launch_bounds__(256)
    k_kron3_fill(tg_kron3_args A, const int32_t *__restrict__ rowof, int32_t *__restrict__ col, double *__restrict__ val) {
  extern __shared__ double s_k3[];                 // [3][A.slot]: v1, v2, column offset of every (j, k) combination
  double *s_v1 = s_k3, *s_v2 = s_k3 + A.slot;
  int64_t *s_c = reinterpret_cast<int64_t *>(s_k3 + 2 * A.slot);
  const int tid = threadIdx.x;
  const int64_t pencil = A.row0 / A.n[0] + blockIdx.x;
  if (pencil >= A.npencils) return;
  const int64_t b = A.d > 1 ? pencil % A.n[1] : 0, c = A.d > 2 ? pencil / A.n[1] : 0;
  const int y0 = A.d > 1 ? A.rp[1][b] : 0, n1 = A.d > 1 ? A.rp[1][b + 1] - y0 : 1;
  const int z0 = A.d > 2 ? A.rp[2][c] : 0, n2 = A.d > 2 ? A.rp[2][c + 1] - z0 : 1;
  const int n12 = n1 * n2;
  for (int lc = tid; lc < n12; lc += 256) {
    const int k = lc / n1, j = lc - k * n1;        // rank lc = k * n1 + j: (k, j) order = column order
    s_v1[lc] = A.d > 1 ? A.cv[1][y0 + j] : 1.0;
    s_v2[lc] = A.d > 2 ? A.cv[2][z0 + k] : 1.0;
    int64_t cjk = A.col_offset;
    if (A.d > 1) cjk += A.cstride[1] * (int64_t)A.ci[1][y0 + j];
    if (A.d > 2) cjk += A.cstride[2] * (int64_t)A.ci[2][z0 + k];
    s_c[lc] = cjk;
  }
  __syncthreads();
  const int64_t g0 = A.n[0] * pencil;              // global row of a = 0
  const int64_t a_lo = max((int64_t)0, A.row0 - g0), a_hi = min(A.n[0], A.row0 + A.nrows - g0);
  if (a_hi <= a_lo || n12 == 0) return;
  const int64_t base = tg_kron3_rowstart(A, 0, b, c) - A.out0;   // entry index of row (0, b, c)
  const int64_t t_lo = (int64_t)n12 * A.ps[0][a_lo], t_hi = (int64_t)n12 * A.ps[0][a_hi];
  for (int64_t t = t_lo + tid; t < t_hi; t += 256) {
    // q = t / n12: the 1-D entry of direction 0 this output entry descends from (32-bit division where it suffices)
    const int64_t q = t < 0x7fffffffll ? (int64_t)((uint32_t)t / (uint32_t)n12) : t / n12;
    const int a = rowof[q];
    const int x0 = A.rp[0][a], n0 = A.rp[0][a + 1] - x0;
    const int e = (int)(t - (int64_t)n12 * x0);    // position inside row a (x0 == ps0[a])
    const int jk = e / n0, i = e - jk * n0;
    double v = A.cv[0][x0 + i];                    // (v0 * v1) * v2, left to right as the reference multiplies
    if (A.d > 1) v = v * s_v1[jk];
    if (A.d > 2) v = v * s_v2[jk];
    col[base + t] = (int32_t)(A.ci[0][x0 + i] + s_c[jk]);
    val[base + t] = v;
  }
}

// the same stream for a Kronecker SUM of terms on one pattern (synthetic FE input: Laplace = K1xM1xM1 + ...):
// value = sum_t x_t[i] * (y_t[j] z_t[k])
#define TG_KRON3_MAXT 9
#ifndef TG_KRON3_Z
#define TG_KRON3_Z 4
#endif
__global__ void __launch_bounds__(256)
    k_kron3_fill_sum(tg_kron3_args A, const int32_t *__restrict__ rowof, int32_t *__restrict__ col,
                     double *__restrict__ val) {
  extern __shared__ double s_dyn[];                // w[nterms][slot] then column offsets (int64) [slot]
  double *s_w = s_dyn;
  int64_t *s_c = reinterpret_cast<int64_t *>(s_dyn + (size_t)A.nterms * A.slot);
  const int tid = threadIdx.x;
  const int64_t pencil = A.row0 / A.n[0] + blockIdx.x;
  if (pencil >= A.npencils) return;
  const int64_t b = A.d > 1 ? pencil % A.n[1] : 0, c = A.d > 2 ? pencil / A.n[1] : 0;
  const int y0 = A.d > 1 ? A.rp[1][b] : 0, n1 = A.d > 1 ? A.rp[1][b + 1] - y0 : 1;
  const int z0 = A.d > 2 ? A.rp[2][c] : 0, n2 = A.d > 2 ? A.rp[2][c + 1] - z0 : 1;
  const int n12 = n1 * n2;
  for (int lc = tid; lc < n12; lc += 256) {
    const int k = lc / n1, j = lc - k * n1;
    for (int t = 0; t < A.nterms; t++) {
      double w = 1.0;
      if (A.d > 1) w *= A.cv[1][t * A.nnz1d[1] + y0 + j];
      if (A.d > 2) w *= A.cv[2][t * A.nnz1d[2] + z0 + k];
      s_w[t * A.slot + lc] = w;
    }
    int64_t cjk = A.col_offset;
    if (A.d > 1) cjk += A.cstride[1] * (int64_t)A.ci[1][y0 + j];
    if (A.d > 2) cjk += A.cstride[2] * (int64_t)A.ci[2][z0 + k];
    s_c[lc] = cjk;
  }
  __syncthreads();
  const int64_t g0 = A.n[0] * pencil;
  const int64_t a_lo = max((int64_t)0, A.row0 - g0), a_hi = min(A.n[0], A.row0 + A.nrows - g0);
  if (a_hi <= a_lo || n12 == 0) return;
  const int64_t base = tg_kron3_rowstart(A, 0, b, c) - A.out0;
  const int64_t t_lo = (int64_t)n12 * A.ps[0][a_lo], t_hi = (int64_t)n12 * A.ps[0][a_hi];
  // TG_KRON3_Z consecutive entries per thread: (row, (j,k) combination, i) of the first by division, of the others by
  // stepping (i, then jk, then row) -- the kernel is bound by this integer arithmetic, not by the stores (9.6 ms per
  // 31 GB launch with two divisions per entry, 7.9 ms with four entries per division; one work item per (row, jk) with
  // its n0 entries written by one thread: 24 ms, the stores of a wave are then 8-byte pieces n0*8 bytes apart) -- and
  // 16-byte stores: one per four columns, one per two values
  typedef int tg_i4 __attribute__((ext_vector_type(4)));
  typedef double tg_d2 __attribute__((ext_vector_type(2)));
  for (int64_t t4 = t_lo + TG_KRON3_Z * (int64_t)tid; t4 < t_hi; t4 += 256 * TG_KRON3_Z) {
    int cc[TG_KRON3_Z];
    double vv[TG_KRON3_Z];
    const int64_t q = t4 < 0x7fffffffll ? (int64_t)((uint32_t)t4 / (uint32_t)n12) : t4 / n12;
    int a = rowof[q];
    int x0 = A.rp[0][a], n0 = A.rp[0][a + 1] - x0;
    const int e = (int)(t4 - (int64_t)n12 * x0);
    int jk = e / n0, i = e - jk * n0;
#pragma unroll
    for (int z = 0; z < TG_KRON3_Z; z++) {
      double sum = 0.0;
      for (int u = 0; u < A.nterms; u++) sum += A.cv[0][u * A.nnz1d[0] + x0 + i] * s_w[u * A.slot + jk];
      cc[z] = (int32_t)(A.ci[0][x0 + i] + s_c[jk]);
      vv[z] = sum;
      if (++i == n0) {
        i = 0;
        if (++jk == n12) {
          jk = 0;
          do {                                   // next non-empty row (an empty 1-D row holds no entries)
            a++;
          } while (a + 1 < (int)A.n[0] && A.rp[0][a + 1] == A.rp[0][a]);
          if (a >= (int)A.n[0]) a = (int)A.n[0] - 1;   // (past the pencil: the value is not stored)
          x0 = A.rp[0][a];
          n0 = max(A.rp[0][a + 1] - x0, 1);
        }
      }
    }
    if (t4 + TG_KRON3_Z - 1 < t_hi) {
#pragma unroll
      for (int z = 0; z < TG_KRON3_Z; z += 4) {
        tg_i4 c4 = {cc[z], cc[z + 1], cc[z + 2], cc[z + 3]};
        tg_d2 v01 = {vv[z], vv[z + 1]}, v23 = {vv[z + 2], vv[z + 3]};
        __builtin_memcpy(col + base + t4 + z, &c4, 16);
        __builtin_memcpy(val + base + t4 + z, &v01, 16);
        __builtin_memcpy(val + base + t4 + z + 2, &v23, 16);
      }
    } else {
      for (int z = 0; z < TG_KRON3_Z && t4 + z < t_hi; z++) {
        col[base + t4 + z] = cc[z];
        val[base + t4 + z] = vv[z];
      }
    }
  }
}

// The same sum with one WAVE per row of the pencil: lanes <-> consecutive entries t = jk * n0 + i of the row, so a store
// instruction of a wave is one contiguous 512-byte (values) / 256-byte (columns) run.
//   * tables: one LDS line of NT+1 doubles per (j,k) combination of the pencil {w_0..w_NT-1, column offset} and one per
//     1-D entry i of the wave's current row {x_0..x_NT-1, column}: two 16-byte LDS reads each at NT = 3, immediate offsets;
//   * everything that is the same for the lanes of a wave (row, its length, its output address, the reciprocal) is forced
//     into scalar registers (readfirstlane): the stores take a scalar base and a 32-bit lane offset;
//   * jk = t / n0 is a multiply-shift (t < 2^12: exact with the 20-bit reciprocal);
//   * the 1-D data of the wave's NEXT row are fetched while the current one is streamed.
// History (31 GB launch, cfg3): thread per group of four entries with stepping decode 7.7 ms; this structure with 60 vector
// instructions per entry (64-bit per-lane addresses, run-time table strides) 7.7 ms -- both bound by the vector ALU, not by
// the stores (SQ_INSTS_VALU x 4 cycles = 78 % of the SIMD cycles of the launch); as written here (29 per entry) 6.2 ms
// where the thread kernel takes 7.7 in the same process, 7.3 in a process whose buffers landed on slower memory (the
// rate of a pure write stream is a property of the buffer: 5.6-6.7 TB/s, tools/mb/write_bw.hip); stores alone 5.5 ms.  Sums run over the terms in ascending order from
// 0.0 like k_kron3_fill_sum: identical values.
#define TG_KRON3_N0 16
template <int NT, int TG_K3_Z>      // terms of the sum; entries per lane and pass
__global__ void __launch_bounds__(256)
    k_kron3_fill_sum_rows(tg_kron3_args A, int32_t *__restrict__ col, double *__restrict__ val) {
  constexpr int S = NT + 1;                        // doubles per table line
  extern __shared__ double s_jk[];                 // [slot][S]
  __shared__ double s_xl[4][TG_KRON3_N0][S];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int64_t pencil = A.row0 / A.n[0] + blockIdx.x;
  if (pencil >= A.npencils) return;
  const int64_t b = A.d > 1 ? pencil % A.n[1] : 0, c = A.d > 2 ? pencil / A.n[1] : 0;
  const int y0 = A.d > 1 ? A.rp[1][b] : 0, n1 = A.d > 1 ? A.rp[1][b + 1] - y0 : 1;
  const int z0 = A.d > 2 ? A.rp[2][c] : 0, n2 = A.d > 2 ? A.rp[2][c + 1] - z0 : 1;
  const int n12 = n1 * n2;
  for (int lc = tid; lc < n12; lc += 256) {
    const int k = lc / n1, j = lc - k * n1;
#pragma unroll
    for (int t = 0; t < NT; t++) {
      double w = 1.0;
      if (A.d > 1) w *= A.cv[1][t * A.nnz1d[1] + y0 + j];
      if (A.d > 2) w *= A.cv[2][t * A.nnz1d[2] + z0 + k];
      s_jk[lc * S + t] = w;
    }
    int64_t cjk = A.col_offset;
    if (A.d > 1) cjk += A.cstride[1] * (int64_t)A.ci[1][y0 + j];
    if (A.d > 2) cjk += A.cstride[2] * (int64_t)A.ci[2][z0 + k];
    s_jk[lc * S + NT] = __longlong_as_double(cjk);
  }
  __syncthreads();
  const int64_t g0 = A.n[0] * pencil;
  const int64_t a_lo = max((int64_t)0, A.row0 - g0), a_hi = min(A.n[0], A.row0 + A.nrows - g0);
  if (a_hi <= a_lo || n12 == 0) return;
  const int64_t base = tg_kron3_rowstart(A, 0, b, c) - A.out0;
  int px0 = 0, pn0 = 0, pci = 0;
  int64_t pps = 0;
  double pcv[NT];
  auto fetch = [&](int64_t a) {
    px0 = A.rp[0][a];
    pn0 = A.rp[0][a + 1] - px0;
    pps = A.ps[0][a];
    if (lane < pn0) {
#pragma unroll
      for (int u = 0; u < NT; u++) pcv[u] = A.cv[0][u * A.nnz1d[0] + px0 + lane];
      pci = A.ci[0][px0 + lane];
    }
  };
  if (a_lo + wave < a_hi) fetch(a_lo + wave);
  for (int64_t a = a_lo + wave; a < a_hi; a += 4) {
    const int n0 = __builtin_amdgcn_readfirstlane(pn0);
    const int64_t ps_a = ((int64_t)__builtin_amdgcn_readfirstlane((int)(pps >> 32)) << 32) |
                         (uint32_t)__builtin_amdgcn_readfirstlane((int)pps);
    // (the reads of the previous row are done before these writes: LDS operations of a wave execute in order)
    if (lane < n0) {
#pragma unroll
      for (int u = 0; u < NT; u++) s_xl[wave][lane][u] = pcv[u];
      s_xl[wave][lane][NT] = __longlong_as_double((int64_t)pci);
    }
    if (a + 4 < a_hi) fetch(a + 4);
    if (n0 == 0) continue;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const int tot = n0 * n12;
    const unsigned magic = ((1u << 20) + (unsigned)n0 - 1u) / (unsigned)n0;
    const int64_t rowbase = base + (int64_t)n12 * ps_a;
    double *__restrict__ vrow = val + rowbase;
    int32_t *__restrict__ crow = col + rowbase;
    const double *xl = &s_xl[wave][0][0];
    // TG_K3_Z entries per lane and pass: all LDS reads of a pass are issued before the first multiply waits.  Full
    // passes run without bounds checks, the last partial one with them.
    auto entry = [&](unsigned t, double &v_out, int32_t &c_out) {
      const unsigned jk = __umul24(t, magic) >> 20, i = t - __umul24(jk, (unsigned)n0);
      const double *pj = s_jk + jk * S, *pi = xl + i * S;
      double xs[NT], ws[NT];
#pragma unroll
      for (int u = 0; u < NT; u++) {
        xs[u] = pi[u];
        ws[u] = pj[u];
      }
      c_out = (int32_t)__double_as_longlong(pi[NT]) + (int32_t)__double_as_longlong(pj[NT]);
      double acc = 0.0;
#pragma unroll
      for (int u = 0; u < NT; u++) acc += xs[u] * ws[u];
      v_out = acc;
    };
    const int full = tot / (64 * TG_K3_Z);           // (scalar)
    int t0 = lane;
    for (int q = 0; q < full; q++, t0 += 64 * TG_K3_Z) {
      double sum[TG_K3_Z];
      int32_t cc[TG_K3_Z];
#pragma unroll
      for (int z = 0; z < TG_K3_Z; z++) entry((unsigned)(t0 + 64 * z), sum[z], cc[z]);
#pragma unroll
      for (int z = 0; z < TG_K3_Z; z++) {
        crow[t0 + 64 * z] = cc[z];
        if (!A.skip_val) vrow[t0 + 64 * z] = sum[z];
      }
    }
#pragma unroll
    for (int z = 0; z < TG_K3_Z; z++) {
      const int t = t0 + 64 * z;
      if (t < tot) {
        double v;
        int32_t cidx;
        entry((unsigned)t, v, cidx);
        crow[t] = cidx;
        if (!A.skip_val) vrow[t] = v;
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
}

static int tg_kron3_build(int d, int nterms, const tg_kron_dir_t *dirs, const int64_t *cdim, int64_t row0, int64_t row1,
                          int64_t col_offset, int64_t ncols_total, tg_csr_t *out);

extern "C" int tg_kron3_csr(int d, const tg_kron_dir_t *dirs, const int64_t *cdim, int64_t row0, int64_t row1,
                            int64_t col_offset, int64_t ncols_total, tg_csr_t *out) {
  return tg_kron3_build(d, 0, dirs, cdim, row0, row1, col_offset, ncols_total, out);
}

// Device copies of the 1-D tables of the last few factor sets: a row-block producer calls tg_kron_sum_csr once per
// sub-slab with the same factors, and the uploads (with the wait that keeps the host arrays alive) stood between the
// kernels of consecutive sub-slabs.
struct tg_kron3_tables {
  uint64_t key = 0;
  int d = 0, nval = 0;
  int64_t n[3] = {0, 0, 0}, nnz1d[3] = {0, 0, 0};
  void *dev[12] = {nullptr};
  int32_t *rowof = nullptr;
};
static std::vector<tg_kron3_tables> g_k3_tables;
#define TG_KRON3_TABLES 4

static void tg_kron3_tables_free(tg_kron3_tables &t) {
  if (!g_tg.ready) return;
  for (int i = 0; i < 12; i++) tg_dfree(t.dev[i]);
  tg_dfree(t.rowof);
}

void tg_kron_cache_clear(void) {
  if (g_tg.ready) hipStreamSynchronize(g_tg.stream);
  for (auto &t : g_k3_tables) tg_kron3_tables_free(t);
  g_k3_tables.clear();
}

static uint64_t tg_kron3_key(int d, int nval, const tg_kron_dir_t *dirs) {
  uint64_t h = 1469598103934665603ull;
  auto mix = [&](uint64_t v) {
    h = (h ^ v) * 1099511628211ull;
    h ^= h >> 29;
  };
  mix((uint64_t)d);
  mix((uint64_t)nval);
  for (int k = 0; k < d; k++) {
    const tg_kron_dir_t &D = dirs[k];
    const int64_t nnz1 = D.rowptr[D.n];
    mix((uint64_t)D.n);
    for (int64_t r = 0; r <= D.n; r++) mix((uint64_t)(uint32_t)D.rowptr[r]);
    for (int64_t q = 0; q < nnz1; q++) mix((uint64_t)(uint32_t)D.col[q]);
    for (int64_t q = 0; q < nnz1 * nval; q++) {
      uint64_t bits;
      memcpy(&bits, &D.val[q], 8);
      mix(bits);
    }
  }
  return h ? h : 1;
}

// nterms == 0: single product with the reference's value order (extraction operators); nterms >= 1: Kronecker sum
static int tg_kron3_build(int d, int nterms, const tg_kron_dir_t *dirs, const int64_t *cdim, int64_t row0, int64_t row1,
                          int64_t col_offset, int64_t ncols_total, tg_csr_t *out) {
  TG_REQUIRE_INIT();
  TG_REQUIRE(d >= 1 && d <= 3 && dirs && cdim && out && nterms >= 0 && nterms <= TG_KRON3_MAXT, "bad arguments to tg_kron3_csr");
  const int nval = nterms > 0 ? nterms : 1;
  const bool skip_val = g_k3_skip_val;       // (one-shot: whatever this call does, the next one writes values again)
  g_k3_skip_val = false;
  tg_kron3_args A;
  memset(&A, 0, sizeof(A));
  A.d = d;
  A.col_offset = col_offset;
  A.nterms = nterms;
  int64_t total = 1, ctotal = 1;
  std::vector<std::vector<int64_t>> hps(3);
  int maxn[3] = {1, 1, 1};
  void *dev[12] = {nullptr};
  int rc = 0;
  for (int k = 0; k < 3; k++) A.n[k] = 1;
  for (int k = 0; k < d; k++)
    TG_REQUIRE(dirs[k].n >= 1 && dirs[k].rowptr && dirs[k].col && dirs[k].val, "bad 1-D factor %d", k);
  // tables of these factors already on the device?
  const uint64_t key = tg_kron3_key(d, nval, dirs);
  int hit = -1;
  for (size_t i = 0; i < g_k3_tables.size() && hit < 0; i++) {
    const tg_kron3_tables &t = g_k3_tables[i];
    bool same = t.key == key && t.d == d && t.nval == nval;
    for (int k = 0; k < d && same; k++) same = t.n[k] == dirs[k].n && t.nnz1d[k] == dirs[k].rowptr[dirs[k].n];
    if (same) hit = (int)i;
  }
  for (int k = 0; k < d && !rc; k++) {
    const tg_kron_dir_t &D = dirs[k];
    const int64_t nnz1 = D.rowptr[D.n];
    A.n[k] = D.n;
    A.nnz1d[k] = nnz1;
    total *= D.n;
    A.cstride[k] = ctotal;
    ctotal *= cdim[k];
    hps[k].assign((size_t)D.n + 1, 0);
    for (int64_t r = 0; r < D.n; r++) {
      const int len = D.rowptr[r + 1] - D.rowptr[r];
      hps[k][(size_t)r + 1] = hps[k][(size_t)r] + len;
      maxn[k] = std::max(maxn[k], len);
      for (int q = D.rowptr[r] + 1; q < D.rowptr[r + 1]; q++)
        TG_REQUIRE(D.col[q] > D.col[q - 1], "tg_kron3_csr: 1-D factor %d row %lld is not in ascending column order", k,
                   (long long)r);
    }
    int32_t *rp = nullptr, *cl = nullptr;
    double *vl = nullptr;
    int64_t *ps = nullptr;
    if (hit >= 0) {
      rp = (int32_t *)g_k3_tables[(size_t)hit].dev[4 * k];
      cl = (int32_t *)g_k3_tables[(size_t)hit].dev[4 * k + 1];
      vl = (double *)g_k3_tables[(size_t)hit].dev[4 * k + 2];
      ps = (int64_t *)g_k3_tables[(size_t)hit].dev[4 * k + 3];
    } else {
      rc = tg_dmalloc(&rp, D.n + 1) || tg_dmalloc(&cl, nnz1) || tg_dmalloc(&vl, nnz1 * nval) || tg_dmalloc(&ps, D.n + 1);
      dev[4 * k] = rp;
      dev[4 * k + 1] = cl;
      dev[4 * k + 2] = vl;
      dev[4 * k + 3] = ps;
      if (rc) break;
      hipMemcpyAsync(rp, D.rowptr, (size_t)(D.n + 1) * sizeof(int32_t), hipMemcpyHostToDevice, g_tg.stream);
      if (nnz1) {
        hipMemcpyAsync(cl, D.col, (size_t)nnz1 * sizeof(int32_t), hipMemcpyHostToDevice, g_tg.stream);
        hipMemcpyAsync(vl, D.val, (size_t)(nnz1 * nval) * sizeof(double), hipMemcpyHostToDevice, g_tg.stream);
      }
      hipMemcpyAsync(ps, hps[k].data(), (size_t)(D.n + 1) * sizeof(int64_t), hipMemcpyHostToDevice, g_tg.stream);
    }
    A.rp[k] = rp;
    A.ci[k] = cl;
    A.cv[k] = vl;
    A.ps[k] = ps;
  }
  tg_csr_s *m = nullptr;
  if (!rc) {
    if (ctotal > ncols_total - col_offset || row0 < 0 || row1 < row0 || row1 > total) {
      tg_set_error("tg_kron3_csr: bad row range / column space");
      rc = 2;
    }
  }
  if (!rc) {
    auto rowstart = [&](int64_t r) -> int64_t {
      if (r >= total) return hps[0][(size_t)A.n[0]] * (d > 1 ? hps[1][(size_t)A.n[1]] : 1) * (d > 2 ? hps[2][(size_t)A.n[2]] : 1);
      const int64_t a = r % A.n[0], bc = r / A.n[0];
      const int64_t b = d > 1 ? bc % A.n[1] : 0, c = d > 2 ? bc / A.n[1] : 0;
      const int64_t t0 = hps[0][(size_t)A.n[0]];
      if (d == 1) return hps[0][(size_t)a];
      const int64_t n1 = hps[1][(size_t)b + 1] - hps[1][(size_t)b];
      if (d == 2) return t0 * hps[1][(size_t)b] + n1 * hps[0][(size_t)a];
      const int64_t t1 = hps[1][(size_t)A.n[1]], n2 = hps[2][(size_t)c + 1] - hps[2][(size_t)c];
      return t0 * t1 * hps[2][(size_t)c] + n2 * (t0 * hps[1][(size_t)b] + n1 * hps[0][(size_t)a]);
    };
    A.row0 = row0;
    A.nrows = row1 - row0;
    A.out0 = rowstart(row0);
    const int64_t nnz = rowstart(row1) - A.out0;
    A.slot = maxn[1] * maxn[2];
    A.L = 1;
    A.npencils = total / A.n[0];
    if (A.slot > TG_KRON3_MAXJK) {
      tg_set_error("tg_kron3_csr: more than %d (j,k) combinations per row", TG_KRON3_MAXJK);
      rc = 2;
    }
    int32_t *d_rowof = hit >= 0 ? g_k3_tables[(size_t)hit].rowof : nullptr;
    if (!rc && hit < 0) {
      const tg_kron_dir_t &D0 = dirs[0];
      std::vector<int32_t> rowof((size_t)std::max<int64_t>(D0.rowptr[D0.n], 1), 0);
      for (int64_t r = 0; r < D0.n; r++)
        for (int q = D0.rowptr[r]; q < D0.rowptr[r + 1]; q++) rowof[(size_t)q] = (int32_t)r;
      rc = tg_dmalloc(&d_rowof, (int64_t)rowof.size());
      if (!rc) {
        hipMemcpyAsync(d_rowof, rowof.data(), rowof.size() * sizeof(int32_t), hipMemcpyHostToDevice, g_tg.stream);
        hipStreamSynchronize(g_tg.stream);
      }
    }
    if (!rc) rc = tg_csr_alloc(A.nrows, ncols_total, nnz, &m);
    if (!rc && A.nrows > 0) {
      hipLaunchKernelGGL(k_kron3_rowptr, dim3((unsigned)tg_cdiv(A.nrows + 1, 64)), dim3(64), 0, g_tg.stream, A, m->rowptr);
      const int64_t p_first = row0 / A.n[0], p_last = (row1 - 1) / A.n[0];
      if (nterms == 0)
        hipLaunchKernelGGL(k_kron3_fill, dim3((unsigned)(p_last - p_first + 1)), dim3(256), (size_t)A.slot * 24, g_tg.stream, A, d_rowof,
                           m->col, m->val);
      else if (nterms <= 3 && maxn[0] <= TG_KRON3_N0 && (int64_t)maxn[0] * A.slot < (1 << 12) &&
               !getenv("TIGAR_KRON3_THREADS")) {
        const dim3 grid((unsigned)(p_last - p_first + 1));
        const size_t lds = (size_t)(nterms + 1) * A.slot * sizeof(double);
#define TG_K3_LAUNCH(NT, Z) hipLaunchKernelGGL((k_kron3_fill_sum_rows<NT, Z>), grid, dim3(256), lds, g_tg.stream, A, m->col, m->val)
        A.skip_val = skip_val ? 1 : 0;
        if (nterms == 1) TG_K3_LAUNCH(1, 2);
        else if (nterms == 2) TG_K3_LAUNCH(2, 2);
        else TG_K3_LAUNCH(3, 2);      // (1, 2 or 4 entries per pass: equal within the spread, A/B inside one process)
#undef TG_K3_LAUNCH
      }
      else
        hipLaunchKernelGGL(k_kron3_fill_sum, dim3((unsigned)(p_last - p_first + 1)), dim3(256),
                           (size_t)(nterms + 1) * A.slot * sizeof(double), g_tg.stream, A, d_rowof, m->col, m->val);
      if (hipGetLastError() != hipSuccess) {
        tg_set_error("tg_kron3_csr: kernel launch failed");
        rc = 1;
      }
    } else if (!rc) {
      const int64_t zero = 0;
      hipMemcpyAsync(m->rowptr, &zero, sizeof(int64_t), hipMemcpyHostToDevice, g_tg.stream);
    }
    if (!rc && m) {
      // certificate of the pattern just written: Kronecker product of the callers' 1-D patterns, rows from row0 on
      const int32_t *rps[3] = {nullptr, nullptr, nullptr}, *cls[3] = {nullptr, nullptr, nullptr};
      int64_t nr[3] = {1, 1, 1}, nc[3] = {1, 1, 1};
      for (int k = 0; k < d; k++) {
        rps[k] = dirs[k].rowptr;
        cls[k] = dirs[k].col;
        nr[k] = dirs[k].n;
        nc[k] = cdim[k];
      }
      m->pattern_tag = tg_pattern_hash(d, nr, nc, rps, cls, col_offset);
      m->pattern_row0 = row0;
    }
    if (hit < 0) {
      // (the host tables above are read by the copies: wait before they go out of scope)
      hipStreamSynchronize(g_tg.stream);
      if (!rc) {
        // keep the device tables for the next call with these factors
        tg_kron3_tables t;
        t.key = key;
        t.d = d;
        t.nval = nval;
        for (int k = 0; k < d; k++) {
          t.n[k] = dirs[k].n;
          t.nnz1d[k] = dirs[k].rowptr[dirs[k].n];
        }
        for (int i = 0; i < 12; i++) t.dev[i] = dev[i];
        t.rowof = d_rowof;
        if (g_k3_tables.size() >= TG_KRON3_TABLES) {
          tg_kron3_tables_free(g_k3_tables.back());
          g_k3_tables.pop_back();
        }
        g_k3_tables.insert(g_k3_tables.begin(), t);
        for (int i = 0; i < 12; i++) dev[i] = nullptr;
        d_rowof = nullptr;
      }
      tg_dfree(d_rowof);
    } else if (hit > 0) {
      std::swap(g_k3_tables[0], g_k3_tables[(size_t)hit]);     // most recent first
    }
  } else
    hipStreamSynchronize(g_tg.stream);
  for (int i = 0; i < 12; i++) tg_dfree(dev[i]);
  if (rc) {
    if (m) tg_csr_destroy(m);
    return rc;
  }
  *out = m;
  return 0;
}
