// FE-side operator assembly on tensor-product patches with a mapped geometry.
//
// In the reference the FE matrix/vector handed to extractMatrix/extractVector comes from
// dolfin.assemble(form) (tIGAr/common.py:1206-1220) with the spline's measures and differential
// operators: F = cpFuncs[i]/cpFuncs[nsd] (common.py:917-921), metric g = DF^T DF, volume element
// sqrt(det g) (calculusUtils.py:66-70), Cartesian gradient through pinv(DF) = g^-1 DF^T
// (calculusUtils.py:56-64).  For the scalar Q_p Lagrange space on the tensor node grid this file
// assembles, with Gauss-Legendre quadrature (nq points per direction),
//     mass      a(u,v) = int u v sqrt(det g) dxi
//     laplace   a(u,v) = int (grad_xi u)^T g^-1 (grad_xi v) sqrt(det g) dxi   (= grad_x u . grad_x v dx)
//     load      L(v)   = int f_h v sqrt(det g) dxi,  f_h the nodal interpolant of given node values
// nsd >= d is allowed (surfaces in 3-D: Laplace-Beltrami), geometry is rational (quotient rule).
//
// One workgroup per element: local control values and the 1-D Lagrange tables go to LDS, one
// thread per quadrature point builds w_q sqrt(det g) g^-1, then one thread per (a,b) pair of
// local nodes sums over the quadrature points and adds into the CSR slot, which is known in closed
// form (the pattern is the Kronecker product of the 1-D element-coupling patterns, columns of a
// row are contiguous per direction).  Plain O((p+1)^(3d)) element integration -- this is the
// caller step before the hot path, not the hot path; the identity-geometry inputs of the
// benchmark configurations use the Kronecker-sum generator (tg_kron.hip) instead.
#include "tg_common.h"
#include <cmath>

#define TG_ASM_MAXLOC 128      // (p+1)^d local nodes: p <= 4 in 3-D, p <= 8 in 2-D (<= 81), any p <= 8 in 1-D
#define TG_ASM_MAXQ1 10        // Gauss points per direction

struct tg_asm_args {
  int d, p, nsd, nq;
  int nel[3], n[3];            // elements / nodes per direction
  const double *verts[3];      // device: element vertices
  const double *cp[4];         // device: nsd+1 control functions on the node grid
  const double *tab;           // device: l[a][q] (p+1)*nq | dl[a][q] (p+1)*nq | w[q] nq   (reference element [0,1])
  int form;                    // 0 mass, 1 laplace, 2 load
  const int64_t *rowptr;       // pattern (matrix forms)
  double *val;
  const double *fnod;          // load: nodal values
  double *bout;
  int colour[3];               // this launch: the elements with el[k] = colour[k] (mod 2) -- they share no node
  int ncol[3];                 // ... of which there are ncol[k] per direction
};

__device__ __forceinline__ void tg_sym_inverse(int d, const double *g, double *gi, double *det) {
  if (d == 1) {
    *det = g[0];
    gi[0] = 1.0 / g[0];
  } else if (d == 2) {
    const double a = g[0], b = g[1], c = g[3];
    const double dt = a * c - b * b;
    *det = dt;
    gi[0] = c / dt;
    gi[1] = gi[2] = -b / dt;
    gi[3] = a / dt;
  } else {
    const double a = g[0], b = g[1], c = g[2], e = g[4], f = g[5], i = g[8];
    const double c00 = e * i - f * f, c01 = c * f - b * i, c02 = b * f - c * e;
    const double dt = a * c00 + b * c01 + c * c02;
    *det = dt;
    gi[0] = c00 / dt;
    gi[1] = gi[3] = c01 / dt;
    gi[2] = gi[6] = c02 / dt;
    gi[4] = (a * i - c * c) / dt;
    gi[5] = gi[7] = (b * c - a * f) / dt;
    gi[8] = (a * e - b * b) / dt;
  }
}

// 1-D pattern of the element-coupling matrix: columns of node r form the contiguous range
// [lo, lo+width): both neighbouring elements for an interior vertex, the own element otherwise
__device__ __forceinline__ void tg_row_range_1d(int r, int p, int n, int *lo, int *width) {
  if (r % p == 0) {
    const int l = max(0, r - p), h = min(n - 1, r + p);
    *lo = l;
    *width = h - l + 1;
  } else {
    *lo = (r / p) * p;
    *width = p + 1;
  }
}

__global__ void __launch_bounds__(256) k_assemble_mapped(tg_asm_args P) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int p1 = P.p + 1, nq1 = P.nq;
  const int d = P.d;
  const int nloc = d == 1 ? p1 : (d == 2 ? p1 * p1 : p1 * p1 * p1);
  const int nqt = d == 1 ? nq1 : (d == 2 ? nq1 * nq1 : nq1 * nq1 * nq1);
  double *tl = reinterpret_cast<double *>(smem);   // l[a][q]
  double *tdl = tl + p1 * nq1;                     // dl[a][q]
  double *tw = tdl + p1 * nq1;                     // w[q]
  double *cpl = tw + nq1;                          // [nsd+1][nloc]
  double *G = cpl + (P.nsd + 1) * nloc;            // [nqt][9]  w sqrt(det g) g^-1
  double *S = G + (size_t)nqt * 9;                 // [nqt]     w sqrt(det g)   (load: times f_h)
  double *fl = S + nqt;                            // [nloc]    load: nodal values
  const int tid = threadIdx.x, nt = blockDim.x;
  // element: the blockIdx-th of this launch's colour.  Elements of one colour share no node, so they add into the output
  // without atomics; the 2^d colours follow each other in a fixed order: the assembled values are bit-reproducible
  // (global floating-point atomics added the contributions of the elements around a node in arrival order)
  int64_t e = blockIdx.x;
  int el[3] = {0, 0, 0};
  el[0] = 2 * (int)(e % P.ncol[0]) + P.colour[0];
  e /= P.ncol[0];
  if (d > 1) {
    el[1] = 2 * (int)(e % P.ncol[1]) + P.colour[1];
    e /= P.ncol[1];
  }
  if (d > 2) el[2] = 2 * (int)e + P.colour[2];
  double h[3] = {1.0, 1.0, 1.0};
  for (int k = 0; k < d; k++) h[k] = P.verts[k][el[k] + 1] - P.verts[k][el[k]];
  for (int s = tid; s < 2 * p1 * nq1 + nq1; s += nt) tl[s] = P.tab[s];
  for (int a = tid; a < nloc; a += nt) {
    const int a0 = a % p1, a1 = (a / p1) % p1, a2 = a / (p1 * p1);
    const int64_t node = (int64_t)(el[0] * P.p + a0) + (int64_t)P.n[0] * ((d > 1 ? el[1] * P.p + a1 : 0) +
                                                                         (int64_t)P.n[1] * (d > 2 ? el[2] * P.p + a2 : 0));
    for (int c = 0; c <= P.nsd; c++) cpl[c * nloc + a] = P.cp[c][node];
    if (P.form == 2) fl[a] = P.fnod[node];
  }
  __syncthreads();
  // quadrature-point data
  for (int q = tid; q < nqt; q += nt) {
    const int qk[3] = {q % nq1, (q / nq1) % nq1, q / (nq1 * nq1)};
    double N[4] = {0, 0, 0, 0}, dN[4][3] = {{0}}, fh = 0.0;
    for (int a = 0; a < nloc; a++) {
      const int ak[3] = {a % p1, (a / p1) % p1, a / (p1 * p1)};
      double l[3] = {1, 1, 1}, dl[3] = {0, 0, 0};
      for (int k = 0; k < d; k++) {
        l[k] = tl[ak[k] * nq1 + qk[k]];
        dl[k] = tdl[ak[k] * nq1 + qk[k]] / h[k];
      }
      const double phi = l[0] * l[1] * l[2];
      const double g0 = dl[0] * l[1] * l[2], g1 = l[0] * dl[1] * l[2], g2 = l[0] * l[1] * dl[2];
      for (int c = 0; c <= P.nsd; c++) {
        const double v = cpl[c * nloc + a];
        N[c] += v * phi;
        dN[c][0] += v * g0;
        dN[c][1] += v * g1;
        dN[c][2] += v * g2;
      }
      if (P.form == 2) fh += fl[a] * phi;
    }
    // DF[i][k] = d(N_i / W)/dxi_k ; metric g = DF^T DF
    const double W = N[P.nsd];
    double g[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = 0; i < P.nsd; i++) {
      double df[3];
      for (int k = 0; k < d; k++) df[k] = (dN[i][k] * W - N[i] * dN[P.nsd][k]) / (W * W);
      for (int k = 0; k < d; k++)
        for (int m = 0; m < d; m++) g[k * d + m] += df[k] * df[m];
    }
    double gi[9], det;
    tg_sym_inverse(d, g, gi, &det);
    double wq = 1.0;
    for (int k = 0; k < d; k++) wq *= tw[qk[k]] * h[k];
    const double s = wq * sqrt(fabs(det));
    for (int k = 0; k < d * d; k++) G[(size_t)q * 9 + k] = s * gi[k];
    S[q] = (P.form == 2) ? s * fh : s;
  }
  __syncthreads();
  if (P.form == 2) {
    for (int a = tid; a < nloc; a += nt) {
      const int ak[3] = {a % p1, (a / p1) % p1, a / (p1 * p1)};
      double acc = 0.0;
      for (int q = 0; q < nqt; q++) {
        const int qk[3] = {q % nq1, (q / nq1) % nq1, q / (nq1 * nq1)};
        double phi = 1.0;
        for (int k = 0; k < d; k++) phi *= tl[ak[k] * nq1 + qk[k]];
        acc += S[q] * phi;
      }
      const int64_t node = (int64_t)(el[0] * P.p + ak[0]) +
                           (int64_t)P.n[0] * ((d > 1 ? el[1] * P.p + ak[1] : 0) + (int64_t)P.n[1] * (d > 2 ? el[2] * P.p + ak[2] : 0));
      P.bout[node] += acc;
    }
    return;
  }
  for (int pr = tid; pr < nloc * nloc; pr += nt) {
    const int a = pr / nloc, b = pr - a * nloc;
    const int ak[3] = {a % p1, (a / p1) % p1, a / (p1 * p1)};
    const int bk[3] = {b % p1, (b / p1) % p1, b / (p1 * p1)};
    double acc = 0.0;
    for (int q = 0; q < nqt; q++) {
      const int qk[3] = {q % nq1, (q / nq1) % nq1, q / (nq1 * nq1)};
      double la[3] = {1, 1, 1}, lb[3] = {1, 1, 1}, da[3] = {0, 0, 0}, db[3] = {0, 0, 0};
      for (int k = 0; k < d; k++) {
        la[k] = tl[ak[k] * nq1 + qk[k]];
        lb[k] = tl[bk[k] * nq1 + qk[k]];
        da[k] = tdl[ak[k] * nq1 + qk[k]] / h[k];
        db[k] = tdl[bk[k] * nq1 + qk[k]] / h[k];
      }
      if (P.form == 0) {
        acc += S[q] * (la[0] * la[1] * la[2]) * (lb[0] * lb[1] * lb[2]);
      } else {
        const double ga[3] = {da[0] * la[1] * la[2], la[0] * da[1] * la[2], la[0] * la[1] * da[2]};
        const double gb[3] = {db[0] * lb[1] * lb[2], lb[0] * db[1] * lb[2], lb[0] * lb[1] * db[2]};
        const double *Gq = G + (size_t)q * 9;
        double t = 0.0;
        for (int k = 0; k < d; k++) {
          double u = 0.0;
          for (int m = 0; m < d; m++) u += Gq[k * d + m] * gb[m];
          t += ga[k] * u;
        }
        acc += t;
      }
    }
    // CSR slot of (row a, col b)
    int64_t row = 0, rstride = 1;
    int pos = 0, pstride = 1;
    for (int k = 0; k < d; k++) {
      const int r = el[k] * P.p + ak[k], c = el[k] * P.p + bk[k];
      int lo, width;
      tg_row_range_1d(r, P.p, P.n[k], &lo, &width);
      row += rstride * r;
      rstride *= P.n[k];
      pos += pstride * (c - lo);
      pstride *= width;
    }
    P.val[P.rowptr[row] + pos] += acc;
  }
}

// Gauss-Legendre points/weights on [0,1] by Newton iteration on P_n
static void tg_gauss01(int n, std::vector<double> &x, std::vector<double> &w) {
  x.resize(n);
  w.resize(n);
  for (int i = 0; i < n; i++) {
    double z = cos(M_PI * (i + 0.75) / (n + 0.5));
    double pp = 0.0;
    for (int it = 0; it < 100; it++) {
      double p1 = 1.0, p2 = 0.0;
      for (int j = 0; j < n; j++) {
        const double p3 = p2;
        p2 = p1;
        p1 = ((2.0 * j + 1.0) * z * p2 - j * p3) / (j + 1.0);
      }
      pp = n * (z * p1 - p2) / (z * z - 1.0);
      const double z1 = z;
      z = z1 - p1 / pp;
      if (fabs(z - z1) < 1e-16) break;
    }
    x[n - 1 - i] = 0.5 * (z + 1.0);
    w[n - 1 - i] = 1.0 / ((1.0 - z * z) * pp * pp);   // = 0.5 * 2/((1-z^2) pp^2)
  }
}

static int tg_assemble_common(const tg_patch_t *pt, int form, tg_csr_t *mout, tg_vec_t fnod, tg_vec_t bout) {
  TG_REQUIRE_INIT();
  TG_REQUIRE(pt && pt->d >= 1 && pt->d <= 3 && pt->p >= 1 && pt->p <= TG_MAX_DEGREE && pt->nsd >= pt->d && pt->nsd <= 3,
             "bad patch description");
  TG_REQUIRE(pt->nq >= 1 && pt->nq <= TG_ASM_MAXQ1, "1..%d Gauss points per direction", TG_ASM_MAXQ1);
  const int d = pt->d, p = pt->p, p1 = p + 1;
  int nloc = 1, nqt = 1;
  for (int k = 0; k < d; k++) {
    nloc *= p1;
    nqt *= pt->nq;
  }
  TG_REQUIRE(nloc <= TG_ASM_MAXLOC, "(p+1)^d = %d local nodes exceed the kernel limit %d", nloc, TG_ASM_MAXLOC);
  tg_asm_args A;
  memset(&A, 0, sizeof(A));
  A.d = d;
  A.p = p;
  A.nsd = pt->nsd;
  A.nq = pt->nq;
  A.form = form;
  int64_t nnodes = 1, nelem = 1;
  for (int k = 0; k < 3; k++) {
    A.nel[k] = 1;
    A.n[k] = 1;
  }
  std::vector<void *> dev;
  auto cleanup = [&]() {
    hipStreamSynchronize(g_tg.stream);
    for (void *q : dev) tg_dfree(q);
  };
  for (int k = 0; k < d; k++) {
    TG_REQUIRE(pt->nverts[k] >= 2 && pt->verts[k], "direction %d needs at least one element", k);
    A.nel[k] = pt->nverts[k] - 1;
    A.n[k] = A.nel[k] * p + 1;
    nnodes *= A.n[k];
    nelem *= A.nel[k];
    double *dv = nullptr;
    if (tg_dmalloc(&dv, pt->nverts[k])) {
      cleanup();
      return 1;
    }
    dev.push_back(dv);
    hipMemcpyAsync(dv, pt->verts[k], pt->nverts[k] * sizeof(double), hipMemcpyHostToDevice, g_tg.stream);
    A.verts[k] = dv;
  }
  TG_REQUIRE(nelem < (1ll << 31), "too many elements for one launch");
  for (int c = 0; c <= pt->nsd; c++) {
    if (!(pt->cp[c] && pt->cp[c]->n == nnodes)) {
      cleanup();
      tg_set_error("control function %d must be a vector on the %lld FE nodes", c, (long long)nnodes);
      return 2;
    }
    A.cp[c] = pt->cp[c]->d;
  }
  // reference-element tables: equispaced Lagrange nodes a/p at the Gauss points
  std::vector<double> gx, gw;
  tg_gauss01(pt->nq, gx, gw);
  std::vector<double> tab(2 * (size_t)p1 * pt->nq + pt->nq);
  for (int a = 0; a < p1; a++)
    for (int q = 0; q < pt->nq; q++) {
      const double t = gx[q];
      double l = 1.0, dl = 0.0;
      for (int m = 0; m < p1; m++)
        if (m != a) l *= (t - (double)m / p) / ((double)a / p - (double)m / p);
      for (int m = 0; m < p1; m++) {
        if (m == a) continue;
        double term = 1.0 / ((double)a / p - (double)m / p);
        for (int r = 0; r < p1; r++)
          if (r != a && r != m) term *= (t - (double)r / p) / ((double)a / p - (double)r / p);
        dl += term;
      }
      tab[(size_t)a * pt->nq + q] = l;
      tab[(size_t)p1 * pt->nq + (size_t)a * pt->nq + q] = dl;
    }
  for (int q = 0; q < pt->nq; q++) tab[2 * (size_t)p1 * pt->nq + q] = gw[q];
  double *dtab = nullptr;
  if (tg_dmalloc(&dtab, (int64_t)tab.size())) {
    cleanup();
    return 1;
  }
  dev.push_back(dtab);
  hipMemcpyAsync(dtab, tab.data(), tab.size() * sizeof(double), hipMemcpyHostToDevice, g_tg.stream);
  A.tab = dtab;

  tg_csr_s *m = nullptr;
  if (form != 2) {
    // pattern: Kronecker product of the 1-D element-coupling patterns, zero values
    std::vector<std::vector<int32_t>> rp(d), cl(d);
    std::vector<std::vector<double>> vl(d);
    tg_kron_dir_t dirs[3];
    for (int k = 0; k < d; k++) {
      const int n = A.n[k];
      rp[k].assign(n + 1, 0);
      for (int r = 0; r < n; r++) {
        int lo, width;
        if (r % p == 0) {
          const int l = std::max(0, r - p), hh = std::min(n - 1, r + p);
          lo = l;
          width = hh - l + 1;
        } else {
          lo = (r / p) * p;
          width = p + 1;
        }
        for (int c = 0; c < width; c++) cl[k].push_back(lo + c);
        rp[k][r + 1] = rp[k][r] + width;
      }
      vl[k].assign(cl[k].size(), 0.0);
      dirs[k].n = n;
      dirs[k].rowptr = rp[k].data();
      dirs[k].col = cl[k].data();
      dirs[k].val = vl[k].data();
    }
    tg_csr_t pat = nullptr;
    const int rc = tg_kron_sum_csr(d, 1, dirs, 0, nnodes, &pat);
    if (rc) {
      cleanup();
      return rc;
    }
    m = pat;
    hipMemsetAsync(m->val, 0, (size_t)m->nnz * sizeof(double), g_tg.stream);
    A.rowptr = m->rowptr;
    A.val = m->val;
  } else {
    if (!(fnod && fnod->n == nnodes && bout && bout->n == nnodes)) {
      cleanup();
      tg_set_error("load assembly needs nodal values and an output vector on the %lld FE nodes", (long long)nnodes);
      return 2;
    }
    A.fnod = fnod->d;
    A.bout = bout->d;
    hipMemsetAsync(bout->d, 0, (size_t)nnodes * sizeof(double), g_tg.stream);
  }
  const size_t lds = ((size_t)2 * p1 * pt->nq + pt->nq + (size_t)(pt->nsd + 1) * nloc + (size_t)nqt * 10 + nloc) * sizeof(double);
  if (lds > 64 * 1024) {
    if (m) tg_csr_destroy(m);
    cleanup();
    tg_set_error("element data (%zu B) does not fit in LDS", lds);
    return 2;
  }
  const int nt = (form == 2) ? 128 : 256;
  bool bad = false;
  for (int c = 0; c < (1 << d) && !bad; c++) {       // one launch per colour (parity of the element index per direction)
    int64_t nblk = 1;
    for (int k = 0; k < 3; k++) {
      A.colour[k] = k < d ? (c >> k) & 1 : 0;
      A.ncol[k] = k < d ? (A.nel[k] - A.colour[k] + 1) / 2 : 1;
      nblk *= A.ncol[k];
    }
    if (nblk == 0) continue;
    hipLaunchKernelGGL(k_assemble_mapped, dim3((unsigned)nblk), dim3(nt), lds, g_tg.stream, A);
    bad = hipGetLastError() != hipSuccess;
  }
  cleanup();
  if (bad) {
    if (m) tg_csr_destroy(m);
    tg_set_error("k_assemble_mapped failed to launch");
    return 1;
  }
  if (mout) *mout = m;
  return 0;
}

extern "C" int tg_assemble_mapped_matrix(const tg_patch_t *patch, int form, tg_csr_t *out) {
  TG_REQUIRE(out && (form == 0 || form == 1), "form: 0 = mass, 1 = laplace");
  return tg_assemble_common(patch, form, out, nullptr, nullptr);
}

extern "C" int tg_assemble_mapped_load(const tg_patch_t *patch, tg_vec_t fnodal, tg_vec_t out) {
  return tg_assemble_common(patch, 2, nullptr, fnodal, out);
}
