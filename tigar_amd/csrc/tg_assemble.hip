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
//     elasticity, block (i, j) of a(u,v) = int lambda div u div v + 2 mu eps(u):eps(v) dx on the d-field space (nsd == d):
//               lambda int d_i phi_a d_j phi_b + mu int d_j phi_a d_i phi_b + delta_ij mu int grad phi_a . grad phi_b
//               with d_i = sum_k Jinv[k][i] d/dxi_k (spline.grad / spline.div, tIGAr/common.py:1022-1040,
//               calculusUtils.py:255-276), i.e. the Laplace integrand with the NON-symmetric coefficient tensor
//               C_km = w |det DF| (lambda Jinv[k][i] Jinv[m][j] + mu Jinv[k][j] Jinv[m][i] + delta_ij mu g^-1[k][m])
//     biharmonic a(u,v) = int (lap u)(lap v) dx element by element, lap = spline.div(spline.grad(.)) (nsd == d;
//               demos/biharmonic/biharmonic.py:100-103, tIGAr/common.py:1022-1040): with Jinv = DF^-1 and the second derivatives
//               H_r,sm of the (rational) map,  lap phi = sum_km g^-1[k][m] d_k d_m phi + sum_k b_k d_k phi,
//               b_k = -sum_r Jinv[k][r] sum_sm g^-1[s][m] H_r,sm   (d_m Jinv = -Jinv (d_m DF) Jinv).  Plain kernel only.
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
#include <utility>

#define TG_ASM_MAXLOC 128      // (p+1)^d local nodes: p <= 4 in 3-D, p <= 8 in 2-D (<= 81), any p <= 8 in 1-D
#define TG_ASM_MAXQ1 10        // Gauss points per direction

struct tg_asm_args {
  int d, p, nsd, nq;
  int nel[3], n[3];            // elements / nodes per direction
  const double *verts[3];      // device: element vertices
  const double *cp[4];         // device: nsd+1 control functions on the node grid
  const double *tab;           // device: l[a][q] (p+1)*nq | dl[a][q] (p+1)*nq | w[q] nq | d2l[a][q] (p+1)*nq  (reference element [0,1])
  int form;                    // 0 mass, 1 laplace, 2 load, 3 block (ei, ej) of the elasticity form, 4 biharmonic
  int ei, ej;
  double lam, mu;
  const int64_t *rowptr;       // pattern (matrix forms)
  double *val;
  const double *fnod;          // load: nodal values
  double *bout;
  int efirst[3];               // this launch: the elements el[k] = efirst[k] + 2 i, i < ncol[k] -- one parity per direction,
  int ncol[3];                 // so they share no node (the colour of the launch = the parities of efirst)
  // row blocks (the z-slab pipeline asks for the FE rows of a range of node planes of the LAST direction):
  int64_t row0, row1;          // rows written: [row0, row1); rowptr / val / bout are those of the block (row - row0)
  int64_t cp_node0;            // the control functions (and fnod) hold the nodes from cp_node0 on
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
  double *td2 = tw + nq1;                          // d2l[a][q]
  double *cpl = td2 + p1 * nq1;                    // [nsd+1][nloc]
  double *G = cpl + (P.nsd + 1) * nloc;            // [nqt][9]  w sqrt(det g) g^-1   (biharmonic: g^-1 (6) | b (3))
  double *S = G + (size_t)nqt * 9;                 // [nqt]     w sqrt(det g)   (load: times f_h)
  double *fl = S + nqt;                            // [nloc]    load: nodal values
  const int tid = threadIdx.x, nt = blockDim.x;
  // element: the blockIdx-th of this launch's colour.  Elements of one colour share no node, so they add into the output
  // without atomics; the 2^d colours follow each other in a fixed order: the assembled values are bit-reproducible
  // (global floating-point atomics added the contributions of the elements around a node in arrival order)
  int64_t e = blockIdx.x;
  int el[3] = {0, 0, 0};
  el[0] = 2 * (int)(e % P.ncol[0]) + P.efirst[0];
  e /= P.ncol[0];
  if (d > 1) {
    el[1] = 2 * (int)(e % P.ncol[1]) + P.efirst[1];
    e /= P.ncol[1];
  }
  if (d > 2) el[2] = 2 * (int)e + P.efirst[2];
  double h[3] = {1.0, 1.0, 1.0};
  for (int k = 0; k < d; k++) h[k] = P.verts[k][el[k] + 1] - P.verts[k][el[k]];
  for (int s = tid; s < 3 * p1 * nq1 + nq1; s += nt) tl[s] = P.tab[s];
  for (int a = tid; a < nloc; a += nt) {
    const int a0 = a % p1, a1 = (a / p1) % p1, a2 = a / (p1 * p1);
    const int64_t node = (int64_t)(el[0] * P.p + a0) + (int64_t)P.n[0] * ((d > 1 ? el[1] * P.p + a1 : 0) +
                                                                         (int64_t)P.n[1] * (d > 2 ? el[2] * P.p + a2 : 0));
    for (int c = 0; c <= P.nsd; c++) cpl[c * nloc + a] = P.cp[c][node - P.cp_node0];
    if (P.form == 2) fl[a] = P.fnod[node - P.cp_node0];
  }
  __syncthreads();
  // quadrature-point data
  for (int q = tid; q < nqt; q += nt) {
    const int qk[3] = {q % nq1, (q / nq1) % nq1, q / (nq1 * nq1)};
    double N[4] = {0, 0, 0, 0}, dN[4][3] = {{0}}, fh = 0.0;
    double d2N[4][6] = {{0}};      // biharmonic: second derivatives 00 01 02 11 12 22 of the control functions
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
      if (P.form == 4) {
        double d2[3] = {0, 0, 0};
        for (int k = 0; k < d; k++) d2[k] = td2[ak[k] * nq1 + qk[k]] / (h[k] * h[k]);
        const double hh[6] = {d2[0] * l[1] * l[2], dl[0] * dl[1] * l[2], dl[0] * l[1] * dl[2],
                              l[0] * d2[1] * l[2], l[0] * dl[1] * dl[2], l[0] * l[1] * d2[2]};
        for (int c = 0; c <= P.nsd; c++) {
          const double v = cpl[c * nloc + a];
          for (int j = 0; j < 6; j++) d2N[c][j] += v * hh[j];
        }
      }
    }
    // DF[i][k] = d(N_i / W)/dxi_k ; metric g = DF^T DF
    const double W = N[P.nsd];
    double g[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, DF[3][3] = {{0}};
    for (int i = 0; i < P.nsd; i++) {
      double *df = DF[i];
      for (int k = 0; k < d; k++) df[k] = (dN[i][k] * W - N[i] * dN[P.nsd][k]) / (W * W);
      for (int k = 0; k < d; k++)
        for (int m = 0; m < d; m++) g[k * d + m] += df[k] * df[m];
    }
    double gi[9], det;
    tg_sym_inverse(d, g, gi, &det);
    double wq = 1.0;
    for (int k = 0; k < d; k++) wq *= tw[qk[k]] * h[k];
    const double s = wq * sqrt(fabs(det));
    if (P.form == 3) {
      // Jinv[k][i] = d xi_k / d x_i = (g^-1 DF^T)[k][i]   (nsd == d: checked on the host)
      double ji[3] = {0, 0, 0}, jj[3] = {0, 0, 0};
      for (int k = 0; k < d; k++)
        for (int m = 0; m < d; m++) {
          ji[k] += gi[k * d + m] * DF[P.ei][m];
          jj[k] += gi[k * d + m] * DF[P.ej][m];
        }
      for (int k = 0; k < d; k++)
        for (int m = 0; m < d; m++)
          G[(size_t)q * 9 + k * d + m] =
              s * (P.lam * ji[k] * jj[m] + P.mu * jj[k] * ji[m] + (P.ei == P.ej ? P.mu * gi[k * d + m] : 0.0));
    } else if (P.form == 4) {
      // second derivatives of F_r = N_r / W:  H_r,km = (N_r,km - F_r,m W,k - F_r,k W,m - F_r W,km) / W
      const int sy[3][3] = {{0, 1, 2}, {1, 3, 4}, {2, 4, 5}};
      double t[3] = {0, 0, 0};                     // t_r = sum_sm g^-1[s][m] H_r,sm
      for (int r = 0; r < d; r++) {
        const double Fr = N[r] / W;
        for (int k = 0; k < d; k++)
          for (int m = 0; m < d; m++) {
            const double Hkm = (d2N[r][sy[k][m]] - DF[r][m] * dN[P.nsd][k] - DF[r][k] * dN[P.nsd][m] - Fr * d2N[P.nsd][sy[k][m]]) / W;
            t[r] += gi[k * d + m] * Hkm;
          }
      }
      double *Gq = G + (size_t)q * 9;
      for (int j = 0; j < 9; j++) Gq[j] = 0.0;
      for (int k = 0; k < d; k++) {
        double bk = 0.0;
        for (int r = 0; r < d; r++) {
          double jkr = 0.0;                        // Jinv[k][r] = (g^-1 DF^T)[k][r]
          for (int m = 0; m < d; m++) jkr += gi[k * d + m] * DF[r][m];
          bk -= jkr * t[r];
        }
        Gq[6 + k] = bk;
        for (int m = k; m < d; m++) Gq[sy[k][m]] = gi[k * d + m];
      }
    } else {
      for (int k = 0; k < d * d; k++) G[(size_t)q * 9 + k] = s * gi[k];
    }
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
      if (node >= P.row0 && node < P.row1) P.bout[node - P.row0] += acc;
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
      } else if (P.form == 4) {
        double ea[3] = {0, 0, 0}, eb[3] = {0, 0, 0};
        for (int k = 0; k < d; k++) {
          ea[k] = td2[ak[k] * nq1 + qk[k]] / (h[k] * h[k]);
          eb[k] = td2[bk[k] * nq1 + qk[k]] / (h[k] * h[k]);
        }
        const double *Gq = G + (size_t)q * 9;
        const double La = Gq[0] * (ea[0] * la[1] * la[2]) + Gq[3] * (la[0] * ea[1] * la[2]) + Gq[5] * (la[0] * la[1] * ea[2]) +
                          2.0 * (Gq[1] * (da[0] * da[1] * la[2]) + Gq[2] * (da[0] * la[1] * da[2]) + Gq[4] * (la[0] * da[1] * da[2])) +
                          Gq[6] * (da[0] * la[1] * la[2]) + Gq[7] * (la[0] * da[1] * la[2]) + Gq[8] * (la[0] * la[1] * da[2]);
        const double Lb = Gq[0] * (eb[0] * lb[1] * lb[2]) + Gq[3] * (lb[0] * eb[1] * lb[2]) + Gq[5] * (lb[0] * lb[1] * eb[2]) +
                          2.0 * (Gq[1] * (db[0] * db[1] * lb[2]) + Gq[2] * (db[0] * lb[1] * db[2]) + Gq[4] * (lb[0] * db[1] * db[2])) +
                          Gq[6] * (db[0] * lb[1] * lb[2]) + Gq[7] * (lb[0] * db[1] * lb[2]) + Gq[8] * (lb[0] * lb[1] * db[2]);
        acc += S[q] * La * Lb;
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
    if (row >= P.row0 && row < P.row1) P.val[P.rowptr[row - P.row0] + pos] += acc;
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


// ------------------------------------------------------------------------------------------------------
// Sum-factorised element matrices of 3-D patches (round 5; SURVEY.md 8f-1: "cfg3 can be fused: element matrices never
// leave LDS" needs an element kernel that is not O((p+1)^(3d)) first).
//
//   A_e[a][b] = sum_q sum_km  d_k phi_a(q) G_km(q) d_m phi_b(q),     G = w_q sqrt(det g) g^-1  (reference coordinates
//   of the element: DF, g and the gradients of the Lagrange functions all refer to xi_hat in [0,1]^3, so no element
//   size appears -- the chain rule holds in any parametrisation, tIGAr/calculusUtils.py:56-70)
//
// with phi_a(q) = l[a0][q0] l[a1][q1] l[a2][q2] and nq = p + 1 Gauss points per direction.  A WAVE walks a piece of a
// LINE of elements along direction 0 (p = 3: one line; two lines side by side at p = 2, eight at p = 1), element by
// element:
//   phase 0  lane = quadrature point: control functions and their xi_hat-gradients by three 1-D contractions through the
//            wave's LDS area (cross-lane), quotient rule, metric, w sqrt(det g) g^-1 -> LDS (6 + 1 doubles per point);
//   phase 1  lane = COLUMN b of the element matrix, the 64 rows in registers:  X_k(q) = sum_m G_km(q) d_m phi_b(q)
//            (G: LDS broadcast; phi_b: lane constants), then the three contractions with the WAVE-UNIFORM 1-D tables
//            (scalar registers) pencil by pencil -- Y[a0] over q0, Z[a0][a1] over q1, acc[a0][a1][a2] over q2 -- with the
//            derivative index merged as soon as two terms share their remaining tables: 2.9e3 fused multiply-adds per
//            lane and element (184e3 per element against 2.4e6 of the plain triple loop), no cross-lane traffic;
//   phase 2  row a of the element leaves as one store of the wave, positions in closed form.  The rows of the element's
//            LAST node plane in direction 0 are shared with the next element of the walk: they stay in registers
//            ("carry") and leave together with that element's first rows, as complete runs of 2p + 1 columns.
// Why the walk: an element writes p + 1 of the 2p + 1 direction-0 columns of a vertex row.  Written element by element
// (round 5's first version: one wave per element, eight colours) half of all stores were 32-byte fragments at 8-byte
// alignment, every 128-byte line of those rows was completed by a LATER launch, and the stores alone took 7.0 of
// the kernel's 7.8 ms at 64^3 elements (1 TB/s).  With the walk the shortest run is 4 (p + 1) x (2p + 1) doubles.
// A piece starts one element early (that element only feeds the carry), so that every FE row is written by exactly one
// wave: no seams, no colouring in direction 0.
// Entries that elements of different lines contribute to (both nodes on a face shared in direction 1 or 2) are STORED
// by the contributor with even index in every shared direction and read, added to and stored again by the others; the
// launches go colour by colour (parities of the element index in directions 1, 2) in ascending order, so the storing
// element always comes first, no two waves of a launch touch one entry and every sum is formed in a fixed order:
// bit-reproducible, no memset, one pass over most entries.  (Global floating-point atomics instead of the read-add-store
// were tried first: 12 of 21 ms -- 1.8e8 lane operations at 15 G/s.)
typedef const double __attribute__((address_space(4))) *tg_cdp4;

struct tg_asf_args {
  int nel[3], n[3];
  const double *cp[4];       // the four homogeneous control functions on the nodes from cp_node0 on
  const double *fnod;        // load: nodal values, same nodes
  int64_t cp_node0;
  const double *tab;         // l[a][q] | dl[a][q] | w[q]   (p+1 points per direction)
  double *val;               // matrix: values of the row block; load: the rows of the vector
  int64_t base;              // matrix: position of the first entry of FE plane za in the whole matrix (subtracted)
  int64_t row0;              // load: first row held by val
  int za, zb;                // FE planes of the last direction whose rows are written
  int efirst[3], ncol[3];    // this launch: elements efirst[k] + 2 i, i < ncol[k] (matrix: k = 1, 2 only)
  int chunk, nchunks;        // matrix: elements per piece of a line, pieces per line
  int ngy;                   // groups of EPW lines (matrix) / EPW elements along direction 0 (load)
  int64_t ngroups;
  int ei, ej;                // elasticity: the block
  double lam, mu;
};

__device__ __forceinline__ void tg_wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

#define TG_ASF_NW 4          // waves per workgroup (each works on its own elements)
// raw buffer access with a per-lane switch: an offset at or beyond the descriptor's range is dropped by the hardware
#define TG_BUF_RANGE 0xfffffff0u
#define TG_BUF_OOB 0xffffffffu
typedef unsigned int tg_v2u __attribute__((ext_vector_type(2)));
__device__ __forceinline__ double tg_buf_load(__amdgpu_buffer_rsrc_t r, unsigned off) {
  return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(r, off, 0, 0));
}
__device__ __forceinline__ void tg_buf_store(double v, __amdgpu_buffer_rsrc_t r, unsigned off) {
  __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(tg_v2u, v), r, off, 0, 0);
}
#define TG_ASF_AREA(NC) (5 * (NC) * 64)   // doubles of LDS per wave: NC nodal functions, first contraction 2 NC, second 3 NC
                                        // (of which NC take the place of the nodal values, dead by then)
#define TG_ASF_CARRY (16 * 64)  // ... and of the carried rows (matrix kernel): (p+1)^2 rows x 64 columns

// NC nodal functions given at the element's nodes in W[c * 64 + slot] (slot = eb + local node) -> value and xi_hat-gradient
// at the lane's quadrature point (x0, x1, x2): three 1-D contractions, each a pass through the wave's LDS area
template <int P1, int NC>
__device__ __forceinline__ void tg_asf_to_points(double *W, const double *TL, const double *TD, int lane, int eb, int x0, int x1,
                                                 int x2, bool active, double *N, double (*dN)[3]) {
  double *B1 = W + NC * 64;
  // the second contraction's 3 NC results: the first NC over the nodal values (dead), the others behind the first's
#define TG_B2(j) (W + ((j) < NC ? (j) : 2 * NC + (j)) * 64)
  if (active) {        // lane (q0, a1, a2): contraction over a0
#pragma unroll
    for (int c = 0; c < NC; c++) {
      double vl = 0.0, vd = 0.0;
#pragma unroll
      for (int a = 0; a < P1; a++) {
        const double v = W[c * 64 + eb + a + P1 * (x1 + P1 * x2)];
        vl = fma(TL[a * P1 + x0], v, vl);
        vd = fma(TD[a * P1 + x0], v, vd);
      }
      B1[c * 64 + lane] = vl;
      B1[(NC + c) * 64 + lane] = vd;
    }
  }
  tg_wave_sync();
  if (active) {        // lane (q0, q1, a2): contraction over a1
#pragma unroll
    for (int c = 0; c < NC; c++) {
      double ll = 0.0, dl = 0.0, ld = 0.0;
#pragma unroll
      for (int a = 0; a < P1; a++) {
        const double vl = B1[c * 64 + eb + x0 + P1 * (a + P1 * x2)];
        const double vd = B1[(NC + c) * 64 + eb + x0 + P1 * (a + P1 * x2)];
        const double tl = TL[a * P1 + x1], td = TD[a * P1 + x1];
        ll = fma(tl, vl, ll);
        dl = fma(tl, vd, dl);
        ld = fma(td, vl, ld);
      }
      TG_B2(c)[lane] = ll;
      TG_B2(NC + c)[lane] = dl;
      TG_B2(2 * NC + c)[lane] = ld;
    }
  }
  tg_wave_sync();
  if (active) {        // lane (q0, q1, q2): contraction over a2
#pragma unroll
    for (int c = 0; c < NC; c++) {
      double v = 0.0, d0 = 0.0, d1 = 0.0, d2 = 0.0;
#pragma unroll
      for (int a = 0; a < P1; a++) {
        const double ll = TG_B2(c)[eb + x0 + P1 * (x1 + P1 * a)];
        const double dl = TG_B2(NC + c)[eb + x0 + P1 * (x1 + P1 * a)];
        const double ld = TG_B2(2 * NC + c)[eb + x0 + P1 * (x1 + P1 * a)];
        const double tl = TL[a * P1 + x2], td = TD[a * P1 + x2];
        v = fma(tl, ll, v);
        d0 = fma(tl, dl, d0);
        d1 = fma(tl, ld, d1);
        d2 = fma(td, ll, d2);
      }
      N[c] = v;
      dN[c][0] = d0;
      dN[c][1] = d1;
      dN[c][2] = d2;
    }
  }
#undef TG_B2
}

// w sqrt(det g) g^-1 (G[0..5]: 00 01 02 11 12 22) and w sqrt(det g) (G[6]) at the lane's quadrature point
__device__ __forceinline__ void tg_asf_metric(const double *N, const double (*dN)[3], double w, double *G) {
  const double Wt = N[3];
  double g[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
  for (int i = 0; i < 3; i++) {
    double df[3];
#pragma unroll
    for (int k = 0; k < 3; k++) df[k] = (dN[i][k] * Wt - N[i] * dN[3][k]) / (Wt * Wt);
#pragma unroll
    for (int k = 0; k < 3; k++)
#pragma unroll
      for (int m = 0; m < 3; m++) g[k * 3 + m] += df[k] * df[m];
  }
  double gi[9], det;
  tg_sym_inverse(3, g, gi, &det);
  const double s = w * sqrt(fabs(det));
  G[0] = s * gi[0];
  G[1] = s * gi[1];
  G[2] = s * gi[2];
  G[3] = s * gi[4];
  G[4] = s * gi[5];
  G[5] = s * gi[8];
  G[6] = s;
}

// block (ei, ej) of the elasticity form: the coefficient tensor C[k][m] (G[3 k + m], row index k = derivative of the TEST
// function) = w |det DF| (lambda Jinv[k][ei] Jinv[m][ej] + mu Jinv[k][ej] Jinv[m][ei] + delta mu sum_l Jinv[k][l] Jinv[m][l]),
// Jinv = DF^-1 by cofactors
__device__ __forceinline__ void tg_asf_elast(const double *N, const double (*dN)[3], double w, int ei, int ej, double lam,
                                             double mu, double *G) {
  const double Wt = N[3];
  double F[3][3];
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int k = 0; k < 3; k++) F[i][k] = (dN[i][k] * Wt - N[i] * dN[3][k]) / (Wt * Wt);
  // J[k][i] = cofactor(F)[i][k] / det
  double J[3][3];
  J[0][0] = F[1][1] * F[2][2] - F[1][2] * F[2][1];
  J[0][1] = F[0][2] * F[2][1] - F[0][1] * F[2][2];
  J[0][2] = F[0][1] * F[1][2] - F[0][2] * F[1][1];
  J[1][0] = F[1][2] * F[2][0] - F[1][0] * F[2][2];
  J[1][1] = F[0][0] * F[2][2] - F[0][2] * F[2][0];
  J[1][2] = F[0][2] * F[1][0] - F[0][0] * F[1][2];
  J[2][0] = F[1][0] * F[2][1] - F[1][1] * F[2][0];
  J[2][1] = F[0][1] * F[2][0] - F[0][0] * F[2][1];
  J[2][2] = F[0][0] * F[1][1] - F[0][1] * F[1][0];
  const double det = F[0][0] * J[0][0] + F[0][1] * J[1][0] + F[0][2] * J[2][0];
  const double rd = 1.0 / det;
#pragma unroll
  for (int k = 0; k < 3; k++)
#pragma unroll
    for (int i = 0; i < 3; i++) J[k][i] *= rd;
  const double s = w * fabs(det);
  // (ei, ej are wave-uniform: selects, no indexed register access)
  double ji[3], jj[3];
#pragma unroll
  for (int k = 0; k < 3; k++) {
    ji[k] = ei == 0 ? J[k][0] : (ei == 1 ? J[k][1] : J[k][2]);
    jj[k] = ej == 0 ? J[k][0] : (ej == 1 ? J[k][1] : J[k][2]);
  }
  const double dm = ei == ej ? mu : 0.0;
#pragma unroll
  for (int k = 0; k < 3; k++)
#pragma unroll
    for (int m = 0; m < 3; m++) {
      const double gkm = J[k][0] * J[m][0] + J[k][1] * J[m][1] + J[k][2] * J[m][2];
      G[3 * k + m] = s * (lam * ji[k] * jj[m] + mu * jj[k] * ji[m] + dm * gkm);
    }
}

// X_k = sum_m C_km f_m at one quadrature point: FORM 1 -- the symmetric tensor of tg_asf_metric (6 values), FORM 2 -- the full
// tensor of tg_asf_elast (9 values)
template <int FORM>
__device__ __forceinline__ void tg_asf_flux(const double *Gq, double f0, double f1, double f2, double &X0, double &X1, double &X2) {
  if (FORM == 2) {
    X0 = fma(Gq[2], f2, fma(Gq[1], f1, Gq[0] * f0));
    X1 = fma(Gq[5], f2, fma(Gq[4], f1, Gq[3] * f0));
    X2 = fma(Gq[8], f2, fma(Gq[7], f1, Gq[6] * f0));
  } else {
    X0 = fma(Gq[2], f2, fma(Gq[1], f1, Gq[0] * f0));
    X1 = fma(Gq[4], f2, fma(Gq[3], f1, Gq[1] * f0));
    X2 = fma(Gq[5], f2, fma(Gq[4], f1, Gq[2] * f0));
  }
}

// 1-D row data of node a of element e (direction with nel elements): vertex shared with a neighbour?, row length,
// position of the element's first column in the row, entries of the 1-D rows before the node
template <int P>
__device__ __forceinline__ void tg_asf_row1d(int a, int e, int nel, bool &v, int &n, int &o, int64_t &rps) {
  v = (a == 0 && e > 0) || (a == P && e < nel - 1);
  n = v ? 2 * P + 1 : P + 1;
  o = (a == 0 && e > 0) ? P : 0;
  rps = (int64_t)(P + 1) * (P * e + a) + (a > 0 ? (int64_t)P * e : (e > 0 ? (int64_t)P * (e - 1) : 0));
}

template <int P1, int EPW, int FORM>
__global__ void __launch_bounds__(64 * TG_ASF_NW) k_asf3(tg_asf_args A) {
  constexpr int P = P1 - 1, NL = P1 * P1 * P1, LPE = 64 / EPW, PP = P1 * P1;
  constexpr int AH = P1;                 // (rows of the first local index a0 handled at once: all)
  constexpr int GS = FORM == 2 ? 10 : 8, GN = FORM == 2 ? 9 : 7;     // doubles per quadrature point in LDS: stride, values
  static_assert(NL <= LPE, "an element needs a lane per local node");
  __shared__ __attribute__((aligned(16))) double s_tab[2 * PP + P1];
  __shared__ __attribute__((aligned(16))) double s_w[TG_ASF_NW][TG_ASF_AREA(4)];
  __shared__ __attribute__((aligned(16))) double s_c[TG_ASF_NW][TG_ASF_CARRY];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);       // (wave-uniform: element indices and row positions are scalars)
  for (int s = tid; s < 2 * PP + P1; s += 64 * TG_ASF_NW) s_tab[s] = A.tab[s];
  __syncthreads();
  const int64_t grp = (int64_t)blockIdx.x * TG_ASF_NW + wv;
  if (grp >= A.ngroups) return;
  const double *TL = s_tab, *TD = s_tab + PP, *TW = s_tab + 2 * PP;
  tg_cdp4 UL = (tg_cdp4)A.tab, UD = (tg_cdp4)A.tab + PP;
  double *W = s_w[wv];
  const int es = lane / LPE, eb = es * LPE;
  // the piece of the walk and the line(s) of this wave
  const int piece = (int)(grp % A.nchunks);
  int64_t rest = grp / A.nchunks;
  const int gy = (int)(rest % A.ngy), i2 = (int)(rest / A.ngy);
  const int i1 = gy * EPW + es;
  const bool evalid = i1 < A.ncol[1];
  const int e1_ = A.efirst[1] + 2 * (evalid ? i1 : A.ncol[1] - 1), e2_ = A.efirst[2] + 2 * i2;
  const int e_lo = piece * A.chunk, e_hi = min(A.nel[0], e_lo + A.chunk);
  const int nel0 = A.nel[0];
  double *CR = s_c[wv];                 // carried rows [a1 + P1 a2][lane]
  double cpn[4];
  {
    const int li = lane - eb, x0 = li % P1, x1 = (li / P1) % P1, x2 = (li / PP) % P1;
    const int64_t nodebase = (int64_t)x0 + (int64_t)A.n[0] * ((e1_ * P + x1) + (int64_t)A.n[1] * (e2_ * P + x2)) - A.cp_node0;
    const int ef = e_lo > 0 ? e_lo - 1 : 0;
#pragma unroll
    for (int c = 0; c < 4; c++) cpn[c] = li < NL ? A.cp[c][nodebase + (int64_t)ef * P] : 0.0;
  }

  for (int e0 = (e_lo > 0 ? e_lo - 1 : 0); e0 < e_hi; e0++) {
    const bool carry_only = e0 < e_lo;
    // Everything that depends on the lane or on the line is re-derived per element from values the compiler cannot see
    // through: hoisted out of the walk, the row positions, lane tables and predicates of phase 2 alone would take more
    // registers than the kernel has.
    int ln = lane, e1 = e1_, e2 = e2_;
    asm volatile("" : "+v"(ln));
    if (EPW == 1)
      asm volatile("" : "+s"(e1));
    else
      asm volatile("" : "+v"(e1));
    asm volatile("" : "+s"(e2));
    const int li = ln - eb;
    const bool active = li < NL;
    const bool live = active && evalid;
    const int x0 = li % P1, x1 = (li / P1) % P1, x2 = (li / PP) % P1;
    const int64_t T0 = (int64_t)P1 * A.n[0] + (int64_t)P * (A.nel[0] - 1), T1 = (int64_t)P1 * A.n[1] + (int64_t)P * (A.nel[1] - 1);
    const int par1 = e1 & 1, par2 = e2 & 1;
    // ---- phase 0 --------------------------------------------------------------------------------------------
    if (active) {
#pragma unroll
      for (int c = 0; c < 4; c++) W[c * 64 + ln] = cpn[c];
    }
    tg_wave_sync();
    if (e0 + 1 < e_hi) {                 // the next element's nodal values travel while this one is integrated
      const int64_t nodebase = (int64_t)x0 + (int64_t)A.n[0] * ((e1 * P + x1) + (int64_t)A.n[1] * (e2 * P + x2)) - A.cp_node0;
#pragma unroll
      for (int c = 0; c < 4; c++) cpn[c] = active ? A.cp[c][nodebase + (int64_t)(e0 + 1) * P] : 0.0;
    }
    {
      double N[4], dN[4][3], G[GN];
#pragma unroll
      for (int j = 0; j < GN; j++) G[j] = 0.0;
      tg_asf_to_points<P1, 4>(W, TL, TD, ln, eb, x0, x1, x2, active, N, dN);
      if (active) {
        if constexpr (FORM == 2)
          tg_asf_elast(N, dN, TW[x0] * TW[x1] * TW[x2], A.ei, A.ej, A.lam, A.mu, G);
        else
          tg_asf_metric(N, dN, TW[x0] * TW[x1] * TW[x2], G);
      }
      tg_wave_sync();      // (the area of the nodal values is free: nobody reads it any more)
      if (active) {
#pragma unroll
        for (int j = 0; j < GN; j++) W[(eb + li) * GS + j] = G[j];
      }
      tg_wave_sync();
    }

    const bool last0 = e0 == nel0 - 1;
    // (Forming the rows in two halves of a0 -- accumulators and intermediates halved, the integrand computed twice, +25 %
    //  multiply-adds -- was meant to fit 256 registers and two waves per SIMD: the compiler shares the integrand between
    //  unrolled halves (1 250 values in scratch) and trades vector for scalar registers in a loop over them (400): 30 ms
    //  against 14 at 64^3 elements.  Dropped; the constant h below is what is left of it.)
    {
      constexpr int h = 0, ab = 0;
      // ---- phase 1: lane = column b = (x0, x1, x2) -------------------------------------------------------------
      double acc[AH * PP];
#pragma unroll
      for (int i = 0; i < AH * PP; i++) acc[i] = 0.0;
      {
        const int xh0 = x0;
        double *Wh = W;
        double l0[P1], d0[P1];
#pragma unroll
        for (int q = 0; q < P1; q++) {
          l0[q] = TL[xh0 * P1 + q];
          d0[q] = TD[xh0 * P1 + q];
        }

#pragma unroll
        for (int q2 = 0; q2 < P1; q2++) {
          const double l2q = TL[x2 * P1 + q2], d2q = TD[x2 * P1 + q2];
          double Zl[AH * P1], Zd[AH * P1];
#pragma unroll
          for (int i = 0; i < AH * P1; i++) Zl[i] = Zd[i] = 0.0;
#pragma unroll
          for (int q1 = 0; q1 < P1; q1++) {
            const double l1q = TL[x1 * P1 + q1], d1q = TD[x1 * P1 + q1];
            if (FORM >= 1) {
              const double mll = l1q * l2q, mdl = d1q * l2q, mld = l1q * d2q;
              double Y0[AH], Y1[AH], Y2[AH];
#pragma unroll
              for (int a = 0; a < AH; a++) Y0[a] = Y1[a] = Y2[a] = 0.0;
#pragma unroll
              for (int q0 = 0; q0 < P1; q0++) {
                const double *Gq = Wh + (eb + q0 + P1 * (q1 + P1 * q2)) * GS;
                const double f0 = d0[q0] * mll, f1 = l0[q0] * mdl, f2 = l0[q0] * mld;
                double X0, X1, X2;
                tg_asf_flux<FORM>(Gq, f0, f1, f2, X0, X1, X2);
#pragma unroll
                for (int a = 0; a < AH; a++) {
                  Y0[a] = fma(UD[a * P1 + q0], X0, Y0[a]);
                  Y1[a] = fma(UL[a * P1 + q0], X1, Y1[a]);
                  Y2[a] = fma(UL[a * P1 + q0], X2, Y2[a]);
                }
              }
#pragma unroll
              for (int a1 = 0; a1 < P1; a1++)
#pragma unroll
                for (int a0 = 0; a0 < AH; a0++) {
                  Zl[a0 + AH * a1] = fma(UD[a1 * P1 + q1], Y1[a0], fma(UL[a1 * P1 + q1], Y0[a0], Zl[a0 + AH * a1]));
                  Zd[a0 + AH * a1] = fma(UL[a1 * P1 + q1], Y2[a0], Zd[a0 + AH * a1]);
                }
            } else {
              const double mll = l1q * l2q;
              double Y0[AH];
#pragma unroll
              for (int a = 0; a < AH; a++) Y0[a] = 0.0;
#pragma unroll
              for (int q0 = 0; q0 < P1; q0++) {
                const double X0 = Wh[(eb + q0 + P1 * (q1 + P1 * q2)) * 8 + 6] * (l0[q0] * mll);
#pragma unroll
                for (int a = 0; a < AH; a++) Y0[a] = fma(UL[a * P1 + q0], X0, Y0[a]);
              }
#pragma unroll
              for (int a1 = 0; a1 < P1; a1++)
#pragma unroll
                for (int a0 = 0; a0 < AH; a0++) Zl[a0 + AH * a1] = fma(UL[a1 * P1 + q1], Y0[a0], Zl[a0 + AH * a1]);
            }
            __builtin_amdgcn_sched_barrier(0);   // (one pencil at a time: the scheduler otherwise pulls the LDS reads of all of them up)
          }
#pragma unroll
          for (int a2 = 0; a2 < P1; a2++)
#pragma unroll
            for (int i = 0; i < AH * P1; i++) {
              if (FORM >= 1)
                acc[i + AH * P1 * a2] = fma(UD[a2 * P1 + q2], Zd[i], fma(UL[a2 * P1 + q2], Zl[i], acc[i + AH * P1 * a2]));
              else
                acc[i + AH * P1 * a2] = fma(UL[a2 * P1 + q2], Zl[i], acc[i + AH * P1 * a2]);
            }
          __builtin_amdgcn_sched_barrier(0);
        }
      }

      // ---- phase 2 -----------------------------------------------------------------------------------------------
      // Positions: one base per node plane of direction 2 (a buffer descriptor), everything below it a 32-bit byte
      // offset (the entries of one plane of rows: (2p+1) T0 T1 < 2^29, checked on the host).  Reads and stores are RAW
      // BUFFER operations: a lane that has nothing to read or write carries an offset beyond the descriptor's range -- the
      // hardware drops the access -- so the whole phase is straight-line code without a branch per row.
      // Two sweeps over the rows of the half: the first issues ALL reads of entries that hold earlier contributions (rows
      // on faces shared in directions 1, 2, lanes of the shared columns), the second adds and stores: one round trip to
      // memory per half, not one per (a1, a2) group of rows (12 of ~3 us per element: 67 % of the wave's cycles).
      const int srcP = min(63, ln - x0 + P);           // the lane that holds column b0 = p of this lane's (b1, b2)
      const unsigned uT0 = (unsigned)T0;
      if (!carry_only) {
        // (p = 3, stiffness: the registers hold the reads of ONE a2 layer of rows at a time -- four round trips per element)
        constexpr bool LAYERED = FORM >= 1 && P1 == 4;
        constexpr int NPASS = LAYERED ? P1 : 1;
        double old[(LAYERED ? P1 : PP) * (AH + 1)];
#pragma unroll
        for (int pass = 0; pass < NPASS; pass++) {
#pragma unroll
          for (int sweep = 0; sweep < 2; sweep++) {
            // (the offsets are formed again in the second sweep: kept from the first, they would sit in registers)
            int xs0 = x0, xs1 = x1, xs2 = x2;
            asm volatile("" : "+v"(xs0), "+v"(xs1), "+v"(xs2));
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int a2 = LAYERED ? pass : 0; a2 < (LAYERED ? pass + 1 : P1); a2++) {
              const int r2 = P * e2 + a2;
              if (r2 < A.za || r2 >= A.zb) continue;                    // (wave-uniform)
              bool v2;
              int n2, o2;
              int64_t rps2;
              tg_asf_row1d<P>(a2, e2, A.nel[2], v2, n2, o2, rps2);
              const bool add2 = v2 && par2 && x2 == a2;
              const __amdgpu_buffer_rsrc_t plane =
                  __builtin_amdgcn_make_buffer_rsrc(A.val + (T0 * T1 * rps2 - A.base), 0, TG_BUF_RANGE, 0x00020000);
#pragma unroll
              for (int a1 = 0; a1 < P1; a1++) {
                bool v1;
                int n1, o1;
                int64_t rps1;
                tg_asf_row1d<P>(a1, e1, A.nel[1], v1, n1, o1, rps1);
                const bool add = live && (add2 || (v1 && par1 && x1 == a1));
                const bool edge12 = a2 == 0 || a2 == P || a1 == 0 || a1 == P;     // rows that may hold earlier contributions
                const unsigned R12 = (unsigned)n2 * uT0 * (unsigned)rps1;
                const unsigned n12 = (unsigned)(n2 * n1);
                const unsigned p12 = (unsigned)((xs2 + o2) * n1 + (xs1 + o1));
                double *og = old + (a1 + (LAYERED ? 0 : P1 * a2)) * (AH + 1);
                // first node plane of the element: the rows of the previous element's last plane, completed
                if (h == 0 && e0 > 0) {
                  const unsigned rps0 = (unsigned)(P1 * P * e0 + P * (e0 - 1));
                  const unsigned ro = R12 + n12 * rps0 + p12 * (2 * P + 1);
                  const unsigned oprev = (ro + (unsigned)xs0) * 8u, ocur = (ro + (unsigned)(P + xs0)) * 8u;
                  if (sweep == 0) {
                    og[0] = edge12 ? tg_buf_load(plane, add ? ocur : TG_BUF_OOB) : 0.0;
                    og[AH] = edge12 ? tg_buf_load(plane, (add && x0 < P) ? oprev : TG_BUF_OOB) : 0.0;
                  } else {
                    const double cP = CR[(a1 + P1 * a2) * 64 + srcP];     // column b0 = p of the previous element = b0 = 0 here
                    double vcur = acc[0 + AH * (a1 + P1 * a2)] + (x0 == 0 ? cP : 0.0);
                    double vprev = CR[(a1 + P1 * a2) * 64 + ln];
                    if (edge12) {
                      vcur += og[0];
                      vprev += og[AH];
                    }
                    tg_buf_store(vprev, plane, (live && x0 < P) ? oprev : TG_BUF_OOB);
                    tg_buf_store(vcur, plane, live ? ocur : TG_BUF_OOB);
                  }
                }
#pragma unroll
                for (int ah = 0; ah < AH; ah++) {
                  const int a0 = ab + ah;                                 // (wave-uniform)
                  if (a0 > P) continue;
                  if (a0 == 0 && e0 > 0) continue;                        // (written above, merged)
                  if (a0 == P && !last0) continue;                        // (carried to the next element)
                  const unsigned rps0 = (unsigned)(P1 * (P * e0 + a0) + (a0 > 0 ? P * e0 : 0));
                  const unsigned off = (R12 + n12 * rps0 + p12 * P1 + (unsigned)xs0) * 8u;
                  if (sweep == 0) {
                    og[ah] = edge12 ? tg_buf_load(plane, add ? off : TG_BUF_OOB) : 0.0;
                  } else {
                    double v = acc[ah + AH * (a1 + P1 * a2)];
                    if (edge12) v += og[ah];
                    tg_buf_store(v, plane, live ? off : TG_BUF_OOB);
                  }
                }
                if (sweep == 1) __builtin_amdgcn_sched_barrier(0);      // (stores group by group: their operands are not collected up front)
              }
            }
          }
        }
      }
      // the element's last node plane in direction 0 becomes the carry
      if (!last0) {
#pragma unroll
        for (int i = 0; i < PP; i++) CR[i * 64 + ln] = acc[P + AH * i];
      }
    }
    tg_wave_sync();        // (the next element's nodal values overwrite the area G sits in)
  }
}

// The same element matrices with ONE ELEMENT PER WAVE and eight colours (parities of the element index in all three
// directions): no walk, no carry -- rows of vertices in direction 0 leave as runs of p + 1 columns.  Kept for the p = 3
// stiffness matrix, where the walk's loop costs more registers than the kernel has (512 and 300 values in scratch;
// 14 ms against this kernel's 7 ms at 64^3 elements); every other (p, form) is faster on the walk (§DESIGN 6b).
template <int P1, int EPW, int FORM>
__global__ void __launch_bounds__(64 * TG_ASF_NW) k_asf3_elem(tg_asf_args A) {
  constexpr int P = P1 - 1, NL = P1 * P1 * P1, LPE = 64 / EPW, PP = P1 * P1;
  constexpr int GS = FORM == 2 ? 10 : 8, GN = FORM == 2 ? 9 : 7;
  static_assert(NL <= LPE, "an element needs a lane per local node");
  __shared__ __attribute__((aligned(16))) double s_tab[2 * PP + P1];
  __shared__ __attribute__((aligned(16))) double s_w[TG_ASF_NW][TG_ASF_AREA(4)];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  for (int s = tid; s < 2 * PP + P1; s += 64 * TG_ASF_NW) s_tab[s] = A.tab[s];
  __syncthreads();
  const int64_t grp = (int64_t)blockIdx.x * TG_ASF_NW + wv;
  if (grp >= A.ngroups) return;
  const double *TL = s_tab, *TD = s_tab + PP, *TW = s_tab + 2 * PP;
  tg_cdp4 UL = (tg_cdp4)A.tab, UD = (tg_cdp4)A.tab + PP;
  double *W = s_w[wv];
  const int es = lane / LPE, li = lane - es * LPE, eb = es * LPE;
  const bool active = li < NL;
  const int x0 = li % P1, x1 = (li / P1) % P1, x2 = (li / PP) % P1;
  const int gx = (int)(grp % A.ngy);
  const int64_t rest = grp / A.ngy;
  const int i1 = (int)(rest % A.ncol[1]), i2 = (int)(rest / A.ncol[1]);
  const int i0 = gx * EPW + es;
  const bool evalid = i0 < A.ncol[0];
  const bool live = active && evalid;
  const int e0 = A.efirst[0] + 2 * (evalid ? i0 : A.ncol[0] - 1), e1 = A.efirst[1] + 2 * i1, e2 = A.efirst[2] + 2 * i2;
  // ---- phase 0 ----------------------------------------------------------------------------------------------
  if (active) {
    const int64_t node = (int64_t)(e0 * P + x0) + (int64_t)A.n[0] * ((e1 * P + x1) + (int64_t)A.n[1] * (e2 * P + x2)) - A.cp_node0;
#pragma unroll
    for (int c = 0; c < 4; c++) W[c * 64 + lane] = A.cp[c][node];
  }
  tg_wave_sync();
  {
    double N[4], dN[4][3], G[GN];
#pragma unroll
    for (int j = 0; j < GN; j++) G[j] = 0.0;
    tg_asf_to_points<P1, 4>(W, TL, TD, lane, eb, x0, x1, x2, active, N, dN);
    if (active) {
      if constexpr (FORM == 2)
        tg_asf_elast(N, dN, TW[x0] * TW[x1] * TW[x2], A.ei, A.ej, A.lam, A.mu, G);
      else
        tg_asf_metric(N, dN, TW[x0] * TW[x1] * TW[x2], G);
    }
    tg_wave_sync();
    if (active) {
#pragma unroll
      for (int j = 0; j < GN; j++) W[(eb + li) * GS + j] = G[j];
    }
    tg_wave_sync();
  }
  // ---- phase 1: lane = column b = (x0, x1, x2) -----------------------------------------------------------------
  double acc[NL];
#pragma unroll
  for (int i = 0; i < NL; i++) acc[i] = 0.0;
  {
    double l0[P1], d0[P1];
#pragma unroll
    for (int q = 0; q < P1; q++) {
      l0[q] = TL[x0 * P1 + q];
      d0[q] = TD[x0 * P1 + q];
    }
#pragma unroll
    for (int q2 = 0; q2 < P1; q2++) {
      const double l2q = TL[x2 * P1 + q2], d2q = TD[x2 * P1 + q2];
      double Zl[PP], Zd[PP];
#pragma unroll
      for (int i = 0; i < PP; i++) Zl[i] = Zd[i] = 0.0;
#pragma unroll
      for (int q1 = 0; q1 < P1; q1++) {
        const double l1q = TL[x1 * P1 + q1], d1q = TD[x1 * P1 + q1];
        if (FORM >= 1) {
          const double mll = l1q * l2q, mdl = d1q * l2q, mld = l1q * d2q;
          double Y0[P1], Y1[P1], Y2[P1];
#pragma unroll
          for (int a = 0; a < P1; a++) Y0[a] = Y1[a] = Y2[a] = 0.0;
#pragma unroll
          for (int q0 = 0; q0 < P1; q0++) {
            const double *Gq = W + (eb + q0 + P1 * (q1 + P1 * q2)) * GS;
            const double f0 = d0[q0] * mll, f1 = l0[q0] * mdl, f2 = l0[q0] * mld;
            double X0, X1, X2;
            tg_asf_flux<FORM>(Gq, f0, f1, f2, X0, X1, X2);
#pragma unroll
            for (int a = 0; a < P1; a++) {
              Y0[a] = fma(UD[a * P1 + q0], X0, Y0[a]);
              Y1[a] = fma(UL[a * P1 + q0], X1, Y1[a]);
              Y2[a] = fma(UL[a * P1 + q0], X2, Y2[a]);
            }
          }
#pragma unroll
          for (int a1 = 0; a1 < P1; a1++)
#pragma unroll
            for (int a0 = 0; a0 < P1; a0++) {
              Zl[a0 + P1 * a1] = fma(UD[a1 * P1 + q1], Y1[a0], fma(UL[a1 * P1 + q1], Y0[a0], Zl[a0 + P1 * a1]));
              Zd[a0 + P1 * a1] = fma(UL[a1 * P1 + q1], Y2[a0], Zd[a0 + P1 * a1]);
            }
        } else {
          const double mll = l1q * l2q;
          double Y0[P1];
#pragma unroll
          for (int a = 0; a < P1; a++) Y0[a] = 0.0;
#pragma unroll
          for (int q0 = 0; q0 < P1; q0++) {
            const double X0 = W[(eb + q0 + P1 * (q1 + P1 * q2)) * 8 + 6] * (l0[q0] * mll);
#pragma unroll
            for (int a = 0; a < P1; a++) Y0[a] = fma(UL[a * P1 + q0], X0, Y0[a]);
          }
#pragma unroll
          for (int a1 = 0; a1 < P1; a1++)
#pragma unroll
            for (int a0 = 0; a0 < P1; a0++) Zl[a0 + P1 * a1] = fma(UL[a1 * P1 + q1], Y0[a0], Zl[a0 + P1 * a1]);
        }
      }
#pragma unroll
      for (int a2 = 0; a2 < P1; a2++)
#pragma unroll
        for (int i = 0; i < PP; i++) {
          if (FORM >= 1)
            acc[i + PP * a2] = fma(UD[a2 * P1 + q2], Zd[i], fma(UL[a2 * P1 + q2], Zl[i], acc[i + PP * a2]));
          else
            acc[i + PP * a2] = fma(UL[a2 * P1 + q2], Zl[i], acc[i + PP * a2]);
        }
    }
  }
  // ---- phase 2: layer by layer of a2; per layer all reads of earlier contributions, then the stores ------------------
  const int64_t T0 = (int64_t)P1 * A.n[0] + (int64_t)P * (A.nel[0] - 1), T1 = (int64_t)P1 * A.n[1] + (int64_t)P * (A.nel[1] - 1);
  const unsigned uT0 = (unsigned)T0;
  const int par0 = e0 & 1, par1 = e1 & 1, par2 = e2 & 1;
#pragma unroll
  for (int a2 = 0; a2 < P1; a2++) {
    const int r2 = P * e2 + a2;
    if (r2 < A.za || r2 >= A.zb) continue;                        // (wave-uniform)
    bool v2;
    int n2, o2;
    int64_t rps2;
    tg_asf_row1d<P>(a2, e2, A.nel[2], v2, n2, o2, rps2);
    const bool add2 = v2 && par2 && x2 == a2;
    const __amdgpu_buffer_rsrc_t plane =
        __builtin_amdgcn_make_buffer_rsrc(A.val + (T0 * T1 * rps2 - A.base), 0, TG_BUF_RANGE, 0x00020000);
    double old[PP];
#pragma unroll
    for (int sweep = 0; sweep < 2; sweep++) {
      int xs0 = x0, xs1 = x1, xs2 = x2;
      asm volatile("" : "+v"(xs0), "+v"(xs1), "+v"(xs2));
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int a1 = 0; a1 < P1; a1++) {
        bool v1;
        int n1, o1;
        int64_t rps1;
        tg_asf_row1d<P>(a1, e1, A.nel[1], v1, n1, o1, rps1);
        const bool add1 = v1 && par1 && x1 == a1;
#pragma unroll
        for (int a0 = 0; a0 < P1; a0++) {
          bool v0;
          int n0, o0;
          int64_t rps0;
          tg_asf_row1d<P>(a0, e0, A.nel[0], v0, n0, o0, rps0);
          const bool add = live && (add2 || add1 || (v0 && par0 && x0 == a0));
          const bool edge = a2 == 0 || a2 == P || a1 == 0 || a1 == P || a0 == 0 || a0 == P;
          const unsigned off = ((unsigned)n2 * (uT0 * (unsigned)rps1 + (unsigned)n1 * (unsigned)rps0) +
                                (unsigned)(((xs2 + o2) * n1 + (xs1 + o1)) * n0 + (xs0 + o0))) * 8u;
          if (sweep == 0) {
            old[a0 + P1 * a1] = edge ? tg_buf_load(plane, add ? off : TG_BUF_OOB) : 0.0;
          } else {
            double v = acc[a0 + P1 * (a1 + P1 * a2)];
            if (edge) v += old[a0 + P1 * a1];
            tg_buf_store(v, plane, live ? off : TG_BUF_OOB);
          }
        }
      }
    }
  }
}

// The element kernel above with the FOUR WAVES OF A WORKGROUP ON FOUR CONSECUTIVE ELEMENTS of a line along direction 0
// (round 6); launches go by the parities of (group index, element index in direction 1, in direction 2).
// Why: with independent elements and eight colours the rows of vertices in direction 0 (half of all rows) leave as 32-byte
// runs whose 56-byte stretches are completed by a LATER launch -- every such line reaches the memory twice, partially
// written (PMC: 4.9 GB written for 2.8 GB of values), and the stores are what the kernel waits for.  Inside a group the
// two halves of a stretch are now written by two waves of one CU within a microsecond of each other and meet in the L2;
// what is left of the fragments are the seams between groups (one face in four).  Unlike the walk (k_asf3) there is no loop
// and no carried row -- that loop needs 300 registers more than exist at p = 3 (and so does a plain loop around this body:
// 1.5 - 3 KB of scratch per lane) -- only the 16 x 16 block of a shared face (both nodes on it) changes hands: the wave whose
// element ends on the face leaves it in LDS, its neighbour adds it to its own and stores the sum; no entry is read back.
// Seams between groups and faces shared in directions 1, 2: stored by the contributor with even parity, read - added -
// stored by the others in launches of ascending colour, as before: bit-reproducible, no atomics.  The groups depend on
// nel[0] only, not on the window [za, zb) of rows: row blocks are bit-identical to the whole matrix.
// LOOP: the workgroup walks a piece of A.chunk groups (faces in three generations of LDS buffers, the next group's nodal
// values requested while the rows of this one leave); the loop fits the registers because its body has no branch.
// Measured and not kept: the four waves on four different LINES, each walking its own piece with the face carried inside
// the wave (no barrier in the loop): waits 17 -> 7 % of the wave cycles, VALU active 46 -> 49 %, but the two halves of a
// stretch are then stored 17 us apart and the L2 has let go of the line: 11.7 instead of 9.4 GB written per matrix at 64^3
// elements, 34.9 - 43 ms instead of 34.4 - 37 ms at 128^3.
#define TG_ASF_FACE (16 * 16)
template <typename F>
__device__ __forceinline__ void tg_asf_each4(F &&f) {
  f(std::integral_constant<int, 0>{});
  f(std::integral_constant<int, 1>{});
  f(std::integral_constant<int, 2>{});
  f(std::integral_constant<int, 3>{});
}
template <int P1, int FORM, int PRE, int LOOP>
__global__ void __launch_bounds__(64 * TG_ASF_NW) k_asf3_quad(tg_asf_args A) {
  constexpr int P = P1 - 1, NL = P1 * P1 * P1, PP = P1 * P1;
  constexpr int GS = FORM == 2 ? 10 : 8, GN = FORM == 2 ? 9 : 7;
  static_assert(NL == 64 && PP <= 16, "one element per wave, a lane per local node");
  __shared__ __attribute__((aligned(16))) double s_tab[2 * PP + P1];
  __shared__ __attribute__((aligned(16))) double s_w[TG_ASF_NW][TG_ASF_AREA(4)];
  // faces in three generations: a wave may write the face of its next element while its neighbour still reads the last one
  __shared__ __attribute__((aligned(16))) double s_f[LOOP ? 3 : 1][TG_ASF_NW][TG_ASF_FACE + 64];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  for (int s = tid; s < 2 * PP + P1; s += 64 * TG_ASF_NW) s_tab[s] = A.tab[s];
  __syncthreads();
  const double *TL = s_tab, *TD = s_tab + PP, *TW = s_tab + 2 * PP;
  tg_cdp4 UL = (tg_cdp4)A.tab, UD = (tg_cdp4)A.tab + PP;
  double *W = s_w[wv];
  const int64_t grp = blockIdx.x;
  const int gx = (int)(grp % A.ngy);
  const int64_t rest = grp / A.ngy;
  const int i1 = (int)(rest % A.ncol[1]), i2 = (int)(rest / A.ncol[1]);
  const int piece = A.efirst[0] + 2 * gx, pc = piece & 1;
  const int e1_ = A.efirst[1] + 2 * i1, e2_ = A.efirst[2] + 2 * i2;
  const int nq = (A.nel[0] + TG_ASF_NW - 1) / TG_ASF_NW;
  const int q_lo = LOOP ? piece * A.chunk : piece, q_hi = LOOP ? min(nq, q_lo + A.chunk) : q_lo + 1;
  const int64_t T0 = (int64_t)P1 * A.n[0] + (int64_t)P * (A.nel[0] - 1), T1 = (int64_t)P1 * A.n[1] + (int64_t)P * (A.nel[1] - 1);
  const unsigned uT0 = (unsigned)T0;
  double cpn[4];
  {
    const int x0 = lane % P1, x1 = (lane / P1) % P1, x2 = lane / PP;
    const int e0 = min(q_lo * TG_ASF_NW + wv, A.nel[0] - 1);
    const int64_t node = (int64_t)(e0 * P + x0) + (int64_t)A.n[0] * ((e1_ * P + x1) + (int64_t)A.n[1] * (e2_ * P + x2)) - A.cp_node0;
#pragma unroll
    for (int c = 0; c < 4; c++) cpn[c] = A.cp[c][node];
  }
  for (int q = q_lo; q < q_hi; q++) {
    // (LOOP: what depends on the lane or on the line is re-derived per group from values the compiler cannot see through --
    //  hoisted out of the loop it would sit in the registers phase 1 needs)
    int ln = lane, e1 = e1_, e2 = e2_;
    if (LOOP) {
      asm volatile("" : "+v"(ln));
      asm volatile("" : "+s"(e1));
      asm volatile("" : "+s"(e2));
    }
    const int x0 = ln % P1, x1 = (ln / P1) % P1, x2 = ln / PP;
    const int e0u = q * TG_ASF_NW + wv;
    const bool live = e0u < A.nel[0];          // (the last group of a line may be short: its idle waves keep the barrier)
    const int e0 = live ? e0u : A.nel[0] - 1;
    const bool firstp = wv == 0 && q == q_lo, lastp = (wv == TG_ASF_NW - 1 && q == q_hi - 1) || e0u >= A.nel[0] - 1;
    const int gen = LOOP ? q % 3 : 0;
    double *FW = s_f[gen][wv];
    const double *CF = wv > 0 ? s_f[gen][wv - 1] : s_f[LOOP ? (q + 2) % 3 : 0][TG_ASF_NW - 1];
    const int par1 = e1 & 1, par2 = e2 & 1;
    // ---- phase 0 ----------------------------------------------------------------------------------------------
#pragma unroll
    for (int c = 0; c < 4; c++) W[c * 64 + ln] = cpn[c];
    tg_wave_sync();
    {
      double N[4], dN[4][3], G[GN];
      tg_asf_to_points<P1, 4>(W, TL, TD, ln, 0, x0, x1, x2, true, N, dN);
      if constexpr (FORM == 2)
        tg_asf_elast(N, dN, TW[x0] * TW[x1] * TW[x2], A.ei, A.ej, A.lam, A.mu, G);
      else
        tg_asf_metric(N, dN, TW[x0] * TW[x1] * TW[x2], G);
      tg_wave_sync();
#pragma unroll
      for (int j = 0; j < GN; j++) W[ln * GS + j] = G[j];
      tg_wave_sync();
    }
    double acc[NL];
#pragma unroll
    for (int i = 0; i < NL; i++) acc[i] = 0.0;
    // ---- phase 2 (defined here, run in two parts): the rows leave.  Two sweeps: the reads of entries that hold earlier
    // contributions (faces shared in directions 1, 2, seams between pieces: written by EARLIER launches, so they can be
    // requested before phase 1 -- PRE -- and arrive while the wave computes; otherwise layer by layer of a2, a round trip
    // each), then the stores.
    double old[PRE ? NL : PP];
    auto rows = [&](auto sweepc, auto a2c) {
      constexpr int sweep = decltype(sweepc)::value, a2 = decltype(a2c)::value;
      const int r2 = P * e2 + a2;
      if (r2 < A.za || r2 >= A.zb) return;                          // (wave-uniform)
      bool v2;
      int n2, o2;
      int64_t rps2;
      tg_asf_row1d<P>(a2, e2, A.nel[2], v2, n2, o2, rps2);
      const __amdgpu_buffer_rsrc_t plane =
          __builtin_amdgcn_make_buffer_rsrc(A.val + (T0 * T1 * rps2 - A.base), 0, TG_BUF_RANGE, 0x00020000);
      int xs0 = x0, xs1 = x1, xs2 = x2;
      asm volatile("" : "+v"(xs0), "+v"(xs1), "+v"(xs2));
      __builtin_amdgcn_sched_barrier(0);
      const bool add2 = v2 && par2 && xs2 == a2;
#pragma unroll
      for (int a1 = 0; a1 < P1; a1++) {
        bool v1;
        int n1, o1;
        int64_t rps1;
        tg_asf_row1d<P>(a1, e1, A.nel[1], v1, n1, o1, rps1);
        const bool add1 = v1 && par1 && xs1 == a1;
#pragma unroll
        for (int a0 = 0; a0 < P1; a0++) {
          bool v0;
          int n0, o0;
          int64_t rps0;
          tg_asf_row1d<P>(a0, e0, A.nel[0], v0, n0, o0, rps0);
          // direction 0: a seam between pieces is a colour (read by the odd piece), a face inside the piece is not
          const bool seam = (a0 == 0 && firstp) || (a0 == P && lastp);
          const bool addx = v0 && seam && pc && xs0 == a0;
          const bool add = add2 || add1 || addx;
          const bool edge = a2 == 0 || a2 == P || a1 == 0 || a1 == P || ((a0 == 0 || a0 == P) && seam);
          const bool inner_hi = a0 == P && !lastp;                // the face goes to the next element of the piece
          const bool inner_lo = a0 == 0 && !firstp;               // ... and arrives from the previous one
          const bool mine = live && !(inner_hi && xs0 == P);
          const unsigned off = ((unsigned)n2 * (uT0 * (unsigned)rps1 + (unsigned)n1 * (unsigned)rps0) +
                                (unsigned)(((xs2 + o2) * n1 + (xs1 + o1)) * n0 + (xs0 + o0))) * 8u;
          double &og = old[a0 + P1 * a1 + (PRE ? PP * a2 : 0)];
          if (sweep == 0) {
            og = edge ? tg_buf_load(plane, (add && mine) ? off : TG_BUF_OOB) : 0.0;
          } else {
            double v = acc[a0 + P1 * (a1 + P1 * a2)];
            if (inner_lo) {
              const double cf = CF[(a1 + P1 * a2) * 16 + xs1 + P1 * xs2];
              v += xs0 == 0 ? cf : 0.0;
            }
            if (edge) v += og;
            tg_buf_store(v, plane, mine ? off : TG_BUF_OOB);
          }
        }
      }
    };
    if constexpr (PRE) {
      tg_asf_each4([&](auto a2c) { rows(std::integral_constant<int, 0>{}, a2c); });
      __builtin_amdgcn_sched_barrier(0);
    }
    // ---- phase 1: lane = column b = (x0, x1, x2) -----------------------------------------------------------------
    {
      double l0[P1], d0[P1];
#pragma unroll
      for (int qq = 0; qq < P1; qq++) {
        l0[qq] = TL[x0 * P1 + qq];
        d0[qq] = TD[x0 * P1 + qq];
      }
#pragma unroll
      for (int q2 = 0; q2 < P1; q2++) {
        const double l2q = TL[x2 * P1 + q2], d2q = TD[x2 * P1 + q2];
        double Zl[PP], Zd[PP];
#pragma unroll
        for (int i = 0; i < PP; i++) Zl[i] = Zd[i] = 0.0;
#pragma unroll
        for (int q1 = 0; q1 < P1; q1++) {
          const double l1q = TL[x1 * P1 + q1], d1q = TD[x1 * P1 + q1];
          const double mll = l1q * l2q, mdl = d1q * l2q, mld = l1q * d2q;
          double Y0[P1], Y1[P1], Y2[P1];
#pragma unroll
          for (int a = 0; a < P1; a++) Y0[a] = Y1[a] = Y2[a] = 0.0;
#pragma unroll
          for (int q0 = 0; q0 < P1; q0++) {
            const double *Gq = W + (q0 + P1 * (q1 + P1 * q2)) * GS;
            const double f0 = d0[q0] * mll, f1 = l0[q0] * mdl, f2 = l0[q0] * mld;
            double X0, X1, X2;
            tg_asf_flux<FORM>(Gq, f0, f1, f2, X0, X1, X2);
#pragma unroll
            for (int a = 0; a < P1; a++) {
              Y0[a] = fma(UD[a * P1 + q0], X0, Y0[a]);
              Y1[a] = fma(UL[a * P1 + q0], X1, Y1[a]);
              Y2[a] = fma(UL[a * P1 + q0], X2, Y2[a]);
            }
          }
#pragma unroll
          for (int a1 = 0; a1 < P1; a1++)
#pragma unroll
            for (int a0 = 0; a0 < P1; a0++) {
              Zl[a0 + P1 * a1] = fma(UD[a1 * P1 + q1], Y1[a0], fma(UL[a1 * P1 + q1], Y0[a0], Zl[a0 + P1 * a1]));
              Zd[a0 + P1 * a1] = fma(UL[a1 * P1 + q1], Y2[a0], Zd[a0 + P1 * a1]);
            }
        }
#pragma unroll
        for (int a2 = 0; a2 < P1; a2++)
#pragma unroll
          for (int i = 0; i < PP; i++)
            acc[i + PP * a2] = fma(UD[a2 * P1 + q2], Zd[i], fma(UL[a2 * P1 + q2], Zl[i], acc[i + PP * a2]));
      }
    }
    // the block of the face this element shares with the next one: to the neighbour wave (straight-line: every lane
    // stores, the lanes off the face into a spare slot -- a branch here splits the block phase 1 is scheduled in and costs
    // 160 registers)
    {
      const int fo = x0 == P ? x1 + P1 * x2 : TG_ASF_FACE + ln;
#pragma unroll
      for (int i = 0; i < PP; i++) FW[(x0 == P ? i * 16 : 0) + fo] = acc[P + P1 * i];
    }
    if (LOOP) {       // the next group's nodal values travel while the rows of this one leave
      const int en = min(min(q + 1, q_hi - 1) * TG_ASF_NW + wv, A.nel[0] - 1);
      const int64_t node = (int64_t)(en * P + x0) + (int64_t)A.n[0] * ((e1 * P + x1) + (int64_t)A.n[1] * (e2 * P + x2)) - A.cp_node0;
#pragma unroll
      for (int c = 0; c < 4; c++) cpn[c] = A.cp[c][node];
    }
    __syncthreads();
    if constexpr (PRE) {
      tg_asf_each4([&](auto a2c) { rows(std::integral_constant<int, 1>{}, a2c); });
    } else {
      tg_asf_each4([&](auto a2c) {
        rows(std::integral_constant<int, 0>{}, a2c);
        rows(std::integral_constant<int, 1>{}, a2c);
      });
    }
  }
}

// L(v) = int f_h v: lane = quadrature point computes w sqrt(det g) f_h, three 1-D contractions back to the nodes (through
// LDS), each node's value added to the vector (zeroed before; colour by colour, elements of a launch share no node)
template <int P1, int EPW>
__global__ void __launch_bounds__(64 * TG_ASF_NW) k_asf3_load(tg_asf_args A) {
  constexpr int P = P1 - 1, NL = P1 * P1 * P1, LPE = 64 / EPW, PP = P1 * P1;
  __shared__ __attribute__((aligned(16))) double s_tab[2 * PP + P1];
  __shared__ __attribute__((aligned(16))) double s_w[TG_ASF_NW][TG_ASF_AREA(5)];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  for (int s = tid; s < 2 * PP + P1; s += 64 * TG_ASF_NW) s_tab[s] = A.tab[s];
  __syncthreads();
  const int64_t grp = (int64_t)blockIdx.x * TG_ASF_NW + wv;
  if (grp >= A.ngroups) return;
  const double *TL = s_tab, *TD = s_tab + PP, *TW = s_tab + 2 * PP;
  double *W = s_w[wv];
  const int es = lane / LPE, li = lane - es * LPE, eb = es * LPE;
  const bool active = li < NL;
  const int x0 = li % P1, x1 = (li / P1) % P1, x2 = (li / PP) % P1;
  const int gx = (int)(grp % A.ngy);
  const int64_t rest = grp / A.ngy;
  const int i1 = (int)(rest % A.ncol[1]), i2 = (int)(rest / A.ncol[1]);
  const int i0 = gx * EPW + es;
  const bool evalid = i0 < A.ncol[0];
  const int e0 = A.efirst[0] + 2 * (evalid ? i0 : A.ncol[0] - 1), e1 = A.efirst[1] + 2 * i1, e2 = A.efirst[2] + 2 * i2;
  const int64_t node = (int64_t)(e0 * P + x0) + (int64_t)A.n[0] * ((e1 * P + x1) + (int64_t)A.n[1] * (e2 * P + x2));
  if (active) {
#pragma unroll
    for (int c = 0; c < 4; c++) W[c * 64 + lane] = A.cp[c][node - A.cp_node0];
    W[4 * 64 + lane] = A.fnod[node - A.cp_node0];
  }
  tg_wave_sync();
  double N[5], dN[5][3], G[7] = {0, 0, 0, 0, 0, 0, 0};
  tg_asf_to_points<P1, 5>(W, TL, TD, lane, eb, x0, x1, x2, active, N, dN);
  if (active) tg_asf_metric(N, dN, TW[x0] * TW[x1] * TW[x2], G);
  tg_wave_sync();
  // s(q) = w sqrt(det g) f_h(q) back to the nodes: lane (a0, q1, q2), then (a0, a1, q2), then (a0, a1, a2)
  double *S0 = W, *S1 = W + 64, *S2 = W + 128;
  if (active) S0[lane] = G[6] * N[4];
  tg_wave_sync();
  if (active) {
    double v = 0.0;
#pragma unroll
    for (int q = 0; q < P1; q++) v = fma(TL[x0 * P1 + q], S0[eb + q + P1 * (x1 + P1 * x2)], v);
    S1[lane] = v;
  }
  tg_wave_sync();
  if (active) {
    double v = 0.0;
#pragma unroll
    for (int q = 0; q < P1; q++) v = fma(TL[x1 * P1 + q], S1[eb + x0 + P1 * (q + P1 * x2)], v);
    S2[lane] = v;
  }
  tg_wave_sync();
  if (active && evalid) {
    double v = 0.0;
#pragma unroll
    for (int q = 0; q < P1; q++) v = fma(TL[x2 * P1 + q], S2[eb + x0 + P1 * (x1 + P1 * q)], v);
    const int r2 = P * e2 + x2;
    if (r2 >= A.za && r2 < A.zb) A.val[node - A.row0] += v;
  }
}

static inline int64_t tg_rps_host(int p, int a) { return (int64_t)(p + 1) * a + (a > 0 ? (int64_t)p * ((a - 1) / p) : 0); }

template <int P1, int EPW>
static void tg_asf_launch(int form, const tg_asf_args &A, unsigned nblk, bool walk, bool line) {
  if constexpr (P1 == 4) {
    const bool pre = !(getenv("TIGAR_ASM_PRE") && atoi(getenv("TIGAR_ASM_PRE")) == 0);
    const bool loop = A.chunk > 1;             // (pieces of several groups: the workgroup loops)
    const dim3 g(nblk), b(64 * TG_ASF_NW);
    if (line && form == 3) {
      if (loop)
        hipLaunchKernelGGL((k_asf3_quad<P1, 2, 1, 1>), g, b, 0, g_tg.stream, A);
      else if (pre)
        hipLaunchKernelGGL((k_asf3_quad<P1, 2, 1, 0>), g, b, 0, g_tg.stream, A);
      else
        hipLaunchKernelGGL((k_asf3_quad<P1, 2, 0, 0>), g, b, 0, g_tg.stream, A);
      return;
    }
    if (line && form == 1) {
      if (loop)
        hipLaunchKernelGGL((k_asf3_quad<P1, 1, 1, 1>), g, b, 0, g_tg.stream, A);
      else if (pre)
        hipLaunchKernelGGL((k_asf3_quad<P1, 1, 1, 0>), g, b, 0, g_tg.stream, A);
      else
        hipLaunchKernelGGL((k_asf3_quad<P1, 1, 0, 0>), g, b, 0, g_tg.stream, A);
      return;
    }
  }
  if (!walk && form == 3)
    hipLaunchKernelGGL((k_asf3_elem<P1, EPW, 2>), dim3(nblk), dim3(64 * TG_ASF_NW), 0, g_tg.stream, A);
  else if (form == 3)
    hipLaunchKernelGGL((k_asf3<P1, EPW, 2>), dim3(nblk), dim3(64 * TG_ASF_NW), 0, g_tg.stream, A);
  else if (!walk && form == 1)
    hipLaunchKernelGGL((k_asf3_elem<P1, EPW, 1>), dim3(nblk), dim3(64 * TG_ASF_NW), 0, g_tg.stream, A);
  else if (!walk && form == 0)
    hipLaunchKernelGGL((k_asf3_elem<P1, EPW, 0>), dim3(nblk), dim3(64 * TG_ASF_NW), 0, g_tg.stream, A);
  else if (form == 1)
    hipLaunchKernelGGL((k_asf3<P1, EPW, 1>), dim3(nblk), dim3(64 * TG_ASF_NW), 0, g_tg.stream, A);
  else if (form == 0)
    hipLaunchKernelGGL((k_asf3<P1, EPW, 0>), dim3(nblk), dim3(64 * TG_ASF_NW), 0, g_tg.stream, A);
  else
    hipLaunchKernelGGL((k_asf3_load<P1, EPW>), dim3(nblk), dim3(64 * TG_ASF_NW), 0, g_tg.stream, A);
}

// reference-element tables l[a][q] | dl[a][q] | w[q]: equispaced Lagrange nodes a/p at the Gauss points
static void tg_asm_tables(int p, int nq, std::vector<double> &tab) {
  const int p1 = p + 1;
  std::vector<double> gx, gw;
  tg_gauss01(nq, gx, gw);
  tab.assign(3 * (size_t)p1 * nq + nq, 0.0);
  for (int a = 0; a < p1; a++)
    for (int q = 0; q < nq; q++) {
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
      // second derivative: sum over ordered pairs (m, r) of the product without both factors
      double d2l = 0.0;
      for (int m = 0; m < p1; m++) {
        if (m == a) continue;
        for (int r = 0; r < p1; r++) {
          if (r == a || r == m) continue;
          double term = 1.0 / (((double)a / p - (double)m / p) * ((double)a / p - (double)r / p));
          for (int u = 0; u < p1; u++)
            if (u != a && u != m && u != r) term *= (t - (double)u / p) / ((double)a / p - (double)u / p);
          d2l += term;
        }
      }
      tab[(size_t)a * nq + q] = l;
      tab[(size_t)p1 * nq + (size_t)a * nq + q] = dl;
      tab[2 * (size_t)p1 * nq + nq + (size_t)a * nq + q] = d2l;
    }
  for (int q = 0; q < nq; q++) tab[2 * (size_t)p1 * nq + q] = gw[q];
}

// Device copies of the small per-call tables (element vertices, reference-element tables) of the last patch description: the
// z-slab pipeline calls once per sub-slab with the same patch, and the uploads with the wait that keeps the host arrays
// alive stood between the kernels of consecutive sub-slabs.
struct tg_asm_cache_t {
  int d = 0, p = 0, nq = 0, nverts[3] = {0, 0, 0};
  std::vector<double> hverts[3];
  double *verts[3] = {nullptr, nullptr, nullptr};
  double *tab = nullptr;
};
static tg_asm_cache_t g_asm_cache;

void tg_asm_cache_clear(void) {
  if (g_tg.ready) hipStreamSynchronize(g_tg.stream);
  for (int k = 0; k < 3; k++) {
    tg_dfree(g_asm_cache.verts[k]);
    g_asm_cache.verts[k] = nullptr;
  }
  tg_dfree(g_asm_cache.tab);
  g_asm_cache = tg_asm_cache_t();
}

static int tg_asm_cache_get(const tg_patch_t *pt) {
  tg_asm_cache_t &C = g_asm_cache;
  bool same = C.d == pt->d && C.p == pt->p && C.nq == pt->nq && C.tab;
  for (int k = 0; k < pt->d && same; k++)
    same = C.nverts[k] == pt->nverts[k] && memcmp(C.hverts[k].data(), pt->verts[k], sizeof(double) * pt->nverts[k]) == 0;
  if (same) return 0;
  tg_asm_cache_clear();
  C.d = pt->d;
  C.p = pt->p;
  C.nq = pt->nq;
  for (int k = 0; k < pt->d; k++) {
    C.nverts[k] = pt->nverts[k];
    C.hverts[k].assign(pt->verts[k], pt->verts[k] + pt->nverts[k]);
    TG_TRY(tg_dmalloc(&C.verts[k], pt->nverts[k]));
    TG_CHECK_HIP(hipMemcpyAsync(C.verts[k], C.hverts[k].data(), pt->nverts[k] * sizeof(double), hipMemcpyHostToDevice, g_tg.stream));
  }
  std::vector<double> tab;
  tg_asm_tables(pt->p, pt->nq, tab);
  TG_TRY(tg_dmalloc(&C.tab, (int64_t)tab.size()));
  TG_CHECK_HIP(hipMemcpyAsync(C.tab, tab.data(), tab.size() * sizeof(double), hipMemcpyHostToDevice, g_tg.stream));
  TG_CHECK_HIP(hipStreamSynchronize(g_tg.stream));      // (the host copies of the tables go out of scope)
  return 0;
}

// rows [row0, row1) of the matrix / vector -- whole node planes of the LAST direction (any range when d == 1); the control
// functions (and fnod) hold the nodes [cp_node0, cp_node0 + n), which must cover every element that touches the rows
struct tg_elast_block {
  int i, j;
  double lam, mu;
};

static int tg_assemble_common(const tg_patch_t *pt, int form, int64_t row0, int64_t row1, int64_t cp_node0, tg_csr_t *mout,
                              tg_vec_t fnod, tg_vec_t bout, const tg_elast_block *eb = nullptr) {
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
  if (form == 4) TG_REQUIRE(pt->nsd == pt->d, "the biharmonic form needs as many physical as parametric directions");
  if (form == 3) {
    TG_REQUIRE(eb && pt->nsd == pt->d && eb->i >= 0 && eb->i < pt->d && eb->j >= 0 && eb->j < pt->d,
               "elasticity block (i, j): as many physical as parametric directions (nsd == d = %d), 0 <= i, j < d", pt->d);
    A.ei = eb->i;
    A.ej = eb->j;
    A.lam = eb->lam;
    A.mu = eb->mu;
  }
  int64_t nnodes = 1, plane = 1;
  for (int k = 0; k < 3; k++) {
    A.nel[k] = 1;
    A.n[k] = 1;
  }
  for (int k = 0; k < d; k++) {
    TG_REQUIRE(pt->nverts[k] >= 2 && pt->verts[k], "direction %d needs at least one element", k);
    A.nel[k] = pt->nverts[k] - 1;
    A.n[k] = A.nel[k] * p + 1;
    nnodes *= A.n[k];
    if (k < d - 1) plane *= A.n[k];
  }
  if (row0 < 0 && row1 < 0) {
    row0 = 0;
    row1 = nnodes;
  }
  TG_REQUIRE(row0 >= 0 && row1 >= row0 && row1 <= nnodes && row0 % plane == 0 && row1 % plane == 0,
             "row range [%lld, %lld): whole node planes of the last direction (%lld nodes each) of the %lld FE nodes",
             (long long)row0, (long long)row1, (long long)plane, (long long)nnodes);
  const int za = (int)(row0 / plane), zb = (int)(row1 / plane);
  // element layers of the last direction that touch the node planes [za, zb), and the nodes they need
  const int nelL = A.nel[d - 1];
  int ez0 = 0, ez1 = 0;
  if (zb > za) {
    ez0 = (za > 0 && za % p == 0) ? za / p - 1 : za / p;
    ez1 = std::min(nelL, (zb - 1) / p + 1);
    if (ez0 >= nelL) ez0 = nelL - 1;           // (the last node plane belongs to the last layer)
  }
  const int64_t need0 = (int64_t)ez0 * p * plane, need1 = (zb > za) ? ((int64_t)ez1 * p + 1) * plane : need0;
  for (int c = 0; c <= pt->nsd; c++) {
    TG_REQUIRE(pt->cp[c], "control function %d missing", c);
    TG_REQUIRE(cp_node0 >= 0 && cp_node0 <= need0 && cp_node0 + pt->cp[c]->n >= need1,
               "control function %d holds the FE nodes [%lld, %lld), the rows need [%lld, %lld)", c, (long long)cp_node0,
               (long long)(cp_node0 + pt->cp[c]->n), (long long)need0, (long long)need1);
    A.cp[c] = pt->cp[c]->d;
  }
  A.cp_node0 = cp_node0;
  A.row0 = row0;
  A.row1 = row1;
  TG_TRY(tg_asm_cache_get(pt));
  for (int k = 0; k < d; k++) A.verts[k] = g_asm_cache.verts[k];
  A.tab = g_asm_cache.tab;
  bool fast = d == 3 && pt->nsd == 3 && pt->nq == p1 && p <= 3 && form != 4 && !getenv("TIGAR_ASM_LEGACY");
  if (fast) {   // the walk kernel addresses the entries of one plane of rows in 32 bits
    const double t0 = (double)p1 * A.n[0] + (double)p * (A.nel[0] - 1), t1 = (double)p1 * A.n[1] + (double)p * (A.nel[1] - 1);
    if ((2.0 * p + 1.0) * t0 * t1 * 8.0 >= 4294967000.0) fast = false;      // (byte offsets inside one plane of rows)
  }

  tg_csr_s *m = nullptr;
  if (form != 2) {
    // pattern: Kronecker product of the 1-D element-coupling patterns (carries the pattern certificate of tg_kron_sum_csr)
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
    if (fast) tg_kron_pattern_only();     // (the sum-factorised kernels store every entry before anybody adds to it)
    TG_TRY(tg_kron_sum_csr(d, 1, dirs, row0, row1, &pat));
    m = pat;
    // (the pattern kernel wrote 0 * 0 * 0 into every value: the plain kernel adds into that; the sum-factorised kernel
    //  stores every entry before anybody adds to it)
    A.rowptr = m->rowptr;
    A.val = m->val;
  } else {
    TG_REQUIRE(fnod && cp_node0 + fnod->n >= need1 && bout && bout->n == row1 - row0,
               "load assembly needs nodal values on the nodes of the control functions and an output vector of %lld rows",
               (long long)(row1 - row0));
    A.fnod = fnod->d;
    A.bout = bout->d;
    TG_CHECK_HIP(hipMemsetAsync(bout->d, 0, (size_t)(row1 - row0) * sizeof(double), g_tg.stream));
  }
  const size_t lds = ((size_t)3 * p1 * pt->nq + pt->nq + (size_t)(pt->nsd + 1) * nloc + (size_t)nqt * 10 + nloc) * sizeof(double);
  if (!fast && lds > 64 * 1024) {
    if (m) tg_csr_destroy(m);
    tg_set_error("element data (%zu B) does not fit in LDS", lds);
    return 2;
  }
  tg_asf_args F;
  memset(&F, 0, sizeof(F));
  if (fast) {
    for (int k = 0; k < 3; k++) {
      F.nel[k] = A.nel[k];
      F.n[k] = A.n[k];
    }
    for (int c = 0; c < 4; c++) F.cp[c] = A.cp[c];
    F.fnod = A.fnod;
    F.cp_node0 = cp_node0;
    F.tab = A.tab;
    F.val = form == 2 ? A.bout : m->val;
    F.row0 = row0;
    F.za = za;
    F.zb = zb;
    const int64_t T0 = (int64_t)p1 * A.n[0] + (int64_t)p * (A.nel[0] - 1), T1 = (int64_t)p1 * A.n[1] + (int64_t)p * (A.nel[1] - 1);
    F.base = T0 * T1 * tg_rps_host(p, za);
    // pieces of a line of elements: long enough that the element a piece starts early with (it only feeds the carry)
    // costs little, short enough that a launch has waves for every SIMD
    int chunk = getenv("TIGAR_ASM_CHUNK") ? atoi(getenv("TIGAR_ASM_CHUNK")) : 16;
    if (chunk < 1) chunk = 1;
    F.chunk = chunk;
    F.nchunks = (A.nel[0] + chunk - 1) / chunk;
    F.ei = A.ei;
    F.ej = A.ej;
    F.lam = A.lam;
    F.mu = A.mu;
  }
  const int nt = (form == 2) ? 128 : 256;
  bool bad = false;
  const bool timeit = getenv("TIGAR_ASM_TIME") != nullptr;
  if (timeit) hipEventRecord(g_tg.ev0[0], g_tg.stream);
  const int epw = p == 3 ? 1 : (p == 2 ? 2 : 8);
  // one launch per colour (parity of the element index per direction), colours in ascending order; the walks of the
  // sum-factorised matrix kernel need no colouring in direction 0
  // the walk along direction 0 for every matrix form but the p = 3 stiffness matrix (see k_asf3_elem); TIGAR_ASM_WALK=0/1 forces
  bool walk = fast && form != 2 && !(p == 3 && (form == 1 || form == 3)) && p != 1;
  if (fast && form != 2 && getenv("TIGAR_ASM_WALK")) walk = atoi(getenv("TIGAR_ASM_WALK")) != 0;
  // p = 3 stiffness / elasticity: the four waves of a workgroup on four consecutive elements of a line (k_asf3_quad;
  // TIGAR_ASM_QUAD=0: independent elements); direction 0 of a colour then counts GROUPS
  bool line = fast && !walk && p == 3 && (form == 1 || form == 3);
  if (line && getenv("TIGAR_ASM_QUAD")) line = atoi(getenv("TIGAR_ASM_QUAD")) != 0;
  if (line) {           // pieces of `chunk` groups of four elements (a function of nothing but the environment: the seams
                        // fix the order of the sums, and row blocks must reproduce the whole matrix bit for bit)
    int chunk = getenv("TIGAR_ASM_QUAD_CHUNK") ? atoi(getenv("TIGAR_ASM_QUAD_CHUNK")) : 8;
    if (chunk < 1) chunk = 1;
    const int nq = (A.nel[0] + TG_ASF_NW - 1) / TG_ASF_NW;
    F.chunk = chunk;
    F.nchunks = (nq + chunk - 1) / chunk;
  }
  for (int c = 0; c < (1 << d) && !bad && zb > za; c++) {
    if (walk && (c & 1)) continue;
    int64_t nblk = 1;
    for (int k = 0; k < 3; k++) {
      const int par = k < d ? (c >> k) & 1 : 0;
      int lo = 0, hi = A.nel[k];
      if (k == d - 1) {
        lo = ez0;
        hi = ez1;
      }
      if (line && k == 0) hi = F.nchunks;
      const int first = lo + (((par - lo) % 2) + 2) % 2;       // first index >= lo with this parity
      A.efirst[k] = F.efirst[k] = k < d ? first : 0;
      A.ncol[k] = F.ncol[k] = k < d ? (hi > first ? (hi - first + 1) / 2 : 0) : 1;
      if (!(walk && k == 0)) nblk *= A.ncol[k];
    }
    if (nblk == 0) continue;
    if (fast) {
      if (walk) {
        F.ngy = (F.ncol[1] + epw - 1) / epw;
        F.ngroups = (int64_t)F.nchunks * F.ngy * F.ncol[2];
      } else {
        F.ngy = (F.ncol[0] + epw - 1) / epw;
        F.ngroups = (int64_t)F.ngy * F.ncol[1] * F.ncol[2];
      }
      const unsigned nb = line ? (unsigned)F.ngroups : (unsigned)((F.ngroups + TG_ASF_NW - 1) / TG_ASF_NW);   // (quad: a workgroup per group)
      if (p == 3)
        tg_asf_launch<4, 1>(form, F, nb, walk, line);
      else if (p == 2)
        tg_asf_launch<3, 2>(form, F, nb, walk, false);
      else
        tg_asf_launch<2, 8>(form, F, nb, walk, false);
    } else {
      TG_REQUIRE(nblk < (1ll << 31), "too many elements for one launch");
      hipLaunchKernelGGL(k_assemble_mapped, dim3((unsigned)nblk), dim3(nt), lds, g_tg.stream, A);
    }
    bad = hipGetLastError() != hipSuccess;
  }
  if (timeit) {
    hipEventRecord(g_tg.ev1[0], g_tg.stream);
    hipEventSynchronize(g_tg.ev1[0]);
    float ms = 0.f;
    hipEventElapsedTime(&ms, g_tg.ev0[0], g_tg.ev1[0]);
    fprintf(stderr, "[tg_assemble] form %d rows [%lld, %lld): element kernels %.3f ms (%s)\n", form, (long long)row0, (long long)row1,
            ms, fast ? "sum-factorised" : "plain");
  }
  if (bad) {
    if (m) tg_csr_destroy(m);
    tg_set_error("the assembly kernel failed to launch");
    return 1;
  }
  if (mout) *mout = m;
  return 0;
}

extern "C" int tg_assemble_mapped_matrix(const tg_patch_t *patch, int form, tg_csr_t *out) {
  TG_REQUIRE(out && (form == 0 || form == 1 || form == 4), "form: 0 = mass, 1 = laplace, 4 = biharmonic");
  return tg_assemble_common(patch, form, -1, -1, 0, out, nullptr, nullptr);
}

extern "C" int tg_assemble_mapped_load(const tg_patch_t *patch, tg_vec_t fnodal, tg_vec_t out) {
  return tg_assemble_common(patch, 2, -1, -1, 0, nullptr, fnodal, out);
}

extern "C" int tg_assemble_mapped_matrix_rows(const tg_patch_t *patch, int form, int64_t row0, int64_t row1, int64_t cp_node0,
                                              tg_csr_t *out) {
  TG_REQUIRE(out && (form == 0 || form == 1 || form == 4), "form: 0 = mass, 1 = laplace, 4 = biharmonic");
  return tg_assemble_common(patch, form, row0, row1, cp_node0, out, nullptr, nullptr);
}

extern "C" int tg_assemble_mapped_elasticity_rows(const tg_patch_t *patch, int fi, int fj, double lambda, double mu, int64_t row0,
                                                  int64_t row1, int64_t cp_node0, tg_csr_t *out) {
  TG_REQUIRE(out, "null output");
  const tg_elast_block eb = {fi, fj, lambda, mu};
  return tg_assemble_common(patch, 3, row0, row1, cp_node0, out, nullptr, nullptr, &eb);
}

extern "C" int tg_assemble_mapped_load_rows(const tg_patch_t *patch, tg_vec_t fnodal, int64_t row0, int64_t row1,
                                            int64_t cp_node0, tg_vec_t out) {
  return tg_assemble_common(patch, 2, row0, row1, cp_node0, nullptr, fnodal, out);
}
