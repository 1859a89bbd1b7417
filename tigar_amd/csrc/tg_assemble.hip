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
  el[0] = 2 * (int)(e % P.ncol[0]) + P.efirst[0];
  e /= P.ncol[0];
  if (d > 1) {
    el[1] = 2 * (int)(e % P.ncol[1]) + P.efirst[1];
    e /= P.ncol[1];
  }
  if (d > 2) el[2] = 2 * (int)e + P.efirst[2];
  double h[3] = {1.0, 1.0, 1.0};
  for (int k = 0; k < d; k++) h[k] = P.verts[k][el[k] + 1] - P.verts[k][el[k]];
  for (int s = tid; s < 2 * p1 * nq1 + nq1; s += nt) tl[s] = P.tab[s];
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
// with phi_a(q) = l[a0][q0] l[a1][q1] l[a2][q2]: one WAVE per element (p = 3; two elements per wave at p = 2, eight at
// p = 1), nq = p + 1 Gauss points per direction.
//   phase 0  lane = quadrature point: control functions and their xi_hat-gradients by three 1-D contractions through the
//            wave's LDS area (cross-lane), quotient rule, metric, w sqrt(det g) g^-1 -> LDS (6 + 1 doubles per point);
//   phase 1  lane = COLUMN b of the element matrix, the 64 rows in registers:  X_k(q) = sum_m G_km(q) d_m phi_b(q)
//            (G: LDS broadcast; phi_b: lane constants), then the three contractions with the WAVE-UNIFORM 1-D tables
//            (scalar registers) pencil by pencil -- Y[a0] over q0, Z[a0][a1] over q1, acc[a0][a1][a2] over q2 -- with the
//            derivative index merged as soon as two terms share their remaining tables: 2.9e3 fused multiply-adds per
//            lane and element (184e3 per element against 2.4e6 of the plain triple loop), no cross-lane traffic;
//   phase 2  row a of the element leaves as ONE store of the wave: the 64 columns are contiguous in the CSR row of an
//            element-interior node (512 B), runs of p + 1 otherwise; positions in closed form.
// Entries that several elements contribute to (both nodes on a shared face) are STORED by the contributor with even
// index in every shared direction and ADDED by the others; the launches go colour by colour (parities of the element
// index) in ascending order, so the storing element always comes first, no two elements of a launch touch one entry and
// the sum is formed in a fixed order: bit-reproducible, no memset, one pass over 83 % of the entries.
typedef const double __attribute__((address_space(4))) *tg_cdp4;
typedef double __attribute__((address_space(1))) *tg_gdp1;

struct tg_asf_args {
  int nel[3], n[3];
  const double *cp[4];       // the four homogeneous control functions on the nodes from cp_node0 on
  int64_t cp_node0;
  const double *tab;         // l[a][q] | dl[a][q] | w[q]   (p+1 points per direction)
  double *val;               // values of the row block
  int64_t base;              // position of the first entry of FE plane za in the whole matrix (subtracted)
  int za, zb;                // FE planes of the last direction whose rows are written
  int efirst[3], ncol[3];
  int ngx;                   // groups of EPW elements per line of direction 0
  int64_t ngroups;
};

__device__ __forceinline__ void tg_wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

#define TG_ASF_NW 4          // waves per workgroup (each works on its own elements)

template <int P1, int EPW, int FORM>
__global__ void __launch_bounds__(64 * TG_ASF_NW) k_asf3(tg_asf_args A) {
  constexpr int P = P1 - 1, NL = P1 * P1 * P1, LPE = 64 / EPW, PP = P1 * P1;
  static_assert(NL <= LPE, "an element needs a lane per local node");
  __shared__ __attribute__((aligned(16))) double s_tab[2 * PP + P1];
  __shared__ __attribute__((aligned(16))) double s_w[TG_ASF_NW][24 * 64];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
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
  const int gx = (int)(grp % A.ngx);
  const int64_t rest = grp / A.ngx;
  const int i1 = (int)(rest % A.ncol[1]), i2 = (int)(rest / A.ncol[1]);
  const int i0 = gx * EPW + es;
  const bool evalid = i0 < A.ncol[0];
  const int e0 = A.efirst[0] + 2 * (evalid ? i0 : A.ncol[0] - 1), e1 = A.efirst[1] + 2 * i1, e2 = A.efirst[2] + 2 * i2;

  // ---- phase 0 ----------------------------------------------------------------------------------------------
  if (active) {
    const int64_t node = (int64_t)(e0 * P + x0) + (int64_t)A.n[0] * ((e1 * P + x1) + (int64_t)A.n[1] * (e2 * P + x2)) - A.cp_node0;
#pragma unroll
    for (int c = 0; c < 4; c++) W[c * 64 + lane] = A.cp[c][node];
  }
  tg_wave_sync();
  if (active) {        // lane (q0, a1, a2): contraction over a0
#pragma unroll
    for (int c = 0; c < 4; c++) {
      double vl = 0.0, vd = 0.0;
#pragma unroll
      for (int a = 0; a < P1; a++) {
        const double v = W[c * 64 + eb + a + P1 * (x1 + P1 * x2)];
        vl = fma(TL[a * P1 + x0], v, vl);
        vd = fma(TD[a * P1 + x0], v, vd);
      }
      W[256 + c * 64 + lane] = vl;
      W[256 + (4 + c) * 64 + lane] = vd;
    }
  }
  tg_wave_sync();
  if (active) {        // lane (q0, q1, a2): contraction over a1
#pragma unroll
    for (int c = 0; c < 4; c++) {
      double ll = 0.0, dl = 0.0, ld = 0.0;
#pragma unroll
      for (int a = 0; a < P1; a++) {
        const double vl = W[256 + c * 64 + eb + x0 + P1 * (a + P1 * x2)];
        const double vd = W[256 + (4 + c) * 64 + eb + x0 + P1 * (a + P1 * x2)];
        const double tl = TL[a * P1 + x1], td = TD[a * P1 + x1];
        ll = fma(tl, vl, ll);
        dl = fma(tl, vd, dl);
        ld = fma(td, vl, ld);
      }
      W[768 + c * 64 + lane] = ll;
      W[768 + (4 + c) * 64 + lane] = dl;
      W[768 + (8 + c) * 64 + lane] = ld;
    }
  }
  tg_wave_sync();
  double G[7] = {0, 0, 0, 0, 0, 0, 0};
  if (active) {        // lane (q0, q1, q2): contraction over a2, then the metric
    double N[4], dN[4][3];
#pragma unroll
    for (int c = 0; c < 4; c++) {
      double v = 0.0, d0 = 0.0, d1 = 0.0, d2 = 0.0;
#pragma unroll
      for (int a = 0; a < P1; a++) {
        const double ll = W[768 + c * 64 + eb + x0 + P1 * (x1 + P1 * a)];
        const double dl = W[768 + (4 + c) * 64 + eb + x0 + P1 * (x1 + P1 * a)];
        const double ld = W[768 + (8 + c) * 64 + eb + x0 + P1 * (x1 + P1 * a)];
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
    const double s = TW[x0] * TW[x1] * TW[x2] * sqrt(fabs(det));
    G[0] = s * gi[0];
    G[1] = s * gi[1];
    G[2] = s * gi[2];
    G[3] = s * gi[4];
    G[4] = s * gi[5];
    G[5] = s * gi[8];
    G[6] = s;
  }
  tg_wave_sync();      // (the area of the nodal values and of the first contraction is free: nobody reads it any more)
  if (active) {
#pragma unroll
    for (int j = 0; j < 7; j++) W[(eb + li) * 8 + j] = G[j];
  }
  tg_wave_sync();

  // ---- phase 1: lane = column b = (x0, x1, x2) -----------------------------------------------------------------
  double acc[NL];
#pragma unroll
  for (int i = 0; i < NL; i++) acc[i] = 0.0;
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
      if (FORM == 1) {
        const double mll = l1q * l2q, mdl = d1q * l2q, mld = l1q * d2q;
        double Y0[P1], Y1[P1], Y2[P1];
#pragma unroll
        for (int a = 0; a < P1; a++) Y0[a] = Y1[a] = Y2[a] = 0.0;
#pragma unroll
        for (int q0 = 0; q0 < P1; q0++) {
          const double *Gq = W + (eb + q0 + P1 * (q1 + P1 * q2)) * 8;
          const double f0 = d0[q0] * mll, f1 = l0[q0] * mdl, f2 = l0[q0] * mld;
          const double X0 = fma(Gq[2], f2, fma(Gq[1], f1, Gq[0] * f0));
          const double X1 = fma(Gq[4], f2, fma(Gq[3], f1, Gq[1] * f0));
          const double X2 = fma(Gq[5], f2, fma(Gq[4], f1, Gq[2] * f0));
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
        if (FORM == 1)
          acc[i + PP * a2] = fma(UD[a2 * P1 + q2], Zd[i], fma(UL[a2 * P1 + q2], Zl[i], acc[i + PP * a2]));
        else
          acc[i + PP * a2] = fma(UL[a2 * P1 + q2], Zl[i], acc[i + PP * a2]);
      }
  }

  // ---- phase 2: the rows of the element, one store of the wave per row ---------------------------------------------
  // 1-D row data of node r = P e + a: n = row length, o = position of the element's first column in the row,
  // rps = entries of the 1-D rows before r; T = entries of all 1-D rows
  const int64_t T0 = (int64_t)P1 * A.n[0] + (int64_t)P * (A.nel[0] - 1), T1 = (int64_t)P1 * A.n[1] + (int64_t)P * (A.nel[1] - 1);
  const int par0 = e0 & 1, par1 = e1 & 1, par2 = e2 & 1;
#pragma unroll
  for (int a2 = 0; a2 < P1; a2++) {
    const int r2 = P * e2 + a2;
    if (r2 < A.za || r2 >= A.zb) continue;                        // (wave-uniform)
    const bool v2 = (a2 == 0 && e2 > 0) || (a2 == P && e2 < A.nel[2] - 1);
    const int n2 = v2 ? 2 * P + 1 : P1, o2 = (a2 == 0 && e2 > 0) ? P : 0;
    const int64_t rps2 = (int64_t)P1 * r2 + (a2 > 0 ? (int64_t)P * e2 : (e2 > 0 ? (int64_t)P * (e2 - 1) : 0));
    const bool add2 = v2 && par2 && x2 == a2;
#pragma unroll
    for (int a1 = 0; a1 < P1; a1++) {
      const int r1 = P * e1 + a1;
      const bool v1 = (a1 == 0 && e1 > 0) || (a1 == P && e1 < A.nel[1] - 1);
      const int n1 = v1 ? 2 * P + 1 : P1, o1 = (a1 == 0 && e1 > 0) ? P : 0;
      const int64_t rps1 = (int64_t)P1 * r1 + (a1 > 0 ? (int64_t)P * e1 : (e1 > 0 ? (int64_t)P * (e1 - 1) : 0));
      const bool add1 = v1 && par1 && x1 == a1;
#pragma unroll
      for (int a0 = 0; a0 < P1; a0++) {
        const int r0 = P * e0 + a0;                               // (per lane when EPW > 1)
        const bool v0 = (a0 == 0 && e0 > 0) || (a0 == P && e0 < A.nel[0] - 1);
        const int n0 = v0 ? 2 * P + 1 : P1, o0 = (a0 == 0 && e0 > 0) ? P : 0;
        const int64_t rps0 = (int64_t)P1 * r0 + (a0 > 0 ? (int64_t)P * e0 : (e0 > 0 ? (int64_t)P * (e0 - 1) : 0));
        const bool add0 = v0 && par0 && x0 == a0;
        const int64_t R = T0 * T1 * rps2 - A.base + (int64_t)n2 * (T0 * rps1 + (int64_t)n1 * rps0);
        const int64_t pos = (int64_t)((x2 + o2) * n1 + (x1 + o1)) * n0 + (x0 + o0);
        const double v = acc[a0 + P1 * (a1 + P1 * a2)];
        if (active && evalid) {
          double *dst = A.val + R + pos;
          if (add0 || add1 || add2)
            __builtin_amdgcn_global_atomic_fadd_f64((tg_gdp1)dst, v);   // (one contributor per launch: an ordered sum)
          else
            *dst = v;
        }
      }
    }
  }
}

static inline int64_t tg_rps_host(int p, int a) { return (int64_t)(p + 1) * a + (a > 0 ? (int64_t)p * ((a - 1) / p) : 0); }

template <int P1, int EPW>
static void tg_asf_launch(int form, const tg_asf_args &A, unsigned nblk) {
  if (form == 1)
    hipLaunchKernelGGL((k_asf3<P1, EPW, 1>), dim3(nblk), dim3(64 * TG_ASF_NW), 0, g_tg.stream, A);
  else
    hipLaunchKernelGGL((k_asf3<P1, EPW, 0>), dim3(nblk), dim3(64 * TG_ASF_NW), 0, g_tg.stream, A);
}

// reference-element tables l[a][q] | dl[a][q] | w[q]: equispaced Lagrange nodes a/p at the Gauss points
static void tg_asm_tables(int p, int nq, std::vector<double> &tab) {
  const int p1 = p + 1;
  std::vector<double> gx, gw;
  tg_gauss01(nq, gx, gw);
  tab.assign(2 * (size_t)p1 * nq + nq, 0.0);
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
      tab[(size_t)a * nq + q] = l;
      tab[(size_t)p1 * nq + (size_t)a * nq + q] = dl;
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
static int tg_assemble_common(const tg_patch_t *pt, int form, int64_t row0, int64_t row1, int64_t cp_node0, tg_csr_t *mout,
                              tg_vec_t fnod, tg_vec_t bout) {
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
  const bool fast = form != 2 && d == 3 && pt->nsd == 3 && pt->nq == p1 && p <= 3 && !getenv("TIGAR_ASM_LEGACY");

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
  const size_t lds = ((size_t)2 * p1 * pt->nq + pt->nq + (size_t)(pt->nsd + 1) * nloc + (size_t)nqt * 10 + nloc) * sizeof(double);
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
    F.cp_node0 = cp_node0;
    F.tab = A.tab;
    F.val = m->val;
    F.za = za;
    F.zb = zb;
    const int64_t T0 = (int64_t)p1 * A.n[0] + (int64_t)p * (A.nel[0] - 1), T1 = (int64_t)p1 * A.n[1] + (int64_t)p * (A.nel[1] - 1);
    F.base = T0 * T1 * tg_rps_host(p, za);
  }
  const int nt = (form == 2) ? 128 : 256;
  bool bad = false;
  // one launch per colour (parity of the element index per direction), colours in ascending order
  for (int c = 0; c < (1 << d) && !bad && zb > za; c++) {
    int64_t nblk = 1;
    for (int k = 0; k < 3; k++) {
      const int par = k < d ? (c >> k) & 1 : 0;
      int lo = 0, hi = A.nel[k];
      if (k == d - 1) {
        lo = ez0;
        hi = ez1;
      }
      const int first = lo + (((par - lo) % 2) + 2) % 2;       // first index >= lo with this parity
      A.efirst[k] = F.efirst[k] = k < d ? first : 0;
      A.ncol[k] = F.ncol[k] = k < d ? (hi > first ? (hi - first + 1) / 2 : 0) : 1;
      nblk *= A.ncol[k];
    }
    if (nblk == 0) continue;
    if (fast) {
      const int epw = p == 3 ? 1 : (p == 2 ? 2 : 8);
      F.ngx = (F.ncol[0] + epw - 1) / epw;
      F.ngroups = (int64_t)F.ngx * F.ncol[1] * F.ncol[2];
      const unsigned nb = (unsigned)((F.ngroups + TG_ASF_NW - 1) / TG_ASF_NW);
      if (p == 3)
        tg_asf_launch<4, 1>(form, F, nb);
      else if (p == 2)
        tg_asf_launch<3, 2>(form, F, nb);
      else
        tg_asf_launch<2, 8>(form, F, nb);
    } else {
      TG_REQUIRE(nblk < (1ll << 31), "too many elements for one launch");
      hipLaunchKernelGGL(k_assemble_mapped, dim3((unsigned)nblk), dim3(nt), lds, g_tg.stream, A);
    }
    bad = hipGetLastError() != hipSuccess;
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
  TG_REQUIRE(out && (form == 0 || form == 1), "form: 0 = mass, 1 = laplace");
  return tg_assemble_common(patch, form, -1, -1, 0, out, nullptr, nullptr);
}

extern "C" int tg_assemble_mapped_load(const tg_patch_t *patch, tg_vec_t fnodal, tg_vec_t out) {
  return tg_assemble_common(patch, 2, -1, -1, 0, nullptr, fnodal, out);
}

extern "C" int tg_assemble_mapped_matrix_rows(const tg_patch_t *patch, int form, int64_t row0, int64_t row1, int64_t cp_node0,
                                              tg_csr_t *out) {
  TG_REQUIRE(out && (form == 0 || form == 1), "form: 0 = mass, 1 = laplace");
  return tg_assemble_common(patch, form, row0, row1, cp_node0, out, nullptr, nullptr);
}

extern "C" int tg_assemble_mapped_load_rows(const tg_patch_t *patch, tg_vec_t fnodal, int64_t row0, int64_t row1,
                                            int64_t cp_node0, tg_vec_t out) {
  return tg_assemble_common(patch, 2, row0, row1, cp_node0, nullptr, fnodal, out);
}
