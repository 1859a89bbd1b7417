// extractMatrix for tensor-product patches whose FE matrix carries the element-coupling pattern of the Q_p
// node grid ("tensor-pattern" fast path of tIGAr/common.py:1176-1204, K = M^T A M with M = M_z (x) M_y (x) M_x).
//
// The general kernels (tg_ptap.hip, tg_ptap_box.hip) decode every column index of A and scatter into LDS
// boxes; each entry of A passes through that instruction stream p+1 times.  When A's pattern is the
// Kronecker product of the 1-D element-coupling patterns -- what dolfin assembles for ANY form on the Q_p
// space; verified here entry by entry, never assumed -- row (a, r1, r2) of A is a dense little tensor
// [c2][c1][c0] in CSR order, the position of every entry is known in closed form, and
//
//     K = P_z^T ( P_y^T ( P_x^T A P_x ) P_y ) P_z
//
// becomes three passes of one "line walk" with NO column decode, NO LDS, NO atomics and NO cross-lane
// traffic: a lane owns one pair of passive column coordinates, walks the contracted direction row by row,
// reads its n_t consecutive values of the row block (the lanes of a line cover the block contiguously),
// applies the column-side contraction with the element's (p+1)x(p+1) local extraction weights (wave-uniform
// scalars) and adds the result to a ring of p+1 live output rows held in registers; an output row is stored
// when the walk leaves its support.  Every entry of A is loaded once; intermediates are dense blocks without
// column indices; K leaves the last pass directly in CSR order at closed-form positions, with
// MatZeroRowsColumns fused.  Accumulation order is fixed, so K is bit-reproducible (SURVEY.md section 7,
// hard part 4).
//
// Everything in this header is plain C++ over (block, lane) indices: the HIP kernels in tg_ptap_tensor.hip
// call it with blockIdx / threadIdx, and tests/emu/tensor_emu.cpp runs the very same code lane by lane on the
// host to pin the index arithmetic in the CPU suite (test infrastructure; the product
// library contains only the device build).
//
// Structure assumed of direction t (checked on the host before this path is taken): CG Lagrange degree P on
// nel elements (nfe = P*nel+1 nodes, vertex nodes shared), open knot vector with simple interior knots, so
// that every node of element e (j = 0..P, node P*e+j) has its non-zero spline functions among dofs e..e+P
// ("local weights" wl[e][j][q], q = dof - e), the vertex nodes additionally none at dof e (j = 0, e > 0).
#pragma once
#include <stdint.h>

#include <math.h>

#ifdef __HIPCC__
#define TT_DEV __device__ __forceinline__
#define TT_MEM __device__ __forceinline__
// stores of the intermediates B1 / B2 (written once, read once by the next pass, far too large for the caches).  Measured at
// cfg3 with -DTT_NT_STORES (nontemporal stores): the PtAP stage 0.33-0.37 s against 0.166-0.173 s, same box, alternating runs
// -- the lanes of a wave write 392-byte segments (49 columns x 8 B) that the L2 merges into full lines; bypassing it writes
// partial lines.  Plain stores it is.
#ifdef TT_NT_STORES
#define TT_ST(p, v) __builtin_nontemporal_store((v), (p))
#else
#define TT_ST(p, v) (*(p) = (v))
#endif
#else
#define TT_DEV static inline
#define TT_MEM inline
#define TT_ST(p, v) (*(p) = (v))
#endif

// Small read-only tables (local weights, prefix sums) are read through the constant address space: with a
// wave-uniform index the loads go through the scalar unit (s_load) and the values live in SGPRs -- as plain global
// loads the compiler keeps a copy per lane (they could alias the stores of the kernel)
#ifdef __HIPCC__
typedef const double __attribute__((address_space(4))) *tt_cdp;
typedef const int32_t __attribute__((address_space(4))) *tt_cip;
#define TT_CD(p) ((tt_cdp)(p))
#define TT_CI(p) ((tt_cip)(p))
typedef const double __attribute__((address_space(1))) *tt_gdp;   // pointer known to point to global memory
#define TT_GD(p) ((tt_gdp)(p))
#else
typedef const double *tt_cdp;
typedef const int32_t *tt_cip;
typedef const double *tt_gdp;
#define TT_CD(p) (p)
#define TT_CI(p) (p)
#define TT_GD(p) (p)
#endif

struct tt_dir_t {
  int nel, nfe, ncp;
  const double *wl;      // [nel][P+1][P+1] local extraction weights (see above): the COLUMN side of the product
  // Row side of the product when it differs (block (f, g) of a space whose fields sit on different spline bases, e.g. the
  // components of a div-conforming B-spline, tIGAr/compatibleSplines.py:21-66: K_fg = M_f^T A_fg M_g): the weights of basis f
  // at the same nodes.  A basis of lower degree than the FE grid is PADDED to P + 1 functions per element (weights 0):
  // its rows / columns beyond the true function count come out as zeros and are left out by the last pass.  Null: wl.
  const double *wlr;
  const int32_t *rps;    // [nfe+1] exclusive prefix sums of the 1-D row lengths of the FE pattern
  const int32_t *kps;    // [ncp+1] exclusive prefix sums of the widths of K's 1-D rows (clipped band)
};

// 1-D element-coupling pattern of the CG degree-P grid: row a couples to columns [lo, lo+n)
template <int P>
TT_DEV int tt_rn(int a, int nfe) {
  return (a % P == 0 && a > 0 && a < nfe - 1) ? 2 * P + 1 : P + 1;
}
// position of row a in that pattern = the prefix sum of the row lengths, in closed form: (P+1) a + P * (interior vertices
// before a).  The walks use it instead of loading rps[a]: the addresses of a row's entries then do not depend on a scalar load.
template <int P>
TT_DEV int tt_rps(int a) {
  return (P + 1) * a + (a > 0 ? P * ((a - 1) / P) : 0);
}
template <int P>
TT_DEV int tt_rlo(int a, int nfe) {
  if (a % P == 0 && a > 0) return a - P;          // vertex between two elements, or the last node
  return (a / P) * P;                             // interior node of element a/P, or node 0
}

// ------------------------------------------------------------------------------------------------------
// The walk along the contracted direction.  IO supplies
//   template <int N> void load(int a, int clo, double *v)   the lane's N values of row a (columns clo..clo+N-1)
//   void emit(int i, const double *row)                     the finished output row i: row[m] <-> column dof i-P+m
// Elements e_begin..e_end-1 are walked; an output row i is emitted after element min(i, nel-1), i.e. complete
// rows are those with all of their elements [i-P, i] inside the walked range (the caller filters).
template <int P, class IO>
TT_DEV void tt_walk(const tt_dir_t &D, int e_begin, int e_end, IO &io) {
  constexpr int W = 2 * P + 1, Q = P + 1, NW = Q * Q;
  double acc[Q][W];
#pragma unroll
  for (int r = 0; r < Q; r++)
#pragma unroll
    for (int m = 0; m < W; m++) acc[r][m] = 0.0;

  tt_cdp wrow = TT_CD(D.wlr ? D.wlr : D.wl);
  if (e_begin == 0) {   // opening vertex: node 0 = node j = 0 of element 0
    tt_cdp we = TT_CD(D.wl);
    double v[Q], C[Q];
    io.template load<Q>(0, 0, v);
#pragma unroll
    for (int q = 0; q < Q; q++) {
      double s = 0.0;
#pragma unroll
      for (int j = 0; j < Q; j++) s = fma(v[j], we[j * Q + q], s);
      C[q] = s;
    }
#pragma unroll
    for (int r = 0; r < Q; r++) {
      const double wr = wrow[r];                  // node j = 0, dof 0 + r
#pragma unroll
      for (int q = 0; q < Q; q++) acc[r][q - r + P] = fma(wr, C[q], acc[r][q - r + P]);
    }
  }
  for (int e = e_begin; e < e_end; e++) {
    tt_cdp we = TT_CD(D.wl) + (int64_t)e * NW;
    tt_cdp wre = wrow + (int64_t)e * NW;
    const int a0 = P * e;
    // interior nodes of element e: columns = the element's P+1 nodes
#pragma unroll
    for (int j = 1; j < P; j++) {
      double v[Q], C[Q];
      io.template load<Q>(a0 + j, a0, v);
#pragma unroll
      for (int q = 0; q < Q; q++) {
        double s = 0.0;
#pragma unroll
        for (int jj = 0; jj < Q; jj++) s = fma(v[jj], we[jj * Q + q], s);
        C[q] = s;
      }
#pragma unroll
      for (int r = 0; r < Q; r++) {
        const double wr = wre[j * Q + r];
#pragma unroll
        for (int q = 0; q < Q; q++) acc[r][q - r + P] = fma(wr, C[q], acc[r][q - r + P]);
      }
    }
    // closing vertex P*(e+1): node j = P of element e; its columns also cover element e+1 unless it is the last node
    if (e + 1 < D.nel) {
      tt_cdp wn = we + NW;
      double v[W], C[Q + 1];
      io.template load<W>(a0 + P, a0, v);
#pragma unroll
      for (int q = 0; q < Q; q++) {
        double s = 0.0;
#pragma unroll
        for (int jj = 0; jj < Q; jj++) s = fma(v[jj], we[jj * Q + q], s);
        C[q] = s;
      }
      C[Q] = 0.0;
#pragma unroll
      for (int q = 1; q <= Q; q++) {              // dofs e+q from the nodes j' = 1..P of element e+1 (dof (e+1)+(q-1))
        double s = C[q];
#pragma unroll
        for (int jj = 1; jj < Q; jj++) s = fma(v[P + jj], wn[jj * Q + (q - 1)], s);
        C[q] = s;
      }
#pragma unroll
      for (int r = 0; r < Q; r++) {
        const double wr = wre[P * Q + r];
#pragma unroll
        for (int q = 0; q <= Q; q++) {
          // column dof e+q against output dof e+r; q - r + P == 2P+1 only for (q, r) = (P+1, 0), whose row weight
          // (the closing vertex at dof e) is structurally absent (host-checked)
          if (q - r + P < W) acc[r][q - r + P] = fma(wr, C[q], acc[r][q - r + P]);
        }
      }
    } else {
      double v[Q], C[Q];
      io.template load<Q>(a0 + P, a0, v);
#pragma unroll
      for (int q = 0; q < Q; q++) {
        double s = 0.0;
#pragma unroll
        for (int jj = 0; jj < Q; jj++) s = fma(v[jj], we[jj * Q + q], s);
        C[q] = s;
      }
#pragma unroll
      for (int r = 0; r < Q; r++) {
        const double wr = wre[P * Q + r];
#pragma unroll
        for (int q = 0; q < Q; q++) acc[r][q - r + P] = fma(wr, C[q], acc[r][q - r + P]);
      }
    }
    io.emit(e, acc[0]);                           // no later element touches dof e
#pragma unroll
    for (int r = 0; r < P; r++)
#pragma unroll
      for (int m = 0; m < W; m++) acc[r][m] = acc[r + 1][m];
#pragma unroll
    for (int m = 0; m < W; m++) acc[P][m] = 0.0;
  }
  if (e_end == D.nel) {
#pragma unroll
    for (int r = 0; r < P; r++) io.emit(D.nel + r, acc[r]);
  }
}

// ------------------------------------------------------------------------------------------------------
// stage X: A (CSR, verified) -> B1.   Rows of B1: (i0, r1, r2), block [c2][m0][c1] (c1 fastest).
struct tt_x_args {
  const int64_t *rowptr;
  const int32_t *col;
  const double *val;
  const int32_t *rps2;   // direction 2 prefix sums
  int aplane0;           // FE plane (direction 2) of A's first row
  tt_dir_t d0;
  int nfe1, nfe2;
  const int32_t *rps1;   // direction 1 prefix sums (addresses in B1)
  const int32_t *lines;  // the lines r1 of this class
  int nlines, L, n1;     // lines per wave, block extent in direction 1 of these lines
  const int32_t *planes; // FE planes r2 of this class (global)
  int n2;                // their block extent in direction 2
  double *b1;
  const int64_t *pb1;    // offset of plane r2 in b1, indexed r2 - z0
  int z0;
  int *status;           // bit 0: pattern mismatch
  // 2-D patches with nF fields (tt_y2_*): direction 2 is the FIELD index -- "dense": every row couples to all nfe2 = nF
  // of them (row block [g][c1][c0]) and nothing is contracted there
  int dense2;
  // the walk in pieces of `ech` elements (0: the whole direction at once), piece index = third argument of tt_x_lane:
  // short directions (2-D patches) have too few lines to fill the chip otherwise; a piece re-reads P elements
  int ech;
};

template <int P, bool V = true>       // V: read and verify the column indices (false: the matrix carries a pattern certificate)
struct tt_io_x {
  const int32_t *col;
  const double *val;
  tt_cip ps0;                 // prefix sums of the 1-D row lengths of direction 0
  int64_t linebase;           // entry index of the line's first row block (+ the lane's offset is added per row)
  int64_t lpl;                // entries of one row block per 1-D entry: n1*n2
  int64_t off_s, off_v;       // lane offset inside a short / vertex row block
  int32_t colbase;
  bool valid;
  int bad;
  int elo, ehi;               // output rows emitted by this piece of the walk
  double *out;
  int64_t ostride_i, ostride_m;
  // Row starts follow in closed form from the row LENGTHS (checked for every row by k_tt_check_rows before this
  // kernel runs; a mismatch stops the pass), so no row pointer is read here and nothing depends on a previous load.
  template <int N>
  TT_MEM void load(int a, int clo, double *v) {
#pragma unroll
    for (int j = 0; j < N; j++) v[j] = 0.0;
    if (!valid) return;
    // (ps0[a] in closed form -- the row lengths are those of the element-coupling pattern, verified by k_tt_check_rows --:
    //  the addresses of the row's values do not wait for a scalar load)
    const int64_t o = linebase + lpl * tt_rps<P>(a) + (N == P + 1 ? off_s : off_v);
    const int32_t c0 = colbase + clo;
    int32_t diff = 0;
#pragma unroll
    for (int j = 0; j < N; j++) {
      v[j] = val[o + j];
      if (V) diff |= col[o + j] ^ (c0 + j);
    }
    bad |= diff;
  }
  TT_MEM void emit(int i, const double *row) {
    if (!valid || i < elo || i >= ehi) return;
    double *d = out + ostride_i * i;
#pragma unroll
    for (int m = 0; m < 2 * P + 1; m++) TT_ST(&d[m * ostride_m], row[m]);
  }
};

// the elements a piece of a chunked walk visits and the rows it emits (ech = 0: everything)
TT_DEV void tt_piece(int ech, int piece, int P, int nel, int &e0, int &e1, int &lo, int &hi) {
  e0 = 0, e1 = nel, lo = 0, hi = 0x7fffffff;
  if (ech > 0) {
    lo = piece * ech;
    e0 = lo - P > 0 ? lo - P : 0;
    e1 = lo + ech < nel ? lo + ech : nel;
    if (e1 < nel) hi = lo + ech;
  }
}

template <int P, bool V = true>
TT_DEV int tt_x_lane(const tt_x_args &A, int bx, int by, int lane, int piece = 0) {
  constexpr int W = 2 * P + 1;
  const int plane = A.planes[by];
  const int lpl = A.n1 * A.n2;
  const int sub = lane / lpl, l = lane - sub * lpl;
  const int li = bx * A.L + sub;
  tt_io_x<P, V> io;
  io.valid = sub < A.L && li < A.nlines;
  const int r1 = io.valid ? A.lines[li] : 0;
  const int c1 = l % A.n1, c2 = l / A.n1;
  io.col = A.col;
  io.val = A.val;
  io.ps0 = TT_CI(A.d0.rps);
  {
    // entry index of row block (a = 0, r1, plane): separable prefix sums of the block sizes n0*n1*n2
    const int64_t t0 = A.d0.rps[A.d0.nfe], t1 = A.rps1[A.nfe1];
    io.linebase = A.rowptr[0] + t0 * t1 * (A.rps2[plane] - A.rps2[A.aplane0]) + (int64_t)A.n2 * t0 * A.rps1[r1];
  }
  io.lpl = lpl;
  io.off_s = (int64_t)l * (P + 1);
  io.off_v = (int64_t)l * W;
  const int rlo2 = A.dense2 ? 0 : tt_rlo<P>(plane, A.nfe2), rn2 = A.dense2 ? A.nfe2 : tt_rn<P>(plane, A.nfe2);
  io.colbase = (int32_t)((int64_t)A.d0.nfe * ((tt_rlo<P>(r1, A.nfe1) + c1) + (int64_t)A.nfe1 * (rlo2 + c2)));
  io.bad = 0;
  // consistency of the class tables with the grid (cheap, uniform)
  if (io.valid && (tt_rn<P>(r1, A.nfe1) != A.n1 || rn2 != A.n2)) io.bad = 1;
  const int64_t wn2 = (int64_t)W * A.n2;
  io.out = A.b1 + A.pb1[plane - A.z0] + wn2 * A.d0.ncp * A.rps1[r1] + (int64_t)c2 * W * A.n1 + c1;
  io.ostride_i = wn2 * A.n1;
  io.ostride_m = A.n1;
  int e0, e1;
  tt_piece(A.ech, piece, P, A.d0.nel, e0, e1, io.elo, io.ehi);
  tt_walk<P>(A.d0, e0, e1, io);
  return io.bad;
}

// ------------------------------------------------------------------------------------------------------
// stage X for an FE matrix GIVEN AS A KRONECKER SUM  A = sum_u F2u (x) F1u (x) F0u  of 1-D matrices on the
// element-coupling pattern (what the forms of tigar_amd.forms assemble on an identity-geometry patch): the matrix is
// never written -- the lane forms its entries while it walks,
//     A[(a, r1, r2), (c0, c1, c2)] = sum_u F0u[a][c0] * (F1u[r1][c1] * F2u[r2][c2]),
// the bracket being constant along the walk (one scalar per term in the lane) and F0u[a][.] wave-uniform (scalar
// loads).  The sum is formed exactly as tg_kron_sum_csr forms it (terms in order, fused multiply-adds), so the
// result is bit for bit what the x pass computes from the materialised matrix.
template <int NT>
struct tt_xg_args {
  tt_dir_t d0;
  const double *cv0, *cv1, *cv2;   // 1-D values, term-major: cv[u * nnz + q]
  int nnz0, nnz1, nnz2;
  const int32_t *rps1, *rps2;      // prefix sums of the 1-D row lengths (= positions of the 1-D rows)
  int nfe1, nfe2;
  const int32_t *lines;
  int nlines, L, n1;
  const int32_t *planes;
  int n2;
  double *b1;
  const int64_t *pb1;
  int z0;
  // the walk in pieces of `ech` elements (0: the whole direction at once), as tt_x_args::ech: with the piece as the
  // slowest-varying block index the workgroups resident at one time read the SAME few KB of the Kronecker-factor rows and
  // local weights, which then stay in the scalar cache
  int ech;
};

template <int P, int NT>
struct tt_io_xg {
  tt_cdp f0;
  tt_cip ps0;
  int nnz0;
  double g[NT];
  bool valid;
  double *out;
  int64_t ostride_i, ostride_m;
  int elo, ehi;                    // rows emitted by this piece of the walk
  template <int N>
  TT_MEM void load(int a, int clo, double *v) {
    (void)clo;
    // position of row a in the 1-D factor: its pattern is the element-coupling pattern (checked on the host,
    // tensorptap.pack_kron_factors), so the prefix sum of the row lengths is closed form -- (P+1) a + P * (interior vertices
    // before a) -- and the address of the row does not wait for a scalar load of ps0[a] (two dependent scalar-memory
    // latencies per node were what the walk waited for most: 36 % VALU busy at 3.3 waves per SIMD)
    tt_cdp f = f0 + tt_rps<P>(a);
#pragma unroll
    for (int j = 0; j < N; j++) {
      double acc = fma(f[j], g[0], 0.0);
#pragma unroll
      for (int u = 1; u < NT; u++) acc = fma(f[(int64_t)u * nnz0 + j], g[u], acc);
      v[j] = valid ? acc : 0.0;
    }
  }
  TT_MEM void emit(int i, const double *row) {
    if (!valid || i < elo || i >= ehi) return;
    double *d = out + ostride_i * i;
#pragma unroll
    for (int m = 0; m < 2 * P + 1; m++) TT_ST(&d[m * ostride_m], row[m]);
  }
};

template <int P, int NT>
TT_DEV void tt_xg_lane(const tt_xg_args<NT> &A, int bx, int by, int lane, int piece = 0) {
  constexpr int W = 2 * P + 1;
  const int plane = A.planes[by];
  const int lpl = A.n1 * A.n2;
  const int sub = lane / lpl, l = lane - sub * lpl;
  const int li = bx * A.L + sub;
  tt_io_xg<P, NT> io;
  io.valid = sub < A.L && li < A.nlines;
  const int r1 = io.valid ? A.lines[li] : 0;
  const int c1 = l % A.n1, c2 = l / A.n1;
  io.f0 = TT_CD(A.cv0);
  io.ps0 = TT_CI(A.d0.rps);
  io.nnz0 = A.nnz0;
  {
    const int q1 = A.rps1[r1] + (io.valid ? c1 : 0), q2 = A.rps2[plane] + (io.valid ? c2 : 0);
#pragma unroll
    for (int u = 0; u < NT; u++) io.g[u] = A.cv1[(int64_t)u * A.nnz1 + q1] * A.cv2[(int64_t)u * A.nnz2 + q2];
  }
  const int64_t wn2 = (int64_t)W * A.n2;
  io.out = A.b1 + A.pb1[plane - A.z0] + wn2 * A.d0.ncp * A.rps1[r1] + (int64_t)c2 * W * A.n1 + c1;
  io.ostride_i = wn2 * A.n1;
  io.ostride_m = A.n1;
  int e0, e1;
  tt_piece(A.ech, piece, P, A.d0.nel, e0, e1, io.elo, io.ehi);
  tt_walk<P>(A.d0, e0, e1, io);
}

// Row lengths of A against the element-coupling pattern: row (a, r1, r2) must hold n0(a)*n1(r1)*n2(r2) entries.
// idx runs over the rows of the planes [z0, z1); returns 1 on a mismatch.
struct tt_check_args {
  const int64_t *rowptr;
  int nfe0, nfe1, nfe2, aplane0, z0;
  int dense2;                // direction 2 = fields (see tt_x_args)
};
template <int P>
TT_DEV int tt_check_row(const tt_check_args &A, int64_t idx) {
  const int64_t pf = (int64_t)A.nfe0 * A.nfe1;
  const int r2 = A.z0 + (int)(idx / pf);
  const int64_t rem = idx % pf;
  const int r1 = (int)(rem / A.nfe0), a = (int)(rem % A.nfe0);
  const int64_t r = idx + (int64_t)(A.z0 - A.aplane0) * pf;
  const int64_t want = (int64_t)tt_rn<P>(a, A.nfe0) * tt_rn<P>(r1, A.nfe1) * (A.dense2 ? A.nfe2 : tt_rn<P>(r2, A.nfe2));
  return (A.rowptr[r + 1] - A.rowptr[r]) != want;
}

// ------------------------------------------------------------------------------------------------------
// stage Y: B1 -> B2.   Rows of B2: (i0, i1, r2), block [m1][m0][c2] (c2 fastest).
struct tt_y_args {
  const double *b1;
  const int64_t *pb1;
  double *b2;
  const int64_t *pb2;
  int z0;
  tt_dir_t d1;
  int ncp0;
  const int32_t *planes;
  int n2, L;
  int ech;                       // the walk in pieces of `ech` elements (0: the whole direction), as tt_x_args::ech
};

template <int P>
struct tt_io_y {
  const double *in;
  tt_cip rps;
  int64_t ustride, clane;
  bool valid;
  int elo, ehi;                  // rows emitted by this piece of the walk
  double *out;
  int64_t ostride_i, ostride_m;
  template <int N>
  TT_MEM void load(int a, int clo, double *v) {
#pragma unroll
    for (int j = 0; j < N; j++) v[j] = 0.0;
    if (!valid) return;
    const int64_t o = ustride * tt_rps<P>(a) + clane * N;      // (closed form: see tt_rps)
#pragma unroll
    for (int j = 0; j < N; j++) v[j] = in[o + j];
  }
  TT_MEM void emit(int i, const double *row) {
    if (!valid || i < elo || i >= ehi) return;
    double *d = out + ostride_i * i;
#pragma unroll
    for (int m = 0; m < 2 * P + 1; m++) TT_ST(&d[m * ostride_m], row[m]);
  }
};

template <int P>
TT_DEV void tt_y_lane(const tt_y_args &A, int bx, int by, int lane, int piece = 0) {
  constexpr int W = 2 * P + 1;
  const int plane = A.planes[by];
  const int lpl = W * A.n2;
  const int sub = lane / lpl, l = lane - sub * lpl;
  const int i0 = bx * A.L + sub;
  tt_io_y<P> io;
  io.valid = sub < A.L && i0 < A.ncp0;
  const int m0 = l % W, c2 = l / W;
  const int64_t wn2 = (int64_t)W * A.n2;
  io.in = A.b1 + A.pb1[plane - A.z0];
  io.rps = TT_CI(A.d1.rps);
  io.ustride = wn2 * A.ncp0;
  io.clane = wn2 * i0 + l;
  io.out = A.b2 + A.pb2[plane - A.z0] + (int64_t)W * wn2 * i0 + (int64_t)m0 * A.n2 + c2;
  io.ostride_i = (int64_t)W * wn2 * A.ncp0;
  io.ostride_m = wn2;
  int e0, e1;
  tt_piece(A.ech, piece, P, A.d1.nel, e0, e1, io.elo, io.ehi);
  tt_walk<P>(A.d1, e0, e1, io);
}

// ------------------------------------------------------------------------------------------------------
// stage Z: B2 planes -> rows of K (CSR order, closed-form positions, MatZeroRowsColumns fused).
struct tt_z_args {
  const double *const *planes;   // pointer to the B2 block of FE plane r2, indexed r2 - plane_lo
  int plane_lo;
  tt_dir_t d2;
  int ncp0, ncp1;
  const int32_t *kps0, *kps1;
  int ka, kb;                    // dof planes whose rows are written
  int L;
  int32_t *kcol;                 // destination arrays, already offset to the first entry of dof plane ka
  double *kval;
  double *kdiag;                 // diagonal of the rows written (index: row - first row of dof plane ka), or null
  const uint8_t *mask;           // zeroDofs as a byte mask over all dofs, or null
  double diag;
  // Blocks with different bases on the row and the column side (see tt_dir_t::wlr); 0 = the square default.  ncp0 / ncp1
  // above stay the PADDED counts nel + P (the layout of B2); these are the true ones and the spline degrees per side:
  // row dof i couples to the column dofs [i - pr, i + pc] (clipped), i.e. to the slots m in [P - min(pr, i), ...) of the walk
  int ncr0, ncr1, ncc0, ncc1;    // true row / column function counts of directions 0, 1
  int pr0, pr1, pr2;             // spline degree of the row side per direction
};

template <int P>
struct tt_io_z {
  const double *const *planes;
  int plane_lo;
  int64_t clane;
  bool valid, inwin;
  // K addressing
  const int32_t *kps2;
  int ka, kb, i0, i1, m0, m1, ncp0, ncp1, w0n, w1n, w0lo, w1lo;
  int ncc0, ncc1, pr2;           // (ncp0 / ncp1 here: the true ROW counts)
  int64_t w01tot, rowoff01;
  int32_t *kcol;
  double *kval;
  double *kdiag;
  const uint8_t *mask;
  double diag;
  template <int N>
  TT_MEM void load(int a, int clo, double *v) {
#pragma unroll
    for (int j = 0; j < N; j++) v[j] = 0.0;
    if (!valid) return;
    tt_gdp pl = TT_GD(planes[a - plane_lo]);      // (a pointer read from a table: say that it is global, not flat)
    const int64_t o = clane * N;
#pragma unroll
    for (int j = 0; j < N; j++) v[j] = pl[o + j];
  }
  TT_MEM void emit(int i2, const double *row) {
    if (!valid || !inwin || i2 < ka || i2 >= kb) return;
    const int w2n = kps2[i2 + 1] - kps2[i2];
    const int w2lo = P - (i2 < pr2 ? i2 : pr2);
    const int64_t rowstart = w01tot * (kps2[i2] - kps2[ka]) + (int64_t)w2n * rowoff01;
    const int64_t R = i0 + (int64_t)ncp0 * (i1 + (int64_t)ncp1 * i2);
    const bool mrow = mask && mask[R];
    const int64_t within = (int64_t)(m1 - w1lo) * w0n + (m0 - w0lo);
    const int64_t c01 = (i0 - P + m0) + (int64_t)ncc0 * (i1 - P + m1);
#pragma unroll
    for (int m2 = 0; m2 < 2 * P + 1; m2++) {
      if (m2 >= w2lo && m2 < w2lo + w2n) {
        const int64_t pos = rowstart + (int64_t)(m2 - w2lo) * w1n * w0n + within;
        const int64_t c = c01 + (int64_t)ncc0 * ncc1 * (i2 - P + m2);
        double v = row[m2];
        if (mask && (mrow || mask[c])) v = (mrow && c == R) ? diag : 0.0;
        kcol[pos] = (int32_t)c;
        kval[pos] = v;
        if (m2 == P && kdiag && m0 == P && m1 == P) kdiag[R - (int64_t)ka * ncp0 * ncp1] = v;   // c == R
      }
    }
  }
};

template <int P>
TT_DEV void tt_z_lane(const tt_z_args &A, int bx, int lane) {
  constexpr int W = 2 * P + 1;
  const int lpl = W * W;
  const int sub = lane / lpl, l = lane - sub * lpl;
  const int64_t line = (int64_t)bx * A.L + sub;
  tt_io_z<P> io;
  io.valid = sub < A.L && line < (int64_t)A.ncp0 * A.ncp1;
  const int i0 = io.valid ? (int)(line % A.ncp0) : 0, i1 = io.valid ? (int)(line / A.ncp0) : 0;
  const int m0 = l % W, m1 = l / W;
  // true function counts and row-side degrees (square blocks: the padded counts, P)
  const int ncr0 = A.ncr0 ? A.ncr0 : A.ncp0, ncr1 = A.ncr1 ? A.ncr1 : A.ncp1;
  const int pr0 = A.pr0 ? A.pr0 : P, pr1 = A.pr1 ? A.pr1 : P;
  const bool real = i0 < ncr0 && i1 < ncr1;       // (rows of functions the padding added are not written)
  const int j0 = real ? i0 : 0, j1 = real ? i1 : 0;
  io.planes = A.planes;
  io.plane_lo = A.plane_lo;
  io.clane = (int64_t)lpl * (i0 + (int64_t)A.ncp0 * i1) + l;
  io.kps2 = A.d2.kps;
  io.ka = A.ka;
  io.kb = A.kb;
  io.i0 = i0;
  io.i1 = i1;
  io.m0 = m0;
  io.m1 = m1;
  io.ncp0 = ncr0;
  io.ncp1 = ncr1;
  io.ncc0 = A.ncc0 ? A.ncc0 : A.ncp0;
  io.ncc1 = A.ncc1 ? A.ncc1 : A.ncp1;
  io.pr2 = A.pr2 ? A.pr2 : P;
  io.w0n = A.kps0[j0 + 1] - A.kps0[j0];
  io.w1n = A.kps1[j1 + 1] - A.kps1[j1];
  io.w0lo = P - (i0 < pr0 ? i0 : pr0);
  io.w1lo = P - (i1 < pr1 ? i1 : pr1);
  io.inwin = real && m0 >= io.w0lo && m0 < io.w0lo + io.w0n && m1 >= io.w1lo && m1 < io.w1lo + io.w1n;
  const int64_t w0tot = A.kps0[ncr0], w1tot = A.kps1[ncr1];
  io.w01tot = w0tot * w1tot;
  io.rowoff01 = w0tot * A.kps1[j1] + (int64_t)io.w1n * A.kps0[j0];
  io.kcol = A.kcol;
  io.kval = A.kval;
  io.kdiag = A.kdiag;
  io.mask = A.mask;
  io.diag = A.diag;
  const int nel = A.d2.nel;
  const int pr2 = A.pr2 ? A.pr2 : P;              // (row function i lives on the elements [i - pr2, i])
  const int e_begin = A.ka - pr2 > 0 ? A.ka - pr2 : 0;
  const int e_end = A.kb < nel ? A.kb : nel;
  tt_walk<P>(A.d2, e_begin, e_end, io);
}

// ------------------------------------------------------------------------------------------------------
// 2-D patches (nF fields on one basis): K = P_y^T (P_x^T A P_x) P_y, the x pass above (direction 2 = field index,
// dense) followed by this FINAL pass along direction 1: B1 rows (i0, r1, f) with block [g][m0][c1] -> rows of K in CSR
// order.  Row (i0, i1, f) of K holds nF * w1n(i1) * w0n(i0) entries [g][m1][m0] (columns ascending).
struct tt_y2_args {
  const double *b1;
  int64_t plane_b1;              // doubles of B1 per field f
  tt_dir_t d1;
  int ncp0, nF;
  const int32_t *kps0;
  int L, ech;
  int32_t *kcol;
  double *kval;
  double *kdiag;                 // diagonal of K (index: row), or null
  const uint8_t *mask;           // zeroDofs as a byte mask over all dofs, or null
  double diag;
  // A block with different bases on the row and the column side (one field each: nF = 1; see tt_dir_t::wlr and tt_z_args),
  // 0 = the square default.  ncp0 / d1.ncp stay the PADDED counts nel + P (the layout of B1); these are the true function
  // counts and the row-side spline degrees: row dof i couples to the column dofs [i - pr, i + pc] (clipped)
  int ncr0, ncr1, ncc0, ncc1, pr0, pr1;
};

template <int P>
struct tt_io_y2 {
  const double *in;
  tt_cip rps;
  int64_t ustride, clane;
  bool valid, inwin;
  int elo, ehi;
  tt_cip kps1;
  int f, g, i0, m0, ncp0, ncp1, nF, w0n, w0lo;
  int ncc0, ncc1, pr1;           // (ncp0 / ncp1 here: the true ROW counts; ncc: the column side's; pr1: row degree of direction 1)
  bool square;
  int64_t w0tot, w1tot, kp0;     // kp0 = kps0[i0]
  int32_t *kcol;
  double *kval;
  double *kdiag;
  const uint8_t *mask;
  double diag;
  template <int N>
  TT_MEM void load(int a, int clo, double *v) {
#pragma unroll
    for (int j = 0; j < N; j++) v[j] = 0.0;
    if (!valid) return;
    const int64_t o = ustride * tt_rps<P>(a) + clane * N;      // (closed form: see tt_rps)
#pragma unroll
    for (int j = 0; j < N; j++) v[j] = in[o + j];
  }
  TT_MEM void emit(int i1, const double *row) {
    if (!valid || !inwin || i1 < elo || i1 >= ehi || i1 >= ncp1) return;       // (ncp1: rows of functions the padding added)
    const int w1n = kps1[i1 + 1] - kps1[i1];
    const int w1lo = P - (i1 < pr1 ? i1 : pr1);
    const int64_t pd = (int64_t)ncp0 * ncp1, pdc = (int64_t)ncc0 * ncc1;
    const int64_t R = i0 + (int64_t)ncp0 * i1 + pd * f;
    const int64_t rowstart = (int64_t)nF * ((int64_t)f * w0tot * w1tot + w0tot * kps1[i1] + (int64_t)w1n * kp0);
    const int64_t base = rowstart + (int64_t)g * w1n * w0n + (m0 - w0lo);
    const int64_t c0g = (i0 - P + m0) + pdc * g;
    const bool mrow = mask && mask[R];
#pragma unroll
    for (int m1 = 0; m1 < 2 * P + 1; m1++) {
      if (m1 >= w1lo && m1 < w1lo + w1n) {
        const int64_t pos = base + (int64_t)(m1 - w1lo) * w0n;
        const int64_t c = c0g + (int64_t)ncc0 * (i1 - P + m1);
        double v = row[m1];
        if (mask && (mrow || mask[c])) v = (mrow && c == R) ? diag : 0.0;
        kcol[pos] = (int32_t)c;
        kval[pos] = v;
        if (kdiag && square && c == R) kdiag[R] = v;
      }
    }
  }
};

// grid: bx = group of lines (i0) x by = field f x piece of the walk
template <int P>
TT_DEV void tt_y2_lane(const tt_y2_args &A, int bx, int by, int piece, int lane) {
  constexpr int W = 2 * P + 1;
  const int lpl = W * A.nF;
  const int sub = lane / lpl, l = lane - sub * lpl;
  const int i0 = bx * A.L + sub;
  tt_io_y2<P> io;
  io.valid = sub < A.L && i0 < A.ncp0;
  const int m0 = l % W, g = l / W;
  const int64_t wn2 = (int64_t)W * A.nF;
  io.in = A.b1 + A.plane_b1 * by;
  io.rps = TT_CI(A.d1.rps);
  io.ustride = wn2 * A.ncp0;
  io.clane = wn2 * (io.valid ? i0 : 0) + l;
  io.kps1 = TT_CI(A.d1.kps);
  io.f = by;
  io.g = g;
  io.i0 = io.valid ? i0 : 0;
  io.m0 = m0;
  // true function counts and row-side degrees (square blocks: the padded counts, P)
  const int ncr0 = A.ncr0 ? A.ncr0 : A.ncp0, ncr1 = A.ncr1 ? A.ncr1 : A.d1.ncp;
  const int pr0 = A.pr0 ? A.pr0 : P;
  const bool real = io.i0 < ncr0;                 // (rows of functions the padding added are not written)
  const int j0 = real ? io.i0 : 0;
  io.ncp0 = ncr0;
  io.ncp1 = ncr1;
  io.ncc0 = A.ncc0 ? A.ncc0 : A.ncp0;
  io.ncc1 = A.ncc1 ? A.ncc1 : A.d1.ncp;
  io.pr1 = A.pr1 ? A.pr1 : P;
  io.square = A.ncr0 == 0;
  io.nF = A.nF;
  io.kp0 = A.kps0[j0];
  io.w0n = A.kps0[j0 + 1] - A.kps0[j0];
  io.w0lo = P - (io.i0 < pr0 ? io.i0 : pr0);
  io.inwin = real && m0 >= io.w0lo && m0 < io.w0lo + io.w0n;
  io.w0tot = A.kps0[ncr0];
  io.w1tot = A.d1.kps[ncr1];
  io.kcol = A.kcol;
  io.kval = A.kval;
  io.kdiag = A.kdiag;
  io.mask = A.mask;
  io.diag = A.diag;
  int e0, e1;
  tt_piece(A.ech, piece, P, A.d1.nel, e0, e1, io.elo, io.ehi);
  tt_walk<P>(A.d1, e0, e1, io);
}

// row pointer of K of a 2-D patch with nF fields: row R = i0 + ncp0*(i1 + ncp1*f)
struct tt_rowptr2_args {
  const int32_t *kps0, *kps1;
  int ncp0, ncp1, nF;
  int64_t *rowptr_out;
};
TT_DEV void tt_rowptr2_one(const tt_rowptr2_args &A, int64_t idx) {
  const int64_t pd = (int64_t)A.ncp0 * A.ncp1;
  const int f = (int)(idx / pd);
  const int64_t rem = idx % pd;
  const int i1 = (int)(rem / A.ncp0), i0 = (int)(rem % A.ncp0);
  const int64_t w0tot = A.kps0[A.ncp0], w1tot = A.kps1[A.ncp1];
  const int w1n = A.kps1[i1 + 1] - A.kps1[i1];
  A.rowptr_out[idx] = (int64_t)A.nF * ((int64_t)f * w0tot * w1tot + w0tot * A.kps1[i1] + (int64_t)w1n * A.kps0[i0]);
}

// row pointer of K for the rows of dof planes [ka, kb): rowptr_out[R - ka*ncp0*ncp1] = base + closed form
struct tt_rowptr_args {
  const int32_t *kps0, *kps1, *kps2;
  int ncp0, ncp1, ka, kb;
  int64_t base;
  int64_t *rowptr_out;
};
TT_DEV void tt_rowptr_one(const tt_rowptr_args &A, int64_t idx) {
  const int64_t pd = (int64_t)A.ncp0 * A.ncp1;
  const int i2 = A.ka + (int)(idx / pd);
  const int64_t rem = idx % pd;
  const int i1 = (int)(rem / A.ncp0), i0 = (int)(rem % A.ncp0);
  const int64_t w0tot = A.kps0[A.ncp0], w1tot = A.kps1[A.ncp1];
  const int w2n = A.kps2[i2 + 1] - A.kps2[i2], w1n = A.kps1[i1 + 1] - A.kps1[i1];
  A.rowptr_out[idx] = A.base + w0tot * w1tot * (A.kps2[i2] - A.kps2[A.ka]) +
                      (int64_t)w2n * (w0tot * A.kps1[i1] + (int64_t)w1n * A.kps0[i0]);
}
