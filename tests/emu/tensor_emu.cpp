// Host emulation of the tensor-pattern PtAP kernels: runs the SAME per-lane code as the HIP kernels
// (tigar_amd/csrc/tg_tensor_body.h) block by block, lane by lane.  Test infrastructure only (CPU suite): pins
// the index arithmetic of the line walks against the oracle without a GPU.  Not part of the product library.
#include "../../tigar_amd/csrc/tg_tensor_body.h"
#include <stdint.h>

template <int P>
static int run_x(const tt_x_args &A, int gx, int gy) {
  int bad = 0;
  for (int by = 0; by < gy; by++)
    for (int bx = 0; bx < gx; bx++)
      for (int lane = 0; lane < 64; lane++) bad |= tt_x_lane<P>(A, bx, by, lane);
  return bad;
}
template <int P>
static void run_y(const tt_y_args &A, int gx, int gy) {
  for (int by = 0; by < gy; by++)
    for (int bx = 0; bx < gx; bx++)
      for (int lane = 0; lane < 64; lane++) tt_y_lane<P>(A, bx, by, lane);
}
template <int P>
static void run_z(const tt_z_args &A, int gx) {
  for (int bx = 0; bx < gx; bx++)
    for (int lane = 0; lane < 64; lane++) tt_z_lane<P>(A, bx, lane);
}

extern "C" {
// one launch of the x pass for one (plane class, line class); returns the mismatch flag
int emu_x(int P, const int64_t *rowptr, const int32_t *col, const double *val, int aplane0, int nel0, int nfe1, int nfe2,
          const double *wl0, const int32_t *rps0, const int32_t *rps1, const int32_t *rps2, const int32_t *lines, int nlines, int L, int n1, const int32_t *planes,
          int nplanes, int n2, double *b1, const int64_t *pb1, int z0, int64_t nrows_slab) {
  tt_x_args A = {};
  A.rowptr = rowptr;
  A.col = col;
  A.val = val;
  A.aplane0 = aplane0;
  A.d0.nel = nel0;
  A.d0.nfe = P * nel0 + 1;
  A.d0.ncp = nel0 + P;
  A.d0.wl = wl0;
  A.d0.rps = rps0;
  A.d0.kps = nullptr;
  A.rps2 = rps2;
  {   // what k_tt_check_rows does before the x pass
    tt_check_args Cq = {};
    Cq.rowptr = rowptr;
    Cq.nfe0 = A.d0.nfe;
    Cq.nfe1 = nfe1;
    Cq.nfe2 = nfe2;
    Cq.aplane0 = aplane0;
    Cq.z0 = z0;
    const int64_t nr = (int64_t)nrows_slab;
    for (int64_t i = 0; i < nr; i++) {
      const int bad = P == 1 ? tt_check_row<1>(Cq, i) : (P == 2 ? tt_check_row<2>(Cq, i) : tt_check_row<3>(Cq, i));
      if (bad) return 1;
    }
  }
  A.nfe1 = nfe1;
  A.nfe2 = nfe2;
  A.rps1 = rps1;
  A.lines = lines;
  A.nlines = nlines;
  A.L = L;
  A.n1 = n1;
  A.planes = planes;
  A.n2 = n2;
  A.b1 = b1;
  A.pb1 = pb1;
  A.z0 = z0;
  A.status = nullptr;
  const int gx = (nlines + L - 1) / L;
  if (P == 1) return run_x<1>(A, gx, nplanes);
  if (P == 2) return run_x<2>(A, gx, nplanes);
  return run_x<3>(A, gx, nplanes);
}

void emu_y(int P, const double *b1, const int64_t *pb1, double *b2, const int64_t *pb2, int z0, int nel1, const double *wl1,
           const int32_t *rps1, int ncp0, const int32_t *planes, int nplanes, int n2, int L) {
  tt_y_args A = {};
  A.b1 = b1;
  A.pb1 = pb1;
  A.b2 = b2;
  A.pb2 = pb2;
  A.z0 = z0;
  A.d1.nel = nel1;
  A.d1.nfe = P * nel1 + 1;
  A.d1.ncp = nel1 + P;
  A.d1.wl = wl1;
  A.d1.rps = rps1;
  A.d1.kps = nullptr;
  A.ncp0 = ncp0;
  A.planes = planes;
  A.n2 = n2;
  A.L = L;
  const int gx = (ncp0 + L - 1) / L;
  if (P == 1) run_y<1>(A, gx, nplanes);
  else if (P == 2) run_y<2>(A, gx, nplanes);
  else run_y<3>(A, gx, nplanes);
}

void emu_z(int P, const double *const *planes, int plane_lo, int nel2, const double *wl2, const int32_t *kps2, int ncp0,
           int ncp1, const int32_t *kps0, const int32_t *kps1, int ka, int kb, int L, int32_t *kcol, double *kval,
           const uint8_t *mask, double diag, int64_t *rowptr_out, int64_t base) {
  tt_z_args A = {};
  A.planes = planes;
  A.plane_lo = plane_lo;
  A.d2.nel = nel2;
  A.d2.nfe = P * nel2 + 1;
  A.d2.ncp = nel2 + P;
  A.d2.wl = wl2;
  A.d2.rps = nullptr;
  A.d2.kps = kps2;
  A.ncp0 = ncp0;
  A.ncp1 = ncp1;
  A.kps0 = kps0;
  A.kps1 = kps1;
  A.ka = ka;
  A.kb = kb;
  A.L = L;
  A.kcol = kcol;
  A.kval = kval;
  A.kdiag = nullptr;
  A.mask = mask;
  A.diag = diag;
  const int64_t pd = (int64_t)ncp0 * ncp1;
  const int gx = (int)((pd + L - 1) / L);
  if (P == 1) run_z<1>(A, gx);
  else if (P == 2) run_z<2>(A, gx);
  else run_z<3>(A, gx);
  tt_rowptr_args R;
  R.kps0 = kps0;
  R.kps1 = kps1;
  R.kps2 = kps2;
  R.ncp0 = ncp0;
  R.ncp1 = ncp1;
  R.ka = ka;
  R.kb = kb;
  R.base = base;
  R.rowptr_out = rowptr_out;
  for (int64_t i = 0; i < (int64_t)(kb - ka) * pd; i++) tt_rowptr_one(R, i);
}

}  // extern "C"

// ---- 2-D patches with nF fields: x pass with the field index as dense third direction, then the final pass
template <int P>
static int run_x2(const tt_x_args &A, int gx, int nF, int npieces) {
  int bad = 0;
  for (int pc = 0; pc < npieces; pc++)
    for (int f = 0; f < nF; f++)
      for (int bx = 0; bx < gx; bx++)
        for (int lane = 0; lane < 64; lane++) bad |= tt_x_lane<P>(A, bx, f, lane, pc);
  return bad;
}
template <int P>
static void run_y2(const tt_y2_args &A, int gx, int npieces) {
  for (int pc = 0; pc < npieces; pc++)
    for (int f = 0; f < A.nF; f++)
      for (int bx = 0; bx < gx; bx++)
        for (int lane = 0; lane < 64; lane++) tt_y2_lane<P>(A, bx, f, pc, lane);
}

extern "C" {
int emu_x2(int P, const int64_t *rowptr, const int32_t *col, const double *val, int nel0, int nfe1, int nF, const double *wl0,
           const int32_t *rps0, const int32_t *rps1, const int32_t *rps2, const int32_t *lines, int nlines, int L, int n1,
           const int32_t *planes, double *b1, const int64_t *pb1, int ech, int check_rows) {
  tt_x_args A = {};
  A.rowptr = rowptr;
  A.col = col;
  A.val = val;
  A.aplane0 = 0;
  A.d0.nel = nel0;
  A.d0.nfe = P * nel0 + 1;
  A.d0.ncp = nel0 + P;
  A.d0.wl = wl0;
  A.d0.rps = rps0;
  A.rps2 = rps2;
  A.nfe1 = nfe1;
  A.nfe2 = nF;
  A.rps1 = rps1;
  A.lines = lines;
  A.nlines = nlines;
  A.L = L;
  A.n1 = n1;
  A.planes = planes;
  A.n2 = nF;
  A.b1 = b1;
  A.pb1 = pb1;
  A.z0 = 0;
  A.dense2 = 1;
  A.ech = ech;
  if (check_rows) {
    tt_check_args Cq = {};
    Cq.rowptr = rowptr;
    Cq.nfe0 = A.d0.nfe;
    Cq.nfe1 = nfe1;
    Cq.nfe2 = nF;
    Cq.dense2 = 1;
    const int64_t nr = (int64_t)A.d0.nfe * nfe1 * nF;
    for (int64_t i = 0; i < nr; i++) {
      const int bad = P == 1 ? tt_check_row<1>(Cq, i) : P == 2 ? tt_check_row<2>(Cq, i) : P == 3 ? tt_check_row<3>(Cq, i) : tt_check_row<4>(Cq, i);
      if (bad) return 1;
    }
  }
  const int gx = (nlines + L - 1) / L;
  const int np = ech ? (nel0 + ech - 1) / ech : 1;
  if (P == 1) return run_x2<1>(A, gx, nF, np);
  if (P == 2) return run_x2<2>(A, gx, nF, np);
  if (P == 3) return run_x2<3>(A, gx, nF, np);
  return run_x2<4>(A, gx, nF, np);
}

void emu_y2(int P, const double *b1, int64_t plane_b1, int nel1, const double *wl1, const int32_t *rps1, const int32_t *kps1,
            int ncp0, int nF, const int32_t *kps0, int L, int ech, int32_t *kcol, double *kval, double *kdiag,
            const uint8_t *mask, double diag, int64_t *rowptr_out) {
  tt_y2_args A = {};
  A.b1 = b1;
  A.plane_b1 = plane_b1;
  A.d1.nel = nel1;
  A.d1.nfe = P * nel1 + 1;
  A.d1.ncp = nel1 + P;
  A.d1.wl = wl1;
  A.d1.rps = rps1;
  A.d1.kps = kps1;
  A.ncp0 = ncp0;
  A.nF = nF;
  A.kps0 = kps0;
  A.L = L;
  A.ech = ech;
  A.kcol = kcol;
  A.kval = kval;
  A.kdiag = kdiag;
  A.mask = mask;
  A.diag = diag;
  const int gx = (ncp0 + L - 1) / L;
  const int np = ech ? (nel1 + ech - 1) / ech : 1;
  if (P == 1) run_y2<1>(A, gx, np);
  else if (P == 2) run_y2<2>(A, gx, np);
  else if (P == 3) run_y2<3>(A, gx, np);
  else run_y2<4>(A, gx, np);
  tt_rowptr2_args R;
  R.kps0 = kps0;
  R.kps1 = kps1;
  R.ncp0 = ncp0;
  R.ncp1 = A.d1.ncp;
  R.nF = nF;
  R.rowptr_out = rowptr_out;
  for (int64_t i = 0; i < (int64_t)nF * ncp0 * A.d1.ncp; i++) tt_rowptr2_one(R, i);
}
}
