// Direct solve of K U = M^T b: banded LU with partial pivoting.
//
// With `linearSolver == None` the reference's solveLinearSystem calls dolfin's `solve(A, x, b)`, i.e. a sparse
// direct LU (tIGAr/common.py:1255-1256 [ext]); every demo of the reference relies on it (biharmonic, shells,
// penalty and saddle-point systems on which Jacobi-Krylov methods stall).  IGA matrices of a tensor-product
// patch are banded in the patch numbering (half-bandwidth ~ p * ncp_x in 2-D), so the direct solver here is a
// dense-band factorisation in LAPACK's dgbtrf storage and pivoting scheme (unblocked dgbtf2: row interchanges
// within the band, U's bandwidth grows to kl+ku), one column at a time:
//   k_lu_pivot   one workgroup: pivot search in column j, row interchange, multipliers
//   k_lu_update  rank-1 update of the trailing (km x (ju-j)) window, many workgroups
// followed by forward / backward substitution in one persistent workgroup each.  Launch-bound (2 launches per
// column), meant for the moderate sizes where the reference uses LU; large 3-D systems belong to the Krylov solvers.
#include "tg_common.h"
#include <algorithm>

struct tg_lu_state {     // device-resident control block
  int ju;                // last column touched by the row interchanges so far (LAPACK's JU), 0-based
  int info;              // 0, or 1 + index of the first exactly-zero pivot
  int km, jp;            // of the current column
  double pivinv;
};

// AB(i, j) of the LAPACK band layout: element (row r, col c) at ab[(kv + r - c) + ldab * c]
__global__ void __launch_bounds__(256) k_lu_scatter(const int64_t *__restrict__ rowptr, const int32_t *__restrict__ col,
                                                    const double *__restrict__ val, int64_t n, int kv, int64_t ldab,
                                                    double *__restrict__ ab) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t r = wave; r < n; r += nwaves)
    for (int64_t q = rowptr[r] + lane; q < rowptr[r + 1]; q += 64) {
      const int64_t c = col[q];
      ab[(kv + r - c) + ldab * c] += val[q];       // (+=: duplicate entries of a row add up, as in MatSetValues ADD)
    }
}

__global__ void __launch_bounds__(256) k_lu_bandwidth(const int64_t *__restrict__ rowptr, const int32_t *__restrict__ col,
                                                      int64_t n, int *__restrict__ kl, int *__restrict__ ku) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n) return;
  const int64_t a = rowptr[r], b = rowptr[r + 1];
  if (b <= a) return;
  // (columns are sorted within a row)
  const int64_t lo = r - col[a], hi = col[b - 1] - r;
  if (lo > 0) atomicMax(kl, (int)lo);
  if (hi > 0) atomicMax(ku, (int)hi);
}

__global__ void __launch_bounds__(256)
    k_lu_pivot(double *__restrict__ ab, int64_t ldab, int64_t n, int kl, int kv, int64_t j, int32_t *__restrict__ ipiv,
               tg_lu_state *st, int swap_trailing) {
  __shared__ double sval[256];
  __shared__ int sidx[256];
  const int tid = threadIdx.x;
  const int km = (int)min((int64_t)kl, n - 1 - j);
  double *cj = ab + kv + ldab * j;                 // cj[i] = A(j+i, j)
  double best = -1.0;
  int bi = 0;
  for (int i = tid; i <= km; i += 256) {
    const double a = fabs(cj[i]);
    if (a > best || (a != a && best == best)) {   // first maximum; NaN wins so that it surfaces
      best = a;
      bi = i;
    }
  }
  sval[tid] = best;
  sidx[tid] = bi;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (tid < o) {
      const double b2 = sval[tid + o];
      const int i2 = sidx[tid + o];
      if (b2 > sval[tid] || (b2 == sval[tid] && i2 < sidx[tid])) {
        sval[tid] = b2;
        sidx[tid] = i2;
      }
    }
    __syncthreads();
  }
  const int jp = sidx[0];
  const double piv = cj[jp];
  int ju = st->ju;
  __syncthreads();
  if (tid == 0) {
    ipiv[j] = (int32_t)(j + jp);
    ju = max(ju, (int)min(j + (int64_t)(kv - kl) + jp, n - 1));     // ku = kv - kl
    st->ju = ju;
    st->km = km;
    st->jp = jp;
    if (piv == 0.0 && st->info == 0) st->info = (int)(j + 1);
    st->pivinv = piv != 0.0 ? 1.0 / piv : 0.0;
    sidx[1] = ju;
  }
  __syncthreads();
  ju = sidx[1];
  if (piv == 0.0) return;
  // interchange rows j and j+jp over the columns j..ju (stride ldab-1 walks along a row of the band)
  if (jp != 0) {
    const int64_t len = swap_trailing ? ju - j + 1 : 1;      // (fused path: the other columns are interchanged by k_lu_step)
    for (int64_t t = tid; t < len; t += 256) {
      double *p = cj + t * (ldab - 1);
      const double a = p[jp], b = p[0];
      p[jp] = b;
      p[0] = a;
    }
  }
  __syncthreads();
  const double pinv = 1.0 / piv;
  for (int i = 1 + tid; i <= km; i += 256) cj[i] *= pinv;
}

__global__ void __launch_bounds__(256)
    k_lu_update(double *__restrict__ ab, int64_t ldab, int kv, int64_t j, const tg_lu_state *__restrict__ st) {
  const int km = st->km;
  const int64_t nc = st->ju - j;                   // trailing columns j+1..ju
  if (km <= 0 || nc <= 0 || st->pivinv == 0.0) return;
  const double *l = ab + kv + ldab * j;            // l[i], i = 1..km: multipliers
  // 2-D tiling: x over rows (contiguous in memory), y over columns
  for (int64_t c = blockIdx.y; c < nc; c += gridDim.y) {
    double *cc = ab + kv + ldab * (j + 1 + c) - (1 + c);     // cc[i] = A(j+i, j+1+c)
    const double u = cc[0];
    if (u == 0.0) continue;
    for (int i = 1 + blockIdx.x * 256 + threadIdx.x; i <= km; i += gridDim.x * 256) cc[i] = fma(-l[i], u, cc[i]);
  }
}

// One launch per column (round 3; the two launches per column were the cost: 67 600 columns x 2 x 8 us at cfg4).
// State of column j (found by the launch before): pivot row jp, km, 1/pivot, ju; column j itself is final (swapped,
// multipliers scaled).  Launch j: every workgroup first applies the row interchange of column j to its trailing
// columns, then the rank-1 update -- exactly the operations of k_lu_pivot + k_lu_update in the same order, so the
// factors are bit for bit the same.  Workgroup (0, 0) owns column j+1: it brings it up to date FIRST, then searches its
// pivot, interchanges inside that column, scales the multipliers and writes the state of column j+1 (the interchange
// of the other columns is left to launch j+1).
__global__ void __launch_bounds__(256)
    k_lu_step(double *__restrict__ ab, int64_t ldab, int64_t n, int kl, int kv, int64_t j, int32_t *__restrict__ ipiv,
              tg_lu_state *st2) {
  __shared__ double sval[256];
  __shared__ int sidx[256];
  const tg_lu_state *st = st2 + (j & 1);
  tg_lu_state *nx = st2 + ((j + 1) & 1);
  const int tid = threadIdx.x;
  const int km = st->km, jp = st->jp, ju = st->ju;
  const bool live = st->pivinv != 0.0;             // (a zero pivot: LAPACK goes on without eliminating)
  const double *l = ab + kv + ldab * j;            // l[i], i = 1..km: multipliers of column j
  const int64_t nc = ju - j;                       // trailing columns j+1..ju
  const bool lead = blockIdx.x == 0 && blockIdx.y == 0;
  // the leader's column j+1 (c = 0) first, alone; the others (and the leader afterwards) share the columns c >= 1
  if (lead && nc >= 1) {
    double *cc = ab + kv + ldab * (j + 1) - 1;     // cc[i] = A(j+i, j+1)
    if (live && jp != 0 && tid == 0) {
      const double a = cc[jp], b = cc[0];
      cc[jp] = b;
      cc[0] = a;
    }
    __syncthreads();
    const double u = cc[0];
    if (live && u != 0.0)
      for (int i = 1 + tid; i <= km; i += 256) cc[i] = fma(-l[i], u, cc[i]);
    __syncthreads();
  }
  if (live)
    for (int64_t c = 1 + blockIdx.y; c < nc; c += gridDim.y) {   // (one workgroup per column: uniform trip count)
      double *cc = ab + kv + ldab * (j + 1 + c) - (1 + c);     // cc[i] = A(j+i, j+1+c)
      double u = cc[0];
      if (jp != 0) {
        const double ajp = cc[jp];
        __syncthreads();                             // everybody has read the two entries
        if (tid == 0) {
          cc[0] = ajp;
          cc[jp] = u;
        }
        u = ajp;
        __syncthreads();
      }
      if (u != 0.0)
        for (int i = 1 + tid; i <= km; i += 256) cc[i] = fma(-l[i], u, cc[i]);
    }
  if (!lead) return;
  // ---- column j+1: pivot search, interchange inside the column, multipliers, state
  const int64_t j1 = j + 1;
  if (j1 >= n) return;
  __syncthreads();
  const int km1 = (int)min((int64_t)kl, n - 1 - j1);
  double *cj = ab + kv + ldab * j1;                // cj[i] = A(j1+i, j1)
  double best = -1.0;
  int bi = 0;
  for (int i = tid; i <= km1; i += 256) {
    const double a = fabs(cj[i]);
    if (a > best || (a != a && best == best)) {
      best = a;
      bi = i;
    }
  }
  sval[tid] = best;
  sidx[tid] = bi;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (tid < o) {
      const double b2 = sval[tid + o];
      const int i2 = sidx[tid + o];
      if (b2 > sval[tid] || (b2 == sval[tid] && i2 < sidx[tid])) {
        sval[tid] = b2;
        sidx[tid] = i2;
      }
    }
    __syncthreads();
  }
  const int jp1 = sidx[0];
  const double piv = cj[jp1];
  __syncthreads();
  if (tid == 0) {
    ipiv[j1] = (int32_t)(j1 + jp1);
    nx->ju = max(ju, (int)min(j1 + (int64_t)(kv - kl) + jp1, n - 1));
    nx->km = km1;
    nx->jp = jp1;
    nx->info = (piv == 0.0 && st->info == 0) ? (int)(j1 + 1) : st->info;
    nx->pivinv = piv != 0.0 ? 1.0 / piv : 0.0;
    if (piv != 0.0 && jp1 != 0) {
      const double a = cj[jp1], b = cj[0];
      cj[jp1] = b;
      cj[0] = a;
    }
  }
  __syncthreads();
  if (piv == 0.0) return;
  const double pinv = 1.0 / piv;
  for (int i = 1 + tid; i <= km1; i += 256) cj[i] *= pinv;
}

// forward substitution with the row interchanges, then backward substitution; one workgroup
__global__ void __launch_bounds__(1024)
    k_lu_solve_global(const double *__restrict__ ab, int64_t ldab, int64_t n, int kl, int kv, const int32_t *__restrict__ ipiv,
               double *__restrict__ x) {
  const int tid = threadIdx.x;
  for (int64_t j = 0; j < n; j++) {
    const int km = (int)min((int64_t)kl, n - 1 - j);
    const int64_t jp = ipiv[j];
    __syncthreads();
    if (tid == 0 && jp != j) {
      const double t = x[jp];
      x[jp] = x[j];
      x[j] = t;
    }
    __syncthreads();
    const double xj = x[j];
    const double *l = ab + kv + ldab * j;
    for (int i = 1 + tid; i <= km; i += 1024) x[j + i] = fma(-l[i], xj, x[j + i]);
  }
  for (int64_t j = n - 1; j >= 0; j--) {
    __syncthreads();
    const double *cj = ab + kv + ldab * j;         // cj[-i] = U(j-i, j)
    if (tid == 0) x[j] = x[j] / cj[0];
    __syncthreads();
    const double xj = x[j];
    const int kk = (int)min((int64_t)kv, j);
    for (int i = 1 + tid; i <= kk; i += 1024) x[j - i] = fma(-cj[-i], xj, x[j - i]);
  }
}

// forward substitution with the row interchanges, then backward substitution; one workgroup.  The entries of x a step
// touches form a window of kl+1 (forward) / kv+1 (backward) consecutive entries that slides by one per step: the window
// lives in LDS as a ring (entry x[q] at q mod W), so a step costs one barrier and no global round trip -- the multipliers
// of the step after next are requested before the barrier (201 -> ~60 ms at cfg4: 67 600 steps each way).
// ------------------------------------------------------------------------------------------------------
// Blocked factorisation (round 3): NB columns per pair of launches instead of one launch per column.
//   k_lu_panel   ONE workgroup factorises the panel (rows j0 .. j0+NB-1+kl of the columns j0 .. j0+NB-1) in LDS: pivot
//                search, interchange, multipliers and the rank-1 updates INSIDE the panel, column by column;
//   k_lu_trail   one workgroup per trailing column (j0+NB .. ju): the panel's interchanges, then the NB eliminations, in
//                the panel's order, on the column held in LDS.
// Every entry receives the same operations in the same order as in the column-by-column algorithm (interchange and
// elimination of step jj, then of step jj+1, ...), so the factors are bit
// for bit the same.  Positions outside the band are never touched: a multiplier of column jj exists for the rows
// jj+1 .. jj+kl only, a trailing column holds the rows >= c - kv only -- no work arrays as in dgbtrf, the loops are
// bounded instead.
// barrier that orders the LDS traffic of the workgroup only: __syncthreads() also waits for every outstanding GLOBAL
// load, i.e. for the multipliers of the next step that are requested on purpose before they are needed
#define TG_LDS_BARRIER()                                            \
  do {                                                              \
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local"); \
    __builtin_amdgcn_s_barrier();                                   \
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local"); \
  } while (0)
// copy of the panel between the band storage and LDS: four columns at a time (four independent loads in flight)
template <bool LOAD>
__device__ __forceinline__ void tg_lu_panel_copy(double *__restrict__ ab, double *P, int64_t ldab, int64_t n, int kl, int kv,
                                                 int64_t j0, int nbc, int H, int tid) {
  const int rmax = (int)min((int64_t)H - 1, n - 1 - j0);
  for (int lc0 = 0; lc0 < nbc; lc0 += 4)
    for (int lr = tid; lr <= rmax; lr += 1024) {
      double v[4];
      bool ok[4];
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const int lc = lc0 + q;
        ok[q] = lc < nbc && lr >= lc - kv && lr <= lc + kl;            // (rows above the band are not stored)
        double *g = ab + kv + lr - lc + ldab * (j0 + lc);              // A(j0 + lr, j0 + lc)
        if (LOAD) v[q] = ok[q] ? *g : 0.0;
        else if (ok[q]) *g = P[lc * H + lr];
      }
      if (LOAD) {
#pragma unroll
        for (int q = 0; q < 4; q++)
          if (ok[q]) P[(lc0 + q) * H + lr] = v[q];
      }
    }
}

// wave-wide maximum of a 64-bit key / minimum of a 32-bit value through DPP (row shifts, then the row broadcasts of
// GFX9); every lane receives the result
__device__ __forceinline__ unsigned long long tg_wave_max_u64(unsigned long long v) {
#define TG_DPP_STEP(ctrl, rmask)                                                                      \
  {                                                                                                   \
    const int lo = __builtin_amdgcn_update_dpp((int)v, (int)v, ctrl, rmask, 0xF, false);              \
    const int hi = __builtin_amdgcn_update_dpp((int)(v >> 32), (int)(v >> 32), ctrl, rmask, 0xF, false); \
    const unsigned long long o = ((unsigned long long)(unsigned)hi << 32) | (unsigned)lo;             \
    v = o > v ? o : v;                                                                                \
  }
  TG_DPP_STEP(0x111, 0xF)   // row_shr:1
  TG_DPP_STEP(0x112, 0xF)   // row_shr:2
  TG_DPP_STEP(0x114, 0xF)   // row_shr:4
  TG_DPP_STEP(0x118, 0xF)   // row_shr:8
  TG_DPP_STEP(0x142, 0xA)   // row_bcast:15 into rows 1, 3
  TG_DPP_STEP(0x143, 0xC)   // row_bcast:31 into rows 2, 3
#undef TG_DPP_STEP
  const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)v, 63);
  const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(v >> 32), 63);
  return ((unsigned long long)hi << 32) | lo;
}
__device__ __forceinline__ int tg_wave_min_i32(int v) {
#define TG_DPP_STEP(ctrl, rmask)                                                   \
  {                                                                                \
    const int o = __builtin_amdgcn_update_dpp(v, v, ctrl, rmask, 0xF, false);      \
    v = o < v ? o : v;                                                             \
  }
  TG_DPP_STEP(0x111, 0xF)
  TG_DPP_STEP(0x112, 0xF)
  TG_DPP_STEP(0x114, 0xF)
  TG_DPP_STEP(0x118, 0xF)
  TG_DPP_STEP(0x142, 0xA)
  TG_DPP_STEP(0x143, 0xC)
#undef TG_DPP_STEP
  return __builtin_amdgcn_readlane(v, 63);
}

// One workgroup on one CU: its four waves per SIMD share the issue slots, so an instruction that all sixteen waves
// execute costs sixteen cycles -- the kernel is bound by the instructions of a step, not by LDS or arithmetic.  Hence:
//   search    ONE wave scans the column (a contiguous chunk per lane, so that the first lane holding the maximum holds its
//             first occurrence), reduces through DPP and publishes pivot row, pivot, A(j, j) and the reciprocal;
//   interchange + multipliers   one thread per column to the right / per pair of rows;
//   rank-1 update               a thread keeps its two multipliers and walks the columns.
// Three barriers per step, all of them waiting for LDS only (__syncthreads() would wait for global stores as well);
// the pivot indices go to global memory at the end.
__global__ void __launch_bounds__(1024)
    k_lu_panel(double *__restrict__ ab, int64_t ldab, int64_t n, int kl, int kv, int64_t j0, int nb,
               int32_t *__restrict__ ipiv, tg_lu_state *st) {
  extern __shared__ double P[];                    // [nb][H], H = kl + nb: P[lc * H + lr] = A(j0 + lr, j0 + lc)
  __shared__ double sh_piv[3];                     // pivot, A(j, j) before the interchange, 1 / pivot
  __shared__ int sh_jp;
  __shared__ int spiv[32];
  const int tid = threadIdx.x;
  const int H = kl + nb;
  const int nbc = (int)min((int64_t)nb, n - j0);
  const int ku = kv - kl;
  int ju = st->ju, info = st->info;                // (every thread carries the same copy)
  tg_lu_panel_copy<true>(ab, P, ldab, n, kl, kv, j0, nbc, H, tid);
  __syncthreads();
  for (int jj = 0; jj < nbc; jj++) {
    const int64_t j = j0 + jj;
    const int km = (int)min((int64_t)kl, n - 1 - j);
    double *cj = P + jj * H + jj;                  // cj[i] = A(j + i, j)
    if (tid < 64) {
      // first maximum of |cj[0..km]|: the bit pattern of |a| orders like |a| (and a NaN above everything, so that it
      // surfaces); ties go to the smallest index
      const int chunk = (km + 64) >> 6;
      const int lo = tid * chunk, hi = min(lo + chunk - 1, km);
      unsigned long long key = 0;
      int bi = -1;
      for (int i8 = lo; i8 <= hi; i8 += 8) {        // eight independent LDS reads in flight
        double v[8];
#pragma unroll
        for (int q = 0; q < 8; q++) v[q] = i8 + q <= hi ? cj[i8 + q] : 0.0;
#pragma unroll
        for (int q = 0; q < 8; q++) {
          const unsigned long long k2 = (unsigned long long)__double_as_longlong(fabs(v[q]));
          if (i8 + q <= hi && (k2 > key || bi < 0)) {
            key = k2;
            bi = i8 + q;
          }
        }
      }
      const unsigned long long wmax = tg_wave_max_u64(key);
      const unsigned long long holders = __ballot(bi >= 0 && key == wmax);
      const int first = __builtin_amdgcn_readfirstlane(__ffsll((long long)holders) - 1);
      const int jp = __shfl(bi, first);
      if (tid == 0) {
        const double piv = cj[jp];
        sh_jp = jp;
        sh_piv[0] = piv;
        sh_piv[1] = cj[0];
        sh_piv[2] = piv != 0.0 ? 1.0 / piv : 0.0;
        spiv[jj] = (int)(j + jp);
      }
    }
    TG_LDS_BARRIER();
    const int jp = sh_jp;
    const double piv = sh_piv[0];
    ju = max(ju, (int)min(j + (int64_t)ku + jp, n - 1));
    if (piv == 0.0 && info == 0) info = (int)(j + 1);
    const int lcl = (int)min((int64_t)(nbc - 1), (int64_t)ju - j0);     // last panel column the step touches
    const int ncol = piv != 0.0 ? lcl - jj : 0;    // (a zero pivot: LAPACK goes on without eliminating)
    if (piv != 0.0) {                              // (uniform)
      const double a0 = sh_piv[1], pinv = sh_piv[2];
      if (jp != 0) {
        if (tid < ncol) {                          // interchange in the columns to the right
          double *q = P + (jj + 1 + tid) * H + jj;
          const double a = q[jp], b = q[0];
          q[jp] = b;
          q[0] = a;
        }
        if (tid == 0) cj[0] = piv;                 // ... and in column jj (row jp: below, through a0)
      }
      for (int i0 = 1 + 2 * tid; i0 <= km; i0 += 2048) {               // multipliers: a pair of rows per thread
        const int i1 = i0 + 1;
        cj[i0] = (i0 == jp ? a0 : cj[i0]) * pinv;
        if (i1 <= km) cj[i1] = (i1 == jp ? a0 : cj[i1]) * pinv;
      }
    }
    TG_LDS_BARRIER();
    if (ncol > 0)
      for (int i0 = 1 + 2 * tid; i0 <= km; i0 += 2048) {               // rank-1 update of the columns jj+1 .. lcl
        const bool h1 = i0 + 1 <= km;
        const double l0 = cj[i0], l1 = h1 ? cj[i0 + 1] : 0.0;
        double *cc = P + (jj + 1) * H + jj;        // cc[i] = A(j + i, j + 1 + c)
        int c = 0;
        for (; c + 4 <= ncol; c += 4, cc += 4 * H) {
          double u[4], x0[4], x1[4];
#pragma unroll
          for (int q = 0; q < 4; q++) {
            u[q] = cc[q * H];
            x0[q] = cc[q * H + i0];
            x1[q] = h1 ? cc[q * H + i0 + 1] : 0.0;
          }
#pragma unroll
          for (int q = 0; q < 4; q++)
            if (u[q] != 0.0) {
              cc[q * H + i0] = fma(-l0, u[q], x0[q]);
              if (h1) cc[q * H + i0 + 1] = fma(-l1, u[q], x1[q]);
            }
        }
        for (; c < ncol; c++, cc += H) {
          const double u = cc[0];
          if (u != 0.0) {
            cc[i0] = fma(-l0, u, cc[i0]);
            if (h1) cc[i0 + 1] = fma(-l1, u, cc[i0 + 1]);
          }
        }
      }
    TG_LDS_BARRIER();
  }
  tg_lu_panel_copy<false>(ab, P, ldab, n, kl, kv, j0, nbc, H, tid);
  if (tid < nbc) ipiv[j0 + tid] = spiv[tid];
  if (tid == 0) {
    st->ju = ju;
    st->info = info;
  }
}

// The panel in REGISTERS (round 3, after the LDS panel above proved bound by LDS bandwidth and by the issue slots that
// sixteen waves share; cfg4: 54 us per 16 columns in LDS, 39 us here): NT threads, thread t owns the panel rows t, t + NT,
// ... (RS of them) of all NB columns.  A step:
// every thread proposes the largest of its own candidates, the waves reduce through DPP, four partial results meet in
// LDS | the owner of the pivot row and the owner of row j publish their rows (columns j .. ) | everyone reads them,
// takes part in the interchange if it owns one of the two rows, scales its multipliers and updates its own entries.
// Two barriers per step, LDS traffic of a few hundred bytes, the update is plain FMAs on registers.  All loops are
// unrolled (the register file is indexed statically).
// a value that is the same in every lane, moved to scalar registers
__device__ __forceinline__ double tg_uniform_f64(double v) {
  const int lo = __builtin_amdgcn_readfirstlane(__double2loint(v)), hi = __builtin_amdgcn_readfirstlane(__double2hiint(v));
  return __hiloint2double(hi, lo);
}
struct tg_lu_panel_ctx {
  unsigned long long *wkey;
  int *widx;
  double *urow, *r0row;
  int *spiv;
  int64_t n, j0;
  int kl, ku, nbc, tid, ju, info;
};
// step JJ of the register panel (a template recursion: the register file is indexed statically)
template <int NB, int RS, int NT, int JJ>
__device__ __forceinline__ void tg_lu_panel_step(double (&a)[RS][NB], tg_lu_panel_ctx &X) {
  if constexpr (JJ < NB) {
    if (JJ >= X.nbc) return;                       // (uniform)
    const int tid = X.tid;
    const int64_t j = X.j0 + JJ;
    const int km = (int)min((int64_t)X.kl, X.n - 1 - j);
    // first maximum of |A(j .. j + km, j)|: the bit pattern of |a| orders like |a| (and a NaN above everything, so that it
    // surfaces); ties go to the smallest row
    unsigned long long key = 0;
    int bi = 0x7fffffff;
#pragma unroll
    for (int s = 0; s < RS; s++) {
      const int lr = tid + NT * s;
      const unsigned long long k2 = (unsigned long long)__double_as_longlong(fabs(a[s][JJ]));
      if (lr >= JJ && lr <= JJ + km && (k2 > key || bi == 0x7fffffff)) {
        key = k2;
        bi = lr - JJ;
      }
    }
    const unsigned long long wmax = tg_wave_max_u64(key);
    const int wi = tg_wave_min_i32(key == wmax ? bi : 0x7fffffff);
    if ((tid & 63) == 0) {
      X.wkey[tid >> 6] = wmax;
      X.widx[tid >> 6] = wi;
    }
    TG_LDS_BARRIER();
    unsigned long long bk = X.wkey[0];
    int jp = X.widx[0];
#pragma unroll
    for (int w = 1; w < NT / 64; w++) {
      const unsigned long long k2 = X.wkey[w];
      const int i2 = X.widx[w];
      if (k2 > bk || (k2 == bk && i2 < jp)) {
        bk = k2;
        jp = i2;
      }
    }
    // everything below that is the same in every lane is moved to scalar registers, so that the branches on it are
    // scalar branches and only the code of the slot that owns a row is executed (the kernel is bound by its instruction
    // count: straight-line code, executed once)
    jp = __builtin_amdgcn_readfirstlane(jp);
    const int lrp = JJ + jp;                       // the pivot row (relative to j0)
    const int sp = lrp / NT, tp = lrp - sp * NT;
#pragma unroll
    for (int s = 0; s < RS; s++)
      if (sp == s) {                               // (scalar branch: the asm keeps it from becoming selects)
        asm volatile("" ::: "memory");
        if (tid == tp) {
#pragma unroll
          for (int c = JJ; c < NB; c++) X.urow[c] = a[s][c];
        }
      }
    if (tid == JJ) {
#pragma unroll
      for (int c = JJ; c < NB; c++) X.r0row[c] = a[0][c];
      X.spiv[JJ] = (int)(j + jp);
    }
    TG_LDS_BARRIER();
    const double piv = tg_uniform_f64(X.urow[JJ]);
    X.ju = max(X.ju, (int)min(j + (int64_t)X.ku + jp, X.n - 1));
    if (piv == 0.0) {                              // (uniform) LAPACK goes on without eliminating
      if (X.info == 0) X.info = (int)(j + 1);
    } else {
      const int lcl = __builtin_amdgcn_readfirstlane((int)min((int64_t)(X.nbc - 1), (int64_t)X.ju - X.j0));    // last panel column the step touches
      const double pinv = 1.0 / piv;
      double uc[NB];
#pragma unroll
      for (int c = JJ + 1; c < NB; c++) uc[c] = tg_uniform_f64(X.urow[c]);
      if (jp != 0) {                               // interchange of the rows JJ and lrp, columns JJ .. (past lcl: zeros)
        asm volatile("" ::: "memory");
        if (tid == JJ) {
          a[0][JJ] = piv;
#pragma unroll
          for (int c = JJ + 1; c < NB; c++) a[0][c] = uc[c];
        }
#pragma unroll
        for (int s = 0; s < RS; s++)
          if (sp == s) {
            asm volatile("" ::: "memory");
            if (tid == tp) {
#pragma unroll
              for (int c = JJ; c < NB; c++) a[s][c] = X.r0row[c];
            }
          }
      }
      // multipliers; a row outside JJ+1 .. JJ+km gets the multiplier 0: for the rows past the band (nothing stored, zeros
      // that are never written back) the update below is then an exact no-op, the rows <= JJ (slot 0 only) are protected
      double l[RS];
#pragma unroll
      for (int s = 0; s < RS; s++) {
        const int lr = tid + NT * s;
        const bool inr = lr > JJ && lr <= JJ + km;
        l[s] = inr ? a[s][JJ] * pinv : 0.0;
        a[s][JJ] = inr ? l[s] : a[s][JJ];
      }
#pragma unroll
      for (int c = JJ + 1; c < NB; c++)
        if (c <= lcl && uc[c] != 0.0) {            // (uniform: a scalar branch around RS FMAs)
          asm volatile("" ::: "memory");
          a[0][c] = tid > JJ ? fma(-l[0], uc[c], a[0][c]) : a[0][c];
#pragma unroll
          for (int s = 1; s < RS; s++) a[s][c] = fma(-l[s], uc[c], a[s][c]);
        }
    }
    tg_lu_panel_step<NB, RS, NT, JJ + 1>(a, X);
  }
}

template <int NB, int RS, int NT>
__global__ void __launch_bounds__(NT)
    k_lu_panel_reg(double *__restrict__ ab, int64_t ldab, int64_t n, int kl, int kv, int64_t j0, int32_t *__restrict__ ipiv,
                   tg_lu_state *st) {
  __shared__ unsigned long long wkey[NT / 64];
  __shared__ int widx[NT / 64];
  __shared__ double urow[NB], r0row[NB];           // the pivot row / row j, columns j .. , before the interchange
  __shared__ int spiv[NB];
  const int tid = threadIdx.x;
  const int nbc = (int)min((int64_t)NB, n - j0);
  const int rmax = (int)min((int64_t)kl + NB - 1, n - 1 - j0);         // last row of the panel (relative to j0)
  double a[RS][NB];                                // a[s][c] = A(j0 + tid + NT s, j0 + c); 0 where nothing is stored
#pragma unroll
  for (int s = 0; s < RS; s++) {
    const int lr = tid + NT * s;
#pragma unroll
    for (int c = 0; c < NB; c++) {
      const bool ok = c < nbc && lr <= rmax && lr >= c - kv && lr <= c + kl;
      a[s][c] = ok ? ab[kv + lr - c + ldab * (j0 + c)] : 0.0;
    }
  }
  tg_lu_panel_ctx X;
  X.wkey = wkey;
  X.widx = widx;
  X.urow = urow;
  X.r0row = r0row;
  X.spiv = spiv;
  X.n = n;
  X.j0 = j0;
  X.kl = kl;
  X.ku = kv - kl;
  X.nbc = nbc;
  X.tid = tid;
  X.ju = st->ju;                                   // (every thread carries the same copy)
  X.info = st->info;
  tg_lu_panel_step<NB, RS, NT, 0>(a, X);
#pragma unroll
  for (int s = 0; s < RS; s++) {
    const int lr = tid + NT * s;
#pragma unroll
    for (int c = 0; c < NB; c++) {
      const bool ok = c < nbc && lr <= rmax && lr >= c - kv && lr <= c + kl;
      if (ok) ab[kv + lr - c + ldab * (j0 + c)] = a[s][c];
    }
  }
  __syncthreads();
  if (tid < nbc) ipiv[j0 + tid] = spiv[tid];
  if (tid == 0) {
    st->ju = X.ju;
    st->info = X.info;
  }
}

// Trailing columns: a workgroup takes TG_LU_TC adjacent columns (in LDS) through the nb steps of the panel; the
// multipliers of a step are fetched a step ahead, with the pivot and the interchange of the step.  Measured at cfg4
// (kl = 1044, nb = 16): 22 us with one column per workgroup, 24 / 32 us with two / four (the columns in LDS make the
// steps LDS-bandwidth bound, 16 B per FMA, whatever the grouping; holding the multipliers of all steps in registers
// halved the occupancy and was slower still).  NP > 0: a thread owns NP pairs of adjacent rows (kl <= 512 NP); NP = 0:
// any kl, multipliers read where they are used.
#define TG_LU_TC 1
template <int NP>
__global__ void __launch_bounds__(256)
    k_lu_trail(double *__restrict__ ab, int64_t ldab, int64_t n, int kl, int kv, int64_t j0, int nb,
               const int32_t *__restrict__ ipiv, const tg_lu_state *__restrict__ st) {
  extern __shared__ double col[];                  // [TG_LU_TC][W]: col[t * W + r - j0], rows j0 .. j0 + nb - 1 + kl
  const int tid = threadIdx.x;
  const int W = kl + nb;
  const int nbc = (int)min((int64_t)nb, n - j0);
  const int64_t ju = st->ju;
  constexpr int NR = NP > 0 ? NP : 1;
  const int64_t rhi = min(n - 1, j0 + nbc - 1 + (int64_t)kl);
  for (int64_t c0 = j0 + nbc + (int64_t)TG_LU_TC * blockIdx.x; c0 <= ju; c0 += (int64_t)TG_LU_TC * gridDim.x) {
    __syncthreads();
    // rows above a column's band (r < c - kv) and columns past ju enter as zeros and are never written back
    for (int64_t r = j0 + tid; r <= rhi; r += 256) {
      double v[TG_LU_TC];
#pragma unroll
      for (int t = 0; t < TG_LU_TC; t++) {
        const int64_t c = c0 + t;
        v[t] = (c <= ju && r >= c - kv) ? ab[kv + r - c + ldab * c] : 0.0;
      }
#pragma unroll
      for (int t = 0; t < TG_LU_TC; t++) col[t * W + (r - j0)] = v[t];
    }
    const int jj0 = (int)max((int64_t)0, c0 - kv - j0);                // steps before: row j is above every band of the group
    double la[NR], lb[NR], d0n = 0.0;
    int64_t r2n = 0;
    if (NP > 0 && jj0 < nbc) {
      const int64_t j = j0 + jj0;
      const double *lj = ab + kv + ldab * j;
      const int km = (int)min((int64_t)kl, n - 1 - j);
      d0n = lj[0];
      r2n = ipiv[j];
#pragma unroll
      for (int q = 0; q < NR; q++) {
        const int i = 1 + 2 * tid + 512 * q;
        la[q] = i <= km ? lj[i] : 0.0;
        lb[q] = i + 1 <= km ? lj[i + 1] : 0.0;
      }
    }
    __syncthreads();
    for (int jj = jj0; jj < nbc; jj++) {
      const int64_t j = j0 + jj;
      const double *lj = ab + kv + ldab * j;       // lj[i] = L(j + i, j); lj[0] = U(j, j)
      const int km = (int)min((int64_t)kl, n - 1 - j);
      double ca[NR], cb[NR], d0;
      int64_t r2;
      if (NP > 0) {
        d0 = d0n;
        r2 = r2n;
#pragma unroll
        for (int q = 0; q < NR; q++) {
          ca[q] = la[q];
          cb[q] = lb[q];
        }
        if (jj + 1 < nbc) {
          const double *l1 = lj + ldab;
          const int km1 = (int)min((int64_t)kl, n - 2 - j);
          d0n = l1[0];
          r2n = ipiv[j + 1];
#pragma unroll
          for (int q = 0; q < NR; q++) {
            const int i = 1 + 2 * tid + 512 * q;
            la[q] = i <= km1 ? l1[i] : 0.0;
            lb[q] = i + 1 <= km1 ? l1[i + 1] : 0.0;
          }
        }
      } else {
        d0 = lj[0];
        r2 = ipiv[j];
      }
      if (d0 == 0.0) continue;                     // zero pivot: the column was neither interchanged nor eliminated
      // the interchange of step jj comes AFTER the eliminations of the steps before it (the multipliers of the columns
      // to the left are stored un-interchanged, as dgbtf2 leaves them), so the two cannot be separated
      if (r2 != j) {
        if (tid < TG_LU_TC) {
          double *q = col + tid * W;
          const double a = q[r2 - j0], b = q[jj];
          q[r2 - j0] = b;
          q[jj] = a;
        }
        TG_LDS_BARRIER();
      }
      double u[TG_LU_TC];
      bool any = false;
#pragma unroll
      for (int t = 0; t < TG_LU_TC; t++) {
        u[t] = col[t * W + jj];
        any = any || u[t] != 0.0;
      }
      if (any) {                                   // (uniform)
        if (NP > 0) {
#pragma unroll
          for (int q = 0; q < NR; q++) {
            const int i = 1 + 2 * tid + 512 * q;
            if (i + 1 <= km) {
#pragma unroll
              for (int t = 0; t < TG_LU_TC; t++)
                if (u[t] != 0.0) {
                  double *x = col + t * W + jj + i;
                  const double x0 = x[0], x1 = x[1];
                  x[0] = fma(-ca[q], u[t], x0);
                  x[1] = fma(-cb[q], u[t], x1);
                }
            } else if (i <= km) {
#pragma unroll
              for (int t = 0; t < TG_LU_TC; t++)
                if (u[t] != 0.0) col[t * W + jj + i] = fma(-ca[q], u[t], col[t * W + jj + i]);
            }
          }
        } else {
          for (int i = 1 + tid; i <= km; i += 256) {
            const double l = lj[i];
#pragma unroll
            for (int t = 0; t < TG_LU_TC; t++)
              if (u[t] != 0.0) col[t * W + jj + i] = fma(-l, u[t], col[t * W + jj + i]);
          }
        }
      }
      TG_LDS_BARRIER();                            // (LDS only: the multipliers of the next step stay in flight)
    }
    for (int64_t r = j0 + tid; r <= rhi; r += 256) {
#pragma unroll
      for (int t = 0; t < TG_LU_TC; t++) {
        const int64_t c = c0 + t;
        if (c <= ju && r >= c - kv) ab[kv + r - c + ldab * c] = col[t * W + (r - j0)];
      }
    }
  }
}

#define TG_LU_CH 1024        // entries of x that enter the window at a time
// ring slot of entry (base entry at slot b) + off, 0 <= off < W   (32-bit: a 64-bit modulo per access costs more than the step)
__device__ __forceinline__ int tg_slot(int b, int off, int W) {
  const int q = b + off;
  return q >= W ? q - W : q;
}
template <int R, int NT>
__device__ __forceinline__ void tg_lu_load_col(const double *__restrict__ col, int sign, int cnt, int tid, double (&v)[R]) {
#pragma unroll
  for (int q = 0; q < R; q++) {
    const int i = 1 + tid + NT * q;
    v[q] = col[sign * (i <= cnt ? i : 0)];         // (unconditional: a branch around a load makes the compiler wait for
  }                                                //  ALL outstanding loads at the next use; entries past cnt are not used)
}

// Substitution with the window of x that a step touches in an LDS ring (one workgroup of NT threads; entries enter
// TG_LU_CH at a time).  A step is one LDS-only barrier (two for a forward step with an interchange) and R guarded FMAs per thread:
// the interchange of a forward step is not carried out first -- every thread reads x[j + jp] as the value of the step,
// and the owner of entry j + jp continues from the old x[j].  The entries of a column (and its pivot) are requested D steps before their use, into the register
// slot the step D before has just emptied.  R = entries of a column per thread (windows of up to R * NT entries).
template <int R, int D, int NT>
__global__ void __launch_bounds__(NT)
    k_lu_fwd(const double *__restrict__ ab, int64_t ldab, int64_t n, int kl, int kv, const int32_t *__restrict__ ipiv,
             double *__restrict__ x) {
  extern __shared__ double ring[];                 // W doubles; entry x[q] lives at slot q mod W
  const int tid = threadIdx.x;
  const int W = kl + 2 + TG_LU_CH;
  // the window [j, j + kl] lies inside the loaded range [j, top); `top` grows by TG_LU_CH entries at a time
  int64_t top = min(n, (int64_t)kl + 1 + TG_LU_CH);
  for (int64_t q = tid; q < top; q += NT) ring[(int)(q % W)] = x[q];
  int b = 0;                                       // slot of entry j
  int tb = (int)(top % W);                         // slot of entry top
  double L[D][R];
  int JP[D];                                       // pivot offsets jp - j
#pragma unroll
  for (int k = 0; k < D; k++) {
    JP[k] = 0;
    if (k < n) {
      JP[k] = (int)(ipiv[k] - k);
      tg_lu_load_col<R, NT>(ab + kv + ldab * k, 1, (int)min((int64_t)kl, n - 1 - k), tid, L[k]);
    }
  }
  __syncthreads();
  for (int64_t j0 = 0; j0 < n; j0 += D) {
#pragma unroll
    for (int k = 0; k < D; k++) {
      const int64_t j = j0 + k;
      if (j < n) {                                 // (uniform)
        const int km = (int)min((int64_t)kl, n - 1 - j);
        const int jp = JP[k];
        const double xj = ring[tg_slot(b, jp, W)], r0 = ring[b];
        if (jp != 0) TG_LDS_BARRIER();             // (uniform) every thread has x[j + jp] before its owner overwrites it
#pragma unroll
        for (int q = 0; q < R; q++) {
          const int i = 1 + tid + NT * q;
          if (i <= km) {
            const int sl = tg_slot(b, i, W);
            const double v = ring[sl];
            ring[sl] = fma(-L[k][q], xj, i == jp ? r0 : v);
          }
        }
        if (tid == 0) x[j] = xj;
        {                                          // the slot is free: column j + D (past the end: the last column again)
          const int64_t j2 = min(j + D, n - 1);
          JP[k] = (int)(ipiv[j2] - j2);
          tg_lu_load_col<R, NT>(ab + kv + ldab * j2, 1, (int)min((int64_t)kl, n - 1 - j2), tid, L[k]);
        }
        if (top < n && j + 1 + kl + 1 > top) {
          const int64_t e1 = min(n, top + TG_LU_CH);
          for (int64_t e = top + tid; e < e1; e += NT) ring[tg_slot(tb, (int)(e - top), W)] = x[e];
          tb = tg_slot(tb, (int)(e1 - top), W);
          top = e1;
        }
        TG_LDS_BARRIER();
        b = tg_slot(b, 1, W);
      }
    }
  }
}

template <int R, int D, int NT>
__global__ void __launch_bounds__(NT)
    k_lu_bwd(const double *__restrict__ ab, int64_t ldab, int64_t n, int kv, double *__restrict__ x) {
  extern __shared__ double ring[];
  const int tid = threadIdx.x;
  const int W = kv + 2 + TG_LU_CH;
  // the window [j - kv, j] lies inside the loaded range [lo, j]
  int64_t lo = max((int64_t)0, n - 1 - kv - TG_LU_CH);
  for (int64_t q = lo + tid; q <= n - 1; q += NT) ring[(int)(q % W)] = x[q];
  int b = (int)((n - 1) % W);                      // slot of entry j
  int lb = (int)(lo % W);                          // slot of entry lo
  double U[D][R], DG[D];
#pragma unroll
  for (int k = 0; k < D; k++) {
    DG[k] = 1.0;
    const int64_t j = n - 1 - k;
    if (j >= 0) {
      const double *cj = ab + kv + ldab * j;
      tg_lu_load_col<R, NT>(cj, -1, (int)min((int64_t)kv, j), tid, U[k]);
      DG[k] = cj[0];
    }
  }
  __syncthreads();
  for (int64_t j0 = n - 1; j0 >= 0; j0 -= D) {
#pragma unroll
    for (int k = 0; k < D; k++) {
      const int64_t j = j0 - k;
      if (j >= 0) {                                // (uniform)
        const int kk = (int)min((int64_t)kv, j);
        const double xj = ring[b] / DG[k];
#pragma unroll
        for (int q = 0; q < R; q++) {
          const int i = 1 + tid + NT * q;
          if (i <= kk) {
            const int sl = tg_slot(b, W - i, W);
            ring[sl] = fma(-U[k][q], xj, ring[sl]);
          }
        }
        if (tid == 0) x[j] = xj;
        {
          const int64_t j2 = max(j - D, (int64_t)0);
          const double *c2 = ab + kv + ldab * j2;
          tg_lu_load_col<R, NT>(c2, -1, (int)min((int64_t)kv, j2), tid, U[k]);
          DG[k] = c2[0];
        }
        if (lo > 0 && lo > j - 1 - kv) {
          const int64_t e0 = max((int64_t)0, lo - TG_LU_CH);
          const int cnt = (int)(lo - e0);
          const int nb_ = tg_slot(lb, W - cnt, W);
          for (int e = tid; e < cnt; e += NT) ring[tg_slot(nb_, e, W)] = x[e0 + e];
          lb = nb_;
          lo = e0;
        }
        TG_LDS_BARRIER();
        b = tg_slot(b, W - 1, W);
      }
    }
  }
}

extern "C" int tg_lu_band_info(tg_csr_t k, int *kl_out, int *ku_out, int64_t *bytes_out) {
  TG_REQUIRE_INIT();
  TG_REQUIRE(k && kl_out && ku_out && bytes_out, "null argument to tg_lu_band_info");
  TG_REQUIRE_CANONICAL(k);
  TG_REQUIRE(k->nrows == k->ncols, "tg_lu_band_info: square matrix expected");
  int *d = (int *)(g_tg.scratch);
  TG_CHECK_HIP(hipMemsetAsync(d, 0, 2 * sizeof(int), g_tg.stream));
  if (k->nrows > 0)
    hipLaunchKernelGGL(k_lu_bandwidth, dim3((unsigned)tg_cdiv(k->nrows, 256)), dim3(256), 0, g_tg.stream, k->rowptr, k->col,
                       k->nrows, d, d + 1);
  TG_LAUNCH_CHECK();
  int h[2];
  TG_CHECK_HIP(hipMemcpyAsync(h, d, 2 * sizeof(int), hipMemcpyDeviceToHost, g_tg.stream));
  TG_CHECK_HIP(hipStreamSynchronize(g_tg.stream));
  *kl_out = h[0];
  *ku_out = h[1];
  *bytes_out = (int64_t)(2 * (int64_t)h[0] + h[1] + 1) * k->nrows * (int64_t)sizeof(double);
  return 0;
}

// status 0 = solved; info > 0: U(info-1, info-1) is exactly zero (the matrix is singular to working precision)
extern "C" int tg_lu_solve(tg_csr_t k, tg_vec_t b, tg_vec_t x, int *info) {
  TG_REQUIRE_INIT();
  TG_REQUIRE(k && b && x && info, "null argument to tg_lu_solve");
  TG_REQUIRE_CANONICAL(k);
  const int64_t n = k->nrows;
  TG_REQUIRE(k->ncols == n && b->n == n && x->n == n, "tg_lu_solve: square system with matching vectors expected");
  *info = 0;
  if (n == 0) return 0;
  int kl = 0, ku = 0;
  int64_t bytes = 0;
  TG_TRY(tg_lu_band_info(k, &kl, &ku, &bytes));
  {   // a symmetric positive definite system: blocked banded Cholesky (tg_chol.hip); otherwise -- or TIGAR_LU_CHOLESKY=0 -- the LU
    int done = 0;
    TG_TRY(tg_chol_try(k, kl, ku, b->d, x->d, &done));
    g_tg.prof_n[TG_PROF_LU_CHOLESKY] += done;
    if (done) return 0;
  }
  const int kv = kl + ku;
  const int64_t ldab = 2 * (int64_t)kl + ku + 1;
  double *ab = nullptr;
  int32_t *ipiv = nullptr;
  tg_lu_state *st = nullptr;
  bool blocked_used = false;
  int rc = tg_dmalloc(&ab, ldab * n);
  if (!rc) rc = tg_dmalloc(&ipiv, n);
  if (!rc) rc = tg_dmalloc_bytes((void **)&st, 2 * sizeof(tg_lu_state));
  if (!rc && hipMemsetAsync(ab, 0, (size_t)(ldab * n) * sizeof(double), g_tg.stream) != hipSuccess) rc = 1;
  if (!rc) {
    tg_lu_state h0;
    h0.ju = 0;
    h0.info = 0;
    h0.km = 0;
    h0.jp = 0;
    h0.pivinv = 0.0;
    if (hipMemcpyAsync(st, &h0, sizeof(h0), hipMemcpyHostToDevice, g_tg.stream) != hipSuccess) rc = 1;
    if (!rc && hipMemcpyAsync(st + 1, &h0, sizeof(h0), hipMemcpyHostToDevice, g_tg.stream) != hipSuccess) rc = 1;
    hipStreamSynchronize(g_tg.stream);
  }
  if (!rc) {
    hipLaunchKernelGGL(k_lu_scatter, dim3((unsigned)std::min<int64_t>(tg_cdiv(n, 4), (int64_t)g_tg.num_cu * 16)), dim3(256), 0,
                       g_tg.stream, k->rowptr, k->col, k->val, n, kv, ldab, ab);
    // column 0: pivot search alone (its state lands in slot 0); then ONE launch per column: launch j interchanges and
    // eliminates with column j and prepares column j+1 (TIGAR_LU_FUSED=0: the two launches per column of round 2)
    const int fused = getenv("TIGAR_LU_FUSED") ? atoi(getenv("TIGAR_LU_FUSED")) : 1;
    const int blocked = getenv("TIGAR_LU_BLOCKED") ? atoi(getenv("TIGAR_LU_BLOCKED")) : 1;
    // panel width: the widest of 16 / 8 / 4 columns whose panel ((kl + nb) x nb doubles) fits in LDS
    int nb = 0;
    for (int cand = 16; cand >= 4 && !nb; cand >>= 1)
      if ((size_t)(kl + cand) * cand * sizeof(double) <= 140 * 1024) nb = cand;
    if (getenv("TIGAR_LU_NB")) nb = std::min(32, atoi(getenv("TIGAR_LU_NB")));
    bool use_blocked = blocked && nb >= 2 && kl > 0;
    typedef void (*trail_fn)(double *, int64_t, int64_t, int, int, int64_t, int, const int32_t *, const tg_lu_state *);
    const trail_fn trail = kl <= 512 ? k_lu_trail<1> : kl <= 1024 ? k_lu_trail<2> : kl <= 1536 ? k_lu_trail<3>
                           : kl <= 2048 ? k_lu_trail<4> : k_lu_trail<0>;
    typedef void (*panel_reg_fn)(double *, int64_t, int64_t, int, int, int64_t, int32_t *, tg_lu_state *);
    panel_reg_fn panel_reg = nullptr;              // (TIGAR_LU_PANEL_REG=0: the panel in LDS)
    int panel_nt = 0;
    if (!(getenv("TIGAR_LU_PANEL_REG") && atoi(getenv("TIGAR_LU_PANEL_REG")) == 0)) {
      const int rows = kl + nb;                    // rows of the panel: NT threads x RS rows each
      if (nb == 16) {
        if (rows <= 256) panel_reg = k_lu_panel_reg<16, 1, 256>, panel_nt = 256;
        else if (rows <= 512) panel_reg = k_lu_panel_reg<16, 2, 256>, panel_nt = 256;
        else if (rows <= 768) panel_reg = k_lu_panel_reg<16, 3, 256>, panel_nt = 256;
        else if (rows <= 1152) panel_reg = k_lu_panel_reg<16, 3, 384>, panel_nt = 384;
      } else if (nb == 8) {
        if (rows <= 1536) panel_reg = k_lu_panel_reg<8, 4, 384>, panel_nt = 384;
        else if (rows <= 2304) panel_reg = k_lu_panel_reg<8, 6, 384>, panel_nt = 384;
      }
    }
    const size_t lds = (size_t)(kl + nb) * nb * sizeof(double);        // the panel
    const size_t ldt = (size_t)TG_LU_TC * (kl + nb) * sizeof(double);  // TG_LU_TC trailing columns
    if (use_blocked &&
        (hipFuncSetAttribute((const void *)k_lu_panel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess ||
         hipFuncSetAttribute((const void *)trail, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldt) != hipSuccess)) {
      (void)hipGetLastError();
      use_blocked = false;
    }
    blocked_used = use_blocked;
    if (use_blocked) {
      const unsigned gt = (unsigned)std::max<int64_t>(1, std::min<int64_t>(tg_cdiv((int64_t)kv + nb, TG_LU_TC), (int64_t)g_tg.num_cu * 8));
      for (int64_t j0 = 0; j0 < n; j0 += nb) {
        if (panel_reg) hipLaunchKernelGGL(panel_reg, dim3(1), dim3(panel_nt), 0, g_tg.stream, ab, ldab, n, kl, kv, j0, ipiv, st);
        else hipLaunchKernelGGL(k_lu_panel, dim3(1), dim3(1024), lds, g_tg.stream, ab, ldab, n, kl, kv, j0, nb, ipiv, st);
        if (j0 + nb >= n) break;
        hipLaunchKernelGGL(trail, dim3(gt), dim3(256), ldt, g_tg.stream, ab, ldab, n, kl, kv, j0, nb, (const int32_t *)ipiv,
                             (const tg_lu_state *)st);
      }
    } else if (fused) {
      hipLaunchKernelGGL(k_lu_pivot, dim3(1), dim3(256), 0, g_tg.stream, ab, ldab, n, kl, kv, (int64_t)0, ipiv, st, 0);
      const unsigned gy = (unsigned)std::max<int64_t>(1, std::min<int64_t>((int64_t)kv, (int64_t)g_tg.num_cu * 8));
      for (int64_t j = 0; j + 1 < n; j++)
        hipLaunchKernelGGL(k_lu_step, dim3(1, gy), dim3(256), 0, g_tg.stream, ab, ldab, n, kl, kv, j, ipiv, st);
    } else {
      // trailing window of a column: at most kl rows x kv columns
      const unsigned gx = (unsigned)std::max<int64_t>(1, std::min<int64_t>(tg_cdiv(kl, 256), 64));
      const unsigned gy = (unsigned)std::max<int64_t>(1, std::min<int64_t>((int64_t)kv, (int64_t)g_tg.num_cu * 8 / gx));
      for (int64_t j = 0; j < n; j++) {
        hipLaunchKernelGGL(k_lu_pivot, dim3(1), dim3(256), 0, g_tg.stream, ab, ldab, n, kl, kv, j, ipiv, st, 1);
        if (kl > 0 && j + 1 < n)
          hipLaunchKernelGGL(k_lu_update, dim3(gx, gy), dim3(256), 0, g_tg.stream, ab, ldab, kv, j, st);
      }
    }
    if (hipGetLastError() != hipSuccess) {
      tg_set_error("tg_lu_solve: kernel launch failed");
      rc = 1;
    }
  }
  if (!rc) {
    tg_lu_state h1;
    // (fused: the state of the last column is in slot (n-1) & 1; `info` is carried from slot to slot)
    const int fused1 = getenv("TIGAR_LU_FUSED") ? atoi(getenv("TIGAR_LU_FUSED")) : 1;
    if (hipMemcpyAsync(&h1, st + ((fused1 && !blocked_used) ? ((n - 1) & 1) : 0), sizeof(h1), hipMemcpyDeviceToHost, g_tg.stream) != hipSuccess) rc = 1;
    if (!rc && hipStreamSynchronize(g_tg.stream) != hipSuccess) {
      tg_set_error("tg_lu_solve: %s", hipGetErrorString(hipGetLastError()));
      rc = 1;
    }
    if (!rc) *info = h1.info;
  }
  if (!rc && *info == 0) {
    if (x->d != b->d &&
        hipMemcpyAsync(x->d, b->d, (size_t)n * sizeof(double), hipMemcpyDeviceToDevice, g_tg.stream) != hipSuccess)
      rc = 1;
    // 1024 threads x R entries each (measured at cfg4: 256 / 512 threads with more entries each are slower, 148 / 103 ms
    // against 89 ms)
    const int nt = 1024;
    const int rf = (int)tg_cdiv(std::max(kl, 1), nt), rb = (int)tg_cdiv(std::max(kv, 1), nt);
    if (rb <= 6 && !getenv("TIGAR_LU_SOLVE_GLOBAL")) {
      typedef void (*fwd_fn)(const double *, int64_t, int64_t, int, int, const int32_t *, double *);
      typedef void (*bwd_fn)(const double *, int64_t, int64_t, int, double *);
      const fwd_fn fwd = rf <= 1 ? k_lu_fwd<1, 16, 1024> : rf == 2 ? k_lu_fwd<2, 16, 1024> : rf == 3 ? k_lu_fwd<3, 8, 1024>
                         : rf == 4 ? k_lu_fwd<4, 8, 1024> : k_lu_fwd<6, 4, 1024>;
      const bwd_fn bwd = rb <= 1 ? k_lu_bwd<1, 16, 1024> : rb == 2 ? k_lu_bwd<2, 16, 1024> : rb == 3 ? k_lu_bwd<3, 8, 1024>
                         : rb == 4 ? k_lu_bwd<4, 8, 1024> : k_lu_bwd<6, 4, 1024>;
      hipLaunchKernelGGL(fwd, dim3(1), dim3(nt), (size_t)(kl + 2 + TG_LU_CH) * sizeof(double), g_tg.stream, (const double *)ab,
                         ldab, n, kl, kv, (const int32_t *)ipiv, x->d);
      hipLaunchKernelGGL(bwd, dim3(1), dim3(nt), (size_t)(kv + 2 + TG_LU_CH) * sizeof(double), g_tg.stream, (const double *)ab,
                         ldab, n, kv, x->d);
    } else
      hipLaunchKernelGGL(k_lu_solve_global, dim3(1), dim3(1024), 0, g_tg.stream, ab, ldab, n, kl, kv, ipiv, x->d);
    if (hipGetLastError() != hipSuccess) rc = 1;
    if (hipStreamSynchronize(g_tg.stream) != hipSuccess) rc = 1;
    if (rc) tg_set_error("tg_lu_solve: substitution failed");
  }
  tg_dfree(ab);
  tg_dfree(ipiv);
  tg_dfree(st);
  return rc;
}
