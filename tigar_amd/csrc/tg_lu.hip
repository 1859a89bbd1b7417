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

// One workgroup on one CU: four waves per SIMD share its issue slots, so the kernel is bound by its instruction
// count -- the search reduces through DPP (no LDS tree), the update walks the columns with the multiplier in a
// register (no index arithmetic per entry).
__global__ void __launch_bounds__(1024)
    k_lu_panel(double *__restrict__ ab, int64_t ldab, int64_t n, int kl, int kv, int64_t j0, int nb,
               int32_t *__restrict__ ipiv, tg_lu_state *st) {
  extern __shared__ double P[];                    // [nb][H], H = kl + nb: P[lc * H + lr] = A(j0 + lr, j0 + lc)
  __shared__ unsigned long long wkey[16];
  __shared__ int widx[16];
  const int tid = threadIdx.x, wv = tid >> 6;
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
    // first maximum of |cj[0..km]|: the bit pattern of |a| orders like |a| (and a NaN above everything, so that it
    // surfaces); ties go to the smallest index
    unsigned long long key = 0;
    int bi = 0x7fffffff;
    for (int i = tid; i <= km; i += 1024) {
      const unsigned long long k2 = (unsigned long long)__double_as_longlong(fabs(cj[i]));
      if (k2 > key || bi == 0x7fffffff) {
        key = k2;
        bi = i;
      }
    }
    const unsigned long long wmax = tg_wave_max_u64(key);
    const int wi = tg_wave_min_i32(key == wmax ? bi : 0x7fffffff);
    if ((tid & 63) == 0) {
      wkey[wv] = wmax;
      widx[wv] = wi;
    }
    __syncthreads();
    unsigned long long bk = wkey[0];
    int jp = widx[0];
#pragma unroll
    for (int w = 1; w < 16; w++) {
      const unsigned long long k2 = wkey[w];
      const int i2 = widx[w];
      if (k2 > bk || (k2 == bk && i2 < jp)) {
        bk = k2;
        jp = i2;
      }
    }
    const double piv = cj[jp], a0 = cj[0];
    ju = max(ju, (int)min(j + (int64_t)ku + jp, n - 1));
    if (tid == 0) ipiv[j] = (int32_t)(j + jp);
    if (piv == 0.0 && info == 0) info = (int)(j + 1);
    __syncthreads();                               // every thread has read the partial maxima, cj[0] and cj[jp]
    if (piv == 0.0) continue;                      // (uniform) LAPACK goes on without eliminating
    const int lcl = (int)min((int64_t)(nbc - 1), (int64_t)ju - j0);     // last panel column the step touches
    const int ncol = lcl - jj;
    // interchange (the other panel columns: one thread each; column jj itself through a0 / piv) and multipliers
    if (jp != 0) {
      if (tid < ncol) {
        double *q = P + (jj + 1 + tid) * H + jj;
        const double a = q[jp], b = q[0];
        q[jp] = b;
        q[0] = a;
      }
      if (tid == 0) cj[0] = piv;
    }
    const double pinv = 1.0 / piv;
    for (int i = 1 + tid; i <= km; i += 1024) cj[i] = (i == jp ? a0 : cj[i]) * pinv;
    __syncthreads();
    // rank-1 update of the panel columns jj+1 .. lcl: a thread keeps its multiplier and walks the columns
    for (int i = 1 + tid; i <= km; i += 1024) {
      const double l = cj[i];
      double *cc = P + (jj + 1) * H + jj;          // cc[i] = A(j + i, j + 1 + c)
      for (int c = 0; c < ncol; c++, cc += H) {
        const double u = cc[0];
        if (u != 0.0) cc[i] = fma(-l, u, cc[i]);
      }
    }
    __syncthreads();
  }
  tg_lu_panel_copy<false>(ab, P, ldab, n, kl, kv, j0, nbc, H, tid);
  if (tid == 0) {
    st->ju = ju;
    st->info = info;
  }
}

// trailing columns, one workgroup each (all of them resident at once: the kernel is bound by its instruction count).
// NP > 0: a thread owns NP pairs of adjacent rows; the multipliers of a step, its pivot and its interchange are
// fetched one step ahead, so that a step costs LDS work only.  NP = 0: any kl.
template <int NP>
__global__ void __launch_bounds__(256)
    k_lu_trail(double *__restrict__ ab, int64_t ldab, int64_t n, int kl, int kv, int64_t j0, int nb,
               const int32_t *__restrict__ ipiv, const tg_lu_state *__restrict__ st) {
  extern __shared__ double col[];                  // col[r - j0], rows j0 .. j0 + nb - 1 + kl
  const int tid = threadIdx.x;
  const int nbc = (int)min((int64_t)nb, n - j0);
  const int64_t ju = st->ju;
  constexpr int NR = NP > 0 ? NP : 1;
  for (int64_t c = j0 + nbc + blockIdx.x; c <= ju; c += gridDim.x) {   // (uniform per workgroup)
    const int64_t rlo = max(j0, c - kv), rhi = min(n - 1, j0 + nbc - 1 + (int64_t)kl);
    double *cc = ab + kv - c + ldab * c;           // cc[r] = A(r, c)
    __syncthreads();
    for (int64_t r = rlo + tid; r <= rhi; r += 256) col[r - j0] = cc[r];
    const int jj0 = (int)(rlo - j0);               // steps before: their row is above this column's band (zero)
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
      double u = col[jj];
      if (r2 != j) {
        const double a = col[r2 - j0];
        __syncthreads();                           // every thread has read both
        if (tid == 0) {
          col[r2 - j0] = u;
          col[jj] = a;
        }
        u = a;
        __syncthreads();
      }
      if (u != 0.0) {
        if (NP > 0) {
#pragma unroll
          for (int q = 0; q < NR; q++) {
            const int i = 1 + 2 * tid + 512 * q;
            if (i + 1 <= km) {
              col[jj + i] = fma(-ca[q], u, col[jj + i]);
              col[jj + i + 1] = fma(-cb[q], u, col[jj + i + 1]);
            } else if (i <= km) {
              col[jj + i] = fma(-ca[q], u, col[jj + i]);
            }
          }
        } else {
          for (int i = 1 + tid; i <= km; i += 256) col[jj + i] = fma(-lj[i], u, col[jj + i]);
        }
      }
      __syncthreads();
    }
    for (int64_t r = rlo + tid; r <= rhi; r += 256) cc[r] = col[r - j0];
  }
}

// barrier that orders the LDS traffic of the workgroup only: __syncthreads() also waits for every outstanding GLOBAL
// load, i.e. for the multipliers of the next step that are requested on purpose before they are needed
#define TG_LDS_BARRIER()                                            \
  do {                                                              \
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local"); \
    __builtin_amdgcn_s_barrier();                                   \
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local"); \
  } while (0)
#define TG_LU_R 6            // entries of a step per thread held in registers: windows of up to 6 * 1024 entries
#define TG_LU_CH 1024        // entries of x that enter the window at a time
// ring slot of entry (base entry at slot b) + off, 0 <= off < W   (32-bit: a 64-bit modulo per access costs more than the step)
__device__ __forceinline__ int tg_slot(int b, int off, int W) {
  const int q = b + off;
  return q >= W ? q - W : q;
}
__device__ __forceinline__ void tg_lu_load_col(const double *__restrict__ col, int sign, int cnt, int tid, double *v) {
#pragma unroll
  for (int q = 0; q < TG_LU_R; q++) {
    const int i = 1 + tid + 1024 * q;
    v[q] = i <= cnt ? col[sign * i] : 0.0;
  }
}

__global__ void __launch_bounds__(1024)
    k_lu_solve(const double *__restrict__ ab, int64_t ldab, int64_t n, int kl, int kv, const int32_t *__restrict__ ipiv,
               double *__restrict__ x) {
  extern __shared__ double ring[];                 // W doubles; entry x[q] lives at slot q mod W
  const int tid = threadIdx.x;
  const int W = kv + 2 + TG_LU_CH;
  double va[TG_LU_R], vb[TG_LU_R];
  // ---- forward: the window [j, j + kl] lies inside the loaded range [j, top); `top` grows by TG_LU_CH entries at a time
  // (all threads load, one wait per TG_LU_CH steps).  The multipliers and the pivot row of step j+1 are requested during
  // step j into the OTHER register set (two steps per trip of the loop: no copies, so nothing waits for them early).
  int64_t top = min(n, (int64_t)kl + 1 + TG_LU_CH);
  for (int64_t q = tid; q < top; q += 1024) ring[(int)(q % W)] = x[q];
  int b = 0;                                       // slot of entry j
  int tb = (int)(top % W);                         // slot of entry top
  int jpa = n > 0 ? (int)(ipiv[0] - 0) : 0, jpb = 0;   // pivot offsets jp - j
  tg_lu_load_col(ab + kv, 1, (int)min((int64_t)kl, n - 1), tid, va);
  __syncthreads();
#define TG_FWD_STEP(J, CUR, NXT, JPC, JPN)                                                      \
  {                                                                                             \
    const int64_t j_ = (J);                                                                     \
    const int km = (int)min((int64_t)kl, n - 1 - j_);                                           \
    if (j_ + 1 < n) {                                                                           \
      JPN = (int)(ipiv[j_ + 1] - (j_ + 1));                                                     \
      tg_lu_load_col(ab + kv + ldab * (j_ + 1), 1, (int)min((int64_t)kl, n - 2 - j_), tid, NXT); \
    }                                                                                           \
    if (tid == 0 && JPC != 0) {                                                                 \
      const int sp = tg_slot(b, JPC, W);                                                        \
      const double t = ring[sp];                                                                \
      ring[sp] = ring[b];                                                                       \
      ring[b] = t;                                                                              \
    }                                                                                           \
    TG_LDS_BARRIER();                                                                           \
    const double xj = ring[b];                                                                  \
    _Pragma("unroll") for (int q = 0; q < TG_LU_R; q++) {                                       \
      const int i = 1 + tid + 1024 * q;                                                         \
      if (i <= km) ring[tg_slot(b, i, W)] = fma(-CUR[q], xj, ring[tg_slot(b, i, W)]);                                       \
    }                                                                                           \
    if (tid == 0) x[j_] = xj;                                                                   \
    if (top < n && j_ + 1 + kl + 1 > top) {                                                     \
      const int64_t e1 = min(n, top + TG_LU_CH);                                                \
      for (int64_t e = top + tid; e < e1; e += 1024) ring[tg_slot(tb, (int)(e - top), W)] = x[e]; \
      tb = tg_slot(tb, (int)(e1 - top), W);                                                     \
      top = e1;                                                                                 \
    }                                                                                           \
    TG_LDS_BARRIER();                                                                           \
    b = tg_slot(b, 1, W);                                                                       \
  }
  {
    int64_t j = 0;
    for (; j + 1 < n; j += 2) {
      TG_FWD_STEP(j, va, vb, jpa, jpb)
      TG_FWD_STEP(j + 1, vb, va, jpb, jpa)
    }
    if (j < n) TG_FWD_STEP(j, va, vb, jpa, jpb)
  }
#undef TG_FWD_STEP
  __syncthreads();                                 // (the forward values written to x are visible to the whole workgroup)
  // ---- backward: the window [j - kv, j] lies inside the loaded range [lo, j]
  int64_t lo = max((int64_t)0, n - 1 - kv - TG_LU_CH);
  for (int64_t q = lo + tid; q <= n - 1; q += 1024) ring[(int)(q % W)] = x[q];
  b = n > 0 ? (int)((n - 1) % W) : 0;              // slot of entry j
  int lb = (int)(lo % W);                          // slot of entry lo
  double da = 1.0, db = 1.0;
  if (n > 0) {
    const double *cj = ab + kv + ldab * (n - 1);
    tg_lu_load_col(cj, -1, (int)min((int64_t)kv, n - 1), tid, va);
    da = cj[0];
  }
  __syncthreads();
#define TG_BWD_STEP(J, CUR, NXT, DC, DN)                                                        \
  {                                                                                             \
    const int64_t j_ = (J);                                                                     \
    const int kk = (int)min((int64_t)kv, j_);                                                   \
    if (j_ >= 1) {                                                                              \
      const double *c1 = ab + kv + ldab * (j_ - 1);                                             \
      tg_lu_load_col(c1, -1, (int)min((int64_t)kv, j_ - 1), tid, NXT);                          \
      DN = c1[0];                                                                               \
    }                                                                                           \
    const double xj = ring[b] / DC;                                                             \
    TG_LDS_BARRIER();                                                                           \
    _Pragma("unroll") for (int q = 0; q < TG_LU_R; q++) {                                       \
      const int i = 1 + tid + 1024 * q;                                                         \
      if (i <= kk) ring[tg_slot(b, W - i, W)] = fma(-CUR[q], xj, ring[tg_slot(b, W - i, W)]);                                   \
    }                                                                                           \
    if (tid == 0) x[j_] = xj;                                                                   \
    if (lo > 0 && lo > j_ - 1 - kv) {                                                           \
      const int64_t e0 = max((int64_t)0, lo - TG_LU_CH);                                        \
      const int cnt = (int)(lo - e0);                                                           \
      const int nb_ = tg_slot(lb, W - cnt, W);                                                  \
      for (int e = tid; e < cnt; e += 1024) ring[tg_slot(nb_, e, W)] = x[e0 + e];               \
      lb = nb_;                                                                                 \
      lo = e0;                                                                                  \
    }                                                                                           \
    TG_LDS_BARRIER();                                                                           \
    b = tg_slot(b, W - 1, W);                                                                   \
  }
  {
    int64_t j = n - 1;
    for (; j >= 1; j -= 2) {
      TG_BWD_STEP(j, va, vb, da, db)
      TG_BWD_STEP(j - 1, vb, va, db, da)
    }
    if (j == 0) TG_BWD_STEP(0, va, vb, da, db)
  }
#undef TG_BWD_STEP
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
    // panel width: the widest of 32 / 16 / 8 / 4 columns whose panel ((kl + nb) x nb doubles) fits in LDS
    int nb = 0;
    for (int cand = 32; cand >= 4 && !nb; cand >>= 1)
      if ((size_t)(kl + cand) * cand * sizeof(double) <= 140 * 1024) nb = cand;
    if (getenv("TIGAR_LU_NB")) nb = atoi(getenv("TIGAR_LU_NB"));
    bool use_blocked = blocked && nb >= 2 && kl > 0;
    if (use_blocked) {
      const size_t lds = (size_t)(kl + nb) * nb * sizeof(double);
      if (hipFuncSetAttribute((const void *)k_lu_panel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
        (void)hipGetLastError();
        use_blocked = false;
      }
    }
    blocked_used = use_blocked;
    if (use_blocked) {
      const size_t lds = (size_t)(kl + nb) * nb * sizeof(double);
      const unsigned gt = (unsigned)std::max<int64_t>(1, std::min<int64_t>((int64_t)kv + nb, (int64_t)g_tg.num_cu * 8));
      for (int64_t j0 = 0; j0 < n; j0 += nb) {
        hipLaunchKernelGGL(k_lu_panel, dim3(1), dim3(1024), lds, g_tg.stream, ab, ldab, n, kl, kv, j0, nb, ipiv, st);
        if (j0 + nb < n) {
          const size_t ldt = (size_t)(kl + nb) * sizeof(double);
#define TG_LU_TRAIL(NI) hipLaunchKernelGGL(k_lu_trail<NI>, dim3(gt), dim3(256), ldt, g_tg.stream, ab, ldab, n, kl, kv, j0, nb, \
                                           (const int32_t *)ipiv, (const tg_lu_state *)st)
          if (kl <= 512) TG_LU_TRAIL(1);
          else if (kl <= 1024) TG_LU_TRAIL(2);
          else if (kl <= 1536) TG_LU_TRAIL(3);
          else if (kl <= 2048) TG_LU_TRAIL(4);
          else TG_LU_TRAIL(0);
#undef TG_LU_TRAIL
        }
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
    if (kv + 1 <= TG_LU_R * 1024 && !getenv("TIGAR_LU_SOLVE_GLOBAL"))
      hipLaunchKernelGGL(k_lu_solve, dim3(1), dim3(1024), (size_t)(kv + 2 + TG_LU_CH) * sizeof(double), g_tg.stream, ab, ldab, n, kl, kv,
                         ipiv, x->d);
    else
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
