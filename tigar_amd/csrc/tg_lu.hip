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
    for (int i = 1 + blockIdx.x * 256 + threadIdx.x; i <= km; i += gridDim.x * 256) cc[i] -= l[i] * u;
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
      for (int i = 1 + tid; i <= km; i += 256) cc[i] -= l[i] * u;
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
        for (int i = 1 + tid; i <= km; i += 256) cc[i] -= l[i] * u;
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
    for (int i = 1 + tid; i <= km; i += 1024) x[j + i] -= l[i] * xj;
  }
  for (int64_t j = n - 1; j >= 0; j--) {
    __syncthreads();
    const double *cj = ab + kv + ldab * j;         // cj[-i] = U(j-i, j)
    if (tid == 0) x[j] = x[j] / cj[0];
    __syncthreads();
    const double xj = x[j];
    const int kk = (int)min((int64_t)kv, j);
    for (int i = 1 + tid; i <= kk; i += 1024) x[j - i] -= cj[-i] * xj;
  }
}

// forward substitution with the row interchanges, then backward substitution; one workgroup.  The entries of x a step
// touches form a window of kl+1 (forward) / kv+1 (backward) consecutive entries that slides by one per step: the window
// lives in LDS as a ring (entry x[q] at q mod W), so a step costs one barrier and no global round trip -- the multipliers
// of the step after next are requested before the barrier (201 -> ~60 ms at cfg4: 67 600 steps each way).
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
      if (i <= km) ring[tg_slot(b, i, W)] -= CUR[q] * xj;                                       \
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
      if (i <= kk) ring[tg_slot(b, W - i, W)] -= CUR[q] * xj;                                   \
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
    static const int fused = getenv("TIGAR_LU_FUSED") ? atoi(getenv("TIGAR_LU_FUSED")) : 1;
    if (fused) {
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
    static const int fused1 = getenv("TIGAR_LU_FUSED") ? atoi(getenv("TIGAR_LU_FUSED")) : 1;
    if (hipMemcpyAsync(&h1, st + (fused1 ? ((n - 1) & 1) : 0), sizeof(h1), hipMemcpyDeviceToHost, g_tg.stream) != hipSuccess) rc = 1;
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
