// Krylov solves of SMALL systems in ONE persistent kernel (round 4).
//
// solveLinearSystem (tIGAr/common.py:1236-1263) on the 2-D configurations -- cfg4: 67 600 unknowns, 65 MB of K; cfg5:
// 51 483 unknowns, 89 MB -- is bound by launches, not by bytes: an iteration of the single-reduction CG of tg_krylov.hip is
// four dependent launches (product, inner product, fold, update: 33 us) around 10 us of memory traffic, and K sits in the
// 256 MB Infinity Cache the whole time.  Here the whole loop is one kernel: one workgroup per CU owns a contiguous block of
// rows -- its part of every vector, its rows of K --, the iteration is
//
//     w = K u (own rows)            partial (r,u), (w,u), (u,u) of the own rows  ->  HBM slot of the workgroup
//     -- grid barrier --
//     every workgroup sums ALL partials in the same fixed order (identical scalars everywhere, no broadcast),
//     decides convergence, forms alpha / beta, updates p, s, x, r, u = B r on its rows
//     -- grid barrier --            (u complete before the next product)
//
// i.e. the recurrence of tg_cg (Chronopoulos & Gear; KSPCG -ksp_cg_single_reduction [ext]) with two device-wide barriers per
// iteration instead of four kernel boundaries.  The barrier is a counter + generation word in HBM (agent-scope atomics,
// release before / acquire after: gfx950 has one L2 per XCD, the fences write back and invalidate what the other XCDs must
// see), waited for with s_sleep and a wall-clock limit -- a barrier that cannot complete (a workgroup not resident) ends the
// kernel with an error instead of hanging the GPU; the launch is cooperative, so residency of all workgroups is checked by
// the runtime.  Sums are formed in a fixed order: the iterates are bit-reproducible run to run (they differ from tg_cg's in
// the last bits: other partial sums).
#include "tg_common.h"
#include <algorithm>

#define PS_NT 512                // threads per workgroup: 2 waves per SIMD, 256 registers per lane -- a CU's share of K fits
#define PS_GROUPS (PS_NT / 32)   // groups of 32 lanes, one row each at a time
#define PS_MAXG 512              // workgroups at most (partials are folded by one pass of a workgroup)

#define PS_AUX_SC1 16            // cache-policy bit of the buffer loads: sc1 (agent scope) on gfx940+
#define PS_FAN 16                // workgroups per first-level counter of the barrier
struct tg_ps_ctrl {
  unsigned gen;
  int abort_flag;
  unsigned pad0[30];
  unsigned top;                  // second level: groups that are complete
  unsigned pad1[31];
  unsigned grp[(PS_MAXG / PS_FAN) * 32];   // first level, one 128-byte line per group of PS_FAN workgroups
  double out[8];                 // [0] iterations, [1] nu at the end, [2] nu0 (reference), [3] status, [4..7] phase ticks
};

struct tg_ps_args {
  const int64_t *rowptr;
  const int32_t *col;
  const double *val;
  int64_t n;
  const double *b;
  double *x;
  double *u;                     // u = B r, all rows (the other workgroups gather it)
  double *partial;               // [G][4]
  tg_ps_ctrl *ctrl;
  double rtol, atol;
  int maxit, jacobi, nonzero_guess, pad;
  long long budget_ticks;        // wall_clock64 ticks the kernel may wait at one barrier
};

// What the workgroups exchange -- the gathered vector, the partial sums, the barrier words -- goes through AGENT-SCOPE relaxed
// atomic stores and loads (sc1 on gfx950: written through to / read from the level that is coherent across the XCDs), so the
// barrier needs NO cache maintenance: every wave waits for its own stores (s_waitcnt vmcnt(0)), the workgroup meets, thread 0
// arrives.  Measured (tools/mb/grid_barrier.hip, 256 workgroups, 2 KB exchanged per workgroup): 2.5 us per barrier; with a
// release fence before and an acquire fence after (buffer_wbl2 / buffer_inv of a whole L2) 8.5 us; with the fences in every
// wave 31 us; plain stores and loads without fences read stale data -- each XCD has its own L2 --, also in memory
// allocated fine-grained.
__device__ __forceinline__ void ps_store(double *p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ double ps_load(const double *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

__device__ __forceinline__ bool ps_barrier(tg_ps_ctrl *c, unsigned G, unsigned &gen, long long budget) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  __shared__ int s_ok;
  if (threadIdx.x == 0) {
    int ok = 1;
    const unsigned target = gen + 1;
    // two levels (atomics on ONE address are served one after the other: 256 of them are 3.7 us, 16 + 16 are 1.7)
    const unsigned grp = blockIdx.x / PS_FAN, ngrp = (G + PS_FAN - 1) / PS_FAN;
    const unsigned gsz = grp + 1 < ngrp ? PS_FAN : G - grp * PS_FAN;
    bool last = false;
    if (__hip_atomic_fetch_add(&c->grp[32 * grp], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gsz - 1) {
      __hip_atomic_store(&c->grp[32 * grp], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (__hip_atomic_fetch_add(&c->top, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == ngrp - 1) last = true;
    }
    if (last) {
      __hip_atomic_store(&c->top, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __hip_atomic_store(&c->gen, target, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      const long long t0 = wall_clock64();
      unsigned spins = 0;
      while (__hip_atomic_load(&c->gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != target) {
        __builtin_amdgcn_s_sleep(1);
        if ((++spins & 1023u) == 0u) {
          if (__hip_atomic_load(&c->abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
            ok = 0;
            break;
          }
          if (wall_clock64() - t0 > budget) {
            __hip_atomic_store(&c->abort_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            ok = 0;
            break;
          }
        }
      }
    }
    s_ok = ok;
  }
  gen++;
  __syncthreads();
  return s_ok != 0;
}

// three sums over the workgroup at once, fixed order (wave shuffles, then the wave sums by threads 0..2); valid in every thread
__device__ __forceinline__ void ps_block_sum3(double &a, double &b, double &c, double *lds /* [3 * 17] */) {
  a = tg_wave_sum(a);
  b = tg_wave_sum(b);
  c = tg_wave_sum(c);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (lane == 0) {
    lds[w] = a;
    lds[17 + w] = b;
    lds[34 + w] = c;
  }
  __syncthreads();
  if (threadIdx.x < 3) {
    double t = 0.0;
    for (int q = 0; q < PS_NT / 64; q++) t += lds[17 * threadIdx.x + q];
    lds[17 * threadIdx.x + 16] = t;
  }
  __syncthreads();
  a = lds[16];
  b = lds[33];
  c = lds[50];
  __syncthreads();                       // (the slots are written again by the next call)
}

// The rows of the workgroup live in REGISTERS: 32 lanes per row, EPR entries of the row per lane, RI rows per group of lanes
// (local row = group + 32 * ri), padded with (value 0, own column) -- 32 slots of (8 B value, 4 B column) per thread, a
// whole CU holds 384 KB of K.  The workgroup's part of every vector lives in LDS; only u = B r goes to HBM (the other
// workgroups gather it), x at the end.
#define PS_ROWS_MAX 1024           // local rows at most (vectors in LDS)
struct ps_lds {
  double x[PS_ROWS_MAX], r[PS_ROWS_MAX], p[PS_ROWS_MAX], s[PS_ROWS_MAX], u[PS_ROWS_MAX], w[PS_ROWS_MAX], dinv[PS_ROWS_MAX];
  double red[3 * 17];
};

template <int EPR, int RI>
__device__ __forceinline__ void ps_product(const double (&v)[RI][EPR], const unsigned (&c)[RI][EPR], const double *__restrict__ xg,
                                           double *__restrict__ w_lds, int nloc) {
  // c: BYTE offsets into xg (uniform base + 32-bit lane offset: no 64-bit address per entry).  The gathers are issued in
  // batches of PS_BATCH rows -- all of them at once (what the scheduler does when left alone) needs two registers per entry
  // for the values in flight on top of the three that hold the entry, and the kernel spills.
  constexpr int PS_BATCH = EPR >= 5 ? 2 : EPR == 4 ? 3 : EPR == 3 ? 4 : EPR == 2 ? 6 : 12;
  const int g = threadIdx.x >> 5, l = threadIdx.x & 31;
  // (raw buffer loads: the descriptor in scalar registers, ONE 32-bit register per entry for the offset -- as pointers the
  //  compiler kept a 64-bit offset per entry: 4 registers per entry instead of 3)
  const __amdgpu_buffer_rsrc_t xb = __builtin_amdgcn_make_buffer_rsrc((void *)xg, 0, 0x7fffffff, 0x00020000);
#pragma unroll
  for (int r0 = 0; r0 < RI; r0 += PS_BATCH) {
    double xs[PS_BATCH][EPR];
#pragma unroll
    for (int ri = r0; ri < r0 + PS_BATCH && ri < RI; ri++)
#pragma unroll
      for (int k = 0; k < EPR; k++)
        xs[ri - r0][k] = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(xb, (int)c[ri][k], 0, PS_AUX_SC1));
#pragma unroll
    for (int ri = r0; ri < r0 + PS_BATCH && ri < RI; ri++) {
      double acc = 0.0;
#pragma unroll
      for (int k = 0; k < EPR; k++) acc = fma(v[ri][k], xs[ri - r0][k], acc);
#pragma unroll
      for (int o = 16; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
      const int lrow = g + PS_GROUPS * ri;
      if (l == 0 && lrow < nloc) w_lds[lrow] = acc;
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}

// The same with the last EL of the EPR entries per lane and row held in LDS (lv / lc: [RI][EL][PS_NT]) instead of registers:
// for rows of 129-160 entries (cfg5: 147) the 65 slots of a lane do not fit the register file next to the GMRES body -- the
// kernel spilled ~100 registers per lane to scratch, re-read in every product.
template <int EPR, int RI, int EL>
__device__ __forceinline__ void ps_product_l(const double (&v)[RI][EPR - EL], const unsigned (&c)[RI][EPR - EL],
                                             const double *__restrict__ lv, const unsigned *__restrict__ lc,
                                             const double *__restrict__ xg, double *__restrict__ w_lds, int nloc) {
  constexpr int ER = EPR - EL;
  constexpr int PS_BATCH = EPR >= 5 ? 2 : EPR == 4 ? 3 : EPR == 3 ? 4 : EPR == 2 ? 6 : 12;
  const int g = threadIdx.x >> 5, l = threadIdx.x & 31;
  const __amdgpu_buffer_rsrc_t xb = __builtin_amdgcn_make_buffer_rsrc((void *)xg, 0, 0x7fffffff, 0x00020000);
#pragma unroll
  for (int r0 = 0; r0 < RI; r0 += PS_BATCH) {
    double xs[PS_BATCH][EPR];
    double lvv[PS_BATCH][EL > 0 ? EL : 1];
#pragma unroll
    for (int ri = r0; ri < r0 + PS_BATCH && ri < RI; ri++) {
#pragma unroll
      for (int k = 0; k < ER; k++)
        xs[ri - r0][k] = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(xb, (int)c[ri][k], 0, PS_AUX_SC1));
#pragma unroll
      for (int k = 0; k < EL; k++) {
        const int at = (ri * EL + k) * PS_NT + threadIdx.x;
        lvv[ri - r0][k] = lv[at];
        xs[ri - r0][ER + k] = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(xb, (int)lc[at], 0, PS_AUX_SC1));
      }
    }
#pragma unroll
    for (int ri = r0; ri < r0 + PS_BATCH && ri < RI; ri++) {
      double acc = 0.0;
#pragma unroll
      for (int k = 0; k < ER; k++) acc = fma(v[ri][k], xs[ri - r0][k], acc);
#pragma unroll
      for (int k = 0; k < EL; k++) acc = fma(lvv[ri - r0][k], xs[ri - r0][ER + k], acc);
#pragma unroll
      for (int o = 16; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
      const int lrow = g + PS_GROUPS * ri;
      if (l == 0 && lrow < nloc) w_lds[lrow] = acc;
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}

template <int EPR, int RI>
__global__ void __launch_bounds__(PS_NT) k_cg_persistent(tg_ps_args A) {
  extern __shared__ double ps_dyn[];
  ps_lds &L = *(ps_lds *)ps_dyn;
  const unsigned G = gridDim.x;
  unsigned gen = 0;
  const int64_t per = (A.n + G - 1) / G;
  const int64_t c0r = (int64_t)blockIdx.x * per, c0 = c0r < A.n ? c0r : A.n, c1 = c0 + per < A.n ? c0 + per : A.n;
  const int nloc = (int)(c1 - c0);
  const int tid = threadIdx.x, g = tid >> 5, l = tid & 31;
  // ---- the rows of K into registers; the Jacobi diagonal (PCJACOBI [ext]: 1 where the diagonal is zero / absent)
  double v[RI][EPR];
  unsigned c[RI][EPR];           // byte offsets of the columns
#pragma unroll
  for (int ri = 0; ri < RI; ri++) {
    const int lrow = g + PS_GROUPS * ri;
    const int64_t row = c0 + lrow;
    const bool live = lrow < nloc;
    const int64_t a = live ? A.rowptr[row] : 0, e = live ? A.rowptr[row + 1] : 0;
    double dd = 0.0;
#pragma unroll
    for (int k = 0; k < EPR; k++) {
      const int64_t q = a + l + 32 * k;
      const bool in = q < e;
      v[ri][k] = in ? A.val[q] : 0.0;
      const int cq = in ? A.col[q] : (int)(live ? row : 0);
      c[ri][k] = 8u * (unsigned)cq;
      if (in && cq == row) dd = v[ri][k];
    }
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) dd += __shfl_xor(dd, o, 64);
    if (l == 0 && live) L.dinv[lrow] = (A.jacobi && dd != 0.0) ? 1.0 / dd : 1.0;
  }
  __syncthreads();
  double bn = 0.0;                         // ||B b||^2 of the own rows (reference norm when a guess is given)
  if (A.nonzero_guess) {
    ps_product<EPR, RI>(v, c, A.x, L.w, nloc);      // r = b - K x0: all of x0 was written before the launch
    __syncthreads();
  }
  double gp = 0.0, np = 0.0, dp = 0.0;     // this thread's terms of (r,u), (u,u), (w,u)
  for (int i = tid; i < nloc; i += PS_NT) {
    const double bi = A.b[c0 + i];
    const double ri = A.nonzero_guess ? bi - L.w[i] : bi;
    const double ui = L.dinv[i] * ri;
    const double ub = L.dinv[i] * bi;
    bn += ub * ub;
    L.x[i] = A.nonzero_guess ? A.x[c0 + i] : 0.0;
    L.r[i] = ri;
    L.u[i] = ui;
    L.p[i] = 0.0;
    L.s[i] = 0.0;
    ps_store(&A.u[c0 + i], ui);
    gp += ri * ui;
    np += ui * ui;
  }
  {
    double z0 = 0.0, z1 = 0.0;
    ps_block_sum3(bn, z0, z1, L.red);
    if (tid == 0) ps_store(&A.partial[4 * blockIdx.x + 3], bn);
  }
  if (!ps_barrier(A.ctrl, G, gen, A.budget_ticks)) return;          // u complete
  double gamma_prev = 0.0, alpha_prev = 0.0, tol2 = 0.0, nu0 = 0.0, nu = 0.0;
  int it = 0, status = -1;
  long long t_spmv = 0, t_bar1 = 0, t_upd = 0, t_bar2 = 0;
#ifdef PS_TIMING
  long long t_mark = wall_clock64();
#endif
#ifdef PS_TIMING
#define PS_MARK(acc)                       \
  do {                                     \
    const long long now_ = wall_clock64(); \
    acc += now_ - t_mark;                  \
    t_mark = now_;                         \
  } while (0)
#else
#define PS_MARK(acc) (void)acc
#endif
  for (;;) {
    // ---- w = K u on the own rows, the three inner products of the own rows
    ps_product<EPR, RI>(v, c, A.u, L.w, nloc);
    __syncthreads();
    dp = 0.0;
    for (int i = tid; i < nloc; i += PS_NT) dp += L.w[i] * L.u[i];
    {
      double a = gp, b = dp, cc = np;
      ps_block_sum3(a, b, cc, L.red);
      if (tid == 0) {
        ps_store(&A.partial[4 * blockIdx.x], a);
        ps_store(&A.partial[4 * blockIdx.x + 1], b);
        ps_store(&A.partial[4 * blockIdx.x + 2], cc);
      }
    }
    PS_MARK(t_spmv);
    if (!ps_barrier(A.ctrl, G, gen, A.budget_ticks)) return;
    PS_MARK(t_bar1);
    double gamma = 0.0, delta = 0.0;
    nu = 0.0;
    if (tid < (int)G) {
      gamma = ps_load(&A.partial[4 * tid]);
      delta = ps_load(&A.partial[4 * tid + 1]);
      nu = ps_load(&A.partial[4 * tid + 2]);
    }
    ps_block_sum3(gamma, delta, nu, L.red);
    if (it == 0) {
      // reference norm: ||B r0||, or ||B b|| with a guess (KSPConvergedDefault [ext]); as tg_cg
      double ref = nu;
      if (A.nonzero_guess) {
        double t = tid < (int)G ? ps_load(&A.partial[4 * tid + 3]) : 0.0, z0 = 0.0, z1 = 0.0;
        ps_block_sum3(t, z0, z1, L.red);
        ref = t;
      }
      nu0 = ref;
      const double tol = fmax(A.rtol * sqrt(ref), A.atol);
      tol2 = tol * tol;
    }
    if (!(nu == nu)) {
      status = -2;
      break;
    }
    if (!(nu > tol2)) {
      const double zn = sqrt(nu);
      status = it == 0 ? (zn <= A.atol ? 1 : 0) : ((zn <= A.atol && !(zn <= A.rtol * sqrt(nu0))) ? 1 : 0);
      break;
    }
    if (it >= A.maxit) break;
    double beta = 0.0, alpha;
    if (it == 0)
      alpha = gamma / delta;
    else {
      beta = gamma / gamma_prev;
      alpha = gamma / (delta - beta * gamma / alpha_prev);
    }
    gamma_prev = gamma;
    alpha_prev = alpha;
    gp = 0.0;
    np = 0.0;
    for (int i = tid; i < nloc; i += PS_NT) {
      const double pi = L.u[i] + beta * L.p[i];
      const double si = L.w[i] + beta * L.s[i];
      L.p[i] = pi;
      L.s[i] = si;
      L.x[i] += alpha * pi;
      const double ri = L.r[i] - alpha * si;
      L.r[i] = ri;
      const double ui = L.dinv[i] * ri;
      L.u[i] = ui;
      ps_store(&A.u[c0 + i], ui);
      gp += ri * ui;
      np += ui * ui;
    }
    it++;
    PS_MARK(t_upd);
    if (!ps_barrier(A.ctrl, G, gen, A.budget_ticks)) return;        // u complete before the next product
    PS_MARK(t_bar2);
  }
  for (int i = tid; i < nloc; i += PS_NT) A.x[c0 + i] = L.x[i];
  if (blockIdx.x == 0 && tid == 0) {
    A.ctrl->out[0] = (double)it;
    A.ctrl->out[1] = nu;
    A.ctrl->out[2] = nu0;
    A.ctrl->out[3] = (double)status;
    A.ctrl->out[4] = (double)t_spmv;
    A.ctrl->out[5] = (double)t_bar1;
    A.ctrl->out[6] = (double)t_upd;
    A.ctrl->out[7] = (double)t_bar2;
  }
}

// ------------------------------------------------------------------------------------------------------------------
// GMRES(m) the same way (tg_gmres of tg_krylov.hip: left Jacobi preconditioning, classical Gram-Schmidt without refinement,
// Givens recurrence for the residual estimate, restart from the true residual -- KSPGMRES defaults [ext]; same counting of
// iterations, same status codes).  On top of the rows of K in registers a workgroup keeps ITS ROWS OF THE WHOLE BASIS
// v_0..v_m in LDS (31 x 256 rows x 8 B = 62 KB); every workgroup folds the partial Gram-Schmidt sums itself, in the same
// order, and runs the Givens recurrence redundantly -- identical H, identical decisions, no broadcast.  Three device-wide
// barriers per inner iteration (v_j complete | coefficients | norm) where the multi-kernel loop has seven launches.
#define PG_M 30                   // restart length at most
#define PG_ROWS 224               // local rows at most (the basis of the workgroup in LDS; 224: room for one layer of K)
struct pg_lds {
  double V[(PG_M + 1) * PG_ROWS];
  double x[PG_ROWS], w[PG_ROWS], dinv[PG_ROWS], b[PG_ROWS];
  double H[(PG_M + 1) * PG_M], cs[PG_M], sn[PG_M], g[PG_M + 1], y[PG_M], h[PG_M + 2];
  double red[3 * 17];
  double st[8];                   // [0] live, [1] res, [2] tol, [3] beta0, [4] its, [5] status, [6] kused
};

struct tg_pg_args {
  tg_ps_args P;
  int m;
  double *pdots;                  // [G][PG_M + 1] partial Gram-Schmidt sums
  double *pnorm;                  // [G][2]        partial norms
};

// sums of nstreams columns of partial[G][ld] into out[0..nstreams): wave w takes the streams w, w + 8, ...; fixed order
__device__ __forceinline__ void pg_fold(const double *__restrict__ partial, int ld, unsigned G, int nstreams, double *out) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll 1
  for (int s = w; s < nstreams; s += PS_NT / 64) {
    double t = 0.0;
#pragma unroll 1
    for (unsigned b = lane; b < G; b += 64) t += ps_load(&partial[(int64_t)b * ld + s]);
    t = tg_wave_sum(t);
    if (lane == 0) out[s] = t;
  }
  __syncthreads();
}

template <int EPR, int RI, int EL>
__global__ void __launch_bounds__(PS_NT) k_gmres_persistent(tg_pg_args Q) {
  extern __shared__ double ps_dyn[];
  pg_lds &L = *(pg_lds *)ps_dyn;
  constexpr int ER = EPR - EL;
  double *const lv = ps_dyn + (sizeof(pg_lds) + 7) / 8;           // [RI][EL][PS_NT] values, then the byte offsets
  unsigned *const lc = (unsigned *)(lv + (size_t)RI * (EL > 0 ? EL : 0) * PS_NT);
  const tg_ps_args &A = Q.P;
  const int m = Q.m;
  const unsigned G = gridDim.x;
  unsigned gen = 0;
  const int64_t per = (A.n + G - 1) / G;
  const int64_t c0r = (int64_t)blockIdx.x * per, c0 = c0r < A.n ? c0r : A.n, c1 = c0 + per < A.n ? c0 + per : A.n;
  const int nloc = (int)(c1 - c0);
  const int tid = threadIdx.x, g = tid >> 5, l = tid & 31, lane = tid & 63, wv = tid >> 6;
  double v[RI][ER];
  unsigned c[RI][ER];
#pragma unroll
  for (int ri = 0; ri < RI; ri++) {
    const int lrow = g + PS_GROUPS * ri;
    const int64_t row = c0 + lrow;
    const bool live = lrow < nloc;
    const int64_t a = live ? A.rowptr[row] : 0, e = live ? A.rowptr[row + 1] : 0;
    double dd = 0.0;
#pragma unroll
    for (int k = 0; k < EPR; k++) {
      const int64_t q = a + l + 32 * k;
      const bool in = q < e;
      const double vq = in ? A.val[q] : 0.0;
      const int cq = in ? A.col[q] : (int)(live ? row : 0);
      if (k < ER) {
        v[ri][k < ER ? k : 0] = vq;
        c[ri][k < ER ? k : 0] = 8u * (unsigned)cq;
      } else {
        lv[(ri * EL + (k - ER)) * PS_NT + tid] = vq;
        lc[(ri * EL + (k - ER)) * PS_NT + tid] = 8u * (unsigned)cq;
      }
      if (in && cq == row) dd = vq;
    }
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) dd += __shfl_xor(dd, o, 64);
    if (l == 0 && live) L.dinv[lrow] = (A.jacobi && dd != 0.0) ? 1.0 / dd : 1.0;
  }
  __syncthreads();
  const bool mine = tid < nloc;            // one row per thread (nloc <= PG_ROWS <= PS_NT)
  double bn = 0.0;
  if (mine) {
    L.b[tid] = A.b[c0 + tid];
    L.x[tid] = A.nonzero_guess ? A.x[c0 + tid] : 0.0;
    const double ub = L.dinv[tid] * L.b[tid];
    bn = ub * ub;
  }
  double bnorm2 = -1.0;
  if (A.nonzero_guess) {
    // reference norm ||B b|| [ext], and x0 where the other workgroups gather it
    double z0 = 0.0, z1 = 0.0;
    ps_block_sum3(bn, z0, z1, L.red);
    if (tid == 0) ps_store(&Q.pnorm[2 * blockIdx.x + 1], bn);
    if (mine) ps_store(&A.u[c0 + tid], L.x[tid]);
    if (!ps_barrier(A.ctrl, G, gen, A.budget_ticks)) return;
    pg_fold(Q.pnorm + 1, 2, G, 1, L.h);
    bnorm2 = L.h[0];
    __syncthreads();
  }
  bool first = true;
  int its = 0, status = -1, kused = 0;
  double res = 0.0, tol = 0.0;
  bool alive = true;
  while (alive) {
    // ---- r = B (b - K x), beta = ||r||, v_0 = r / beta
    if (!(first && !A.nonzero_guess)) {
      if (!first) {
        if (mine) ps_store(&A.u[c0 + tid], L.x[tid]);
        if (!ps_barrier(A.ctrl, G, gen, A.budget_ticks)) return;
      }
      ps_product_l<EPR, RI, EL>(v, c, lv, lc, A.u, L.w, nloc);
      __syncthreads();
    }
    double rr = 0.0;
    if (mine) {
      const double ri = L.dinv[tid] * ((first && !A.nonzero_guess) ? L.b[tid] : L.b[tid] - L.w[tid]);
      L.V[tid] = ri;
      rr = ri * ri;
    }
    {
      double z0 = 0.0, z1 = 0.0;
      ps_block_sum3(rr, z0, z1, L.red);
      if (tid == 0) ps_store(&Q.pnorm[2 * blockIdx.x], rr);
    }
    if (!ps_barrier(A.ctrl, G, gen, A.budget_ticks)) return;
    pg_fold(Q.pnorm, 2, G, 1, L.h);
    const double beta = sqrt(L.h[0]);
    __syncthreads();
    res = beta;
    if (first) {
      const double beta0 = bnorm2 >= 0.0 ? sqrt(bnorm2) : beta;
      tol = fmax(A.rtol * beta0, A.atol);
      if (!(beta == beta) || !(beta0 == beta0)) {
        status = -2;
        break;
      }
      if (beta0 <= A.atol) {
        status = 1;
        break;
      }
      first = false;
    }
    if (beta <= tol) {
      status = 0;
      break;
    }
    if (tid == 0) {
      L.g[0] = beta;
      #pragma unroll 1
      for (int i = 1; i <= m; i++) L.g[i] = 0.0;
    }
    if (mine) {
      const double v0 = L.V[tid] / beta;
      L.V[tid] = v0;
      ps_store(&A.u[c0 + tid], v0);
    }
    if (!ps_barrier(A.ctrl, G, gen, A.budget_ticks)) return;          // v_0 complete
    kused = 0;
    #pragma unroll 1
    for (int j = 0; j < m && alive; j++) {
      // ---- w = B K v_j
      ps_product_l<EPR, RI, EL>(v, c, lv, lc, A.u, L.w, nloc);
      __syncthreads();
      if (mine) L.w[tid] *= L.dinv[tid];
      __syncthreads();
      // ---- Gram-Schmidt coefficients h_i = (v_i, w), i = 0..j: wave wv takes i = wv, wv + 8, ...
      #pragma unroll 1
      for (int i = wv; i <= j; i += PS_NT / 64) {
        double t = 0.0;
        #pragma unroll 1
        for (int r = lane; r < nloc; r += 64) t += L.V[i * PG_ROWS + r] * L.w[r];
        t = tg_wave_sum(t);
        if (lane == 0) ps_store(&Q.pdots[(int64_t)blockIdx.x * (PG_M + 1) + i], t);
      }
      if (!ps_barrier(A.ctrl, G, gen, A.budget_ticks)) return;
      pg_fold(Q.pdots, PG_M + 1, G, j + 1, L.h);
      // ---- w -= sum h_i v_i, ||w||^2
      double ss = 0.0;
      if (mine) {
        double wi = L.w[tid];
        #pragma unroll 1
        for (int i = 0; i <= j; i++) wi -= L.h[i] * L.V[i * PG_ROWS + tid];
        L.w[tid] = wi;
        ss = wi * wi;
      }
      {
        double z0 = 0.0, z1 = 0.0;
        ps_block_sum3(ss, z0, z1, L.red);
        if (tid == 0) ps_store(&Q.pnorm[2 * blockIdx.x], ss);
      }
      if (!ps_barrier(A.ctrl, G, gen, A.budget_ticks)) return;
      pg_fold(Q.pnorm, 2, G, 1, L.h + j + 1);
      // ---- column j of H: earlier rotations, the new one, the residual estimate, the decision (thread 0; k_gm_givens)
      if (tid == 0) {
        double *Hj = L.H + j * (PG_M + 1);
        #pragma unroll 1
        for (int i = 0; i <= j; i++) Hj[i] = L.h[i];
        Hj[j + 1] = sqrt(L.h[j + 1]);
        #pragma unroll 1
        for (int i = 0; i < j; i++) {
          const double t = L.cs[i] * Hj[i] + L.sn[i] * Hj[i + 1];
          Hj[i + 1] = -L.sn[i] * Hj[i] + L.cs[i] * Hj[i + 1];
          Hj[i] = t;
        }
        const double den = hypot(Hj[j], Hj[j + 1]);
        double live = 1.0, st_status = -1.0;
        if (den == 0.0 || !(den == den)) {
          st_status = -2.0;
          live = 0.0;
        } else {
          L.cs[j] = Hj[j] / den;
          L.sn[j] = Hj[j + 1] / den;
          Hj[j] = den;
          Hj[j + 1] = 0.0;
          L.g[j + 1] = -L.sn[j] * L.g[j];
          L.g[j] = L.cs[j] * L.g[j];
          L.st[4] = (double)(its + 1);
          L.st[6] = (double)(j + 1);
          L.st[1] = fabs(L.g[j + 1]);
          if (L.st[1] <= tol) {
            st_status = 0.0;
            live = 0.0;
          } else if (its + 1 >= A.maxit)
            live = 0.0;
        }
        if (st_status == -2.0) {               // breakdown: the iteration does not count
          L.st[4] = (double)its;
          L.st[6] = (double)kused;
          L.st[1] = res;
        }
        L.st[0] = live;
        L.st[5] = st_status;
      }
      __syncthreads();
      its = (int)L.st[4];
      kused = (int)L.st[6];
      res = L.st[1];
      if (L.st[0] == 0.0) {
        alive = false;
        status = (int)L.st[5];
      }
      __syncthreads();
      if (alive && j + 1 < m) {
        // (the last vector of a cycle is never multiplied: the cycle closes below)
        if (mine) {
          const double nrm = sqrt(L.h[j + 1]);
          const double vn = (nrm > 0.0 ? 1.0 / nrm : 0.0) * L.w[tid];
          L.V[(j + 1) * PG_ROWS + tid] = vn;
          ps_store(&A.u[c0 + tid], vn);
        }
        if (!ps_barrier(A.ctrl, G, gen, A.budget_ticks)) return;        // v_{j+1} complete
      }
    }
    // ---- close the cycle: y from the triangular system of the columns used, x += V y
    if (tid == 0) {
      #pragma unroll 1
      for (int i = kused - 1; i >= 0; i--) {
        double t = L.g[i];
        #pragma unroll 1
        for (int cc = i + 1; cc < kused; cc++) t -= L.H[cc * (PG_M + 1) + i] * L.y[cc];
        L.y[i] = t / L.H[i * (PG_M + 1) + i];
      }
    }
    __syncthreads();
    if (mine) {
      double t = L.x[tid];
      #pragma unroll 1
      for (int i = 0; i < kused; i++) t += L.y[i] * L.V[i * PG_ROWS + tid];
      L.x[tid] = t;
    }
    __syncthreads();
  }
  if (mine) A.x[c0 + tid] = L.x[tid];
  if (blockIdx.x == 0 && tid == 0) {
    A.ctrl->out[0] = (double)its;
    A.ctrl->out[1] = res;
    A.ctrl->out[3] = (double)status;
  }
}

// ------------------------------------------------------------------------------------------------------------------
// BiCGStab the same way (tg_bicgstab of tg_krylov.hip: KSPBCGS [ext] with left Jacobi preconditioning, the two fused
// reductions per iteration -- (rhat,v), then (t,s), (t,t), (rhat,s), (rhat,t), (s,s) --, the same breakdown and convergence
// rules and status codes).  The multi-kernel loop reads both reductions back on the HOST (two round trips per iteration,
// 90 us); here every workgroup folds them itself: four device-wide barriers per iteration (p complete | (rhat,v) | s
// complete | the five sums), the workgroup's rows of r, rhat, p, v, s, t, x in LDS.
struct pb_lds {
  double x[PS_ROWS_MAX], r[PS_ROWS_MAX], rhat[PS_ROWS_MAX], p[PS_ROWS_MAX], v[PS_ROWS_MAX], s[PS_ROWS_MAX], t[PS_ROWS_MAX],
      w[PS_ROWS_MAX], dinv[PS_ROWS_MAX];
  double red[3 * 17];
  double f[8];
};

template <int EPR, int RI, int EL>
__global__ void __launch_bounds__(PS_NT) k_bicgstab_persistent(tg_pg_args Q) {
  extern __shared__ double ps_dyn[];
  pb_lds &L = *(pb_lds *)ps_dyn;
  constexpr int ER = EPR - EL;
  double *const lv = ps_dyn + (sizeof(pb_lds) + 7) / 8;           // [RI][EL][PS_NT] values of K held in LDS, then the offsets
  unsigned *const lc = (unsigned *)(lv + (size_t)RI * (EL > 0 ? EL : 0) * PS_NT);
  const tg_ps_args &A = Q.P;
  const unsigned G = gridDim.x;
  unsigned gen = 0;
  const int64_t per = (A.n + G - 1) / G;
  const int64_t c0r = (int64_t)blockIdx.x * per, c0 = c0r < A.n ? c0r : A.n, c1 = c0 + per < A.n ? c0 + per : A.n;
  const int nloc = (int)(c1 - c0);
  const int tid = threadIdx.x, g = tid >> 5, l = tid & 31;
  double v[RI][ER];
  unsigned c[RI][ER];
#pragma unroll
  for (int ri = 0; ri < RI; ri++) {
    const int lrow = g + PS_GROUPS * ri;
    const int64_t row = c0 + lrow;
    const bool live = lrow < nloc;
    const int64_t a = live ? A.rowptr[row] : 0, e = live ? A.rowptr[row + 1] : 0;
    double dd = 0.0;
#pragma unroll
    for (int k = 0; k < EPR; k++) {
      const int64_t q = a + l + 32 * k;
      const bool in = q < e;
      const double vq = in ? A.val[q] : 0.0;
      const int cq = in ? A.col[q] : (int)(live ? row : 0);
      if (k < ER) {
        v[ri][k < ER ? k : 0] = vq;
        c[ri][k < ER ? k : 0] = 8u * (unsigned)cq;
      } else {
        lv[(ri * EL + (k - ER)) * PS_NT + tid] = vq;
        lc[(ri * EL + (k - ER)) * PS_NT + tid] = 8u * (unsigned)cq;
      }
      if (in && cq == row) dd = vq;
    }
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) dd += __shfl_xor(dd, o, 64);
    if (l == 0 && live) L.dinv[lrow] = (A.jacobi && dd != 0.0) ? 1.0 / dd : 1.0;
  }
  __syncthreads();
  // ---- reference norm ||B b||, r = B (b - K x0), rhat = r
  double bn = 0.0;
#pragma unroll 1
  for (int i = tid; i < nloc; i += PS_NT) {
    const double ub = L.dinv[i] * A.b[c0 + i];
    bn += ub * ub;
    L.x[i] = A.nonzero_guess ? A.x[c0 + i] : 0.0;
  }
  if (A.nonzero_guess) {
    ps_product_l<EPR, RI, EL>(v, c, lv, lc, A.x, L.w, nloc);
    __syncthreads();
  }
  double rn = 0.0;
#pragma unroll 1
  for (int i = tid; i < nloc; i += PS_NT) {
    const double ri = L.dinv[i] * (A.nonzero_guess ? A.b[c0 + i] - L.w[i] : A.b[c0 + i]);
    L.r[i] = ri;
    L.rhat[i] = ri;
    L.p[i] = 0.0;
    L.v[i] = 0.0;
    rn += ri * ri;
  }
  {
    double z = 0.0;
    ps_block_sum3(bn, rn, z, L.red);
    if (tid == 0) {
      ps_store(&Q.pnorm[2 * blockIdx.x], bn);
      ps_store(&Q.pnorm[2 * blockIdx.x + 1], rn);
    }
  }
  if (!ps_barrier(A.ctrl, G, gen, A.budget_ticks)) return;
  pg_fold(Q.pnorm, 2, G, 2, L.f);
  const double bnorm = sqrt(L.f[0]);
  double rho = L.f[1], alpha = 1.0, omega = 1.0, beta = 0.0, znorm = sqrt(L.f[1]);
  __syncthreads();
  const double tol = fmax(A.rtol * bnorm, A.atol);
  int its = 0, status = -1;
  if (!(znorm == znorm))
    status = -2;
  else if (znorm <= tol)
    status = (znorm <= A.atol && !(znorm <= A.rtol * bnorm)) ? 1 : 0;
  else {
#pragma unroll 1
    for (int it = 1; it <= A.maxit; it++) {
      // p = r + beta (p - omega v); into the gather buffer
#pragma unroll 1
      for (int i = tid; i < nloc; i += PS_NT) {
        const double pi = L.r[i] + beta * (L.p[i] - omega * L.v[i]);
        L.p[i] = pi;
        ps_store(&A.u[c0 + i], pi);
      }
      if (!ps_barrier(A.ctrl, G, gen, A.budget_ticks)) return;
      ps_product_l<EPR, RI, EL>(v, c, lv, lc, A.u, L.w, nloc);
      __syncthreads();
      double hv = 0.0;
#pragma unroll 1
      for (int i = tid; i < nloc; i += PS_NT) {
        const double vi = L.dinv[i] * L.w[i];
        L.v[i] = vi;
        hv += L.rhat[i] * vi;
      }
      {
        double z0 = 0.0, z1 = 0.0;
        ps_block_sum3(hv, z0, z1, L.red);
        if (tid == 0) ps_store(&Q.pnorm[2 * blockIdx.x], hv);
      }
      if (!ps_barrier(A.ctrl, G, gen, A.budget_ticks)) return;
      pg_fold(Q.pnorm, 2, G, 1, L.f);
      const double rv = L.f[0];
      __syncthreads();
      if (!(rv == rv) || rv == 0.0) {
        status = -2;
        its = it;
        break;
      }
      alpha = rho / rv;
      // s = r - alpha v; into the gather buffer
#pragma unroll 1
      for (int i = tid; i < nloc; i += PS_NT) {
        const double si = L.r[i] - alpha * L.v[i];
        L.s[i] = si;
        ps_store(&A.u[c0 + i], si);
      }
      if (!ps_barrier(A.ctrl, G, gen, A.budget_ticks)) return;
      ps_product_l<EPR, RI, EL>(v, c, lv, lc, A.u, L.w, nloc);
      __syncthreads();
      double ts = 0.0, tt = 0.0, hs = 0.0, ht = 0.0, ss = 0.0;
#pragma unroll 1
      for (int i = tid; i < nloc; i += PS_NT) {
        const double ti = L.dinv[i] * L.w[i], si = L.s[i], hi = L.rhat[i];
        L.t[i] = ti;
        ts += ti * si;
        tt += ti * ti;
        hs += hi * si;
        ht += hi * ti;
        ss += si * si;
      }
      {
        ps_block_sum3(ts, tt, hs, L.red);
        double z = 0.0;
        ps_block_sum3(ht, ss, z, L.red);
        if (tid == 0) {
          double *o = Q.pdots + (int64_t)blockIdx.x * (PG_M + 1);
          ps_store(o + 0, ts);
          ps_store(o + 1, tt);
          ps_store(o + 2, hs);
          ps_store(o + 3, ht);
          ps_store(o + 4, ss);
        }
      }
      if (!ps_barrier(A.ctrl, G, gen, A.budget_ticks)) return;
      pg_fold(Q.pdots, PG_M + 1, G, 5, L.f);
      ts = L.f[0];
      tt = L.f[1];
      hs = L.f[2];
      ht = L.f[3];
      ss = L.f[4];
      __syncthreads();
      omega = tt > 0.0 ? ts / tt : 0.0;
      // x += alpha p + omega s ; r = s - omega t
#pragma unroll 1
      for (int i = tid; i < nloc; i += PS_NT) {
        L.x[i] += alpha * L.p[i] + omega * L.s[i];
        L.r[i] = L.s[i] - omega * L.t[i];
      }
      const double rho_new = hs - omega * ht;
      // (fmax returns the other argument for a NaN: a NaN in the sums -- an Inf in K, say -- must stay one)
      const double rn2raw = (ss - 2.0 * omega * ts + omega * omega * tt) + 0.0 * (ts + tt + hs + ht);
      const double rn2 = rn2raw < 0.0 ? 0.0 : rn2raw;
      znorm = sqrt(rn2);
      its = it;
      if (!(znorm == znorm)) {
        status = -2;
        break;
      }
      if (znorm <= tol) {
        status = (znorm <= A.atol && !(znorm <= A.rtol * bnorm)) ? 1 : 0;
        break;
      }
      if (omega == 0.0 || rho == 0.0 || !(rho_new == rho_new)) {
        status = -2;
        break;
      }
      beta = (rho_new / rho) * (alpha / omega);
      rho = rho_new;
    }
  }
  __syncthreads();
#pragma unroll 1
  for (int i = tid; i < nloc; i += PS_NT) A.x[c0 + i] = L.x[i];
  if (blockIdx.x == 0 && tid == 0) {
    A.ctrl->out[0] = (double)its;
    A.ctrl->out[1] = znorm;
    A.ctrl->out[3] = (double)status;
  }
}

// ------------------------------------------------------------------------------------------------------------------
// CG with the Chebyshev polynomial preconditioner the same way (tg_pcg_cheb of tg_krylov.hip: u = p_m(D^-1 K) D^-1 r by m - 1
// steps of the Chebyshev iteration on [theta - delta, theta + delta] -- the interval comes from the caller's Lanczos steps --,
// then the single-reduction CG recurrence).  The inner steps need no inner product: ONE device-wide barrier per product (the
// iterate complete), the gather vector in two buffers used alternately so that nobody overwrites what a slower workgroup
// still reads; the outer step adds the barrier of the three sums.
struct pc_lds {
  double x[PS_ROWS_MAX], r[PS_ROWS_MAX], p[PS_ROWS_MAX], s[PS_ROWS_MAX], u[PS_ROWS_MAX], w[PS_ROWS_MAX], dinv[PS_ROWS_MAX],
      g[PS_ROWS_MAX], d[PS_ROWS_MAX];
  double red[3 * 17];
  double f[8];
};
struct tg_pc_args {
  tg_ps_args P;
  int m;
  double theta, delta;
  double *u2;                     // the second gather buffer
  double *psum;                   // [G][4] partial sums
};

template <int EPR, int RI>
__global__ void __launch_bounds__(PS_NT) k_pcg_cheb_persistent(tg_pc_args Q) {
  extern __shared__ double ps_dyn[];
  pc_lds &L = *(pc_lds *)ps_dyn;
  const tg_ps_args &A = Q.P;
  const unsigned G = gridDim.x;
  unsigned gen = 0;
  const int64_t per = (A.n + G - 1) / G;
  const int64_t c0r = (int64_t)blockIdx.x * per, c0 = c0r < A.n ? c0r : A.n, c1 = c0 + per < A.n ? c0 + per : A.n;
  const int nloc = (int)(c1 - c0);
  const int tid = threadIdx.x, g = tid >> 5, l = tid & 31;
  double v[RI][EPR];
  unsigned c[RI][EPR];
#pragma unroll
  for (int ri = 0; ri < RI; ri++) {
    const int lrow = g + PS_GROUPS * ri;
    const int64_t row = c0 + lrow;
    const bool live = lrow < nloc;
    const int64_t a = live ? A.rowptr[row] : 0, e = live ? A.rowptr[row + 1] : 0;
    double dd = 0.0;
#pragma unroll
    for (int k = 0; k < EPR; k++) {
      const int64_t q = a + l + 32 * k;
      const bool in = q < e;
      v[ri][k] = in ? A.val[q] : 0.0;
      const int cq = in ? A.col[q] : (int)(live ? row : 0);
      c[ri][k] = 8u * (unsigned)cq;
      if (in && cq == row) dd = v[ri][k];
    }
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) dd += __shfl_xor(dd, o, 64);
    if (l == 0 && live) L.dinv[lrow] = dd != 0.0 ? 1.0 / dd : 1.0;       // (the polynomial is in D^-1 K: always Jacobi)
  }
  __syncthreads();
  int flip = 0;                                    // which gather buffer the next write goes to
  double *const gb[2] = {A.u, Q.u2};
  const double inv_theta = 1.0 / Q.theta, sigma = Q.theta / Q.delta;
  bool ok = true;
  // u = B r on the own rows (L.u); every product reads the iterate of ALL rows from a gather buffer
  auto apply_pc = [&]() {
#pragma unroll 1
    for (int i = tid; i < nloc; i += PS_NT) {
      const double gi = L.dinv[i] * L.r[i];
      L.g[i] = gi;
      const double zi = gi * inv_theta;
      L.u[i] = zi;
      L.d[i] = zi;
    }
    double rho = 1.0 / sigma;
#pragma unroll 1
    for (int j = 1; j < Q.m && ok; j++) {
      const double rho_new = 1.0 / (2.0 * sigma - rho);
      const double cc1 = rho_new * rho, cc2 = 2.0 * rho_new / Q.delta;
      double *buf = gb[flip];
      flip ^= 1;
#pragma unroll 1
      for (int i = tid; i < nloc; i += PS_NT) ps_store(&buf[c0 + i], L.u[i]);
      if (!ps_barrier(A.ctrl, G, gen, A.budget_ticks)) {
        ok = false;
        break;
      }
      ps_product<EPR, RI>(v, c, buf, L.w, nloc);
      __syncthreads();
#pragma unroll 1
      for (int i = tid; i < nloc; i += PS_NT) {
        const double di = cc1 * L.d[i] + cc2 * (L.g[i] - L.dinv[i] * L.w[i]);
        L.d[i] = di;
        L.u[i] += di;
      }
      rho = rho_new;
    }
    __syncthreads();
  };
  // w = K y on the own rows (y of all rows through a gather buffer)
  auto product_of = [&](const double *own) {
    double *buf = gb[flip];
    flip ^= 1;
#pragma unroll 1
    for (int i = tid; i < nloc; i += PS_NT) ps_store(&buf[c0 + i], own[i]);
    if (!ps_barrier(A.ctrl, G, gen, A.budget_ticks)) {
      ok = false;
      return;
    }
    ps_product<EPR, RI>(v, c, buf, L.w, nloc);
    __syncthreads();
  };
  // ---- reference norm ||B b||: u = B b
#pragma unroll 1
  for (int i = tid; i < nloc; i += PS_NT) {
    L.r[i] = A.b[c0 + i];
    L.x[i] = A.nonzero_guess ? A.x[c0 + i] : 0.0;
    L.p[i] = 0.0;
    L.s[i] = 0.0;
  }
  __syncthreads();
  apply_pc();
  if (!ok) return;
  double bn = 0.0;
#pragma unroll 1
  for (int i = tid; i < nloc; i += PS_NT) bn += L.u[i] * L.u[i];
  {
    double z0 = 0.0, z1 = 0.0;
    ps_block_sum3(bn, z0, z1, L.red);
    if (tid == 0) ps_store(&Q.psum[4 * blockIdx.x + 3], bn);
  }
  if (A.nonzero_guess) {
    product_of(L.x);
    if (!ok) return;
#pragma unroll 1
    for (int i = tid; i < nloc; i += PS_NT) L.r[i] = A.b[c0 + i] - L.w[i];
    __syncthreads();
    apply_pc();
    if (!ok) return;
  }
  double gamma_prev = 1.0, alpha_prev = 1.0, znorm = 0.0, tol = 0.0, bnorm = 0.0;
  int its = 0, status = -1;
#pragma unroll 1
  for (int it = 0; it <= A.maxit; it++) {
    product_of(L.u);
    if (!ok) return;
    double a = 0.0, b = 0.0, cc = 0.0;
#pragma unroll 1
    for (int i = tid; i < nloc; i += PS_NT) {
      const double ui = L.u[i];
      a += L.r[i] * ui;
      b += L.w[i] * ui;
      cc += ui * ui;
    }
    ps_block_sum3(a, b, cc, L.red);
    if (tid == 0) {
      ps_store(&Q.psum[4 * blockIdx.x], a);
      ps_store(&Q.psum[4 * blockIdx.x + 1], b);
      ps_store(&Q.psum[4 * blockIdx.x + 2], cc);
    }
    if (!ps_barrier(A.ctrl, G, gen, A.budget_ticks)) return;
    pg_fold(Q.psum, 4, G, 4, L.f);
    const double gamma = L.f[0], delta_ = L.f[1], nu = L.f[2];
    if (it == 0) {
      bnorm = sqrt(L.f[3]);
      tol = fmax(A.rtol * bnorm, A.atol);
    }
    __syncthreads();
    znorm = sqrt(nu);
    its = it;
    if (!(nu == nu) || !(gamma == gamma)) {
      status = -2;
      break;
    }
    if (znorm <= tol) {
      status = (znorm <= A.atol && !(znorm <= A.rtol * bnorm)) ? 1 : 0;
      break;
    }
    if (it == A.maxit) break;
    double beta = 0.0, alpha;
    if (it == 0)
      alpha = gamma / delta_;
    else {
      beta = gamma / gamma_prev;
      alpha = gamma / (delta_ - beta * gamma / alpha_prev);
    }
    if (!(alpha == alpha) || alpha == 0.0 || !(gamma > 0.0)) {
      status = -2;
      break;
    }
    gamma_prev = gamma;
    alpha_prev = alpha;
#pragma unroll 1
    for (int i = tid; i < nloc; i += PS_NT) {
      const double pi = L.u[i] + beta * L.p[i];
      const double si = L.w[i] + beta * L.s[i];
      L.p[i] = pi;
      L.s[i] = si;
      L.x[i] += alpha * pi;
      L.r[i] -= alpha * si;
    }
    __syncthreads();
    apply_pc();
    if (!ok) return;
  }
#pragma unroll 1
  for (int i = tid; i < nloc; i += PS_NT) A.x[c0 + i] = L.x[i];
  if (blockIdx.x == 0 && tid == 0) {
    A.ctrl->out[0] = (double)its;
    A.ctrl->out[1] = znorm;
    A.ctrl->out[3] = (double)status;
  }
}

// Whether a system is taken by the persistent loop: one rank, at least a thousand rows (below, the launches are not what
// the solve costs), rows of at most 128 entries, and all of K in the registers of one workgroup per CU
// (TIGAR_KSP_PERSISTENT=0 turns it off, =1 lifts the lower limit).
static const int PS_RI[5][4] = {{16, 32, 48, 56}, {8, 16, 24, 28}, {6, 12, 17, 18}, {4, 8, 12, 14}, {4, 8, 11, 13}};   // rows per group of lanes, by EPR
static bool ps_shape(const tg_csr_s *k, int64_t rows_max, int *epr_out, int *ri_out, int *g_out) {
  const int64_t n = k->nrows;
  const int maxlen = k->max_row_nnz;
  if (maxlen < 1 || maxlen > 160) return false;
  const int epr = (maxlen + 31) / 32;
  int G = std::min<int>(g_tg.num_cu, PS_MAXG);
  G = (int)std::min<int64_t>(G, std::max<int64_t>(1, tg_cdiv(n, PS_GROUPS)));
  const int64_t per = tg_cdiv(n, G);
  if (per > rows_max) return false;
  for (int q = 0; q < 4; q++)
    if (per <= (int64_t)PS_RI[epr - 1][q] * PS_GROUPS) {
      *epr_out = epr;
      *ri_out = PS_RI[epr - 1][q];
      *g_out = G;
      return true;
    }
  return false;
}

bool tg_cg_persistent_applies(const tg_csr_s *k) {
  const int mode = getenv("TIGAR_KSP_PERSISTENT") ? atoi(getenv("TIGAR_KSP_PERSISTENT")) : -1;
  if (mode == 0 || k->nrows < 1 || k->nrows != k->ncols || k->nnz < 1) return false;
  return mode == 1 || k->nrows >= 1024;
}

// Returns 0 with the results set, 100 when the kernel could not be used (K beyond the registers, workgroups not resident,
// barrier time-out): the caller runs the multi-kernel loop instead.
int tg_cg_persistent(tg_csr_s *k, tg_vec_s *b, tg_vec_s *x, int pc, double rtol, double atol, int maxit, int nonzero_guess,
                     int *iters, double *resnorm, int *status) {
  const int64_t n = k->nrows;
  TG_TRY(tg_spmv_plan(k));                   // (longest row)
  int epr = 0, ri = 0, G = 0;
  if (!ps_shape(k, PS_ROWS_MAX, &epr, &ri, &G)) return 100;
  double *buf = nullptr;
  const int64_t ctrl_doubles = (int64_t)(sizeof(tg_ps_ctrl) + 7) / 8 + 16;
  TG_TRY(tg_dmalloc(&buf, n + 4 * (int64_t)G + 16 + ctrl_doubles));
  tg_ps_args A;
  memset(&A, 0, sizeof(A));
  A.rowptr = k->rowptr;
  A.col = k->col;
  A.val = k->val;
  A.n = n;
  A.b = b->d;
  A.x = x->d;
  A.u = buf;
  A.partial = buf + n;
  A.ctrl = (tg_ps_ctrl *)(((uintptr_t)(buf + n + 4 * (int64_t)G) + 127) & ~(uintptr_t)127);
  A.rtol = rtol;
  A.atol = atol;
  A.maxit = maxit;
  A.jacobi = pc == TG_PC_JACOBI ? 1 : 0;
  A.nonzero_guess = nonzero_guess;
  A.budget_ticks = 100000000ll * 5;          // wall_clock64 runs at 100 MHz: 5 s at one barrier
  hipMemsetAsync(A.ctrl, 0, sizeof(tg_ps_ctrl), g_tg.stream);
  void *params[] = {&A};
  const void *fn = nullptr;
#define PS_PICK(E, R) \
  if (epr == E && ri == R) fn = (const void *)k_cg_persistent<E, R>
  PS_PICK(1, 16); PS_PICK(1, 32); PS_PICK(1, 48); PS_PICK(1, 56);
  PS_PICK(2, 8); PS_PICK(2, 16); PS_PICK(2, 24); PS_PICK(2, 28);
  PS_PICK(3, 6); PS_PICK(3, 12); PS_PICK(3, 17); PS_PICK(3, 18);
  PS_PICK(4, 4); PS_PICK(4, 8); PS_PICK(4, 12); PS_PICK(4, 14);
  PS_PICK(5, 4); PS_PICK(5, 8); PS_PICK(5, 11); PS_PICK(5, 13);
#undef PS_PICK
  if (!fn) {
    tg_dfree(buf);
    return 100;
  }
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  hipEventCreate(&ev0);
  hipEventCreate(&ev1);
  hipEventRecord(ev0, g_tg.stream);
  hipError_t e = hipLaunchCooperativeKernel(fn, dim3((unsigned)G), dim3(PS_NT), params, sizeof(ps_lds), g_tg.stream);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    hipEventDestroy(ev0);
    hipEventDestroy(ev1);
    tg_dfree(buf);
    return 100;
  }
  hipEventRecord(ev1, g_tg.stream);
  tg_ps_ctrl h;
  e = hipMemcpyAsync(&h, A.ctrl, sizeof(h), hipMemcpyDeviceToHost, g_tg.stream);
  const hipError_t e2 = hipStreamSynchronize(g_tg.stream);
  float ems = 0.f;
  if (e2 == hipSuccess && hipEventElapsedTime(&ems, ev0, ev1) != hipSuccess) ems = 0.f;
  hipEventDestroy(ev0);
  hipEventDestroy(ev1);
  tg_dfree(buf);
  if (e != hipSuccess || e2 != hipSuccess) {
    tg_set_error("persistent CG: %s", hipGetErrorString(e2 != hipSuccess ? e2 : e));
    return 1;
  }
  if (h.abort_flag) return 100;
  if (getenv("TIGAR_TRACE"))
    fprintf(stderr, "[trace] persistent cg: %d its, workgroup 0 per iteration: product+dots %.2f us, barrier %.2f us, fold+update %.2f us, "
            "barrier %.2f us (%d workgroups, %d entries per lane and row, %d rows per group of lanes)\n", (int)h.out[0], h.out[4] / 100.0 / std::max(1.0, h.out[0]),
            h.out[5] / 100.0 / std::max(1.0, h.out[0]), h.out[6] / 100.0 / std::max(1.0, h.out[0]),
            h.out[7] / 100.0 / std::max(1.0, h.out[0]), G, epr, ri);
  g_tg.prof_n[TG_PROF_KSP_PERSISTENT] += 1;
  g_tg.prof_ms[TG_PROF_KSP_PERSISTENT] += ems;
  g_tg.prof_n[TG_PROF_KSP_SPMV] += (int64_t)h.out[0] + 1 + (nonzero_guess ? 1 : 0);
  g_tg.prof_ms[TG_PROF_KSP_SPMV] += ems;
  *iters = (int)h.out[0];
  *resnorm = sqrt(h.out[1]);
  *status = (int)h.out[3];
  return 0;
}

// GMRES(m), m <= 30, one rank, no stagnation guard (restart >= 1), or BiCGStab (restart = 0): as tg_cg_persistent
// (100 = not taken, the caller runs tg_gmres / tg_bicgstab)
static int ps_run_pg(bool bicgstab, tg_csr_s *k, tg_vec_s *b, tg_vec_s *x, int pc, double rtol, double atol, int maxit, int restart,
                     int nonzero_guess, int *iters, double *resnorm, int *status) {
  const int64_t n = k->nrows;
  if (!bicgstab && (restart < 1 || restart > PG_M)) return 100;
  TG_TRY(tg_spmv_plan(k));
  int epr = 0, ri = 0, G = 0;
  if (!ps_shape(k, bicgstab ? PS_ROWS_MAX : PG_ROWS, &epr, &ri, &G)) return 100;
  size_t lds_bytes = bicgstab ? sizeof(pb_lds) : sizeof(pg_lds);
  double *buf = nullptr;
  const int64_t ctrl_doubles = (int64_t)(sizeof(tg_ps_ctrl) + 7) / 8 + 16;
  const int64_t npart = (int64_t)G * (PG_M + 1) + 2 * (int64_t)G;
  TG_TRY(tg_dmalloc(&buf, n + npart + 16 + ctrl_doubles));
  tg_pg_args Q;
  memset(&Q, 0, sizeof(Q));
  tg_ps_args &A = Q.P;
  A.rowptr = k->rowptr;
  A.col = k->col;
  A.val = k->val;
  A.n = n;
  A.b = b->d;
  A.x = x->d;
  A.u = buf;
  Q.pdots = buf + n;
  Q.pnorm = Q.pdots + (int64_t)G * (PG_M + 1);
  A.ctrl = (tg_ps_ctrl *)(((uintptr_t)(buf + n + npart) + 127) & ~(uintptr_t)127);
  A.rtol = rtol;
  A.atol = atol;
  A.maxit = maxit;
  A.jacobi = pc == TG_PC_JACOBI ? 1 : 0;
  A.nonzero_guess = nonzero_guess;
  A.budget_ticks = 100000000ll * 5;
  Q.m = restart;
  hipMemsetAsync(A.ctrl, 0, sizeof(tg_ps_ctrl), g_tg.stream);
  void *params[] = {&Q};
  const void *fn = nullptr;
  int el = 0;                                // layers of K in LDS (GMRES with rows of 129-160 entries)
#define PS_PICK(E, R)                                                                                  \
  if (epr == E && ri == R) {                                                                           \
    if (bicgstab && E == 5 && R >= 11) {                                                               \
      fn = (const void *)k_bicgstab_persistent<E, R, (E == 5 && R >= 11) ? 1 : 0>;                     \
      el = 1;                                                                                          \
    } else if (bicgstab)                                                                               \
      fn = (const void *)k_bicgstab_persistent<E, R, 0>;                                               \
    else if (E == 5 && R >= 11) {                                                                      \
      fn = (const void *)k_gmres_persistent<E, R, (E == 5 && R >= 11) ? 1 : 0>;                        \
      el = 1;                                                                                          \
    } else                                                                                             \
      fn = (const void *)k_gmres_persistent<E, R, 0>;                                                  \
  }
  PS_PICK(1, 16); PS_PICK(1, 32); PS_PICK(1, 48); PS_PICK(1, 56);
  PS_PICK(2, 8); PS_PICK(2, 16); PS_PICK(2, 24); PS_PICK(2, 28);
  PS_PICK(3, 6); PS_PICK(3, 12); PS_PICK(3, 17); PS_PICK(3, 18);
  PS_PICK(4, 4); PS_PICK(4, 8); PS_PICK(4, 12); PS_PICK(4, 14);
  PS_PICK(5, 4); PS_PICK(5, 8); PS_PICK(5, 11); PS_PICK(5, 13);
#undef PS_PICK
  if (el) lds_bytes = (((bicgstab ? sizeof(pb_lds) : sizeof(pg_lds)) + 7) / 8) * 8 + (size_t)ri * el * PS_NT * 12;
  if (!fn || lds_bytes > 160 * 1024 || hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes) != hipSuccess) {
    (void)hipGetLastError();
    tg_dfree(buf);
    return 100;
  }
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  hipEventCreate(&ev0);
  hipEventCreate(&ev1);
  hipEventRecord(ev0, g_tg.stream);
  hipError_t e = hipLaunchCooperativeKernel(fn, dim3((unsigned)G), dim3(PS_NT), params, lds_bytes, g_tg.stream);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    hipEventDestroy(ev0);
    hipEventDestroy(ev1);
    tg_dfree(buf);
    return 100;
  }
  hipEventRecord(ev1, g_tg.stream);
  tg_ps_ctrl h;
  e = hipMemcpyAsync(&h, A.ctrl, sizeof(h), hipMemcpyDeviceToHost, g_tg.stream);
  const hipError_t e2 = hipStreamSynchronize(g_tg.stream);
  float ems = 0.f;
  if (e2 == hipSuccess && hipEventElapsedTime(&ems, ev0, ev1) != hipSuccess) ems = 0.f;
  hipEventDestroy(ev0);
  hipEventDestroy(ev1);
  tg_dfree(buf);
  if (e != hipSuccess || e2 != hipSuccess) {
    tg_set_error("persistent %s: %s", bicgstab ? "BiCGStab" : "GMRES", hipGetErrorString(e2 != hipSuccess ? e2 : e));
    return 1;
  }
  if (h.abort_flag) return 100;
  if (getenv("TIGAR_TRACE"))
    fprintf(stderr, "[trace] persistent %s(%d): %d its in %.3f ms (%d workgroups, %d entries per lane and row, %d rows per group)\n",
            bicgstab ? "bicgstab" : "gmres", restart, (int)h.out[0], ems, G, epr, ri);
  g_tg.prof_n[TG_PROF_KSP_PERSISTENT] += 1;
  g_tg.prof_ms[TG_PROF_KSP_PERSISTENT] += ems;
  g_tg.prof_n[TG_PROF_KSP_SPMV] += (bicgstab ? 2 : 1) * (int64_t)h.out[0] + 1;
  g_tg.prof_ms[TG_PROF_KSP_SPMV] += ems;
  *iters = (int)h.out[0];
  *resnorm = h.out[1];
  *status = (int)h.out[3];
  return 0;
}

int tg_gmres_persistent(tg_csr_s *k, tg_vec_s *b, tg_vec_s *x, int pc, double rtol, double atol, int maxit, int restart,
                        int nonzero_guess, int *iters, double *resnorm, int *status) {
  return ps_run_pg(false, k, b, x, pc, rtol, atol, maxit, restart, nonzero_guess, iters, resnorm, status);
}
int tg_bicgstab_persistent(tg_csr_s *k, tg_vec_s *b, tg_vec_s *x, int pc, double rtol, double atol, int maxit, int nonzero_guess,
                           int *iters, double *resnorm, int *status) {
  return ps_run_pg(true, k, b, x, pc, rtol, atol, maxit, 0, nonzero_guess, iters, resnorm, status);
}


// CG with the Chebyshev polynomial preconditioner of degree m on [theta - delta, theta + delta] (from tg_pcg_cheb's Lanczos
// steps); 100 = not taken
int tg_pcg_cheb_persistent(tg_csr_s *k, tg_vec_s *b, tg_vec_s *x, int m, double theta, double delta, double rtol, double atol,
                           int maxit, int nonzero_guess, int *iters, double *resnorm, int *status) {
  const int64_t n = k->nrows;
  TG_TRY(tg_spmv_plan(k));
  int epr = 0, ri = 0, G = 0;
  if (!ps_shape(k, PS_ROWS_MAX, &epr, &ri, &G) || !(delta > 0.0) || !(theta > delta)) return 100;
  double *buf = nullptr;
  const int64_t ctrl_doubles = (int64_t)(sizeof(tg_ps_ctrl) + 7) / 8 + 16;
  TG_TRY(tg_dmalloc(&buf, 2 * n + 4 * (int64_t)G + 16 + ctrl_doubles));
  tg_pc_args Q;
  memset(&Q, 0, sizeof(Q));
  tg_ps_args &A = Q.P;
  A.rowptr = k->rowptr;
  A.col = k->col;
  A.val = k->val;
  A.n = n;
  A.b = b->d;
  A.x = x->d;
  A.u = buf;
  Q.u2 = buf + n;
  Q.psum = buf + 2 * n;
  A.ctrl = (tg_ps_ctrl *)(((uintptr_t)(buf + 2 * n + 4 * (int64_t)G) + 127) & ~(uintptr_t)127);
  A.rtol = rtol;
  A.atol = atol;
  A.maxit = maxit;
  A.jacobi = 1;
  A.nonzero_guess = nonzero_guess;
  A.budget_ticks = 100000000ll * 5;
  Q.m = m;
  Q.theta = theta;
  Q.delta = delta;
  hipMemsetAsync(A.ctrl, 0, sizeof(tg_ps_ctrl), g_tg.stream);
  void *params[] = {&Q};
  const void *fn = nullptr;
#define PS_PICK(E, R) \
  if (epr == E && ri == R) fn = (const void *)k_pcg_cheb_persistent<E, R>
  PS_PICK(1, 16); PS_PICK(1, 32); PS_PICK(1, 48); PS_PICK(1, 56);
  PS_PICK(2, 8); PS_PICK(2, 16); PS_PICK(2, 24); PS_PICK(2, 28);
  PS_PICK(3, 6); PS_PICK(3, 12); PS_PICK(3, 17); PS_PICK(3, 18);
  PS_PICK(4, 4); PS_PICK(4, 8); PS_PICK(4, 12); PS_PICK(4, 14);
  PS_PICK(5, 4); PS_PICK(5, 8); PS_PICK(5, 11); PS_PICK(5, 13);
#undef PS_PICK
  if (!fn || hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(pc_lds)) != hipSuccess) {
    (void)hipGetLastError();
    tg_dfree(buf);
    return 100;
  }
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  hipEventCreate(&ev0);
  hipEventCreate(&ev1);
  hipEventRecord(ev0, g_tg.stream);
  hipError_t e = hipLaunchCooperativeKernel(fn, dim3((unsigned)G), dim3(PS_NT), params, sizeof(pc_lds), g_tg.stream);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    hipEventDestroy(ev0);
    hipEventDestroy(ev1);
    tg_dfree(buf);
    return 100;
  }
  hipEventRecord(ev1, g_tg.stream);
  tg_ps_ctrl h;
  e = hipMemcpyAsync(&h, A.ctrl, sizeof(h), hipMemcpyDeviceToHost, g_tg.stream);
  const hipError_t e2 = hipStreamSynchronize(g_tg.stream);
  float ems = 0.f;
  if (e2 == hipSuccess && hipEventElapsedTime(&ems, ev0, ev1) != hipSuccess) ems = 0.f;
  hipEventDestroy(ev0);
  hipEventDestroy(ev1);
  tg_dfree(buf);
  if (e != hipSuccess || e2 != hipSuccess) {
    tg_set_error("persistent Chebyshev-CG: %s", hipGetErrorString(e2 != hipSuccess ? e2 : e));
    return 1;
  }
  if (h.abort_flag) return 100;
  if (getenv("TIGAR_TRACE"))
    fprintf(stderr, "[trace] persistent chebyshev(%d)-cg: %d its in %.3f ms (%d workgroups)\n", m, (int)h.out[0], ems, G);
  g_tg.prof_n[TG_PROF_KSP_PERSISTENT] += 1;
  g_tg.prof_ms[TG_PROF_KSP_PERSISTENT] += ems;
  g_tg.prof_n[TG_PROF_KSP_SPMV] += ((int64_t)h.out[0] + 2) * m;
  g_tg.prof_ms[TG_PROF_KSP_SPMV] += ems;
  *iters = (int)h.out[0];
  *resnorm = h.out[1];
  *status = (int)h.out[3];
  return 0;
}
