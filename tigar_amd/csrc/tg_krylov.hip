// Krylov solve of K U = M^T b  (solveLinearSystem, tIGAr/common.py:1236-1263; solver seam
// b-4 `self.linearSolver.solve(MTAM, MTU, MTb)`, common.py:1255-1258).
//
// Semantics restated from PETSc KSP as configured by dolfin's PETScKrylovSolver [ext]:
// zero initial guess, left preconditioning, convergence on the PRECONDITIONED residual
// 2-norm  ||B r_k|| <= max(rtol * ||B b||, atol).  CG (KSPCG) and restarted GMRES
// (KSPGMRES, classical Gram-Schmidt, restart 30) with PCJACOBI / PCNONE.
//
// Device design: everything stays in HBM; scalars (alpha, beta) live on the device so the
// only host round trip per iteration is the 8-byte residual norm.  Per CG iteration:
//   halo exchange (multi-GPU)  ->  SpMV  ->  p.Kp partials on a fixed grid -> fold (+all-reduce)
//   ->  fused x/r update + r.z and z.z partials  ->  fold (+all-reduce)  ->  p update.
// Vector traffic is 11 doubles per dof on top of the 12 B/nnz of the SpMV.
#include "tg_dist.h"
#include <math.h>
#include <algorithm>
#include <chrono>

// The products of one solve go through the sliced, pattern-compressed copy of tg_sell.hip when the
// matrix has the structure (built here, dropped when the solve returns: the copy is a snapshot of
// the values).  A copy requested by the caller (tg_spmv_sell) is used and left alone.
struct tg_sell_guard {
  tg_csr_s *k;
  bool temp = false;
  int rc = 0;
  explicit tg_sell_guard(tg_csr_s *m, bool skip = false) : k(m) {
    if (k->sell_state == 0 && !skip) {
      rc = tg_sell_plan(k);
      temp = true;
    }
  }
  ~tg_sell_guard() {
    if (temp) {
      tg_sell_drop(k);
      k->sell_state = 0;
    }
  }
};

static int g_ksp_symmetric_hint = 0;      // TG_KSP_SYMMETRIC of the solve under way (tg_krylov_solve_flags)
// CG only (a symmetric K is its premise): the half-storage copy of tg_symgrid.hip when K is a box stencil on a 3-D grid
// and this rank holds all of it; built for the solve like the sliced copy, which is then not needed.
struct tg_symgrid_guard {
  tg_symgrid_s *s = nullptr;
  ~tg_symgrid_guard() { tg_symgrid_free(s); }
};

static double tk_now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

#define TG_VEC_BLOCKS 1024

// folds nb partials of `nstreams` interleaved streams into out[0..nstreams)
__global__ void __launch_bounds__(256) k_fold(const double *partial, int nb, int nstreams, double *out) {
  __shared__ double lds4[4];
  for (int k = 0; k < nstreams; k++) {
    double s = 0.0;
    for (int b = threadIdx.x; b < nb; b += 256) s += partial[(int64_t)b * nstreams + k];
    s = tg_block_sum256(s, lds4);
    if (threadIdx.x == 0) out[k] = s;
  }
}

// dinv from a recorded diagonal (tg_csr_s::diag_cache)
__global__ void __launch_bounds__(256) k_jacobi_from_diag(const double *__restrict__ diag, int64_t n, int use_jacobi,
                                                         double *__restrict__ dinv) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    const double d = diag[i];
    dinv[i] = (use_jacobi && d != 0.0) ? 1.0 / d : 1.0;
  }
}

__global__ void __launch_bounds__(256) k_jacobi_setup(const int64_t *__restrict__ rowptr,
                                                      const int32_t *__restrict__ col,
                                                      const double *__restrict__ val, int64_t nrows, int64_t row0,
                                                      int use_jacobi, double *__restrict__ dinv) {
  // PCJACOBI [ext]: inverse of the diagonal, 1 where the diagonal is zero / absent
  const int lane = threadIdx.x & 63;
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t r = wave; r < nrows; r += nwaves) {
    double d = 0.0;
    if (use_jacobi)
      for (int64_t q = rowptr[r] + lane; q < rowptr[r + 1]; q += 64)
        if (col[q] == r + row0) d = val[q];
    d = tg_wave_sum(d);
    if (lane == 0) dinv[r] = (use_jacobi && d != 0.0) ? 1.0 / d : 1.0;
  }
}

// ---------------------------------------------------------------------------------------- CG
// Single-reduction CG (Chronopoulos & Gear 1989; PETSc: KSPCG with -ksp_cg_single_reduction [ext]): the
// same iterates as the textbook recurrence in exact arithmetic, but the three inner products of an
// iteration -- gamma = (r,u), delta = (K u, u), nu = (u,u), u = B r -- are available at one point, so a
// multi-GPU iteration costs ONE all-reduce of three doubles (24 B) instead of two, and nothing in the loop
// needs the host: step lengths are computed on the device from the reduced scalars by every workgroup of
// the update kernel, which also freezes the state once nu <= tol^2.  The host enqueues iterations ahead of
// the device and reads the norm history two iterations late (pinned ring + events); after convergence the
// at most two extra iterations in flight are no-ops, so x is the iterate of the converged iteration.
//
//   u = B r, w = K u                                  (r = b - K x0)
//   p = u + beta p ; s = w + beta s                   (s = K p)
//   x += alpha p ; r -= alpha s ; u = B r ; w = K u
//   beta' = gamma'/gamma ; alpha' = gamma' / (delta' - beta' gamma' / alpha)
struct tg_cg_scal {     // device-resident scalars of one parity
  double gamma, delta, nu;       // reduced over blocks (and ranks) for the CURRENT u
  double gamma_prev, alpha_prev; // state of the recurrence
};

// r = b - kx0 (kx0 may be null) ; u = dinv r ; p = s = 0 ; partials of (r,u), (u,u)
__global__ void __launch_bounds__(256) k_cg1_init(const double *__restrict__ b, const double *__restrict__ kx0,
                                                  const double *__restrict__ dinv, double *__restrict__ r,
                                                  double *__restrict__ u, double *__restrict__ p,
                                                  double *__restrict__ s, int64_t n, double *__restrict__ partial) {
  __shared__ double lds4[4];
  double g = 0.0, nu = 0.0;
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    const double ri = kx0 ? b[i] - kx0[i] : b[i];
    const double ui = dinv[i] * ri;
    r[i] = ri;
    u[i] = ui;
    p[i] = 0.0;
    s[i] = 0.0;
    g += ri * ui;
    nu += ui * ui;
  }
  g = tg_block_sum256(g, lds4);
  nu = tg_block_sum256(nu, lds4);
  if (threadIdx.x == 0) {
    partial[2 * blockIdx.x] = g;
    partial[2 * blockIdx.x + 1] = nu;
  }
}

// partial of (w, u)
__global__ void __launch_bounds__(256) k_cg1_dot(const double *__restrict__ w, const double *__restrict__ u, int64_t n,
                                                 double *__restrict__ partial) {
  __shared__ double lds4[4];
  double d = 0.0;
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) d += w[i] * u[i];
  d = tg_block_sum256(d, lds4);
  if (threadIdx.x == 0) partial[blockIdx.x] = d;
}

// folds the (gamma, nu) partials of the update kernel and the delta partials of the dot kernel into cur->{gamma,delta,nu}
// (`hist`: the history entry of the iteration in pinned host memory, written here when one rank runs alone -- a
// device-to-host copy of 24 bytes is a blit kernel of its own, 4 us per iteration)
__global__ void __launch_bounds__(256) k_cg1_fold(const double *__restrict__ part_gn, const double *__restrict__ part_d,
                                                  int nb, tg_cg_scal *cur, double *hist) {
  __shared__ double lds4[4];
  double g = 0.0, nu = 0.0, d = 0.0;
  for (int b = threadIdx.x; b < nb; b += 256) {
    g += part_gn[2 * b];
    nu += part_gn[2 * b + 1];
    d += part_d[b];
  }
  g = tg_block_sum256(g, lds4);
  d = tg_block_sum256(d, lds4);
  nu = tg_block_sum256(nu, lds4);
  if (threadIdx.x == 0) {
    cur->gamma = g;
    cur->delta = d;
    cur->nu = nu;
    if (hist) {
      hist[0] = g;
      hist[1] = d;
      hist[2] = nu;
    }
  }
}

// one CG step from the scalars of `cur`; writes the recurrence state of the next parity into `nxt`
__global__ void __launch_bounds__(256)
    k_cg1_update(const tg_cg_scal *__restrict__ cur, tg_cg_scal *__restrict__ nxt, double tol2, int first, int lead,
                 const double *__restrict__ w, const double *__restrict__ dinv, double *__restrict__ u,
                 double *__restrict__ p, double *__restrict__ s, double *__restrict__ x, double *__restrict__ r,
                 int64_t n, double *__restrict__ partial) {
  __shared__ double lds4[4];
  const double gamma = cur->gamma, delta = cur->delta, nu = cur->nu;
  // converged (or broken down: NaN) before this step: keep everything as it is
  const bool frozen = !(nu > tol2);
  double beta = 0.0, alpha;
  if (first)
    alpha = gamma / delta;
  else {
    beta = gamma / cur->gamma_prev;
    alpha = gamma / (delta - beta * gamma / cur->alpha_prev);
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    nxt->gamma_prev = frozen ? cur->gamma_prev : gamma;
    nxt->alpha_prev = frozen ? cur->alpha_prev : alpha;
  }
  if (frozen) {
    // the fold AND the all-reduce of the next parity must reproduce the same gamma, nu: they are already summed over
    // the ranks, so only the leading rank hands them on (every other rank contributes zeros; all ranks take the same
    // branch because they compare the same reduced nu)
    if (threadIdx.x == 0) {
      const bool carrier = lead && blockIdx.x == 0;
      partial[2 * blockIdx.x] = carrier ? gamma : 0.0;
      partial[2 * blockIdx.x + 1] = carrier ? nu : 0.0;
    }
    return;
  }
  double g = 0.0, nn = 0.0;
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    const double pi = u[i] + beta * p[i];
    const double si = w[i] + beta * s[i];
    p[i] = pi;
    s[i] = si;
    x[i] += alpha * pi;
    const double ri = r[i] - alpha * si;
    r[i] = ri;
    const double ui = dinv[i] * ri;
    u[i] = ui;
    g += ri * ui;
    nn += ui * ui;
  }
  g = tg_block_sum256(g, lds4);
  nn = tg_block_sum256(nn, lds4);
  if (threadIdx.x == 0) {
    partial[2 * blockIdx.x] = g;
    partial[2 * blockIdx.x + 1] = nn;
  }
}

static inline int tg_vec_grid(int64_t n) {
  int64_t g = tg_cdiv(n, 256);
  if (g > TG_VEC_BLOCKS) g = TG_VEC_BLOCKS;
  if (g < 1) g = 1;
  return (int)g;
}

struct tg_krylov_ws {
  double *buf = nullptr;
  ~tg_krylov_ws() {
    if (buf) {
      hipStreamSynchronize(g_tg.stream);
      tg_dfree(buf);
    }
  }
};

static int tg_read_scalars(const double *dev, int n, double *host) {
  TG_CHECK_HIP(hipMemcpyAsync(g_tg.host_pinned, dev, n * sizeof(double), hipMemcpyDeviceToHost, g_tg.stream));
  TG_CHECK_HIP(hipStreamSynchronize(g_tg.stream));
  for (int i = 0; i < n; i++) host[i] = g_tg.host_pinned[i];
  return 0;
}

#define TG_CG_RING 8
struct tg_cg_ring {     // per-iteration events: norm history copies and SpMV timing
  hipEvent_t done[TG_CG_RING], t0[TG_CG_RING], t1[TG_CG_RING];
  bool ok = false;
  int init() {
    for (int i = 0; i < TG_CG_RING; i++) {
      TG_CHECK_HIP(hipEventCreateWithFlags(&done[i], hipEventDisableTiming));
      TG_CHECK_HIP(hipEventCreate(&t0[i]));
      TG_CHECK_HIP(hipEventCreate(&t1[i]));
    }
    ok = true;
    return 0;
  }
  ~tg_cg_ring() {
    if (!ok) return;
    for (int i = 0; i < TG_CG_RING; i++) {
      hipEventDestroy(done[i]);
      hipEventDestroy(t0[i]);
      hipEventDestroy(t1[i]);
    }
  }
};

// out[0] = 1 + the last row with an entry left of column c_lo, out[1] = the first row with an entry at or right of
// column c_hi (rows in canonical form: ascending columns); out preset to {0, n}
__global__ void __launch_bounds__(256)
    k_halo_rows(const int64_t *__restrict__ rowptr, const int32_t *__restrict__ col, int64_t n, int64_t c_lo,
                int64_t c_hi, int *out) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  int lo = 0, hi = 0x7fffffff;
  for (; i < n; i += stride) {
    const int64_t a = rowptr[i], e = rowptr[i + 1];
    if (e <= a) continue;
    if ((int64_t)col[a] < c_lo) lo = max(lo, (int)i + 1);
    if ((int64_t)col[e - 1] >= c_hi) hi = min(hi, (int)i);
  }
  if (lo > 0) atomicMax(&out[0], lo);
  if (hi != 0x7fffffff) atomicMin(&out[1], hi);
}

static int tg_cg(tg_csr_s *k, tg_vec_s *b, tg_vec_s *x, int pc, double rtol, double atol, int maxit, int nonzero_guess,
                 tg_comm_s *comm, int *iters, double *resnorm, int *status) {
  const int64_t n = k->nrows;
  const int64_t hlo = comm ? comm->halo_lo : 0, hhi = comm ? comm->halo_hi : 0;
  const int64_t row0 = comm ? comm->g0 : 0;
  const int64_t next = hlo + n + hhi;
  tg_krylov_ws ws;
  // layout: uext[next] | r[n] | w[n] | p[n] | s[n] | dinv[n]
  TG_TRY(tg_dmalloc(&ws.buf, next + 5 * n));
  double *uext = ws.buf, *u = uext + hlo, *r = uext + next, *w = r + n, *p = w + n, *s = p + n, *dinv = s + n;
  TG_CHECK_HIP(hipMemsetAsync(uext, 0, (size_t)next * sizeof(double), g_tg.stream));
  double *part_gn = g_tg.scratch;                          // 2 * TG_VEC_BLOCKS
  double *part_d = g_tg.scratch + 2 * TG_VEC_BLOCKS;       // TG_VEC_BLOCKS
  tg_cg_scal *sc = (tg_cg_scal *)(g_tg.scratch + TG_SCRATCH_DOUBLES - 2048);   // two parities
  const int vg = tg_vec_grid(n);
  TG_TRY(tg_spmv_plan(k));
  tg_symgrid_guard sym;
  {
    // (TIGAR_SPMV_SYM: 0 = off, 1 = systems of at least 65536 rows [default], 2 = every size the plan accepts)
    const int sym_on = getenv("TIGAR_SPMV_SYM") ? atoi(getenv("TIGAR_SPMV_SYM")) : 1;
    const int sym_verify = getenv("TIGAR_SPMV_SYM_VERIFY") ? atoi(getenv("TIGAR_SPMV_SYM_VERIFY")) : !g_ksp_symmetric_hint;
    // (several ranks: every rank decides for its own z slab -- the products are local once the halo of u has arrived)
    if (sym_on && (n >= 65536 || sym_on > 1) && k->sell_state != 1)
      TG_TRY(tg_symgrid_build(k, row0, sym_verify, &sym.s));
    if (sym.s) g_tg.prof_n[TG_PROF_KSP_SYMGRID] += 1;
  }
  tg_sell_guard sell_guard(k, sym.s != nullptr);   // sliced copy of the values for the products of this solve
  TG_TRY(sell_guard.rc);
  tg_cg_ring ring;
  TG_TRY(ring.init());
  double *hist = g_tg.host_pinned + 8;                     // TG_CG_RING x 3 doubles (pinned)
  double *hist_dev = nullptr;                              // the same ring as a kernel addresses it (one rank only)
  if (!(comm && comm->world > 1) && !getenv("TIGAR_CG_HIST_COPY"))
    TG_CHECK_HIP(hipHostGetDevicePointer((void **)&hist_dev, hist, 0));
  const double *ushift = uext - (row0 - hlo);              // u addressed by global column index
  const int64_t cmin = row0 - hlo, cmax = row0 - hlo + next - 1;

  if (n > 0 && k->diag_cache && k->diag_rows == n) {
    hipLaunchKernelGGL(k_jacobi_from_diag, dim3(tg_vec_grid(n)), dim3(256), 0, g_tg.stream, k->diag_cache, n,
                       pc == TG_PC_JACOBI ? 1 : 0, dinv);
  } else if (n > 0) {
    const unsigned jg = (unsigned)std::min<int64_t>(tg_cdiv(n, 4), (int64_t)g_tg.num_cu * 16);
    hipLaunchKernelGGL(k_jacobi_setup, dim3(jg), dim3(256), 0, g_tg.stream, k->rowptr, k->col, k->val, n, row0,
                       pc == TG_PC_JACOBI ? 1 : 0, dinv);
  }
  // reference norm: ||B b|| -- also with a non-zero initial guess (KSPConvergedDefault without
  // -ksp_converged_use_initial_residual_norm [ext])
  double bnorm2 = 0.0;
  if (nonzero_guess) {
    hipLaunchKernelGGL(k_cg1_init, dim3(vg), dim3(256), 0, g_tg.stream, b->d, (const double *)nullptr, dinv, r, u, p, s,
                       n, part_gn);
    hipLaunchKernelGGL(k_fold, dim3(1), dim3(256), 0, g_tg.stream, part_gn, vg, 2, (double *)&sc[0]);
    TG_LAUNCH_CHECK();
    TG_TRY(tg_comm_allreduce_dev(comm, (double *)&sc[0], 2));
    double h2[2];
    TG_TRY(tg_read_scalars((double *)&sc[0], 2, h2));
    bnorm2 = h2[1];
    // r = b - K x0: x0 with its halo goes through uext, the product lands in w
    TG_CHECK_HIP(hipMemcpyAsync(u, x->d, (size_t)n * sizeof(double), hipMemcpyDeviceToDevice, g_tg.stream));
    TG_TRY(tg_comm_halo_exchange(comm, uext));
    TG_TRY(tg_spmv_raw(k, ushift, cmin, cmax, w));
    hipLaunchKernelGGL(k_cg1_init, dim3(vg), dim3(256), 0, g_tg.stream, b->d, (const double *)w, dinv, r, u, p, s, n,
                       part_gn);
  } else {
    TG_CHECK_HIP(hipMemsetAsync(x->d, 0, (size_t)std::max<int64_t>(n, 1) * sizeof(double), g_tg.stream));
    hipLaunchKernelGGL(k_cg1_init, dim3(vg), dim3(256), 0, g_tg.stream, b->d, (const double *)nullptr, dinv, r, u, p, s,
                       n, part_gn);
  }
  TG_LAUNCH_CHECK();
  double tol2_dev = 0.0;        // the tolerance the gated products compare with (set once the reference norm is known)
  // Rows [in0, in1) of this block have no entry in a halo column: their part of the product runs while the halo of u
  // is exchanged on the communicator's stream; the rows at the two ends follow when it has arrived.  (Needs the
  // sliced copy, whose launches can be restricted to row ranges; TIGAR_CG_OVERLAP=0 keeps the exchange in front.)
  int64_t in0 = 0, in1 = 0;
  static int overlap_on = getenv("TIGAR_CG_OVERLAP") ? atoi(getenv("TIGAR_CG_OVERLAP")) : 1;
  if (comm && comm->world > 1 && overlap_on && k->sell_state == 1 && k->sell && n > 0 && n < 0x7fffffffll) {
    int *d_ends = (int *)(g_tg.scratch + TG_SCRATCH_DOUBLES - 2048 + 64);
    const int ends0[2] = {0, (int)n};
    int ends[2];
    TG_CHECK_HIP(hipMemcpyAsync(d_ends, ends0, sizeof(ends0), hipMemcpyHostToDevice, g_tg.stream));
    hipLaunchKernelGGL(k_halo_rows, dim3(vg), dim3(256), 0, g_tg.stream, k->rowptr, k->col, n, row0, row0 + n, d_ends);
    TG_LAUNCH_CHECK();
    TG_CHECK_HIP(hipMemcpyAsync(ends, d_ends, sizeof(ends), hipMemcpyDeviceToHost, g_tg.stream));
    TG_CHECK_HIP(hipStreamSynchronize(g_tg.stream));
    const int64_t C = tg_sell_slice_rows();
    in0 = tg_cdiv((int64_t)ends[0], C) * C;
    in1 = ((int64_t)ends[1] / C) * C;
    if (in1 <= in0) in0 = in1 = 0;
    if (getenv("TIGAR_TRACE"))
      fprintf(stderr, "[trace] cg rank %d: rows without halo columns [%lld, %lld) of %lld\n", comm->rank, (long long)in0,
              (long long)in1, (long long)n);
  }
  // `gate`: the norm (device) the update in front of this product tested; with that update frozen (converged while the
  // host was still enqueueing) u is unchanged and so is w = K u: the launch returns at once
  const bool sliced = k->sell_state == 1 && k->sell && n > 0;
  auto product = [&](const double *gate) -> int {
    if (in1 > in0) {
      TG_TRY(tg_comm_halo_begin(comm, uext));
      int rc = tg_sell_spmv_rows(k, ushift, cmin, cmax, w, in0, in1, gate, tol2_dev);
      TG_TRY(tg_comm_halo_end(comm, uext));      // (also after a failed launch: the exchange is collective)
      TG_TRY(rc);
      TG_TRY(tg_sell_spmv_rows(k, ushift, cmin, cmax, w, 0, in0, gate, tol2_dev));
      TG_TRY(tg_sell_spmv_rows(k, ushift, cmin, cmax, w, in1, n, gate, tol2_dev));
      g_tg.prof_n[TG_PROF_KSP_OVERLAPPED] += 1;
      return 0;
    }
    if (sym.s && comm && comm->world > 1 && overlap_on && tg_symgrid_chunks(sym.s) > 1) {
      // all z chunks but the last read no halo plane of u: they run while it travels
      TG_TRY(tg_comm_halo_begin(comm, uext));
      int rc = tg_symgrid_spmv(sym.s, k, ushift, cmin, cmax, w, 1, gate, tol2_dev);
      TG_TRY(tg_comm_halo_end(comm, uext));
      TG_TRY(rc);
      g_tg.prof_n[TG_PROF_KSP_OVERLAPPED] += 1;
      return tg_symgrid_spmv(sym.s, k, ushift, cmin, cmax, w, 2, gate, tol2_dev);
    }
    TG_TRY(tg_comm_halo_exchange(comm, uext));
    if (sym.s) return tg_symgrid_spmv(sym.s, k, ushift, cmin, cmax, w, 0, gate, tol2_dev);
    if (sliced && gate) return tg_sell_spmv_rows(k, ushift, cmin, cmax, w, 0, n, gate, tol2_dev);
    return tg_spmv_raw(k, ushift, cmin, cmax, w);
  };
  // w = K u, delta; scalars of parity 0
  auto product_and_reduce = [&](tg_cg_scal *cur, int slot, const double *gate) -> int {
    hipEventRecord(ring.t0[slot], g_tg.stream);
    TG_TRY(product(gate));
    hipEventRecord(ring.t1[slot], g_tg.stream);
    hipLaunchKernelGGL(k_cg1_dot, dim3(vg), dim3(256), 0, g_tg.stream, w, u, n, part_d);
    hipLaunchKernelGGL(k_cg1_fold, dim3(1), dim3(256), 0, g_tg.stream, part_gn, part_d, vg, cur,
                       hist_dev ? hist_dev + 3 * slot : (double *)nullptr);
    TG_LAUNCH_CHECK();
    if (!hist_dev) {                             // several ranks: the entry is what the all-reduce leaves
      TG_TRY(tg_comm_allreduce_dev(comm, (double *)cur, 3));
      TG_CHECK_HIP(hipMemcpyAsync(hist + 3 * slot, cur, 3 * sizeof(double), hipMemcpyDeviceToHost, g_tg.stream));
    }
    TG_CHECK_HIP(hipEventRecord(ring.done[slot], g_tg.stream));
    return 0;
  };
  auto account = [&](int slot) {
    float ems = 0.f;
    if (hipEventElapsedTime(&ems, ring.t0[slot], ring.t1[slot]) == hipSuccess) {
      g_tg.prof_ms[TG_PROF_KSP_SPMV] += ems;
      g_tg.prof_n[TG_PROF_KSP_SPMV] += 1;
    }
  };
  TG_TRY(product_and_reduce(&sc[0], 0, nullptr));
  TG_CHECK_HIP(hipEventSynchronize(ring.done[0]));
  account(0);
  const double nu0 = hist[2];
  const double znorm_init = sqrt(nu0);
  const double znorm0 = nonzero_guess ? sqrt(bnorm2) : znorm_init;
  const double tol = std::max(rtol * znorm0, atol);
  // host and device take the SAME decision: both compare nu = ||B r||^2 with this tol2
  const double tol2 = tol * tol;
  tol2_dev = tol2;
  *iters = 0;
  *status = 0;
  *resnorm = znorm_init;
  if (!(znorm_init == znorm_init) || !(znorm0 == znorm0)) {
    *status = -2;
    return 0;
  }
  if (!(nu0 > tol2)) {
    *status = (znorm_init <= atol) ? 1 : 0;     // (b = 0: the absolute tolerance is what was met)
    return 0;
  }
  // iterations are enqueued `look` ahead of the one whose norm the host has seen
  // (TIGAR_CG_LOOK=0 makes the host read every norm before it enqueues the next iteration: nothing runs past convergence;
  //  the tests compare the two modes bit for bit)
  static const int look_env = getenv("TIGAR_CG_LOOK") ? std::max(0, std::min(TG_CG_RING - 2, atoi(getenv("TIGAR_CG_LOOK")))) : 2;
  const int look = look_env;
  *status = -1;
  int seen = 0;           // iterations whose norm the host has read
  int it_conv = -1;
  double znorm = znorm_init;
  const double t_all0 = tk_now();
  auto observe = [&](int it) -> bool {   // reads iteration `it`; true when the loop has to stop
    const int slot = it % TG_CG_RING;
    if (hipEventSynchronize(ring.done[slot]) != hipSuccess) return true;
    account(slot);
    const double nu = hist[3 * slot + 2];
    znorm = sqrt(nu);
    seen = it;
    if (!(nu == nu)) {
      *status = -2;
      it_conv = it;
      return true;
    }
    if (!(nu > tol2)) {
      *status = (znorm <= atol && !(znorm <= rtol * znorm0)) ? 1 : 0;
      it_conv = it;
      return true;
    }
    return false;
  };
  int it = 0;
  bool stop = false;
  const int lead = (!comm || comm->rank == 0) ? 1 : 0;
  for (it = 1; it <= maxit && !stop; it++) {
    tg_cg_scal *cur = &sc[(it - 1) & 1], *nxt = &sc[it & 1];
    hipLaunchKernelGGL(k_cg1_update, dim3(vg), dim3(256), 0, g_tg.stream, cur, nxt, tol2, it == 1 ? 1 : 0, lead, w, dinv,
                       u, p, s, x->d, r, n, part_gn);
    TG_TRY(product_and_reduce(nxt, it % TG_CG_RING, &cur->nu));
    if (it - look >= 1) stop = observe(it - look);
  }
  const int enq = it - 1;                       // iterations enqueued
  for (int j = seen + 1; j <= enq && it_conv < 0; j++) observe(j);
  *iters = it_conv >= 0 ? it_conv : std::min(enq, maxit);
  *resnorm = znorm;
  TG_CHECK_HIP(hipStreamSynchronize(g_tg.stream));
  TG_TRY(tg_comm_check(comm));
  if (getenv("TIGAR_TRACE"))
    fprintf(stderr, "[trace] cg: %d its (%d enqueued), loop %.3f s\n", *iters, enq, tk_now() - t_all0);
  return 0;
}

// ---------------------------------------------------------------------------------------- GMRES
// partial[b*(k+1)+j] = sum_i V_j[i] * w[i],  j = 0..k
__global__ void __launch_bounds__(256) k_multi_dot(const double *__restrict__ V, int64_t ld, int kp1,
                                                   const double *__restrict__ w, int64_t n,
                                                   double *__restrict__ partial) {
  __shared__ double lds4[4];
  for (int j = 0; j < kp1; j++) {
    const double *vj = V + (int64_t)j * ld;
    double s = 0.0;
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) s += vj[i] * w[i];
    s = tg_block_sum256(s, lds4);
    if (threadIdx.x == 0) partial[(int64_t)blockIdx.x * kp1 + j] = s;
  }
}

// w -= sum_j h[j] V_j ; partial of w.w
__global__ void __launch_bounds__(256) k_gs_update(const double *__restrict__ V, int64_t ld, int kp1,
                                                   const double *__restrict__ h, double *__restrict__ w, int64_t n,
                                                   double *__restrict__ partial) {
  __shared__ double lds4[4];
  double ss = 0.0;
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    double wi = w[i];
    for (int j = 0; j < kp1; j++) wi -= h[j] * V[(int64_t)j * ld + i];
    w[i] = wi;
    ss += wi * wi;
  }
  ss = tg_block_sum256(ss, lds4);
  if (threadIdx.x == 0) partial[blockIdx.x] = ss;
}

__global__ void k_scale_to(double *__restrict__ dst, const double *__restrict__ src, double a, int64_t n) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) dst[i] = a * src[i];
}

// dst = dinv * (b - kx)   (kx may be null: dst = dinv*b); partial of dst.dst
__global__ void __launch_bounds__(256) k_prec_residual(const double *__restrict__ b, const double *__restrict__ kx,
                                                       const double *__restrict__ dinv, double *__restrict__ dst,
                                                       int64_t n, double *__restrict__ partial) {
  __shared__ double lds4[4];
  double ss = 0.0;
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    const double v = dinv[i] * (kx ? b[i] - kx[i] : b[i]);
    dst[i] = v;
    ss += v * v;
  }
  ss = tg_block_sum256(ss, lds4);
  if (threadIdx.x == 0) partial[blockIdx.x] = ss;
}

__global__ void k_mul_inplace(double *__restrict__ w, const double *__restrict__ dinv, int64_t n) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) w[i] *= dinv[i];
}

// x += sum_j y[j] V_j
__global__ void k_lincomb_add(double *__restrict__ x, const double *__restrict__ V, int64_t ld, int k,
                              const double *__restrict__ y, int64_t n) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    double s = x[i];
    for (int j = 0; j < k; j++) s += y[j] * V[(int64_t)j * ld + i];
    x[i] = s;
  }
}

// ---- host-free GMRES(m): the Hessenberg column, the Givens recurrence, the residual estimate and the
// convergence decision live on the device (one single-thread kernel per inner iteration); the host enqueues
// iterations ahead of the one whose residual it has read (pinned ring, two iterations late, as in tg_cg) and
// everything enqueued past convergence is gated on the device flag `live`.  Two reductions per inner iteration
// (classical Gram-Schmidt coefficients, then the norm of the orthogonalised vector -- PETSc's default
// KSPGMRESClassicalGramSchmidtOrthogonalization without refinement [ext]), each one all-reduce with several ranks.
struct tg_gm_state {
  double live;        // 1.0 while iterating; 0.0 once converged / broken down / out of iterations (gate of the products)
  double res;         // current estimate of the preconditioned residual norm
  double tol;
  double beta0;       // reference norm
  int its, status, kused, cycle;
};

// cycle start from the reduced ||r||^2 in scal[0]; `first`: also fixes the tolerance (bnorm2 >= 0: reference norm^2)
__global__ void k_gm_start(tg_gm_state *st, const double *scal, int first, double rtol, double atol, double bnorm2,
                           double *g, int m) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  if (!first && st->live == 0.0) return;
  const double beta = sqrt(scal[0]);
  st->res = beta;
  if (first) {
    const double beta0 = bnorm2 >= 0.0 ? sqrt(bnorm2) : beta;
    st->beta0 = beta0;
    st->tol = fmax(rtol * beta0, atol);
    st->its = 0;
    st->kused = 0;
    st->cycle = 0;
    st->status = -1;
    st->live = 1.0;
    if (!(beta == beta) || !(beta0 == beta0)) {
      st->status = -2;
      st->live = 0.0;
      return;
    }
    if (beta0 <= atol) {
      st->status = 1;
      st->live = 0.0;
      return;
    }
  }
  if (beta <= st->tol) {
    st->status = 0;
    st->live = 0.0;
    return;
  }
  st->cycle += 1;
  g[0] = beta;
  for (int i = 1; i <= m; i++) g[i] = 0.0;
}

// dst = src / norm   (norm = st->res when `norm2` is null, else sqrt(*norm2)); nothing when the solve is over
__global__ void __launch_bounds__(256) k_gm_scale(const tg_gm_state *st, double *__restrict__ dst,
                                                  const double *__restrict__ src, const double *norm2, int64_t n) {
  if (st->live == 0.0) return;
  const double nrm = norm2 ? sqrt(*norm2) : st->res;
  const double a = nrm > 0.0 ? 1.0 / nrm : 0.0;
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) dst[i] = a * src[i];
}

// w <- dinv * w (preconditioning fused) and partial[b*(k+1)+j] = sum_i V_j[i] * w[i],  j = 0..k
__global__ void __launch_bounds__(256) k_gm_dots(const tg_gm_state *st, const double *__restrict__ V, int64_t ld, int kp1,
                                                 double *__restrict__ w, const double *__restrict__ dinv, int64_t n,
                                                 double *__restrict__ partial) {
  __shared__ double lds4[4];
  if (st->live == 0.0) return;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int j = 0; j < kp1; j++) {
    const double *vj = V + (int64_t)j * ld;
    double s = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
      double wi = w[i];
      if (j == 0) {               // (a thread always revisits its own entries: the scaled value is what it reads back)
        wi *= dinv[i];
        w[i] = wi;
      }
      s += vj[i] * wi;
    }
    s = tg_block_sum256(s, lds4);
    if (threadIdx.x == 0) partial[(int64_t)blockIdx.x * kp1 + j] = s;
  }
}

__global__ void __launch_bounds__(256) k_gm_fold(const tg_gm_state *st, const double *partial, int nb, int nstreams,
                                                 double *out) {
  __shared__ double lds4[4];
  if (st && st->live == 0.0) return;
  for (int k = 0; k < nstreams; k++) {
    double s = 0.0;
    for (int b = threadIdx.x; b < nb; b += 256) s += partial[(int64_t)b * nstreams + k];
    s = tg_block_sum256(s, lds4);
    if (threadIdx.x == 0) out[k] = s;
  }
}

// w -= sum_j h[j] V_j ; partial of w.w
__global__ void __launch_bounds__(256) k_gm_orth(const tg_gm_state *st, const double *__restrict__ V, int64_t ld, int kp1,
                                                 const double *__restrict__ h, double *__restrict__ w, int64_t n,
                                                 double *__restrict__ partial) {
  __shared__ double lds4[4];
  if (st->live == 0.0) return;
  double ss = 0.0;
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    double wi = w[i];
    for (int j = 0; j < kp1; j++) wi -= h[j] * V[(int64_t)j * ld + i];
    w[i] = wi;
    ss += wi * wi;
  }
  ss = tg_block_sum256(ss, lds4);
  if (threadIdx.x == 0) partial[blockIdx.x] = ss;
}

// column j of the Hessenberg matrix: rotations of the earlier columns, the new rotation, the residual estimate, the
// decision (hcol[0..j] = Gram-Schmidt coefficients, hcol[j+1] = ||w||^2)
// One workgroup: folds the partials of ||w||^2 first (hcol[j+1]; `reduced`: several ranks -- fold and all-reduce were
// done in front), then thread 0 runs the recurrence; the history entry goes straight to pinned host memory.
__global__ void __launch_bounds__(256)
    k_gm_givens(tg_gm_state *st, int j, int m, double *hcol, const double *partial, int nb, int reduced, double *H,
                double *cs, double *sn, double *g, int maxit, double *hist) {
  __shared__ double lds4[4];
  if (!reduced && st->live != 0.0) {
    double ssum = 0.0;
    for (int b = threadIdx.x; b < nb; b += 256) ssum += partial[b];
    ssum = tg_block_sum256(ssum, lds4);
    if (threadIdx.x == 0) hcol[j + 1] = ssum;
  }
  __syncthreads();
  if (threadIdx.x != 0) return;
  if (st->live != 0.0) {
    double *Hj = H + (int64_t)j * (m + 1);
    for (int i = 0; i <= j; i++) Hj[i] = hcol[i];
    Hj[j + 1] = sqrt(hcol[j + 1]);
    for (int i = 0; i < j; i++) {
      const double t = cs[i] * Hj[i] + sn[i] * Hj[i + 1];
      Hj[i + 1] = -sn[i] * Hj[i] + cs[i] * Hj[i + 1];
      Hj[i] = t;
    }
    const double den = hypot(Hj[j], Hj[j + 1]);
    if (den == 0.0 || !(den == den)) {
      st->status = -2;
      st->live = 0.0;
    } else {
      cs[j] = Hj[j] / den;
      sn[j] = Hj[j + 1] / den;
      Hj[j] = den;
      Hj[j + 1] = 0.0;
      g[j + 1] = -sn[j] * g[j];
      g[j] = cs[j] * g[j];
      st->its += 1;
      st->kused = j + 1;
      st->res = fabs(g[j + 1]);
      if (st->res <= st->tol) {
        st->status = 0;
        st->live = 0.0;
      } else if (st->its >= maxit)
        st->live = 0.0;        // (status stays -1: iteration limit)
    }
  }
  hist[0] = st->res;
  hist[1] = st->live;
  hist[2] = (double)st->its;
  hist[3] = (double)st->cycle;
}

// end of a cycle: y from the triangular system of the columns used, x += V y, the cycle is closed (kused = 0)
__global__ void k_gm_solve_y(const tg_gm_state *st, int m, const double *H, const double *g, double *y) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  const int ku = st->kused;
  for (int i = ku - 1; i >= 0; i--) {
    double s = g[i];
    for (int c = i + 1; c < ku; c++) s -= H[(int64_t)c * (m + 1) + i] * y[c];
    y[i] = s / H[(int64_t)i * (m + 1) + i];
  }
}
__global__ void __launch_bounds__(256) k_gm_update_x(const tg_gm_state *st, double *__restrict__ x,
                                                     const double *__restrict__ V, int64_t ld, const double *__restrict__ y,
                                                     int64_t n) {
  const int ku = st->kused;
  if (ku <= 0) return;
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    double s = x[i];
    for (int j = 0; j < ku; j++) s += y[j] * V[(int64_t)j * ld + i];
    x[i] = s;
  }
}
__global__ void k_gm_close(tg_gm_state *st) {
  if (threadIdx.x == 0 && blockIdx.x == 0) st->kused = 0;
}
// dst = dinv * (b - kx) (kx may be null); partial of dst.dst; nothing once the solve is over (st may be null)
__global__ void __launch_bounds__(256) k_gm_residual(const tg_gm_state *st, const double *__restrict__ b,
                                                     const double *__restrict__ kx, const double *__restrict__ dinv,
                                                     double *__restrict__ dst, int64_t n, double *__restrict__ partial) {
  __shared__ double lds4[4];
  if (st && st->live == 0.0) return;
  double ss = 0.0;
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    const double v = dinv[i] * (kx ? b[i] - kx[i] : b[i]);
    dst[i] = v;
    ss += v * v;
  }
  ss = tg_block_sum256(ss, lds4);
  if (threadIdx.x == 0) partial[blockIdx.x] = ss;
}

static int tg_gmres(tg_csr_s *k, tg_vec_s *b, tg_vec_s *x, int pc, double rtol, double atol, int maxit, int restart,
                    int nonzero_guess, int stagnation_guard, tg_comm_s *comm, int *iters, double *resnorm, int *status) {
  const int64_t n = k->nrows;
  const int64_t hlo = comm ? comm->halo_lo : 0, hhi = comm ? comm->halo_hi : 0;
  const int64_t row0 = comm ? comm->g0 : 0;
  const int64_t next = hlo + n + hhi;
  const int m = restart;
  tg_krylov_ws ws;
  // layout: ext[next] (SpMV input with halo) | w[n] | dinv[n] | V[(m+1) n] | small: H[(m+1) m] cs[m] sn[m] g[m+1] y[m]
  //         hcol[m+2] scal[4] state
  const int64_t nsmall = (int64_t)(m + 1) * m + 3 * (int64_t)m + (m + 1) + (m + 2) + 4 + 16;
  TG_TRY(tg_dmalloc(&ws.buf, next + 2 * n + (int64_t)(m + 1) * n + nsmall));
  double *ext = ws.buf, *xin = ext + hlo, *w = ext + next, *dinv = w + n, *V = dinv + n;
  double *H = V + (int64_t)(m + 1) * n, *cs = H + (int64_t)(m + 1) * m, *sn = cs + m, *g = sn + m, *y = g + (m + 1),
         *hdev = y + m, *scal = hdev + (m + 2);
  tg_gm_state *st = (tg_gm_state *)(scal + 4);
  TG_CHECK_HIP(hipMemsetAsync(ext, 0, (size_t)next * sizeof(double), g_tg.stream));
  TG_CHECK_HIP(hipMemsetAsync(H, 0, (size_t)nsmall * sizeof(double), g_tg.stream));
  double *partial = g_tg.scratch;
  const int vg = std::min(tg_vec_grid(n), 256);
  TG_TRY(tg_spmv_plan(k));
  tg_sell_guard sell_guard(k);   // sliced copy of the values for the products of this solve
  TG_TRY(sell_guard.rc);
  const bool sliced = k->sell_state == 1 && k->sell && n > 0;
  tg_cg_ring ring;
  TG_TRY(ring.init());
  double *hist = g_tg.host_pinned + 8;                     // TG_CG_RING x 4 doubles (pinned)
  double *hist_dev = nullptr;                              // the same ring as the kernels address it
  TG_CHECK_HIP(hipHostGetDevicePointer((void **)&hist_dev, hist, 0));
  const double *xshift = ext - (row0 - hlo);
  const int64_t cmin = row0 - hlo, cmax = row0 - hlo + next - 1;
  if (n > 0 && k->diag_cache && k->diag_rows == n) {
    hipLaunchKernelGGL(k_jacobi_from_diag, dim3(tg_vec_grid(n)), dim3(256), 0, g_tg.stream, k->diag_cache, n,
                       pc == TG_PC_JACOBI ? 1 : 0, dinv);
  } else if (n > 0) {
    const unsigned jg = (unsigned)std::min<int64_t>(tg_cdiv(n, 4), (int64_t)g_tg.num_cu * 16);
    hipLaunchKernelGGL(k_jacobi_setup, dim3(jg), dim3(256), 0, g_tg.stream, k->rowptr, k->col, k->val, n, row0,
                       pc == TG_PC_JACOBI ? 1 : 0, dinv);
  }
  // w = K * (the vector in xin, halo exchanged); products past the end of the solve return at once (sliced copy)
  // (slot >= 0: timed with the event pair of that ring slot -- the product of an inner iteration, accounted when the
  //  host reads that iteration's history entry)
  // w = K src.  One rank: the product reads src where it lies; several: src goes into the extended vector first and
  // its halo is exchanged.
  auto product = [&](const double *src, bool gated, int slot) -> int {
    const double *xs = src;
    if (comm) {
      TG_CHECK_HIP(hipMemcpyAsync(xin, src, (size_t)n * sizeof(double), hipMemcpyDeviceToDevice, g_tg.stream));
      xs = xshift;
    }
    if (slot >= 0) hipEventRecord(ring.t0[slot], g_tg.stream);
    TG_TRY(tg_comm_halo_exchange(comm, ext));
    if (sliced && gated)
      TG_TRY(tg_sell_spmv_rows(k, xs, cmin, cmax, w, 0, n, &st->live, 0.5));
    else
      TG_TRY(tg_spmv_raw(k, xs, cmin, cmax, w));
    if (slot >= 0) hipEventRecord(ring.t1[slot], g_tg.stream);
    return 0;
  };
  // reference norm ||B b|| when the initial guess is not zero [ext]
  double bnorm2 = -1.0;
  if (nonzero_guess) {
    hipLaunchKernelGGL(k_gm_residual, dim3(vg), dim3(256), 0, g_tg.stream, (const tg_gm_state *)nullptr, b->d,
                       (const double *)nullptr, dinv, V, n, partial);
    hipLaunchKernelGGL(k_gm_fold, dim3(1), dim3(256), 0, g_tg.stream, (const tg_gm_state *)nullptr, partial, vg, 1, scal);
    TG_LAUNCH_CHECK();
    TG_TRY(tg_comm_allreduce_dev(comm, scal, 1));
    TG_TRY(tg_read_scalars(scal, 1, &bnorm2));
  } else
    TG_CHECK_HIP(hipMemsetAsync(x->d, 0, (size_t)std::max<int64_t>(n, 1) * sizeof(double), g_tg.stream));

  static const int look_env = getenv("TIGAR_CG_LOOK") ? std::max(0, std::min(TG_CG_RING - 2, atoi(getenv("TIGAR_CG_LOOK")))) : 2;
  const int look = look_env;
  int enq = 0, seen = 0;          // inner iterations enqueued / read back
  bool stop = false;
  double h_res = 0.0, h_live = 1.0;
  int h_its = 0;
  // stagnation guard (opt-in): 25 restart cycles in a row that each reduce the residual by less than 0.1 %
  double cycle_res = -1.0;
  int stagnant = 0, last_cycle_seen = 0;
  bool stagnated = false;
  auto observe = [&](int it) -> bool {
    const int slot = it % TG_CG_RING;
    if (hipEventSynchronize(ring.done[slot]) != hipSuccess) return true;
    if (h_live != 0.0) {          // (products enqueued past the end return at once: not counted)
      float ems = 0.f;
      if (hipEventElapsedTime(&ems, ring.t0[slot], ring.t1[slot]) == hipSuccess) {
        g_tg.prof_ms[TG_PROF_KSP_SPMV] += ems;
        g_tg.prof_n[TG_PROF_KSP_SPMV] += 1;
      }
    }
    h_res = hist[4 * slot];
    h_live = hist[4 * slot + 1];
    h_its = (int)hist[4 * slot + 2];
    const int cyc = (int)hist[4 * slot + 3];
    seen = it;
    if (stagnation_guard && cyc != last_cycle_seen) {     // first iteration of a new cycle: compare the cycles' ends
      if (cycle_res >= 0.0) {
        stagnant = (h_res > 0.999 * cycle_res) ? stagnant + 1 : 0;
        if (stagnant >= 25) stagnated = true;
      }
      cycle_res = h_res;
      last_cycle_seen = cyc;
    }
    return h_live == 0.0 || stagnated;
  };
  bool first = true;
  while (!stop) {
    // r = B (b - K x), beta = ||r||, v_0 = r / beta
    if (first && !nonzero_guess) {
      hipLaunchKernelGGL(k_gm_residual, dim3(vg), dim3(256), 0, g_tg.stream, (const tg_gm_state *)nullptr, b->d,
                         (const double *)nullptr, dinv, V, n, partial);
    } else {
      TG_TRY(product(x->d, !first, -1));
      hipLaunchKernelGGL(k_gm_residual, dim3(vg), dim3(256), 0, g_tg.stream, first ? (const tg_gm_state *)nullptr : st, b->d,
                         (const double *)w, dinv, V, n, partial);
    }
    hipLaunchKernelGGL(k_gm_fold, dim3(1), dim3(256), 0, g_tg.stream, first ? (const tg_gm_state *)nullptr : st, partial, vg,
                       1, scal);
    TG_LAUNCH_CHECK();
    TG_TRY(tg_comm_allreduce_dev(comm, scal, 1));
    hipLaunchKernelGGL(k_gm_start, dim3(1), dim3(1), 0, g_tg.stream, st, scal, first ? 1 : 0, rtol, atol, bnorm2, g, m);
    hipLaunchKernelGGL(k_gm_scale, dim3(vg), dim3(256), 0, g_tg.stream, st, V, V, (const double *)nullptr, n);
    TG_LAUNCH_CHECK();
    if (first) {
      // the tolerance is fixed now; an immediate end (b = 0, converged start) is read before anything is enqueued
      tg_gm_state h0;
      TG_CHECK_HIP(hipMemcpyAsync(g_tg.host_pinned + 40, st, sizeof(tg_gm_state), hipMemcpyDeviceToHost, g_tg.stream));
      TG_CHECK_HIP(hipStreamSynchronize(g_tg.stream));
      memcpy(&h0, g_tg.host_pinned + 40, sizeof(h0));
      h_res = h0.res;
      if (h0.live == 0.0) {
        *iters = 0;
        *resnorm = h0.res;
        *status = h0.status;
        return tg_comm_check(comm);
      }
      first = false;
    }
    for (int j = 0; j < m && !stop; j++) {
      // w = B K v_j ; Gram-Schmidt against v_0..v_j ; v_{j+1} ; column j of H
      TG_TRY(product(V + (int64_t)j * n, true, (enq + 1) % TG_CG_RING));
      hipLaunchKernelGGL(k_gm_dots, dim3(vg), dim3(256), 0, g_tg.stream, st, V, n, j + 1, w, dinv, n, partial);
      hipLaunchKernelGGL(k_gm_fold, dim3(1), dim3(256), 0, g_tg.stream, st, partial, vg, j + 1, hdev);
      TG_LAUNCH_CHECK();
      TG_TRY(tg_comm_allreduce_dev(comm, hdev, j + 1));
      hipLaunchKernelGGL(k_gm_orth, dim3(vg), dim3(256), 0, g_tg.stream, st, V, n, j + 1, hdev, w, n, partial);
      if (comm) {
        hipLaunchKernelGGL(k_gm_fold, dim3(1), dim3(256), 0, g_tg.stream, st, partial, vg, 1, hdev + j + 1);
        TG_LAUNCH_CHECK();
        TG_TRY(tg_comm_allreduce_dev(comm, hdev + j + 1, 1));
      }
      enq++;
      const int slot = enq % TG_CG_RING;
      // the recurrence BEFORE v_{j+1} is scaled: both read ||w||^2 from hcol[j+1], and the scaling is skipped once the
      // recurrence has ended the solve (the basis vector is not needed then)
      hipLaunchKernelGGL(k_gm_givens, dim3(1), dim3(256), 0, g_tg.stream, st, j, m, hdev, partial, vg, comm ? 1 : 0, H, cs,
                         sn, g, maxit, hist_dev + 4 * slot);
      hipLaunchKernelGGL(k_gm_scale, dim3(vg), dim3(256), 0, g_tg.stream, st, V + (int64_t)(j + 1) * n, w,
                         (const double *)(hdev + j + 1), n);
      TG_LAUNCH_CHECK();
      TG_CHECK_HIP(hipEventRecord(ring.done[slot], g_tg.stream));
      if (enq - look >= 1) stop = observe(enq - look);
    }
    // close the cycle: x += V y for the columns used (also after convergence inside the cycle)
    hipLaunchKernelGGL(k_gm_solve_y, dim3(1), dim3(1), 0, g_tg.stream, (const tg_gm_state *)st, m, H, g, y);
    hipLaunchKernelGGL(k_gm_update_x, dim3(vg), dim3(256), 0, g_tg.stream, (const tg_gm_state *)st, x->d, V, n, y, n);
    hipLaunchKernelGGL(k_gm_close, dim3(1), dim3(1), 0, g_tg.stream, st);
    TG_LAUNCH_CHECK();
    if (!stop && enq >= maxit + look) stop = true;   // (the device stops counting at maxit; do not enqueue for ever)
  }
  for (int j = seen + 1; j <= enq; j++)
    if (observe(j) && h_live == 0.0) break;
  TG_CHECK_HIP(hipMemcpyAsync(g_tg.host_pinned + 40, st, sizeof(tg_gm_state), hipMemcpyDeviceToHost, g_tg.stream));
  TG_CHECK_HIP(hipStreamSynchronize(g_tg.stream));
  tg_gm_state hf;
  memcpy(&hf, g_tg.host_pinned + 40, sizeof(hf));
  *iters = hf.its;
  *resnorm = hf.res;
  *status = (stagnated && hf.live != 0.0) ? -3 : hf.status;
  TG_TRY(tg_comm_check(comm));
  return 0;
}

// ------------------------------------------------------------------------ CG with a polynomial preconditioner
// PCG whose preconditioner is m steps of the Chebyshev iteration for D^-1 K z = D^-1 r from z = 0 on the interval
// [lmax / ratio, 1.1 lmax] (PETSc: PCKSP of KSPCHEBYSHEV + PCJACOBI, the GPU-natural stand-in for SOR / ILU sweeps: only
// products and vector updates, no triangular solves, a FIXED polynomial in D^-1 K -- so it is a linear, symmetric positive
// definite operator and plain PCG applies).  In products the method costs what Jacobi-CG costs (CG is optimal in the
// Krylov space: measured 0.93-1.3 x, tools/ notes in DESIGN 4b), but an outer iteration -- the reductions, the host's look
// at the norm, the launches of the update kernels -- happens m times less often: that is the gain for systems whose
// iteration is bound by launch latency (cfg4: 6 128 Jacobi-CG iterations of 34 us).  The interval comes from 12 Lanczos steps
// (upper end x 1.1, capped by the Gershgorin bound: an underestimate would make the polynomial indefinite).
// Same recurrence as tg_cg (one reduction of gamma, delta, nu per iteration), scalars through the host.
__global__ void __launch_bounds__(256) k_row_abs_bound(const int64_t *__restrict__ rowptr, const double *__restrict__ val,
                                                       const double *__restrict__ dinv, int64_t n, double *__restrict__ partial) {
  // partial[block] = max over its rows of |dinv_i| * sum_j |k_ij|   (Gershgorin bound of D^-1 K)
  __shared__ double lds[256];
  double m = 0.0;
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    double a = 0.0;
    for (int64_t q = rowptr[i]; q < rowptr[i + 1]; q++) a += fabs(val[q]);
    m = fmax(m, fabs(dinv[i]) * a);
  }
  lds[threadIdx.x] = m;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) lds[threadIdx.x] = fmax(lds[threadIdx.x], lds[threadIdx.x + o]);
    __syncthreads();
  }
  if (threadIdx.x == 0) partial[blockIdx.x] = lds[0];
}
__global__ void __launch_bounds__(256) k_fold_max(const double *partial, int nb, double *out) {
  __shared__ double lds[256];
  double m = 0.0;
  for (int b = threadIdx.x; b < nb; b += 256) m = fmax(m, partial[b]);
  lds[threadIdx.x] = m;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) lds[threadIdx.x] = fmax(lds[threadIdx.x], lds[threadIdx.x + o]);
    __syncthreads();
  }
  if (threadIdx.x == 0) out[0] = lds[0];
}
// y = a * dinv .* x ; partial of (y, y)
__global__ void __launch_bounds__(256) k_scaled_copy_norm(const double *x, const double *__restrict__ dinv, double a,
                                                          int64_t n, double *y, double *__restrict__ partial) {   // (y may be x)
  __shared__ double lds4[4];
  double nn = 0.0;
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    const double v = a * (dinv ? dinv[i] : 1.0) * x[i];
    y[i] = v;
    nn += v * v;
  }
  nn = tg_block_sum256(nn, lds4);
  if (threadIdx.x == 0) partial[blockIdx.x] = nn;
}
// Lanczos on D^-1/2 K D^-1/2 (eigenvalue estimates): v0 = sqrt(dinv) .* b .* (1 + h/16) (unnormalised), partial of (v0, v0).
// h in [-1, 1) is a hash of the GLOBAL row index: a right-hand side that is (close to) one eigenvector -- the sine load of
// the demos on a uniform degree-1 patch is exactly that -- would end the recurrence after one step with the upper end of
// the spectrum unseen, and a Chebyshev interval that ends below lambda_max amplifies the rounding noise of every other
// component until the recurrence breaks down.  The modulation puts all frequencies into the start vector (a sixteenth is
// enough for the upper end -- Lanczos needs one more step to set the dominant component aside -- and leaves the estimate of the
// lower end, which tells an easy system from a hard one, to the content of b); the same bits on any number of ranks.
__global__ void __launch_bounds__(256) k_lz_start(const double *__restrict__ b, const double *__restrict__ dinv, int64_t n, int64_t row0,
                                                  double *__restrict__ v, double *__restrict__ partial) {
  __shared__ double lds4[4];
  double nn = 0.0;
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    unsigned long long z = (unsigned long long)(row0 + i) + 0x9e3779b97f4a7c15ull;      // splitmix64
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    z ^= z >> 31;
    const double h = (double)(long long)(z >> 11) * (1.0 / 4503599627370496.0) - 1.0;   // [-1, 1)
    const double vi = sqrt(fabs(dinv[i])) * b[i] * (1.0 + 0.0625 * h);
    v[i] = vi;
    nn += vi * vi;
  }
  nn = tg_block_sum256(nn, lds4);
  if (threadIdx.x == 0) partial[blockIdx.x] = nn;
}
// v <- v * inv_norm ; operand = sqrt(dinv) .* v
__global__ void __launch_bounds__(256) k_lz_operand(const double *__restrict__ dinv, double inv_norm, int64_t n, double *__restrict__ v,
                                                    double *__restrict__ op) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    const double vi = v[i] * inv_norm;
    v[i] = vi;
    op[i] = sqrt(fabs(dinv[i])) * vi;
  }
}
// w <- sqrt(dinv) .* w - beta vprev ; partial of (w, v)
__global__ void __launch_bounds__(256) k_lz_alpha(const double *__restrict__ dinv, const double *__restrict__ vprev, double beta,
                                                  const double *__restrict__ v, int64_t n, double *__restrict__ w,
                                                  double *__restrict__ partial) {
  __shared__ double lds4[4];
  double a = 0.0;
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    const double wi = sqrt(fabs(dinv[i])) * w[i] - beta * vprev[i];
    w[i] = wi;
    a += wi * v[i];
  }
  a = tg_block_sum256(a, lds4);
  if (threadIdx.x == 0) partial[blockIdx.x] = a;
}
// w <- w - alpha v ; vprev <- v ; v <- w ; partial of (w, w)
__global__ void __launch_bounds__(256) k_lz_next(double alpha, int64_t n, double *__restrict__ w, double *__restrict__ v,
                                                 double *__restrict__ vprev, double *__restrict__ partial) {
  __shared__ double lds4[4];
  double nn = 0.0;
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    const double vi = v[i];
    const double wi = w[i] - alpha * vi;
    vprev[i] = vi;
    v[i] = wi;
    nn += wi * wi;
  }
  nn = tg_block_sum256(nn, lds4);
  if (threadIdx.x == 0) partial[blockIdx.x] = nn;
}
// first Chebyshev step: g = dinv r ; z = g / theta ; d = z
__global__ void __launch_bounds__(256) k_cheb_first(const double *__restrict__ r, const double *__restrict__ dinv, double inv_theta,
                                                    int64_t n, double *__restrict__ g, double *__restrict__ z, double *__restrict__ d) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    const double gi = dinv[i] * r[i];
    g[i] = gi;
    const double zi = gi * inv_theta;
    z[i] = zi;
    d[i] = zi;
  }
}
// next step: d = c1 d + c2 (g - dinv Kz) ; z += d
__global__ void __launch_bounds__(256) k_cheb_step(const double *__restrict__ g, const double *__restrict__ dinv,
                                                   const double *__restrict__ kz, double c1, double c2, int64_t n,
                                                   double *__restrict__ z, double *__restrict__ d) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    const double di = c1 * d[i] + c2 * (g[i] - dinv[i] * kz[i]);
    d[i] = di;
    z[i] += di;
  }
}
// partials of (r,u), (w,u), (u,u), interleaved
__global__ void __launch_bounds__(256) k_dots3(const double *__restrict__ r, const double *__restrict__ u, const double *__restrict__ w,
                                               int64_t n, double *__restrict__ partial) {
  __shared__ double lds4[4];
  double a = 0.0, b = 0.0, c = 0.0;
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    const double ui = u[i];
    a += r[i] * ui;
    b += w[i] * ui;
    c += ui * ui;
  }
  a = tg_block_sum256(a, lds4);
  b = tg_block_sum256(b, lds4);
  c = tg_block_sum256(c, lds4);
  if (threadIdx.x == 0) {
    partial[3 * blockIdx.x] = a;
    partial[3 * blockIdx.x + 1] = b;
    partial[3 * blockIdx.x + 2] = c;
  }
}
// p = u + beta p ; s = w + beta s ; x += alpha p ; r -= alpha s
__global__ void __launch_bounds__(256) k_pcg_update(const double *__restrict__ u, const double *__restrict__ w, double alpha, double beta,
                                                    int64_t n, double *__restrict__ p, double *__restrict__ s, double *__restrict__ x,
                                                    double *__restrict__ r) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    const double pi = u[i] + beta * p[i];
    const double si = w[i] + beta * s[i];
    p[i] = pi;
    s[i] = si;
    x[i] += alpha * pi;
    r[i] -= alpha * si;
  }
}
__global__ void __launch_bounds__(256) k_residual(const double *__restrict__ b, const double *__restrict__ kx, int64_t n,
                                                  double *__restrict__ r) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) r[i] = kx ? b[i] - kx[i] : b[i];
}

static int tg_pcg_cheb(tg_csr_s *k, tg_vec_s *b, tg_vec_s *x, int degree, double rtol, double atol, int maxit, int nonzero_guess,
                       tg_comm_s *comm, int *iters, double *resnorm, int *status) {
  const int64_t n = k->nrows;
  const int64_t hlo = comm ? comm->halo_lo : 0, hhi = comm ? comm->halo_hi : 0;
  const int64_t row0 = comm ? comm->g0 : 0;
  const int64_t next = hlo + n + hhi;
  const int m = std::max(1, std::min(degree, 64));
  tg_krylov_ws ws;
  // layout: uext[next] | r | w | p | s | dinv | g | d | kz
  TG_TRY(tg_dmalloc(&ws.buf, next + 8 * n));
  double *uext = ws.buf, *u = uext + hlo, *r = uext + next, *w = r + n, *p = w + n, *s = p + n, *dinv = s + n, *g = dinv + n,
         *dd = g + n, *kz = dd + n;
  TG_CHECK_HIP(hipMemsetAsync(ws.buf, 0, (size_t)(next + 8 * n) * sizeof(double), g_tg.stream));
  double *part = g_tg.scratch;                       // 3 * TG_VEC_BLOCKS
  double *sc = g_tg.scratch + TG_SCRATCH_DOUBLES - 2048;
  const int vg = tg_vec_grid(n);
  TG_TRY(tg_spmv_plan(k));
  tg_symgrid_guard sym;        // (CG with a polynomial preconditioner: the same premise, the same half-storage copy)
  {
    const int sym_on = getenv("TIGAR_SPMV_SYM") ? atoi(getenv("TIGAR_SPMV_SYM")) : 1;
    const int sym_verify = getenv("TIGAR_SPMV_SYM_VERIFY") ? atoi(getenv("TIGAR_SPMV_SYM_VERIFY")) : !g_ksp_symmetric_hint;
    if (sym_on && (n >= 65536 || sym_on > 1) && k->sell_state != 1) TG_TRY(tg_symgrid_build(k, row0, sym_verify, &sym.s));
    if (sym.s) g_tg.prof_n[TG_PROF_KSP_SYMGRID] += 1;
  }
  tg_sell_guard sell_guard(k, sym.s != nullptr);
  TG_TRY(sell_guard.rc);
  const double *ushift = uext - (row0 - hlo);
  const int64_t cmin = row0 - hlo, cmax = row0 - hlo + next - 1;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  TG_CHECK_HIP(hipEventCreate(&e0));
  TG_CHECK_HIP(hipEventCreate(&e1));
  struct ev_guard {
    hipEvent_t a, b;
    ~ev_guard() {
      hipEventDestroy(a);
      hipEventDestroy(b);
    }
  } evg{e0, e1};
  auto product = [&](double *out) -> int {            // out = K (vector in uext)
    TG_TRY(tg_comm_halo_exchange(comm, uext));
    if (sym.s) return tg_symgrid_spmv(sym.s, k, ushift, cmin, cmax, out, 0, nullptr, 0.0);
    return tg_spmv_raw(k, ushift, cmin, cmax, out);
  };
  auto reduce = [&](int cnt, double *host) -> int {    // folds `cnt` interleaved partial streams, sums over ranks, reads
    hipLaunchKernelGGL(k_fold, dim3(1), dim3(256), 0, g_tg.stream, part, vg, cnt, sc);
    TG_LAUNCH_CHECK();
    TG_TRY(tg_comm_allreduce_dev(comm, sc, cnt));
    return tg_read_scalars(sc, cnt, host);
  };
  if (n > 0) {
    const unsigned jg = (unsigned)std::min<int64_t>(tg_cdiv(n, 4), (int64_t)g_tg.num_cu * 16);
    hipLaunchKernelGGL(k_jacobi_setup, dim3(jg), dim3(256), 0, g_tg.stream, k->rowptr, k->col, k->val, n, row0, 1, dinv);
  }
  // ---- extreme eigenvalues of D^-1 K: 12 Lanczos steps on D^-1/2 K D^-1/2 from D^1/2-scaled b (both ends of the spectrum at
  // once; the power method needs 20 products for the upper end alone).  theta_max approaches lambda_max from below (safety
  // 1.1, capped by the Gershgorin bound: an interval that ends below lambda_max makes the polynomial indefinite), theta_min
  // approaches lambda_min from above: it only tells an easy system (few Jacobi-CG iterations) from a hard one.
  double lmax = 0.0, gersh = 0.0, theta_min = 0.0;
  {
    hipLaunchKernelGGL(k_row_abs_bound, dim3(vg), dim3(256), 0, g_tg.stream, k->rowptr, k->val, dinv, n, part);
    hipLaunchKernelGGL(k_fold_max, dim3(1), dim3(256), 0, g_tg.stream, part, vg, sc);
    TG_LAUNCH_CHECK();
    TG_TRY(tg_read_scalars(sc, 1, &gersh));
    if (comm) {
      double gg = gersh;                              // (the sum of the per-rank bounds is a bound as well)
      TG_TRY(tg_comm_allreduce_sum(comm, &gg, 1));
      gersh = gg;
    }
    // v (Lanczos vector) in g, previous one in dd, operand D^-1/2 v in u, D^-1/2 K D^-1/2 v in kz
    constexpr int LZ = 12;
    double al[LZ], be[LZ + 1];
    int kdim = 0;
    hipLaunchKernelGGL(k_lz_start, dim3(vg), dim3(256), 0, g_tg.stream, b->d, dinv, n, (int64_t)row0, g, part);
    double nn = 0.0;
    TG_TRY(reduce(1, &nn));
    if (nn != nn) {                                   // NaN in b or on the diagonal of K: breakdown, not "b = 0"
      *iters = 0;
      *resnorm = nn;
      *status = -2;
      return 0;
    }
    if (!(nn > 0.0)) {                                // b = 0
      TG_CHECK_HIP(hipMemsetAsync(x->d, 0, (size_t)std::max<int64_t>(n, 1) * sizeof(double), g_tg.stream));
      *iters = 0;
      *resnorm = 0.0;
      *status = 1;
      return 0;
    }
    double inv_norm = 1.0 / sqrt(nn), beta = 0.0;
    be[0] = 0.0;
    for (int j = 0; j < LZ; j++) {
      // g <- g * inv_norm ; u = sqrt(dinv) g
      hipLaunchKernelGGL(k_lz_operand, dim3(vg), dim3(256), 0, g_tg.stream, dinv, inv_norm, n, g, u);
      TG_TRY(product(kz));
      // kz <- sqrt(dinv) kz - beta dd ; alpha = (kz, g)
      hipLaunchKernelGGL(k_lz_alpha, dim3(vg), dim3(256), 0, g_tg.stream, dinv, dd, beta, g, n, kz, part);
      double alpha = 0.0;
      TG_TRY(reduce(1, &alpha));
      // kz <- kz - alpha g ; dd <- g ; g <- kz ; beta' = ||kz||
      hipLaunchKernelGGL(k_lz_next, dim3(vg), dim3(256), 0, g_tg.stream, alpha, n, kz, g, dd, part);
      TG_TRY(reduce(1, &nn));
      al[j] = alpha;
      kdim = j + 1;
      if (!(nn > 1e-28 * alpha * alpha) || !(nn == nn)) break;      // invariant subspace
      beta = sqrt(nn);
      be[j + 1] = beta;
      inv_norm = 1.0 / beta;
    }
    // extreme eigenvalues of the tridiagonal (alpha_j, beta_j) by Sturm bisection
    auto count_below = [&](double lam) {
      int c = 0;
      double q = 1.0;
      for (int j = 0; j < kdim; j++) {
        const double b2 = j ? be[j] * be[j] : 0.0;
        q = (al[j] - lam) - (j ? b2 / (q == 0.0 ? 1e-300 : q) : 0.0);
        if (q < 0.0) c++;
      }
      return c;
    };
    double lo = 0.0, hi = 0.0;
    for (int j = 0; j < kdim; j++) {
      const double rad = fabs(al[j]) + (j ? fabs(be[j]) : 0.0) + (j + 1 < kdim ? fabs(be[j + 1]) : 0.0);
      hi = std::max(hi, rad);
    }
    lo = -hi;
    auto kth = [&](int kk) {                           // kk-th smallest eigenvalue (0-based)
      double a = lo, bq = hi;
      for (int it = 0; it < 100; it++) {
        const double mid = 0.5 * (a + bq);
        if (count_below(mid) > kk) bq = mid; else a = mid;
      }
      return 0.5 * (a + bq);
    };
    if (kdim > 0) {
      lmax = kth(kdim - 1);
      theta_min = kth(0);
    }
    lmax *= 1.1;
    if (gersh > 0.0 && (lmax > gersh || !(lmax > 0.0))) lmax = gersh;
  }
  static const double ratio_env = getenv("TIGAR_CHEB_RATIO") ? atof(getenv("TIGAR_CHEB_RATIO")) : 0.0;
  // the interval [lmax / ratio, lmax]: 4 m^2 for a hard system (measured: within 10 % of the best ratio for m = 4 ... 16 on
  // squared Poisson operators), less when the Lanczos estimate of the lower end says the spectrum is narrow
  double ratio = std::max(10.0, 4.0 * m * m);
  if (theta_min > 0.0 && lmax > 0.0) ratio = std::min(ratio, std::max(4.0, 3.0 * lmax / theta_min));
  if (ratio_env > 1.0) ratio = ratio_env;
  const double lmin = lmax / ratio;
  const double theta = 0.5 * (lmax + lmin), delta = 0.5 * (lmax - lmin);
  if (!comm && tg_cg_persistent_applies(k)) {
    // small systems: the loop below as one persistent kernel (tg_krylov_small.hip); 100 = not taken
    const int rcp = tg_pcg_cheb_persistent(k, b, x, m, theta, delta, rtol, atol, maxit, nonzero_guess, iters, resnorm, status);
    if (rcp != 100) return rcp;
  }
  // u = B r
  auto apply_pc = [&]() -> int {
    hipLaunchKernelGGL(k_cheb_first, dim3(vg), dim3(256), 0, g_tg.stream, r, dinv, 1.0 / theta, n, g, u, dd);
    const double sigma = theta / delta;
    double rho = 1.0 / sigma;
    for (int j = 1; j < m; j++) {
      const double rho_new = 1.0 / (2.0 * sigma - rho);
      TG_TRY(product(kz));
      hipLaunchKernelGGL(k_cheb_step, dim3(vg), dim3(256), 0, g_tg.stream, g, dinv, kz, rho_new * rho, 2.0 * rho_new / delta, n, u, dd);
      rho = rho_new;
    }
    TG_LAUNCH_CHECK();
    return 0;
  };
  // ---- reference norm ||B b|| and the initial residual
  double h3[3];
  hipLaunchKernelGGL(k_residual, dim3(vg), dim3(256), 0, g_tg.stream, b->d, (const double *)nullptr, n, r);
  TG_TRY(apply_pc());
  hipLaunchKernelGGL(k_dots3, dim3(vg), dim3(256), 0, g_tg.stream, r, u, u, n, part);
  TG_TRY(reduce(3, h3));
  const double bnorm = sqrt(h3[2]);
  if (nonzero_guess) {
    TG_CHECK_HIP(hipMemcpyAsync(u, x->d, (size_t)n * sizeof(double), hipMemcpyDeviceToDevice, g_tg.stream));
    TG_TRY(product(w));
    hipLaunchKernelGGL(k_residual, dim3(vg), dim3(256), 0, g_tg.stream, b->d, (const double *)w, n, r);
    TG_TRY(apply_pc());
  } else {
    TG_CHECK_HIP(hipMemsetAsync(x->d, 0, (size_t)std::max<int64_t>(n, 1) * sizeof(double), g_tg.stream));
  }
  const double tol = std::max(rtol * bnorm, atol);
  *iters = 0;
  *status = -1;
  double gamma_prev = 1.0, alpha_prev = 1.0, znorm = bnorm;
  TG_CHECK_HIP(hipMemsetAsync(p, 0, (size_t)(2 * n) * sizeof(double), g_tg.stream));   // p, s adjacent
  for (int it = 0; it <= maxit; it++) {
    // w = K u ; gamma = (r,u), delta = (w,u), nu = (u,u)
    hipEventRecord(e0, g_tg.stream);
    TG_TRY(product(w));
    hipEventRecord(e1, g_tg.stream);
    hipLaunchKernelGGL(k_dots3, dim3(vg), dim3(256), 0, g_tg.stream, r, u, w, n, part);
    TG_TRY(reduce(3, h3));
    {
      float ems = 0.f;
      if (hipEventElapsedTime(&ems, e0, e1) == hipSuccess) {
        g_tg.prof_ms[TG_PROF_KSP_SPMV] += ems;
        g_tg.prof_n[TG_PROF_KSP_SPMV] += 1;
      }
    }
    const double gamma = h3[0], delta_ = h3[1], nu = h3[2];
    znorm = sqrt(nu);
    *iters = it;
    if (!(nu == nu) || !(gamma == gamma)) {
      *status = -2;
      break;
    }
    if (znorm <= tol) {
      *status = (znorm <= atol && !(znorm <= rtol * bnorm)) ? 1 : 0;
      break;
    }
    if (it == maxit) break;
    double beta = 0.0, alpha;
    if (it == 0)
      alpha = gamma / delta_;
    else {
      beta = gamma / gamma_prev;
      alpha = gamma / (delta_ - beta * gamma / alpha_prev);
    }
    if (!(alpha == alpha) || alpha == 0.0 || !(gamma > 0.0)) {     // (a polynomial that is not positive definite, breakdown)
      *status = -2;
      break;
    }
    gamma_prev = gamma;
    alpha_prev = alpha;
    hipLaunchKernelGGL(k_pcg_update, dim3(vg), dim3(256), 0, g_tg.stream, u, w, alpha, beta, n, p, s, x->d, r);
    TG_TRY(apply_pc());
  }
  *resnorm = znorm;
  TG_CHECK_HIP(hipStreamSynchronize(g_tg.stream));
  TG_TRY(tg_comm_check(comm));
  return 0;
}

// ------------------------------------------------------------------------------------ BiCGStab
// KSPBCGS [ext] with left preconditioning (Jacobi or none): the method runs on B K x = B b, the convergence test reads the
// preconditioned residual like CG / GMRES above.  Two fused reductions per iteration, one per half step:
//   (rhat, v)                                         -> alpha
//   (t,s), (t,t), (rhat,s), (rhat,t), (s,s)           -> omega, rho' = (rhat,s) - omega (rhat,t),
//                                                        ||r'||^2 = (s,s) - 2 omega (t,s) + omega^2 (t,t)
// Non-symmetric systems with a short recurrence (no basis of 30 vectors).
__global__ void __launch_bounds__(256) k_bcgs_p(const double *__restrict__ r, const double *__restrict__ v, double beta, double omega,
                                                int64_t n, double *__restrict__ p) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) p[i] = r[i] + beta * (p[i] - omega * v[i]);
}
// v = dinv .* kp ; partial of (rhat, v)
__global__ void __launch_bounds__(256) k_bcgs_v(const double *__restrict__ kp, const double *__restrict__ dinv,
                                                const double *__restrict__ rhat, int64_t n, double *__restrict__ v,
                                                double *__restrict__ partial) {
  __shared__ double lds4[4];
  double a = 0.0;
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    const double vi = dinv[i] * kp[i];
    v[i] = vi;
    a += rhat[i] * vi;
  }
  a = tg_block_sum256(a, lds4);
  if (threadIdx.x == 0) partial[blockIdx.x] = a;
}
// s = r - alpha v (written into the product operand)
__global__ void __launch_bounds__(256) k_bcgs_s(const double *__restrict__ r, const double *__restrict__ v, double alpha, int64_t n,
                                                double *__restrict__ s) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) s[i] = r[i] - alpha * v[i];
}
// t = dinv .* ks ; partials of (t,s), (t,t), (rhat,s), (rhat,t), (s,s)
__global__ void __launch_bounds__(256) k_bcgs_t(const double *__restrict__ ks, const double *__restrict__ dinv, const double *__restrict__ s,
                                                const double *__restrict__ rhat, int64_t n, double *__restrict__ t,
                                                double *__restrict__ partial) {
  __shared__ double lds4[4];
  double a = 0.0, b = 0.0, c = 0.0, d = 0.0, e = 0.0;
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    const double ti = dinv[i] * ks[i], si = s[i], hi = rhat[i];
    t[i] = ti;
    a += ti * si;
    b += ti * ti;
    c += hi * si;
    d += hi * ti;
    e += si * si;
  }
  a = tg_block_sum256(a, lds4);
  b = tg_block_sum256(b, lds4);
  c = tg_block_sum256(c, lds4);
  d = tg_block_sum256(d, lds4);
  e = tg_block_sum256(e, lds4);
  if (threadIdx.x == 0) {
    double *q = partial + 5 * blockIdx.x;
    q[0] = a, q[1] = b, q[2] = c, q[3] = d, q[4] = e;
  }
}
// x += alpha p + omega s ; r = s - omega t
__global__ void __launch_bounds__(256) k_bcgs_x(const double *__restrict__ p, const double *__restrict__ s, const double *__restrict__ t,
                                                double alpha, double omega, int64_t n, double *__restrict__ x, double *__restrict__ r) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    x[i] += alpha * p[i] + omega * s[i];
    r[i] = s[i] - omega * t[i];
  }
}

static int tg_bicgstab(tg_csr_s *k, tg_vec_s *b, tg_vec_s *x, int pc, double rtol, double atol, int maxit, int nonzero_guess,
                       tg_comm_s *comm, int *iters, double *resnorm, int *status) {
  const int64_t n = k->nrows;
  const int64_t hlo = comm ? comm->halo_lo : 0, hhi = comm ? comm->halo_hi : 0;
  const int64_t row0 = comm ? comm->g0 : 0;
  const int64_t next = hlo + n + hhi;
  tg_krylov_ws ws;
  // layout: opext[next] (operand of the products: p, then s) | r | rhat | v | t | kp | dinv | pvec
  TG_TRY(tg_dmalloc(&ws.buf, next + 7 * n));
  double *opext = ws.buf, *op = opext + hlo, *r = opext + next, *rhat = r + n, *v = rhat + n, *t = v + n, *kp = t + n,
         *dinv = kp + n, *pv = dinv + n;
  TG_CHECK_HIP(hipMemsetAsync(ws.buf, 0, (size_t)(next + 7 * n) * sizeof(double), g_tg.stream));
  double *part = g_tg.scratch;                       // 5 * TG_VEC_BLOCKS
  double *sc = g_tg.scratch + TG_SCRATCH_DOUBLES - 2048;
  const int vg = tg_vec_grid(n);
  TG_TRY(tg_spmv_plan(k));
  tg_sell_guard sell_guard(k);
  TG_TRY(sell_guard.rc);
  const double *opshift = opext - (row0 - hlo);
  const int64_t cmin = row0 - hlo, cmax = row0 - hlo + next - 1;
  auto product = [&](double *out) -> int {
    TG_TRY(tg_comm_halo_exchange(comm, opext));
    return tg_spmv_raw(k, opshift, cmin, cmax, out);
  };
  auto reduce = [&](int cnt, double *host) -> int {
    hipLaunchKernelGGL(k_fold, dim3(1), dim3(256), 0, g_tg.stream, part, vg, cnt, sc);
    TG_LAUNCH_CHECK();
    TG_TRY(tg_comm_allreduce_dev(comm, sc, cnt));
    return tg_read_scalars(sc, cnt, host);
  };
  if (n > 0) {
    const unsigned jg = (unsigned)std::min<int64_t>(tg_cdiv(n, 4), (int64_t)g_tg.num_cu * 16);
    hipLaunchKernelGGL(k_jacobi_setup, dim3(jg), dim3(256), 0, g_tg.stream, k->rowptr, k->col, k->val, n, row0,
                       pc == TG_PC_JACOBI ? 1 : 0, dinv);
  }
  // reference norm ||B b||; r = B (b - K x0)
  double h5[5];
  hipLaunchKernelGGL(k_scaled_copy_norm, dim3(vg), dim3(256), 0, g_tg.stream, b->d, (const double *)dinv, 1.0, n, r, part);
  TG_TRY(reduce(1, h5));
  const double bnorm = sqrt(h5[0]);
  double rnorm2 = h5[0];
  if (nonzero_guess) {
    TG_CHECK_HIP(hipMemcpyAsync(op, x->d, (size_t)n * sizeof(double), hipMemcpyDeviceToDevice, g_tg.stream));
    TG_TRY(product(kp));
    hipLaunchKernelGGL(k_residual, dim3(vg), dim3(256), 0, g_tg.stream, b->d, (const double *)kp, n, t);
    hipLaunchKernelGGL(k_scaled_copy_norm, dim3(vg), dim3(256), 0, g_tg.stream, t, (const double *)dinv, 1.0, n, r, part);
    TG_TRY(reduce(1, h5));
    rnorm2 = h5[0];
  } else {
    TG_CHECK_HIP(hipMemsetAsync(x->d, 0, (size_t)std::max<int64_t>(n, 1) * sizeof(double), g_tg.stream));
  }
  TG_CHECK_HIP(hipMemcpyAsync(rhat, r, (size_t)n * sizeof(double), hipMemcpyDeviceToDevice, g_tg.stream));
  const double tol = std::max(rtol * bnorm, atol);
  double rho = rnorm2, alpha = 1.0, omega = 1.0, beta = 0.0, znorm = sqrt(rnorm2);
  *iters = 0;
  *status = -1;
  if (!(znorm == znorm)) {
    *status = -2;
  } else if (znorm <= tol) {
    *status = (znorm <= atol && !(znorm <= rtol * bnorm)) ? 1 : 0;
  } else {
    for (int it = 1; it <= maxit; it++) {
      // p = r + beta (p - omega v)   (first step: p = r since p = v = 0, beta = 0)
      hipLaunchKernelGGL(k_bcgs_p, dim3(vg), dim3(256), 0, g_tg.stream, r, v, beta, omega, n, pv);
      TG_CHECK_HIP(hipMemcpyAsync(op, pv, (size_t)n * sizeof(double), hipMemcpyDeviceToDevice, g_tg.stream));
      TG_TRY(product(kp));
      hipLaunchKernelGGL(k_bcgs_v, dim3(vg), dim3(256), 0, g_tg.stream, kp, dinv, rhat, n, v, part);
      TG_TRY(reduce(1, h5));
      const double rv = h5[0];
      if (!(rv == rv) || rv == 0.0) {
        *status = -2;
        *iters = it;
        break;
      }
      alpha = rho / rv;
      hipLaunchKernelGGL(k_bcgs_s, dim3(vg), dim3(256), 0, g_tg.stream, r, v, alpha, n, op);
      TG_TRY(product(kp));
      hipLaunchKernelGGL(k_bcgs_t, dim3(vg), dim3(256), 0, g_tg.stream, kp, dinv, op, rhat, n, t, part);
      TG_TRY(reduce(5, h5));
      const double ts = h5[0], tt = h5[1], hs = h5[2], ht = h5[3], ss = h5[4];
      omega = tt > 0.0 ? ts / tt : 0.0;
      hipLaunchKernelGGL(k_bcgs_x, dim3(vg), dim3(256), 0, g_tg.stream, pv, op, t, alpha, omega, n, x->d, r);
      TG_LAUNCH_CHECK();
      const double rho_new = hs - omega * ht;
      // (max / fmax return the other argument for a NaN: a NaN in the sums -- an Inf in K, say -- must stay one)
      const double rn2raw = (ss - 2.0 * omega * ts + omega * omega * tt) + 0.0 * (ts + tt + hs + ht);
      const double rn2 = rn2raw < 0.0 ? 0.0 : rn2raw;
      znorm = sqrt(rn2);
      *iters = it;
      if (!(znorm == znorm)) {
        *status = -2;
        break;
      }
      if (znorm <= tol) {
        *status = (znorm <= atol && !(znorm <= rtol * bnorm)) ? 1 : 0;
        break;
      }
      if (omega == 0.0 || rho == 0.0 || !(rho_new == rho_new)) {
        *status = -2;
        break;
      }
      beta = (rho_new / rho) * (alpha / omega);
      rho = rho_new;
    }
  }
  *resnorm = znorm;
  TG_CHECK_HIP(hipStreamSynchronize(g_tg.stream));
  TG_TRY(tg_comm_check(comm));
  return 0;
}

extern "C" int tg_krylov_solve(tg_csr_t k, tg_vec_t b, tg_vec_t x, int method, int pc, double rtol, double atol,
                               int maxit, int restart, tg_comm_t comm, int *iters, double *resnorm, int *status) {
  return tg_krylov_solve_flags(k, b, x, method, pc, rtol, atol, maxit, restart, 0, comm, iters, resnorm, status);
}

extern "C" int tg_krylov_solve_flags(tg_csr_t k, tg_vec_t b, tg_vec_t x, int method, int pc, double rtol, double atol,
                                     int maxit, int restart, int flags, tg_comm_t comm, int *iters, double *resnorm,
                                     int *status) {
  const int g_krylov_nonzero_guess = (flags & TG_KSP_NONZERO_GUESS) ? 1 : 0;
  g_ksp_symmetric_hint = (flags & TG_KSP_SYMMETRIC) ? 1 : 0;
  TG_REQUIRE_INIT();
  TG_REQUIRE(k && b && x && iters && resnorm && status, "null argument to tg_krylov_solve");
  TG_REQUIRE(b->n == k->nrows && x->n == k->nrows, "tg_krylov_solve: vector length != local rows");
  if (comm && comm->world > 1) {
    TG_REQUIRE(comm->slab_set, "tg_comm_set_slab() must precede a distributed solve");
    TG_REQUIRE(comm->g1 - comm->g0 == k->nrows && comm->nglobal == k->ncols, "slab does not match the matrix block");
  } else {
    TG_REQUIRE(k->nrows == k->ncols, "tg_krylov_solve: matrix must be square");
    comm = nullptr;
  }
  if (method == TG_KSP_CG && pc == TG_PC_CHEBYSHEV)
    // (`restart` carries the degree of the polynomial: the number of products per application + 1)
    return tg_pcg_cheb(k, b, x, restart, rtol, atol, maxit, g_krylov_nonzero_guess, comm, iters, resnorm, status);
  TG_REQUIRE(pc == TG_PC_NONE || pc == TG_PC_JACOBI, "the Chebyshev polynomial preconditioner serves CG only");
  if (method == TG_KSP_CG && !comm && tg_cg_persistent_applies(k)) {
    // small systems (K in the Infinity Cache): the whole loop in one persistent kernel; 100 = could not run, the loop below does
    const int rcp = tg_cg_persistent(k, b, x, pc, rtol, atol, maxit, g_krylov_nonzero_guess, iters, resnorm, status);
    if (rcp != 100) return rcp;
  }
  if (method == TG_KSP_CG) return tg_cg(k, b, x, pc, rtol, atol, maxit, g_krylov_nonzero_guess, comm, iters, resnorm, status);
  if (method == TG_KSP_BICGSTAB && !comm && tg_cg_persistent_applies(k)) {
    const int rcp = tg_bicgstab_persistent(k, b, x, pc, rtol, atol, maxit, g_krylov_nonzero_guess, iters, resnorm, status);
    if (rcp != 100) return rcp;
  }
  if (method == TG_KSP_BICGSTAB)
    return tg_bicgstab(k, b, x, pc, rtol, atol, maxit, g_krylov_nonzero_guess, comm, iters, resnorm, status);
  if (method == TG_KSP_GMRES) {
    TG_REQUIRE(restart >= 1 && restart <= 200, "GMRES restart out of range");
    if (!comm && !(flags & TG_KSP_STAGNATION_GUARD) && tg_cg_persistent_applies(k)) {
      const int rcp = tg_gmres_persistent(k, b, x, pc, rtol, atol, maxit, restart, g_krylov_nonzero_guess, iters, resnorm, status);
      if (rcp != 100) return rcp;
    }
    return tg_gmres(k, b, x, pc, rtol, atol, maxit, restart, g_krylov_nonzero_guess, (flags & TG_KSP_STAGNATION_GUARD) ? 1 : 0,
                    comm, iters, resnorm, status);
  }
  tg_set_error("unknown Krylov method %d", method);
  return 2;
}
