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
  explicit tg_sell_guard(tg_csr_s *m) : k(m) {
    if (k->sell_state == 0) {
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

// r = b ; z = dinv*r ; p = z ; partial sums of r.z and z.z
__global__ void __launch_bounds__(256) k_cg_init(const double *__restrict__ b, const double *__restrict__ dinv,
                                                 double *__restrict__ r, double *__restrict__ p, double *__restrict__ x,
                                                 int64_t n, double *__restrict__ partial) {
  __shared__ double lds4[4];
  double rz = 0.0, zz = 0.0;
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    const double ri = b[i];
    const double zi = dinv[i] * ri;
    r[i] = ri;
    p[i] = zi;
    x[i] = 0.0;
    rz += ri * zi;
    zz += zi * zi;
  }
  rz = tg_block_sum256(rz, lds4);
  zz = tg_block_sum256(zz, lds4);
  if (threadIdx.x == 0) {
    partial[2 * blockIdx.x] = rz;
    partial[2 * blockIdx.x + 1] = zz;
  }
}

// alpha = rz_old / pKp ; x += alpha p ; r -= alpha Kp ; z = dinv r ; partials of r.z, z.z
__global__ void __launch_bounds__(256)
    k_cg_update(const double *__restrict__ scal_rz_old, const double *__restrict__ scal_pkp,
                const double *__restrict__ p, const double *__restrict__ kp, const double *__restrict__ dinv,
                double *__restrict__ x, double *__restrict__ r, int64_t n, double *__restrict__ partial) {
  __shared__ double lds4[4];
  const double alpha = scal_rz_old[0] / scal_pkp[0];
  double rz = 0.0, zz = 0.0;
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    x[i] += alpha * p[i];
    const double ri = r[i] - alpha * kp[i];
    r[i] = ri;
    const double zi = dinv[i] * ri;
    rz += ri * zi;
    zz += zi * zi;
  }
  rz = tg_block_sum256(rz, lds4);
  zz = tg_block_sum256(zz, lds4);
  if (threadIdx.x == 0) {
    partial[2 * blockIdx.x] = rz;
    partial[2 * blockIdx.x + 1] = zz;
  }
}

// beta = rz_new / rz_old ; p = dinv r + beta p
__global__ void __launch_bounds__(256)
    k_cg_direction(const double *__restrict__ scal_rz_new, const double *__restrict__ scal_rz_old,
                   const double *__restrict__ r, const double *__restrict__ dinv, double *__restrict__ p, int64_t n) {
  const double beta = scal_rz_new[0] / scal_rz_old[0];
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) p[i] = dinv[i] * r[i] + beta * p[i];
}

__global__ void __launch_bounds__(256) k_dot2_partial(const double *__restrict__ a, const double *__restrict__ b,
                                                      int64_t n, double *__restrict__ partial) {
  __shared__ double lds4[4];
  double s = 0.0;
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) s += a[i] * b[i];
  s = tg_block_sum256(s, lds4);
  if (threadIdx.x == 0) partial[blockIdx.x] = s;
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

static int tg_cg(tg_csr_s *k, tg_vec_s *b, tg_vec_s *x, int pc, double rtol, double atol, int maxit, tg_comm_s *comm,
                 int *iters, double *resnorm, int *status) {
  const int64_t n = k->nrows;
  const int64_t hlo = comm ? comm->halo_lo : 0, hhi = comm ? comm->halo_hi : 0;
  const int64_t row0 = comm ? comm->g0 : 0;
  const int64_t next = hlo + n + hhi;
  tg_krylov_ws ws;
  // layout: pext[next] | r[n] | kp[n] | dinv[n]
  TG_TRY(tg_dmalloc(&ws.buf, next + 3 * n));
  double *pext = ws.buf, *p = pext + hlo, *r = pext + next, *kp = r + n, *dinv = kp + n;
  TG_CHECK_HIP(hipMemsetAsync(pext, 0, (size_t)next * sizeof(double), g_tg.stream));
  double *partial = g_tg.scratch;                       // up to 32768 partial doubles
  double *scal = g_tg.scratch + TG_SCRATCH_DOUBLES - 2048;  // rz[2], pkp, rz/zz pair
  double *s_rz = scal;        // [0],[1] alternate old/new
  double *s_pkp = scal + 2;
  double *s_pair = scal + 4;  // (rz_new, zz) written together
  const int vg = tg_vec_grid(n);
  TG_TRY(tg_spmv_plan(k));
  tg_sell_guard sell_guard(k);   // sliced copy of the values for the products of this solve
  TG_TRY(sell_guard.rc);

  if (n > 0) {
    const unsigned jg = (unsigned)std::min<int64_t>(tg_cdiv(n, 4), (int64_t)g_tg.num_cu * 16);
    hipLaunchKernelGGL(k_jacobi_setup, dim3(jg), dim3(256), 0, g_tg.stream, k->rowptr, k->col, k->val, n, row0,
                       pc == TG_PC_JACOBI ? 1 : 0, dinv);
  }
  hipLaunchKernelGGL(k_cg_init, dim3(vg), dim3(256), 0, g_tg.stream, b->d, dinv, r, p, x->d, n, partial);
  hipLaunchKernelGGL(k_fold, dim3(1), dim3(256), 0, g_tg.stream, partial, vg, 2, s_pair);
  TG_LAUNCH_CHECK();
  TG_TRY(tg_comm_allreduce_dev(comm, s_pair, 2));
  double h[2];
  TG_TRY(tg_read_scalars(s_pair, 2, h));
  // rz_old <- rz
  TG_CHECK_HIP(hipMemcpyAsync(s_rz, s_pair, sizeof(double), hipMemcpyDeviceToDevice, g_tg.stream));
  const double znorm0 = sqrt(h[1]);
  const double tol = std::max(rtol * znorm0, atol);
  double znorm = znorm0;
  *iters = 0;
  *status = 0;
  if (!(znorm0 == znorm0)) {
    *status = -2;
    *resnorm = znorm0;
    return 0;
  }
  if (znorm0 <= atol) {
    *status = 1;
    *resnorm = znorm0;
    return 0;
  }
  int it = 0;
  *status = -1;
  double t_launch = 0, t_sync = 0, t_all0 = tk_now();
  for (it = 1; it <= maxit; it++) {
    const double _ta = tk_now();
    double *rz_old = s_rz + ((it - 1) & 1), *rz_new = s_rz + (it & 1);
    TG_TRY(tg_comm_halo_exchange(comm, pext));
    // Kp = K p   (x addressed by global column index); then p . Kp on a fixed grid so the
    // reduction order (and the result) does not depend on the matrix size
    hipEventRecord(g_tg.pev0, g_tg.stream);
    TG_TRY(tg_spmv_raw(k, pext - (row0 - hlo), row0 - hlo, row0 - hlo + next - 1, kp));
    hipEventRecord(g_tg.pev1, g_tg.stream);
    hipLaunchKernelGGL(k_dot2_partial, dim3(vg), dim3(256), 0, g_tg.stream, p, kp, n, partial);
    hipLaunchKernelGGL(k_fold, dim3(1), dim3(256), 0, g_tg.stream, partial, vg, 1, s_pkp);
    TG_TRY(tg_comm_allreduce_dev(comm, s_pkp, 1));
    hipLaunchKernelGGL(k_cg_update, dim3(vg), dim3(256), 0, g_tg.stream, rz_old, s_pkp, p, kp, dinv, x->d, r, n,
                       partial);
    hipLaunchKernelGGL(k_fold, dim3(1), dim3(256), 0, g_tg.stream, partial, vg, 2, s_pair);
    TG_LAUNCH_CHECK();
    TG_TRY(tg_comm_allreduce_dev(comm, s_pair, 2));
    TG_CHECK_HIP(hipMemcpyAsync(rz_new, s_pair, sizeof(double), hipMemcpyDeviceToDevice, g_tg.stream));
    const double _tb = tk_now();
    TG_TRY(tg_read_scalars(s_pair, 2, h));
    t_launch += _tb - _ta;
    t_sync += tk_now() - _tb;
    {
      float ems = 0.f;  // the sync above also completed this iteration's SpMV event pair
      if (hipEventElapsedTime(&ems, g_tg.pev0, g_tg.pev1) == hipSuccess) {
        g_tg.prof_ms[TG_PROF_KSP_SPMV] += ems;
        g_tg.prof_n[TG_PROF_KSP_SPMV] += 1;
      }
    }
    znorm = sqrt(h[1]);
    if (!(znorm == znorm)) {
      *status = -2;
      break;
    }
    if (znorm <= tol) {
      *status = (znorm <= atol && !(znorm <= rtol * znorm0)) ? 1 : 0;
      break;
    }
    hipLaunchKernelGGL(k_cg_direction, dim3(vg), dim3(256), 0, g_tg.stream, rz_new, rz_old, r, dinv, p, n);
    TG_LAUNCH_CHECK();
  }
  *iters = std::min(it, maxit);
  *resnorm = znorm;
  TG_CHECK_HIP(hipStreamSynchronize(g_tg.stream));
  if (getenv("TIGAR_TRACE"))
    fprintf(stderr, "[trace] cg: %d its, loop %.3f s (launch %.3f s, sync wait %.3f s)\n", it, tk_now() - t_all0,
            t_launch, t_sync);
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

static int tg_gmres(tg_csr_s *k, tg_vec_s *b, tg_vec_s *x, int pc, double rtol, double atol, int maxit, int restart,
                    tg_comm_s *comm, int *iters, double *resnorm, int *status) {
  const int64_t n = k->nrows;
  const int64_t hlo = comm ? comm->halo_lo : 0, hhi = comm ? comm->halo_hi : 0;
  const int64_t row0 = comm ? comm->g0 : 0;
  const int64_t next = hlo + n + hhi;
  const int m = restart;
  tg_krylov_ws ws;
  // layout: ext[next] (SpMV input with halo) | w[n] | dinv[n] | V[(m+1) n] | hdev[m+2]
  TG_TRY(tg_dmalloc(&ws.buf, next + 2 * n + (int64_t)(m + 1) * n + (m + 2)));
  double *ext = ws.buf, *xin = ext + hlo, *w = ext + next, *dinv = w + n, *V = dinv + n,
         *hdev = V + (int64_t)(m + 1) * n;
  TG_CHECK_HIP(hipMemsetAsync(ext, 0, (size_t)next * sizeof(double), g_tg.stream));
  double *partial = g_tg.scratch;
  double *scal = g_tg.scratch + TG_SCRATCH_DOUBLES - 2048;
  const int vg = std::min(tg_vec_grid(n), 256);
  TG_TRY(tg_spmv_plan(k));
  tg_sell_guard sell_guard(k);   // sliced copy of the values for the products of this solve
  TG_TRY(sell_guard.rc);
  if (n > 0) {
    const unsigned jg = (unsigned)std::min<int64_t>(tg_cdiv(n, 4), (int64_t)g_tg.num_cu * 16);
    hipLaunchKernelGGL(k_jacobi_setup, dim3(jg), dim3(256), 0, g_tg.stream, k->rowptr, k->col, k->val, n, row0,
                       pc == TG_PC_JACOBI ? 1 : 0, dinv);
  }
  TG_CHECK_HIP(hipMemsetAsync(x->d, 0, (size_t)std::max<int64_t>(n, 1) * sizeof(double), g_tg.stream));
  std::vector<double> H((size_t)(m + 1) * m, 0.0), cs(m), sn(m), g(m + 1), y(m), hcol(m + 2);
  auto Hat = [&](int i, int j) -> double & { return H[(size_t)i * m + j]; };
  double beta0 = -1.0, tol = 0.0, res = 0.0;
  int its = 0;
  *status = -1;
  bool first = true;
  while (its < maxit) {
    // r = B (b - K x)
    if (first) {
      hipLaunchKernelGGL(k_prec_residual, dim3(vg), dim3(256), 0, g_tg.stream, b->d, (const double *)nullptr, dinv, V,
                         n, partial);
    } else {
      TG_CHECK_HIP(hipMemcpyAsync(xin, x->d, (size_t)n * sizeof(double), hipMemcpyDeviceToDevice, g_tg.stream));
      TG_TRY(tg_comm_halo_exchange(comm, ext));
      TG_TRY(tg_spmv_raw(k, ext - (row0 - hlo), row0 - hlo, row0 - hlo + next - 1, w));
      hipLaunchKernelGGL(k_prec_residual, dim3(vg), dim3(256), 0, g_tg.stream, b->d, (const double *)w, dinv, V, n,
                         partial);
    }
    hipLaunchKernelGGL(k_fold, dim3(1), dim3(256), 0, g_tg.stream, partial, vg, 1, scal);
    TG_LAUNCH_CHECK();
    TG_TRY(tg_comm_allreduce_dev(comm, scal, 1));
    double hh;
    TG_TRY(tg_read_scalars(scal, 1, &hh));
    const double beta = sqrt(hh);
    res = beta;
    if (first) {
      beta0 = beta;
      tol = std::max(rtol * beta0, atol);
      first = false;
      if (!(beta == beta)) {
        *status = -2;
        break;
      }
      if (beta0 <= atol) {
        *status = 1;
        break;
      }
    }
    if (beta <= tol) {
      *status = 0;
      break;
    }
    hipLaunchKernelGGL(k_scale_to, dim3(vg), dim3(256), 0, g_tg.stream, V, V, 1.0 / beta, n);
    std::fill(g.begin(), g.end(), 0.0);
    g[0] = beta;
    int kused = 0;
    bool done = false;
    for (int j = 0; j < m; j++) {
      // w = B K v_j
      TG_CHECK_HIP(hipMemcpyAsync(xin, V + (int64_t)j * n, (size_t)n * sizeof(double), hipMemcpyDeviceToDevice,
                                  g_tg.stream));
      TG_TRY(tg_comm_halo_exchange(comm, ext));
      TG_TRY(tg_spmv_raw(k, ext - (row0 - hlo), row0 - hlo, row0 - hlo + next - 1, w));
      hipLaunchKernelGGL(k_mul_inplace, dim3(vg), dim3(256), 0, g_tg.stream, w, dinv, n);
      // classical Gram-Schmidt (PETSc default, no refinement [ext])
      hipLaunchKernelGGL(k_multi_dot, dim3(vg), dim3(256), 0, g_tg.stream, V, n, j + 1, w, n, partial);
      hipLaunchKernelGGL(k_fold, dim3(1), dim3(256), 0, g_tg.stream, partial, vg, j + 1, hdev);
      TG_TRY(tg_comm_allreduce_dev(comm, hdev, j + 1));
      hipLaunchKernelGGL(k_gs_update, dim3(vg), dim3(256), 0, g_tg.stream, V, n, j + 1, hdev, w, n, partial);
      hipLaunchKernelGGL(k_fold, dim3(1), dim3(256), 0, g_tg.stream, partial, vg, 1, hdev + j + 1);
      TG_LAUNCH_CHECK();
      TG_TRY(tg_comm_allreduce_dev(comm, hdev + j + 1, 1));
      TG_TRY(tg_read_scalars(hdev, j + 2, hcol.data()));
      for (int i = 0; i <= j; i++) Hat(i, j) = hcol[i];
      const double hn = sqrt(hcol[j + 1]);
      Hat(j + 1, j) = hn;
      if (hn != 0.0)
        hipLaunchKernelGGL(k_scale_to, dim3(vg), dim3(256), 0, g_tg.stream, V + (int64_t)(j + 1) * n, w, 1.0 / hn, n);
      for (int i = 0; i < j; i++) {
        const double t = cs[i] * Hat(i, j) + sn[i] * Hat(i + 1, j);
        Hat(i + 1, j) = -sn[i] * Hat(i, j) + cs[i] * Hat(i + 1, j);
        Hat(i, j) = t;
      }
      const double den = hypot(Hat(j, j), Hat(j + 1, j));
      if (den == 0.0 || !(den == den)) {
        *status = -2;
        done = true;
        kused = j;
        break;
      }
      cs[j] = Hat(j, j) / den;
      sn[j] = Hat(j + 1, j) / den;
      Hat(j, j) = den;
      Hat(j + 1, j) = 0.0;
      g[j + 1] = -sn[j] * g[j];
      g[j] = cs[j] * g[j];
      its++;
      kused = j + 1;
      res = fabs(g[j + 1]);
      if (res <= tol) {
        *status = 0;
        done = true;
        break;
      }
      if (its >= maxit) {
        done = true;
        break;
      }
    }
    // back substitution, x += V y
    for (int i = kused - 1; i >= 0; i--) {
      double s = g[i];
      for (int c = i + 1; c < kused; c++) s -= Hat(i, c) * y[c];
      y[i] = s / Hat(i, i);
    }
    if (kused > 0) {
      TG_CHECK_HIP(hipMemcpyAsync(hdev, y.data(), kused * sizeof(double), hipMemcpyHostToDevice, g_tg.stream));
      hipLaunchKernelGGL(k_lincomb_add, dim3(vg), dim3(256), 0, g_tg.stream, x->d, V, n, kused, hdev, n);
      TG_LAUNCH_CHECK();
      TG_CHECK_HIP(hipStreamSynchronize(g_tg.stream));
    }
    if (done) break;
  }
  *iters = its;
  *resnorm = res;
  TG_CHECK_HIP(hipStreamSynchronize(g_tg.stream));
  return 0;
}

extern "C" int tg_krylov_solve(tg_csr_t k, tg_vec_t b, tg_vec_t x, int method, int pc, double rtol, double atol,
                               int maxit, int restart, tg_comm_t comm, int *iters, double *resnorm, int *status) {
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
  if (method == TG_KSP_CG) return tg_cg(k, b, x, pc, rtol, atol, maxit, comm, iters, resnorm, status);
  if (method == TG_KSP_GMRES) {
    TG_REQUIRE(restart >= 1 && restart <= 200, "GMRES restart out of range");
    return tg_gmres(k, b, x, pc, rtol, atol, maxit, restart, comm, iters, resnorm, status);
  }
  tg_set_error("unknown Krylov method %d", method);
  return 2;
}
