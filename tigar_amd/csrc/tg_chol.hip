// Direct solve of K U = M^T b when K is symmetric positive definite: blocked banded Cholesky on the matrix cores.
//
// With `linearSolver == None` the reference calls dolfin's `solve(A, x, b)`, a sparse direct LU (tIGAr/common.py:1255-1256
// [ext]); tg_lu.hip restates it as a banded LU with partial pivoting, whose column-by-column pivot search is a chain of
// 67 600 dependent steps at cfg4 (0.26 s of panel + trailing kernels, 0.09 s of substitution).  The systems the reference's
// demos hand to it are mostly Galerkin matrices of coercive forms with Dirichlet rows and columns replaced by the identity
// (MatZeroRowsColumns): symmetric positive definite.  For those no pivoting is needed and the factorisation is dense block
// algebra: per block of NB = 32 columns a 32 x 32 Cholesky factor, a triangular solve for the kl rows below it, and a
// symmetric rank-32 update of the trailing kl x kl triangle on `v_mfma_f64_16x16x4_f64` -- two launches per 32 columns
// instead of two per 16 with a pivot search each (k_chol_panel + k_chol_syrk), ONE where the band is narrow and the launches
// are a chain of latencies (k_chol_fused: the update's workgroups solve the panel rows they need, the next diagonal block is
// factorised by the first tile), groups of 2 / 4 panels per pass where it is wide and the pass is bound by the triangle it
// moves -- and substitutions that go block by block on 8 .. 64 workgroups (k_chol_sweep).
//
// tg_lu_solve tries this first: K's values are compared with their transposes (relative 1e-11: a K that is symmetric "to
// rounding", as M^T A M comes out, qualifies -- the factor is that of the lower triangle, a perturbation of the order of the
// rounding errors K carries anyway); a pivot that is not positive ends the attempt and the LU runs as before
// (TIGAR_LU_CHOLESKY=0: never tried).  The solution is that of the same system, to the rounding of a backward stable
// direct method; PETSc users select it by hand (-pc_type cholesky [ext]).
//
// Storage: LAPACK's dpbtrf 'L' layout, lb[(r - c) + ldl * c] for c <= r <= c + kl, ldl = kl + 1.
#include "tg_common.h"
#include <algorithm>

#define CH_NB 32
typedef double ch_v4d __attribute__((ext_vector_type(4)));
typedef unsigned int ch_u2 __attribute__((ext_vector_type(2)));

struct ch_stats {
  unsigned long long lower, upper;     // entries strictly below / above the diagonal
  unsigned long long maxabs;           // bit pattern of max |a|
  int asym;                            // an entry differs from its transpose
  int notpd;                           // 1 + column of the first pivot that is not positive
};

__global__ void __launch_bounds__(256)
    k_chol_scatter(const int64_t *__restrict__ rowptr, const int32_t *__restrict__ col, const double *__restrict__ val, int64_t n,
                   int64_t ldl, double *__restrict__ lb, ch_stats *st) {
  const int lane = threadIdx.x & 63;
  const int64_t nw = ((int64_t)gridDim.x * blockDim.x) >> 6;
  unsigned long long lo = 0, up = 0;
  double mx = 0.0;
  for (int64_t r = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6; r < n; r += nw)
    for (int64_t q = rowptr[r] + lane; q < rowptr[r + 1]; q += 64) {
      const int64_t c = col[q];
      const double v = val[q];
      mx = fmax(mx, fabs(v));
      if (c <= r) {
        lb[(r - c) + ldl * c] += v;       // (+=: duplicate entries of a row add up, as in MatSetValues ADD)
        lo += c < r;
      } else
        up++;
    }
  // (one atomic per wave and counter)
  for (int o = 32; o > 0; o >>= 1) {
    lo += __shfl_down(lo, o);
    up += __shfl_down(up, o);
    mx = fmax(mx, __shfl_down(mx, o));
  }
  if (lane == 0) {
    if (lo) atomicAdd(&st->lower, lo);
    if (up) atomicAdd(&st->upper, up);
    atomicMax(&st->maxabs, (unsigned long long)__double_as_longlong(mx));
  }
}

// every entry above the diagonal against the entry below it that the scatter kernel has placed
__global__ void __launch_bounds__(256)
    k_chol_symcheck(const int64_t *__restrict__ rowptr, const int32_t *__restrict__ col, const double *__restrict__ val, int64_t n,
                    int kl, int64_t ldl, const double *__restrict__ lb, double tol_rel, ch_stats *st) {
  const int lane = threadIdx.x & 63;
  const int64_t nw = ((int64_t)gridDim.x * blockDim.x) >> 6;
  const double tol = tol_rel * __longlong_as_double((long long)st->maxabs);
  bool bad = false;
  for (int64_t r = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6; r < n; r += nw)
    for (int64_t q = rowptr[r] + lane; q < rowptr[r + 1]; q += 64) {
      const int64_t c = col[q];
      if (c > r) bad |= (c - r > kl) || !(fabs(val[q] - lb[(c - r) + ldl * r]) <= tol);
    }
  if (bad) atomicExch(&st->asym, 1);
}

__device__ __forceinline__ double ch_readlane(double v, int l) {
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}

// Cholesky factor of the NB x NB block in D (lower triangle; rows >= nbc: identity) by ONE wave: a lane per row (lanes 32 .. 63
// shadow 0 .. 31), the row in registers; column t of the factor is final after step t and every later entry of the row is
// updated with it -- L(c, t) comes from lane c by readlane (a scalar operand of the multiply-add), and 1 / sqrt(d) from
// v_rsq_f64 + two Newton steps: the 32 steps are a chain of dependent latencies, and the LDS round trip of the column plus the
// full-precision sqrt and division made a step ~600 cycles (8 us per block, a quarter of a block's two launches at cfg4).
// FILL: the factor (lower triangle, zeros above) and the reciprocals of its diagonal into D / dinv for a row solve.
// l11 (if not null): the rows as they stand in the registers (entries above the diagonal: never used), dinv_g: 1 / L(j, j).
template <bool FILL>
__device__ __forceinline__ void ch_potrf_wave(double (*D)[CH_NB + 1], double *dinv, int lane, int nbc, int64_t j0, double *l11,
                                              double *dinv_g, ch_stats *st) {
  const int rr = lane & (CH_NB - 1);
  double row[CH_NB];
#pragma unroll
  for (int c = 0; c < CH_NB; c++) row[c] = D[rr][c];
  int bad = 0;
  double myinv = 1.0;
#pragma unroll
  for (int t = 0; t < CH_NB; t++) {
    const double dtt = ch_readlane(row[t], t);
    if (!(dtt > 0.0) && t < nbc && bad == 0) bad = t + 1;
    double inv = __builtin_amdgcn_rsq(dtt);
#pragma unroll
    for (int it = 0; it < 2; it++) inv = fma(0.5 * inv, fma(-dtt * inv, inv, 1.0), inv);
    const double lrt = rr == t ? dtt * inv : row[t] * inv;
    row[t] = lrt;
    myinv = rr == t ? inv : myinv;
#pragma unroll
    for (int c = t + 1; c < CH_NB; c++) row[c] = fma(-lrt, ch_readlane(lrt, c), row[c]);     // (above the diagonal: never used)
  }
  if (FILL && lane < CH_NB) {
#pragma unroll
    for (int c = 0; c < CH_NB; c++) D[rr][c] = c <= rr ? row[c] : 0.0;
    dinv[rr] = myinv;
  }
  if (lane < CH_NB && l11) {
#pragma unroll
    for (int c = 0; c < CH_NB; c++) l11[rr * CH_NB + c] = row[c];
    if (rr < nbc) dinv_g[j0 + rr] = myinv;
  }
  if (bad && lane == 0 && l11) atomicCAS(&st->notpd, 0, (int)(j0 + bad));
}

// One block of NB columns: every workgroup factorises the NB x NB diagonal block (wave 0, a lane per row, the row in
// registers, a finished column goes round through LDS; workgroup 0 hands the factor on), then its 256 threads take a row of
// the panel below each: x L11^T = a, column by column (x_u final -> every later entry of the row updated: independent
// multiply-adds, L11 broadcast from LDS).  Rows of the bottom triangle (r > j0 + kl) start at column r - kl.
// (The factor of the diagonal block goes to `l11` first, not into the band: the other workgroups read the block while
//  workgroup 0 would overwrite it.  The update kernel, which does not touch the block, puts it in place.  dinv: 1 / L(j, j),
//  what the substitutions multiply with.)
__global__ void __launch_bounds__(256)
    k_chol_panel(double *__restrict__ lb, int64_t ldl, int64_t n, int kl, int64_t j0, double *__restrict__ l11,
                 double *__restrict__ dinv_g, ch_stats *st, const double *__restrict__ l11_in) {
  __shared__ double D[2 * CH_NB][CH_NB + 1];  // the diagonal block, then its factor (lower triangle); rows 32 .. 63: spare
                                             // (the shadow lanes of wave 0 store there: no branch in the loop)
  __shared__ double dinv[CH_NB];
  const int tid = threadIdx.x, lane = tid & 63;
  const int nbc = (int)min((int64_t)CH_NB, n - j0);
  // the thread's row of the panel is requested before the diagonal block is factorised
  const int64_t r = j0 + nbc + (int64_t)blockIdx.x * 256 + tid;
  const int64_t rmax = min(n - 1, j0 + nbc - 1 + kl);
  const bool mine = r <= rmax;
  const int t0 = (int)max((int64_t)0, r - kl - j0);          // first column of the block this row holds
  double a[CH_NB];
  {
    // (one clamped address per entry and a select: no branch around a load)
    const int64_t rc = mine ? r : rmax;
#pragma unroll
    for (int t = 0; t < CH_NB; t++) {
      const bool in = mine && t >= t0 && t < nbc;
      const int tc = in ? t : (int)min((int64_t)nbc - 1, max((int64_t)t0, (int64_t)0));
      const double v = lb[(rc - j0 - tc) + ldl * (j0 + tc)];
      a[t] = in ? v : 0.0;
    }
  }
  if (l11_in) {
    // the factor of this block came out of the previous update kernel (its first tile holds the block: see k_chol_syrk)
    for (int i = tid; i < CH_NB * CH_NB; i += 256) {
      const int rr = i / CH_NB, c = i % CH_NB;
      D[rr][c] = c <= rr ? l11_in[i] : 0.0;
    }
    if (tid < CH_NB) dinv[tid] = tid < nbc ? dinv_g[j0 + tid] : 1.0;
  } else {
    for (int i = tid; i < CH_NB * CH_NB; i += 256) {
      const int rr = i / CH_NB, c = i % CH_NB;
      const bool in = rr < nbc && c <= rr && rr - c <= kl;
      D[rr][c] = in ? lb[(rr - c) + ldl * (j0 + c)] : ((c == rr) ? 1.0 : 0.0);
    }
    __syncthreads();
    if (tid < 64) ch_potrf_wave<true>(D, dinv, lane, nbc, j0, blockIdx.x == 0 ? l11 : nullptr, dinv_g, st);
  }
  __syncthreads();
  if (!mine) return;
  // (column by column -- the scheduler otherwise requests all 496 entries of L11 at once -- with the next column requested
  //  before the multiply-adds of this one: two register copies of a column that change roles)
  double dcol[2][CH_NB], dv[2];
#pragma unroll
  for (int t = 1; t < CH_NB; t++) dcol[0][t] = D[t][0];
  dv[0] = dinv[0];
#pragma unroll
  for (int u = 0; u < CH_NB; u++) {
    if (u + 1 < CH_NB) {
#pragma unroll
      for (int t = u + 2; t < CH_NB; t++) dcol[(u + 1) & 1][t] = D[t][u + 1];
      dv[(u + 1) & 1] = dinv[u + 1];
    }
    const double xu = a[u] * dv[u & 1];
    a[u] = xu;
#pragma unroll
    for (int t = u + 1; t < CH_NB; t++) a[t] = fma(-xu, dcol[u & 1][t], a[t]);
    __builtin_amdgcn_sched_barrier(0);
  }
  // (the stores go through a buffer descriptor per column, a lane without an entry there carries an offset beyond its range,
  //  which the hardware drops: 32 divergent `if`s after the solve split the block it is scheduled in -- 2.4 KB of scratch)
#pragma unroll
  for (int t = 0; t < CH_NB; t++) {
    const __amdgpu_buffer_rsrc_t col = __builtin_amdgcn_make_buffer_rsrc(lb + ldl * (j0 + min(t, nbc - 1)), 0,
                                                                         (unsigned)(kl + 1) * 8u, 0x00020000);
    const unsigned off = (t >= t0 && t < nbc) ? (unsigned)(r - j0 - t) * 8u : 0xffffffffu;
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(ch_u2, a[t]), col, off, 0, 0);
  }
}

// trailing update A(r, c) -= sum_{t < nk} L(r, j0 + t) L(c, j0 + t) for jc <= c <= r <= rmax, c < jc + cw, in tiles of 64 x 64
// (anchored at jc) on the matrix cores; the tile is computed transposed (D[c][r]: the lanes of a result register run along r,
// contiguous in the band).  nk = 32: one panel; nk = 64 / 128: two / four panels per pass over the trailing triangle (wide
// bands, where the pass is bound by the triangle it reads and writes: 4 flop per byte with one panel) -- then the columns of
// the later panels of the group have had their share of the earlier ones from launches with col0 = 1, cw = 32 (the tiles of
// the first tile column only).
// The last workgroup puts the factor of the diagonal block at column jl (nbl columns) in place.
template <bool AHEAD>
__global__ void __launch_bounds__(256)
    k_chol_syrk(double *__restrict__ lb, int64_t ldl, int64_t n, int kl, int64_t j0, int nk, int64_t jc, int cw, int ntile, int col0,
                const double *__restrict__ l11, int64_t jl, int nbl, double *__restrict__ l11_next, double *__restrict__ dinv_g,
                ch_stats *st) {
  __shared__ double sm[2][CH_NB][64 + 1];
  double(*Lr)[64 + 1] = sm[0], (*Lc)[64 + 1] = sm[1];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 15, lk = lane >> 4;
  if ((int)blockIdx.x == (col0 ? ntile : ntile * (ntile + 1) / 2)) {
    for (int i = tid; i < CH_NB * CH_NB; i += 256) {
      const int r = i / CH_NB, c = i % CH_NB;
      if (r < nbl && c <= r && r - c <= kl) lb[(r - c) + ldl * (jl + c)] = l11[i];
    }
    return;
  }
  // linear index -> (ti >= tj)
  int ti = 0, rest = blockIdx.x;
  if (col0) {
    ti = rest;
    rest = 0;
  } else
    while (rest > ti) {
      rest -= ti + 1;
      ti++;
    }
  const int tj = rest;
  const int64_t rmax = min(n - 1, j0 + nk - 1 + kl), cmax = jc + cw - 1;
  const int64_t r0 = jc + 64 * (int64_t)ti, c0 = jc + 64 * (int64_t)tj;
  // the tile of A is requested first: the read-modify-write after the products waited a full memory latency with nothing else
  // in flight.  (The products are summed on their own and subtracted once: accumulating on A itself rounds every one of the
  // 32 / 64 terms at the magnitude of A -- residuals of the 3-D solves 6-12 x larger.)
  ch_v4d acc[4], apre[4];
#pragma unroll
  for (int q = 0; q < 4; q++) {
    acc[q] = (ch_v4d){0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const int64_t c = c0 + 16 * wave + lk + 4 * i, r = r0 + 16 * q + lr;
      const bool ok = r >= c && r <= rmax && c <= cmax;
      const double v = lb[ok ? (r - c) + ldl * c : 0];
      apre[q][i] = ok ? v : 0.0;
    }
  }
  for (int kh = 0; kh < nk; kh += CH_NB) {
    if (kh) __syncthreads();
    const int nkc = min(CH_NB, nk - kh);
    const int64_t jk = j0 + kh;
    for (int i = tid; i < CH_NB * 64; i += 256) {
      const int k = i >> 6, l = i & 63;
      const int64_t rr = r0 + l, rc = c0 + l;
      Lr[k][l] = (k < nkc && rr <= rmax && rr - jk - k <= kl) ? lb[(rr - jk - k) + ldl * (jk + k)] : 0.0;
      Lc[k][l] = (k < nkc && rc <= rmax && rc - jk - k <= kl) ? lb[(rc - jk - k) + ldl * (jk + k)] : 0.0;
    }
    __syncthreads();
#pragma unroll
    for (int k4 = 0; k4 < CH_NB / 4; k4++) {
      const double a = Lc[4 * k4 + lk][16 * wave + lr];        // A operand: rows = matrix columns c of this wave
#pragma unroll
      for (int q = 0; q < 4; q++) acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, Lr[4 * k4 + lk][16 * q + lr], acc[q], 0, 0, 0);
    }
  }
#pragma unroll
  for (int q = 0; q < 4; q++)
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const int64_t c = c0 + 16 * wave + lk + 4 * i, r = r0 + 16 * q + lr;
      if (r >= c && r <= rmax && c <= cmax) lb[(r - c) + ldl * c] = apre[q][i] - acc[q][i];
    }
  // Look-ahead (narrow bands: the two launches per block are a chain of latencies): the first tile holds the diagonal block of
  // the NEXT panel, final with this update -- its factor is computed here, by one wave of one workgroup while the others still
  // update, and the next panel kernel starts with its row solve (6 of its 20 us at cfg4 were the 32 dependent steps of the
  // factorisation).  Needs the whole block inside the rows this update reaches (the host checks kl >= 64).
  if (AHEAD && l11_next && blockIdx.x == 0) {
    double(*D)[CH_NB + 1] = (double(*)[CH_NB + 1]) & sm[0][0][0];          // [2 NB][NB + 1] + dinv[NB]: inside sm
    double *dinv = &sm[0][0][0] + 2 * CH_NB * (CH_NB + 1);
    static_assert(2 * CH_NB * (CH_NB + 1) + CH_NB <= 2 * CH_NB * (64 + 1), "the block and its reciprocals fit into the operand tiles");
    const int nbn = (int)min((int64_t)CH_NB, n - jc);
    __syncthreads();                       // (every wave is done with the operand tiles)
    if (wave < 2) {
#pragma unroll
      for (int q = 0; q < 2; q++)
#pragma unroll
        for (int i = 0; i < 4; i++) {
          const int cc = 16 * wave + lk + 4 * i, rr = 16 * q + lr;
          D[rr][cc] = (rr < nbn && cc <= rr) ? apre[q][i] - acc[q][i] : (rr == cc ? 1.0 : 0.0);
        }
    }
    __syncthreads();
    if (tid < 64) ch_potrf_wave<false>(D, dinv, lane, nbn, jc, l11_next, dinv_g, st);
  }
}

// One launch per block of 32 columns (narrow bands, where a block's two launches are a chain of latencies): every workgroup
// of the trailing update solves the 2 x 64 rows of the panel its tile needs ITSELF -- x L11^T = a with the factor of the
// diagonal block that the previous launch's first tile computed; a row costs 496 multiply-adds, a tile 131 k -- instead of
// waiting for a panel kernel.  The solved rows cannot go into the band at once (other workgroups still read the unsolved ones):
// the workgroups of tile column 0 put them into a side buffer (`pan`, column-major like the band), and CH_NCOPY extra
// workgroups of the NEXT launch (or k_chol_pancopy) copy them into place.  The rest is k_chol_syrk<true>.
#define CH_NCOPY 8
__device__ __forceinline__ void ch_pancopy(double *__restrict__ lb, int64_t ldl, int64_t n, int kl, int64_t jp, const double *__restrict__ pan,
                                           int64_t ph, int w, int tid) {
  const int64_t jp1 = jp + CH_NB, rows = min(n - 1, jp + CH_NB - 1 + kl) - jp1 + 1;
  for (int64_t idx = (int64_t)w * 256 + tid; idx < rows * CH_NB; idx += CH_NCOPY * 256) {
    const int t = (int)(idx / rows);
    const int64_t i = idx - t * rows, r = jp1 + i;           // (i fastest: contiguous in the band and in pan)
    if (r - (jp + t) <= kl) lb[(r - jp - t) + ldl * (jp + t)] = pan[t * ph + i];
  }
}
__global__ void __launch_bounds__(256)
    k_chol_pancopy(double *__restrict__ lb, int64_t ldl, int64_t n, int kl, int64_t jp, const double *__restrict__ pan, int64_t ph) {
  ch_pancopy(lb, ldl, n, kl, jp, pan, ph, blockIdx.x, threadIdx.x);
}
__global__ void __launch_bounds__(256)
    k_chol_fused(double *__restrict__ lb, int64_t ldl, int64_t n, int kl, int64_t j0, int ntile, const double *__restrict__ l11,
                 double *__restrict__ l11_next, double *__restrict__ dinv_g, ch_stats *st, double *__restrict__ pan, int64_t ph,
                 const double *__restrict__ pan_prev, int64_t j0_prev) {
  __shared__ double sm[2][CH_NB][64 + 1];
  __shared__ double Dl[CH_NB][CH_NB + 1];
  __shared__ double dls[CH_NB];
  double(*Lr)[64 + 1] = sm[0], (*Lc)[64 + 1] = sm[1];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 15, lk = lane >> 4;
  const int ntri = ntile * (ntile + 1) / 2;
  if ((int)blockIdx.x == ntri) {                             // the factor of this block's diagonal block into the band
    for (int i = tid; i < CH_NB * CH_NB; i += 256) {
      const int r = i / CH_NB, c = i % CH_NB;
      if (c <= r && r - c <= kl) lb[(r - c) + ldl * (j0 + c)] = l11[i];
    }
    return;
  }
  if ((int)blockIdx.x > ntri) {                              // the solved panel of the previous block into the band
    if (pan_prev) ch_pancopy(lb, ldl, n, kl, j0_prev, pan_prev, ph, (int)blockIdx.x - ntri - 1, tid);
    return;
  }
  int ti = 0, rest = blockIdx.x;
  while (rest > ti) {
    rest -= ti + 1;
    ti++;
  }
  const int tj = rest;
  const int64_t j1 = j0 + CH_NB, rmax = min(n - 1, j0 + CH_NB - 1 + kl);
  const int64_t r0 = j1 + 64 * (int64_t)ti, c0 = j1 + 64 * (int64_t)tj;
  ch_v4d acc[4], apre[4];
#pragma unroll
  for (int q = 0; q < 4; q++) {
    acc[q] = (ch_v4d){0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const int64_t c = c0 + 16 * wave + lk + 4 * i, r = r0 + 16 * q + lr;
      const bool ok = r >= c && r <= rmax;
      const double v = lb[ok ? (r - c) + ldl * c : 0];
      apre[q][i] = ok ? v : 0.0;
    }
  }
  // waves 0 / 1: a thread per row of the tile's row set / column set
  const int64_t myr = (tid < 64 ? r0 : c0) + lane;
  const bool mine = tid < 128 && myr <= rmax;
  const int t0 = (int)max((int64_t)0, myr - kl - j0);        // first column of the block the row holds
  double a[CH_NB];
  if (tid < 128) {
#pragma unroll
    for (int t = 0; t < CH_NB; t++) {                        // (a safe address and a select: no branch around a load)
      const bool in = mine && t >= t0;
      const double v = lb[in ? (myr - j0 - t) + ldl * (j0 + t) : 0];
      a[t] = in ? v : 0.0;
    }
  }
  for (int i = tid; i < CH_NB * CH_NB; i += 256) {
    const int rr = i / CH_NB, c = i % CH_NB;
    Dl[rr][c] = c <= rr ? l11[i] : 0.0;
  }
  if (tid < CH_NB) dls[tid] = dinv_g[j0 + tid];
  __syncthreads();
  if (tid < 128) {
    double dcol[2][CH_NB], dv[2];
#pragma unroll
    for (int t = 1; t < CH_NB; t++) dcol[0][t] = Dl[t][0];
    dv[0] = dls[0];
#pragma unroll
    for (int u = 0; u < CH_NB; u++) {
      if (u + 1 < CH_NB) {
#pragma unroll
        for (int t = u + 2; t < CH_NB; t++) dcol[(u + 1) & 1][t] = Dl[t][u + 1];
        dv[(u + 1) & 1] = dls[u + 1];
      }
      const double xu = a[u] * dv[u & 1];
      a[u] = xu;
#pragma unroll
      for (int t = u + 1; t < CH_NB; t++) a[t] = fma(-xu, dcol[u & 1][t], a[t]);
      __builtin_amdgcn_sched_barrier(0);
    }
    double *dst = tid < 64 ? &Lr[0][0] : &Lc[0][0];
#pragma unroll
    for (int t = 0; t < CH_NB; t++) dst[t * (64 + 1) + lane] = a[t];
    // (the side buffer through a descriptor: a lane that has nothing to put there carries an offset beyond its range)
    const __amdgpu_buffer_rsrc_t pb = __builtin_amdgcn_make_buffer_rsrc(pan, 0, (unsigned)(ph * CH_NB) * 8u, 0x00020000);
    const bool put = mine && tid < 64 && tj == 0;
#pragma unroll
    for (int t = 0; t < CH_NB; t++) {
      const unsigned off = put ? (unsigned)(t * ph + (myr - j1)) * 8u : 0xffffffffu;
      __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(ch_u2, a[t]), pb, off, 0, 0);
    }
  }
  __syncthreads();
#pragma unroll
  for (int k4 = 0; k4 < CH_NB / 4; k4++) {
    const double av = Lc[4 * k4 + lk][16 * wave + lr];
#pragma unroll
    for (int q = 0; q < 4; q++) acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, Lr[4 * k4 + lk][16 * q + lr], acc[q], 0, 0, 0);
  }
#pragma unroll
  for (int q = 0; q < 4; q++)
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const int64_t c = c0 + 16 * wave + lk + 4 * i, r = r0 + 16 * q + lr;
      if (r >= c && r <= rmax) lb[(r - c) + ldl * c] = apre[q][i] - acc[q][i];
    }
  if (blockIdx.x == 0) {                                     // the factor of the next diagonal block (see k_chol_syrk)
    double(*D)[CH_NB + 1] = (double(*)[CH_NB + 1]) & sm[0][0][0];
    double *dinv = &sm[0][0][0] + 2 * CH_NB * (CH_NB + 1);
    const int nbn = (int)min((int64_t)CH_NB, n - j1);
    __syncthreads();
    if (wave < 2) {
#pragma unroll
      for (int q = 0; q < 2; q++)
#pragma unroll
        for (int i = 0; i < 4; i++) {
          const int cc = 16 * wave + lk + 4 * i, rr = 16 * q + lr;
          D[rr][cc] = (rr < nbn && cc <= rr) ? apre[q][i] - acc[q][i] : (rr == cc ? 1.0 : 0.0);
        }
    }
    __syncthreads();
    if (tid < 64) ch_potrf_wave<false>(D, dinv, lane, nbn, j1, l11_next, dinv_g, st);
  }
}

// L y = b block by block (one workgroup, NT threads): the window of x that the blocks ahead still change lives in an LDS ring
// (entry r at r mod W).  Per block: the entries of the thread's row of the panel are requested first, wave 0 solves the
// NB x NB triangle (a lane per unknown, broadcasts by readlane, reciprocals of the diagonal from the factorisation),
// everyone subtracts its row's share.  Nothing in a step waits for a load that was not issued a phase earlier.
#define CH_NT 1024
static_assert(CH_NT == CH_NB * CH_NB, "the substitution kernels load one entry of the diagonal block per thread");
__global__ void __launch_bounds__(CH_NT) k_chol_fwd(const double *__restrict__ lb, const double *__restrict__ dinv_g, int64_t ldl,
                                                     int64_t n, int kl, int W, double *__restrict__ x) {
  extern __shared__ double ch_sm[];
  double *xw = ch_sm;                    // [W]
  double *yb = ch_sm + W;                // [NB]
  double *Ls = yb + CH_NB;               // [NB][NB + 1]: the diagonal block of the factor
  const int tid = threadIdx.x, lane = tid & 63;
  const int RT = (kl + CH_NT - 1) / CH_NT;       // rows of the panel per thread
  for (int64_t r = tid; r < min(n, (int64_t)CH_NB + kl); r += CH_NT) xw[r & (W - 1)] = x[r];
  __syncthreads();
  for (int64_t j0 = 0; j0 < n; j0 += CH_NB) {
    const int nbc = (int)min((int64_t)CH_NB, n - j0);
    const int64_t rmax = min(n - 1, j0 + nbc - 1 + kl);
    // the first panel row of every thread, requested before the triangle is solved
    double pv[CH_NB];
    {
      const int64_t r = j0 + nbc + tid;
      const int t0 = (int)max((int64_t)0, r - kl - j0);
      const int64_t rc = min(r, rmax);
#pragma unroll
      for (int t = 0; t < CH_NB; t++) {      // (a clamped address and a select: no branch around a load)
        const bool in = r <= rmax && t >= t0 && t < nbc;
        const int tc = in ? t : nbc - 1;
        const double v = lb[(rc - j0 - tc) + ldl * (j0 + tc)];
        pv[t] = in ? v : 0.0;
      }
    }
    {
      const int rr = tid >> 5, c = tid & (CH_NB - 1);          // (CH_NT = NB * NB: an entry of the block per thread)
      Ls[rr * (CH_NB + 1) + c] = (rr < nbc && c < rr && rr - c <= kl) ? lb[(rr - c) + ldl * (j0 + c)] : 0.0;
    }
    __syncthreads();
    if (tid < 64) {
      const int rr = lane & (CH_NB - 1);
      const double di = rr < nbc ? dinv_g[j0 + rr] : 1.0;
      double v = rr < nbc ? xw[(j0 + rr) & (W - 1)] : 0.0;
#pragma unroll
      for (int s8 = 0; s8 < CH_NB; s8 += 8) {
        double l[8];                         // entries s8 .. s8 + 7 of row rr of L11 (0 for the lanes above)
#pragma unroll
        for (int i = 0; i < 8; i++) l[i] = Ls[rr * (CH_NB + 1) + s8 + i];
#pragma unroll
        for (int i = 0; i < 8; i++) {
          const double ys = ch_readlane(v * di, s8 + i);
          v = rr == s8 + i ? ys : fma(-l[i], ys, v);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      if (lane < CH_NB) yb[lane] = lane < nbc ? v : 0.0;       // (all of yb: the panel products read every entry)
      if (lane < nbc) x[j0 + lane] = v;
    }
    __syncthreads();
    {
      const int64_t r = j0 + nbc + tid;
      if (r <= rmax) {
        double s = 0.0;
#pragma unroll
        for (int t = 0; t < CH_NB; t++) s = fma(pv[t], yb[t], s);
        xw[r & (W - 1)] -= s;
      }
    }
    for (int q = 1; q < RT; q++) {                   // (bands wider than the workgroup)
      const int64_t r = j0 + nbc + tid + (int64_t)q * CH_NT;
      const int t0 = (int)max((int64_t)0, r - kl - j0);
      const int64_t rc = min(r, rmax);
#pragma unroll
      for (int t = 0; t < CH_NB; t++) {
        const bool in = r <= rmax && t >= t0 && t < nbc;
        const int tc = in ? t : nbc - 1;
        const double v = lb[(rc - j0 - tc) + ldl * (j0 + tc)];
        pv[t] = in ? v : 0.0;
      }
      double s = 0.0;
#pragma unroll
      for (int t = 0; t < CH_NB; t++) s = fma(pv[t], yb[t], s);
      if (r <= rmax) xw[r & (W - 1)] -= s;
    }
    // the rows that enter the window with the next block
    {
      const int64_t r = j0 + CH_NB + kl + tid;
      if (tid < CH_NB && r < n) xw[r & (W - 1)] = x[r];
    }
    __syncthreads();
  }
}

// L^T x = y from the last block to the first: column c of L below the block is contiguous, a wave takes two columns of the
// block, multiplies them with the window of x (eight independent loads at a time) and reduces; wave 0 solves the transposed
// triangle.
__global__ void __launch_bounds__(CH_NT) k_chol_bwd(const double *__restrict__ lb, const double *__restrict__ dinv_g, int64_t ldl,
                                                     int64_t n, int kl, int W, double *__restrict__ x) {
  extern __shared__ double ch_sm[];
  double *xw = ch_sm;                    // [W]: final x of the rows behind the current block
  double *sb = ch_sm + W;                // [NB]
  double *Ls = sb + CH_NB;               // [NB][NB + 1]: Ls[c][u] = L11[u][c], the diagonal block transposed
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t nblk = (n + CH_NB - 1) / CH_NB;
  for (int64_t b = nblk - 1; b >= 0; b--) {
    const int64_t j0 = b * CH_NB;
    const int nbc = (int)min((int64_t)CH_NB, n - j0);
    {
      const int u = tid >> 5, c = tid & (CH_NB - 1);
      Ls[c * (CH_NB + 1) + u] = (u < nbc && c < u && u - c <= kl) ? lb[(u - c) + ldl * (j0 + c)] : 0.0;
    }
    // s_t = sum over the rows r below the block (r <= j0 + t + kl) of L(r, j0 + t) x_r
    for (int t = wave; t < nbc; t += CH_NT / 64) {
      const int64_t c = j0 + t, ra = j0 + nbc, rb = min(n - 1, c + kl);
      const double *lc = lb + ldl * c - c;             // lc[r] = L(r, c)
      double s = 0.0;
      for (int64_t r8 = ra + lane; r8 <= rb; r8 += 8 * 64) {
        double lv[8];
#pragma unroll
        for (int i = 0; i < 8; i++) lv[i] = r8 + 64 * i <= rb ? lc[r8 + 64 * i] : 0.0;
#pragma unroll
        for (int i = 0; i < 8; i++) {          // (beyond the column's last row the ring holds whatever the LDS held: no 0 * NaN)
          const double xv = xw[(r8 + 64 * i) & (W - 1)];
          s = fma(lv[i], r8 + 64 * i <= rb ? xv : 0.0, s);
        }
      }
      s = tg_wave_sum(s);
      if (lane == 0) sb[t] = s;
    }
    __syncthreads();
    if (tid < 64) {
      const int rr = lane & (CH_NB - 1);
      const double di = rr < nbc ? dinv_g[j0 + rr] : 1.0;
      double v = rr < nbc ? x[j0 + rr] - sb[rr] : 0.0;
#pragma unroll
      for (int s8 = CH_NB - 8; s8 >= 0; s8 -= 8) {
        double l[8];                         // L11[s8 + i][rr] (0 for the lanes below)
#pragma unroll
        for (int i = 0; i < 8; i++) l[i] = Ls[rr * (CH_NB + 1) + s8 + i];
#pragma unroll
        for (int i = 7; i >= 0; i--) {
          const double xs = ch_readlane(v * di, s8 + i);
          v = rr == s8 + i ? xs : fma(-l[i], xs, v);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      if (lane < nbc) {
        x[j0 + lane] = v;
        xw[(j0 + lane) & (W - 1)] = v;
      }
    }
    __syncthreads();
  }
}

// ---- the substitutions on SEVERAL workgroups (round 6, after the single-workgroup sweeps above proved bound by what one CU
// streams: 0.56 GB of factor in 25.8 + 17.8 ms at cfg4 = 22 - 31 GB/s).  Blocks of 32 rows / columns are owned cyclically by
// G workgroups (8 .. 64: a quarter of the blocks a block's tiles reach).  Forward (L y = b), right-looking: the owner of block J solves its 32 x 32 triangle, publishes y_J (global
// memory + a flag per block, release / acquire at agent scope) and every workgroup subtracts the tiles L(b, J) y_J from the
// row blocks b it owns; backward (L^T x = y) the same from the last block to the first, with the tiles transposed and the
// pending sums owned by column block.  A workgroup takes the blocks in order, so whoever owns block J + 1 has applied
// everything up to J before it solves: no dead-lock as long as the G workgroups are resident (<= 64 on 256 CUs).  The waits
// give up after two seconds (err[0] = 1: the caller falls back to the single-workgroup sweeps).
#define CH_SG_MIN 8
#define CH_SG_MAX 64
// (what is published is the data itself: `pub` starts as NaNs and the 32 threads that need y_J poll their own entry -- one
//  round trip through the L2 instead of a flag's and then the data's; the tiles of block J and the diagonal block of the next
//  block this workgroup owns are static data and are requested BEFORE the wait.  Loads return in order within a wave: the wave
//  that polls, solves and publishes -- wave 3 -- requests nothing else but the 32 reciprocals (and right-hand sides) of its next
//  block, a step ahead; the tiles, the diagonal blocks and the rows that enter the window belong to waves 0 .. 2.  With the
//  polls queued behind a wave's 32 tile loads a step took 6.5 us at cfg4.)
#define CH_SW_GRP 6                      // groups of 32 threads that take tiles (waves 0 .. 2)
template <bool FWD>
__global__ void __launch_bounds__(256) k_chol_sweep(const double *__restrict__ lb, const double *__restrict__ dinv_g, int64_t ldl,
                                                     int64_t n, int kl, int W, double *x, double *pub, int *err) {
  extern __shared__ double ch_sm[];
  double *xw = ch_sm;                    // [W]: FWD: running right-hand side of the owned rows; else pending sums of owned columns
  double *yb = ch_sm + W;                // [NB]: the block just published
  double *Ls = yb + CH_NB;               // [NB][NB + 1]
  __shared__ int s_abort;
  const int tid = threadIdx.x, lane = tid & 63, g = blockIdx.x, G = gridDim.x;
  const bool lat = tid >= 32 * CH_SW_GRP;                  // the wave of the hand-over
  if (tid == 0) s_abort = 0;
  const int64_t nblk = (n + CH_NB - 1) / CH_NB;
  const int reach = (kl + 2 * CH_NB - 1) / CH_NB;          // blocks a block's tiles reach
  const long long tmo = 2ll * 100000000ll;                 // (wall_clock64 counts at 100 MHz)
  if (FWD) {
    for (int64_t r = tid; r < min(n, (int64_t)CH_NB + kl); r += 256)
      if ((r / CH_NB) % G == g) xw[r & (W - 1)] = x[r];
  } else {
    for (int i = tid; i < W; i += 256) xw[i] = 0.0;
  }
  // the diagonal block of block J into Ls (waves 0 .. 2): FWD Ls[a][c] = L11[a][c] (c < a), else Ls[a][c] = L11[c][a] (a < c)
  auto load_diag = [&](int64_t J) {
    const int64_t j0 = J * CH_NB;
    const int nbc = (int)min((int64_t)CH_NB, n - j0);
#pragma unroll
    for (int k = 0; k < (CH_NB * CH_NB + 32 * CH_SW_GRP - 1) / (32 * CH_SW_GRP); k++) {
      const int i = min(tid + 32 * CH_SW_GRP * k, CH_NB * CH_NB - 1), a = i >> 5, c = i & (CH_NB - 1);     // (the tail: twice)
      const int rr = FWD ? a : c, cc = FWD ? c : a;
      const bool in = rr < nbc && cc < rr && rr - cc <= kl;
      const double v = lb[(in ? rr - cc : 0) + ldl * (j0 + (in ? cc : 0))];
      Ls[a * (CH_NB + 1) + c] = in ? v : 0.0;
    }
  };
  // what the hand-over wave needs of a block it owns: the reciprocals of the diagonal, and (backward) the right-hand side
  double dcur = 1.0, xcur = 0.0;
  auto small = [&](int64_t J, double &d, double &xv) {
    const int64_t j0 = J * CH_NB;
    const int nbc = (int)min((int64_t)CH_NB, n - j0);
    const int rr = lane & (CH_NB - 1);
    const double dl = dinv_g[j0 + min(rr, nbc - 1)];
    d = rr < nbc ? dl : 1.0;
    if (!FWD) {
      const double xl = x[j0 + min(rr, nbc - 1)];
      xv = rr < nbc ? xl : 0.0;
    }
  };
  {
    const int64_t J0 = FWD ? 0 : nblk - 1;
    if (J0 % G == g) {
      if (lat) small(J0, dcur, xcur);
      else load_diag(J0);
    }
  }
  // the tiles of a step: of the blocks this workgroup owns among those block J reaches -- FWD rows of the blocks b in
  // (J, J + reach], else columns of the blocks b in [J - reach, J) -- a group of 32 threads per block; the first pass is
  // requested at the top of the step.  (Requested a step ahead into a second register array -- so that the update would wait for
  // nothing but y_J -- the kernel came out at 256 registers + 7 spilled and the sweeps took 26.9 + 16.4 ms instead of 11.4 + 10.4.)
  struct geom_t {
    int64_t J, j0, b_hi, bb1;
    int nbc;
  };
  auto geom = [&](int64_t step) {
    geom_t e;
    e.J = FWD ? step : nblk - 1 - step;
    e.j0 = e.J * CH_NB;
    e.nbc = (int)min((int64_t)CH_NB, n - e.j0);
    const int64_t b_lo = FWD ? e.J + 1 : max((int64_t)0, e.J - reach);
    e.b_hi = FWD ? min(nblk - 1, e.J + reach) : e.J - 1;
    const int64_t b0 = b_lo + ((g - b_lo % G) % G + G) % G;
    e.bb1 = b0 + (int64_t)(tid >> 5) * G;
    return e;
  };
  auto tile = [&](const geom_t &e, int64_t q, bool on, double *pv) {
#pragma unroll
    for (int t = 0; t < CH_NB; t++) {
      if (FWD) {
        const int64_t c = e.j0 + t;
        const bool in = on && t < e.nbc && q - c <= kl;
        const double v = lb[(in ? q - c : 0) + ldl * (in ? c : e.j0)];
        pv[t] = in ? v : 0.0;
      } else {
        const int64_t rr = e.j0 + t;
        const bool in = on && t < e.nbc && rr - q <= kl;
        const double v = lb[(in ? rr - q : 0) + ldl * (in ? q : e.j0)];
        pv[t] = in ? v : 0.0;
      }
    }
  };
  auto first_tiles = [&](int64_t step, double *pv) {
    if (step >= nblk) return;
    const geom_t e = geom(step);
    const int64_t q1 = e.bb1 * CH_NB + (tid & (CH_NB - 1));        // FWD: the row; else the column
    tile(e, q1, e.bb1 <= e.b_hi && q1 < n, pv);
  };
  // one step; false: a wait gave up
  auto body = [&](int64_t step, double *pv) -> bool {
    const geom_t e = geom(step);
    const int64_t J = e.J, j0 = e.j0;
    const int nbc = e.nbc;
    const bool owner = J % G == g;
    const int64_t Jn = FWD ? J + 1 : J - 1;
    const bool own_next = Jn >= 0 && Jn < nblk && Jn % G == g;
    double xin = 0.0;
    const int64_t rin = j0 + CH_NB + kl + tid;          // FWD: an owned row that the next block reaches for the first time
    const bool in_on = FWD && tid < CH_NB && rin < n && (rin / CH_NB) % G == g;
    if (!lat) {
      first_tiles(step, pv);
      if (FWD && tid < CH_NB) {
        const double v = x[in_on ? rin : 0];
        xin = in_on ? v : 0.0;
      }
      if (own_next) load_diag(Jn);                      // (this workgroup does not own J then: Ls is free)
    } else if (owner) {
      const int rr = lane & (CH_NB - 1);
      double v = 0.0;
      if (rr < nbc) v = FWD ? xw[(j0 + rr) & (W - 1)] : xcur - xw[(j0 + rr) & (W - 1)];
#pragma unroll
      for (int s8 = 0; s8 < CH_NB; s8 += 8) {
        double l[8];
#pragma unroll
        for (int i = 0; i < 8; i++) l[i] = Ls[rr * (CH_NB + 1) + (FWD ? s8 + i : CH_NB - 1 - s8 - i)];
#pragma unroll
        for (int i = 0; i < 8; i++) {
          const int sidx = FWD ? s8 + i : CH_NB - 1 - s8 - i;
          const double ys = ch_readlane(v * dcur, sidx);
          v = rr == sidx ? ys : fma(-l[i], ys, v);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      if (lane < CH_NB) {
        const double out = lane < nbc ? v : 0.0;
        yb[lane] = out;
        __hip_atomic_store(&pub[J * CH_NB + lane], out, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (lane < nbc) {
          x[j0 + lane] = v;
          if (!FWD) xw[(j0 + lane) & (W - 1)] = 0.0;    // (the slot serves a later column)
        }
      }
    } else if (lane < CH_NB) {
      const long long t0 = wall_clock64();
      double v = __hip_atomic_load(&pub[J * CH_NB + lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      int spins = 0;
      while (v != v) {
        if ((++spins & 63) == 0 && (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0 || wall_clock64() - t0 > tmo)) {
          __hip_atomic_store(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          s_abort = 1;
          v = 0.0;
          break;
        }
        v = __hip_atomic_load(&pub[J * CH_NB + lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      yb[lane] = v;
    }
    __syncthreads();
    if (s_abort) return false;           // (the same decision in every thread)
    if (lat) {
      if (own_next) small(Jn, dcur, xcur);              // (in flight while the others update; the next polls queue behind it)
    } else {
      for (int64_t bb = e.bb1; bb <= e.b_hi; bb += CH_SW_GRP * G) {
        const int64_t q = bb * CH_NB + (tid & (CH_NB - 1));
        if (bb != e.bb1) tile(e, q, q < n, pv);
        if (q < n) {
          double sacc = 0.0;
#pragma unroll
          for (int t = 0; t < CH_NB; t++) sacc = fma(pv[t], yb[t], sacc);
          if (FWD) {
            if (q <= min(n - 1, j0 + nbc - 1 + kl)) xw[q & (W - 1)] -= sacc;
          } else {
            if (j0 - q <= kl) xw[q & (W - 1)] += sacc;     // (columns the block does not reach share their slots with others)
          }
        }
      }
      if (in_on) xw[rin & (W - 1)] = xin;
    }
    __syncthreads();
    return true;
  };
  double pv[CH_NB];
  __syncthreads();
  for (int64_t step = 0; step < nblk; step++)
    if (!body(step, pv)) return;
}

// 0 = solved by Cholesky (*done = 1) or not applicable (*done = 0: the caller goes on with the LU)
int tg_chol_try(tg_csr_s *k, int kl, int ku, const double *b, double *x, int *done) {
  *done = 0;
  if (getenv("TIGAR_LU_CHOLESKY") && atoi(getenv("TIGAR_LU_CHOLESKY")) == 0) return 0;
  const int64_t n = k->nrows;
  if (kl != ku || kl < 8 || n < 2 * CH_NB) return 0;
  int W = 64;
  while (W < kl + 2 * CH_NB) W <<= 1;
  const size_t lds = (size_t)(W + CH_NB + CH_NB * (CH_NB + 1)) * sizeof(double);
  if (lds > 150 * 1024) return 0;
  const bool trace = getenv("TIGAR_TRACE") != nullptr;
  const int64_t ldl = (int64_t)kl + 1;
  double *lb = nullptr, *l11 = nullptr, *l11b = nullptr, *dinv = nullptr;
  ch_stats *st = nullptr;
  if (tg_dmalloc(&lb, ldl * n)) {
    (void)hipGetLastError();
    return 0;                        // (no room: the LU reports it)
  }
  int rc = tg_dmalloc_bytes((void **)&st, sizeof(ch_stats)) || tg_dmalloc(&l11, CH_NB * CH_NB) || tg_dmalloc(&l11b, CH_NB * CH_NB) ||
           tg_dmalloc(&dinv, n);
  ch_stats h;
  memset(&h, 0, sizeof(h));
  if (!rc && (hipMemsetAsync(lb, 0, (size_t)(ldl * n) * sizeof(double), g_tg.stream) != hipSuccess ||
              hipMemcpyAsync(st, &h, sizeof(h), hipMemcpyHostToDevice, g_tg.stream) != hipSuccess))
    rc = 1;
  if (!rc) {
    const unsigned g = (unsigned)std::min<int64_t>(tg_cdiv(n, 4), (int64_t)g_tg.num_cu * 16);
    hipLaunchKernelGGL(k_chol_scatter, dim3(g), dim3(256), 0, g_tg.stream, k->rowptr, k->col, k->val, n, ldl, lb, st);
    hipLaunchKernelGGL(k_chol_symcheck, dim3(g), dim3(256), 0, g_tg.stream, k->rowptr, k->col, k->val, n, kl, ldl, lb, 1e-11, st);
    if (hipGetLastError() != hipSuccess || hipMemcpyAsync(&h, st, sizeof(h), hipMemcpyDeviceToHost, g_tg.stream) != hipSuccess ||
        hipStreamSynchronize(g_tg.stream) != hipSuccess)
      rc = 1;
  }
  bool go = !rc && !h.asym && h.lower == h.upper;
  if (!rc && !go && trace) fprintf(stderr, "[trace] cholesky: the matrix is not symmetric: LU\n");
  if (go) {
    if (hipFuncSetAttribute((const void *)k_chol_fwd, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess ||
        hipFuncSetAttribute((const void *)k_chol_bwd, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
      (void)hipGetLastError();
      go = false;
    }
  }
  if (go) {
    // wide bands: groups of 2 / 4 panels per pass over the trailing triangle (TIGAR_CHOL_PAIR_KL: the half-width from which on
    // two, twice that: four; TIGAR_CHOL_GROUP=g: always g).  Inside a group the 32 columns of the next block get their share of
    // the panels before them from the tiles of one tile column (col0 = 1, cw = 32).
    const int pair_kl = getenv("TIGAR_CHOL_PAIR_KL") ? atoi(getenv("TIGAR_CHOL_PAIR_KL")) : 2048;
    int ng = kl >= 2 * (int64_t)pair_kl ? 4 : kl >= pair_kl ? 2 : 1;
    if (getenv("TIGAR_CHOL_GROUP") && atoi(getenv("TIGAR_CHOL_GROUP")) > 0) ng = std::min(atoi(getenv("TIGAR_CHOL_GROUP")), 16);
    const int big = 1 << 30;
    // one panel per update: the factor of the next diagonal block comes out of the update kernel (TIGAR_CHOL_LOOKAHEAD=0: not)
    const bool ahead = ng == 1 && kl >= 64 && l11b && !(getenv("TIGAR_CHOL_LOOKAHEAD") && atoi(getenv("TIGAR_CHOL_LOOKAHEAD")) == 0);
    double *lcur = l11, *lnext = l11b;
    bool have = false;
    // ... and with its factor known a block takes ONE launch: the update kernel solves the rows of the panel it needs itself
    // (k_chol_fused; TIGAR_CHOL_FUSED=0: a panel and an update kernel per block)
    const int64_t ph = (int64_t)kl + 2 * CH_NB;
    double *pan[2] = {nullptr, nullptr};
    bool fused = ahead && !(getenv("TIGAR_CHOL_FUSED") && atoi(getenv("TIGAR_CHOL_FUSED")) == 0);
    if (fused && (tg_dmalloc(&pan[0], ph * CH_NB) || tg_dmalloc(&pan[1], ph * CH_NB))) {
      (void)hipGetLastError();
      fused = false;
    }
    int pcur = 0;
    bool pending = false;                  // a solved panel waits in pan[pcur ^ 1] for its copy into the band
    int64_t jpend = 0;
    auto flush = [&]() {
      if (pending)
        hipLaunchKernelGGL(k_chol_pancopy, dim3(CH_NCOPY), dim3(256), 0, g_tg.stream, lb, ldl, n, kl, jpend, (const double *)pan[pcur ^ 1], ph);
      pending = false;
    };
    for (int64_t j0 = 0; j0 < n;) {
      if (fused && have) {
        // (ng == 1; `have`: the previous launch saw rows below its block, so this block is a full one)
        const int64_t j1 = j0 + CH_NB;
        const int64_t m = std::max<int64_t>(0, std::min<int64_t>(n - 1, j0 + CH_NB - 1 + kl) - j1 + 1);
        const int nt = (int)tg_cdiv(m, 64);
        if (nt > 0) {
          hipLaunchKernelGGL(k_chol_fused, dim3((unsigned)(nt * (nt + 1) / 2 + 1 + CH_NCOPY)), dim3(256), 0, g_tg.stream, lb, ldl, n, kl,
                             j0, nt, (const double *)lcur, lnext, dinv, st, pan[pcur], ph,
                             pending ? (const double *)pan[pcur ^ 1] : (const double *)nullptr, jpend);
          pending = true;
          jpend = j0;
          pcur ^= 1;
          std::swap(lcur, lnext);
          j0 = j1;
          continue;
        }
      }
      flush();
      int64_t jg = j0;
      int nk = 0;
      for (int g = 0; g < ng && jg < n; g++) {
        const int nbc = (int)std::min<int64_t>(CH_NB, n - jg);
        const int64_t mrows = std::min<int64_t>(kl, n - jg - nbc);        // rows below the block
        hipLaunchKernelGGL(k_chol_panel, dim3((unsigned)std::max<int64_t>(1, tg_cdiv(mrows, 256))), dim3(256), 0, g_tg.stream, lb, ldl,
                           n, kl, jg, lcur, dinv, st, have ? (const double *)lcur : (const double *)nullptr);
        nk += nbc;
        const int64_t jn = jg + nbc;
        const int64_t m = std::max<int64_t>(0, std::min<int64_t>(n - 1, j0 + nk - 1 + kl) - jn + 1);
        const int nt = (int)tg_cdiv(m, 64);
        have = ahead && nt > 0;
        if (g + 1 < ng && jn < n)
          hipLaunchKernelGGL(k_chol_syrk<false>, dim3((unsigned)(nt + 1)), dim3(256), 0, g_tg.stream, lb, ldl, n, kl, j0, nk, jn, CH_NB,
                             nt, 1, (const double *)lcur, jg, nbc, (double *)nullptr, dinv, st);
        else if (have)
          hipLaunchKernelGGL(k_chol_syrk<true>, dim3((unsigned)(nt * (nt + 1) / 2 + 1)), dim3(256), 0, g_tg.stream, lb, ldl, n, kl, j0,
                             nk, jn, big, nt, 0, (const double *)lcur, jg, nbc, lnext, dinv, st);
        else
          hipLaunchKernelGGL(k_chol_syrk<false>, dim3((unsigned)(nt * (nt + 1) / 2 + 1)), dim3(256), 0, g_tg.stream, lb, ldl, n, kl, j0,
                             nk, jn, big, nt, 0, (const double *)lcur, jg, nbc, (double *)nullptr, dinv, st);
        if (have) std::swap(lcur, lnext);
        jg = jn;
      }
      j0 = jg;
    }
    flush();
    if (hipGetLastError() != hipSuccess || hipMemcpyAsync(&h, st, sizeof(h), hipMemcpyDeviceToHost, g_tg.stream) != hipSuccess ||
        hipStreamSynchronize(g_tg.stream) != hipSuccess)
      rc = 1;
    if (!rc && h.notpd) {
      if (trace) fprintf(stderr, "[trace] cholesky: pivot %d is not positive: LU\n", h.notpd - 1);
      go = false;
    }
    tg_dfree(pan[0]);
    tg_dfree(pan[1]);
  }
  if (!rc && go) {
    if (x != b && hipMemcpyAsync(x, b, (size_t)n * sizeof(double), hipMemcpyDeviceToDevice, g_tg.stream) != hipSuccess) rc = 1;
    if (!rc) {
      // several workgroups with a flag per block (TIGAR_CHOL_SWEEP=0, or a wait that gave up: one workgroup)
      bool swept = false;
      const int64_t nblk = tg_cdiv(n, CH_NB);
      // a workgroup's six groups of 32 threads take the tiles of ~4 owned blocks per step in one pass
      int sg = (int)std::min<int64_t>(CH_SG_MAX, std::max<int64_t>(CH_SG_MIN, (kl + 2 * CH_NB - 1) / CH_NB / 4));
      if (getenv("TIGAR_CHOL_SWEEP_WGS") && atoi(getenv("TIGAR_CHOL_SWEEP_WGS")) > 1) sg = std::min(atoi(getenv("TIGAR_CHOL_SWEEP_WGS")), 128);
      if (!(getenv("TIGAR_CHOL_SWEEP") && atoi(getenv("TIGAR_CHOL_SWEEP")) == 0) && nblk >= 4 * sg && g_tg.num_cu >= 2 * sg) {
        double *pub = nullptr, *xsave = nullptr;
        int *errf = nullptr;
        const int64_t npub = 2 * nblk * CH_NB;
        if (!tg_dmalloc(&pub, npub) && !tg_dmalloc(&xsave, n) && !tg_dmalloc(&errf, 4) &&
            hipMemsetAsync(pub, 0xff, (size_t)npub * sizeof(double), g_tg.stream) == hipSuccess &&          // (NaNs: nothing published)
            hipMemsetAsync(errf, 0, 4 * sizeof(int), g_tg.stream) == hipSuccess &&
            hipMemcpyAsync(xsave, x, (size_t)n * sizeof(double), hipMemcpyDeviceToDevice, g_tg.stream) == hipSuccess &&
            hipFuncSetAttribute((const void *)k_chol_sweep<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) == hipSuccess &&
            hipFuncSetAttribute((const void *)k_chol_sweep<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) == hipSuccess) {
          hipLaunchKernelGGL(k_chol_sweep<true>, dim3(sg), dim3(256), lds, g_tg.stream, (const double *)lb, (const double *)dinv, ldl,
                             n, kl, W, x, pub, errf);
          hipLaunchKernelGGL(k_chol_sweep<false>, dim3(sg), dim3(256), lds, g_tg.stream, (const double *)lb, (const double *)dinv, ldl,
                             n, kl, W, x, pub + nblk * CH_NB, errf);
          int herr = 1;
          if (hipGetLastError() == hipSuccess && hipMemcpyAsync(&herr, errf, sizeof(int), hipMemcpyDeviceToHost, g_tg.stream) == hipSuccess &&
              hipStreamSynchronize(g_tg.stream) == hipSuccess && herr == 0)
          {
            swept = true;
            if (trace) fprintf(stderr, "[trace] cholesky: substitutions on %d workgroups\n", sg);
          } else {
            (void)hipGetLastError();
            if (trace) fprintf(stderr, "[trace] cholesky: the sweeps on several workgroups gave up: one workgroup\n");
            if (hipMemcpyAsync(x, xsave, (size_t)n * sizeof(double), hipMemcpyDeviceToDevice, g_tg.stream) != hipSuccess) rc = 1;
          }
        } else
          (void)hipGetLastError();
        tg_dfree(pub);
        tg_dfree(errf);
        tg_dfree(xsave);
      }
      if (!rc && !swept) {
        hipLaunchKernelGGL(k_chol_fwd, dim3(1), dim3(CH_NT), lds, g_tg.stream, (const double *)lb, (const double *)dinv, ldl, n, kl, W, x);
        hipLaunchKernelGGL(k_chol_bwd, dim3(1), dim3(CH_NT), lds, g_tg.stream, (const double *)lb, (const double *)dinv, ldl, n, kl, W, x);
      }
      if (hipGetLastError() != hipSuccess || hipStreamSynchronize(g_tg.stream) != hipSuccess) rc = 1;
    }
    if (!rc) *done = 1;
    if (!rc && trace) fprintf(stderr, "[trace] cholesky: %lld x %lld, kl = %d, band %.2f GB\n", (long long)n, (long long)n, kl, ldl * n * 8e-9);
  }
  if (rc) tg_set_error("tg_chol_try: a kernel or a copy failed");
  tg_dfree(lb);
  tg_dfree(l11);
  tg_dfree(l11b);
  tg_dfree(dinv);
  tg_dfree(st);
  return rc;
}

extern "C" int tg_chol_solve(tg_csr_t k, tg_vec_t b, tg_vec_t x, int *done) {
  TG_REQUIRE_INIT();
  TG_REQUIRE(k && b && x && done, "null argument to tg_chol_solve");
  TG_REQUIRE_CANONICAL(k);
  const int64_t n = k->nrows;
  TG_REQUIRE(k->ncols == n && b->n == n && x->n == n, "tg_chol_solve: square system with matching vectors expected");
  *done = 0;
  if (n == 0) return 0;
  int kl = 0, ku = 0;
  int64_t bytes = 0;
  TG_TRY(tg_lu_band_info(k, &kl, &ku, &bytes));
  TG_TRY(tg_chol_try(k, kl, ku, b->d, x->d, done));
  g_tg.prof_n[TG_PROF_LU_CHOLESKY] += *done;
  return 0;
}
