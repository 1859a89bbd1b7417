// Internal definitions shared by the HIP translation units of libtigar_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <stdlib.h>
#include <vector>
#include "../../include/tigar_hip.h"

#define TG_MAX_DEGREE 8            // per-direction spline degree limit of the kernels
#define TG_WAVE 64
#define TG_CSR_PAD 264            // col/val allocations are padded: vector loads and the unrolled
                                  // row loops of the box PtAP (64 lanes x 4) may over-read

struct tg_ctx_t {
  int device = -1;
  hipStream_t stream = nullptr;    // the CURRENT stream: every launch / copy of the library goes here
  // Optional second stream (tg_stream_set): lets a producer of inputs (the FE input generator of the
  // next sub-slab: a pure HBM write stream) run beside the issue-bound PtAP kernels of the current
  // one.  `stream` and `scratch` alias the pair selected last; the allocator orders the reuse of
  // freed blocks across the two streams with events once the second stream has been used.
  hipStream_t streams[2] = {nullptr, nullptr};
  double *scratches[2] = {nullptr, nullptr};
  hipEvent_t xev = nullptr;
  int cur_stream = 0;
  bool multi = false;
  int num_cu = 0;
  bool ready = false;
  hipEvent_t ev0[8], ev1[8];
  hipEvent_t pev0 = nullptr, pev1 = nullptr;   // per-kernel accounting
  double prof_ms[TG_PROF_NSLOTS] = {0};
  int64_t prof_n[TG_PROF_NSLOTS] = {0};
  // small persistent device scratch for reductions / scalars
  double *scratch = nullptr;       // TG_SCRATCH_DOUBLES doubles
  double *host_pinned = nullptr;   // 64 doubles, pinned
};
#define TG_SCRATCH_DOUBLES (1 << 16)
extern tg_ctx_t g_tg;

void tg_set_error(const char *fmt, ...);

#define TG_CHECK_HIP(expr)                                                         \
  do {                                                                             \
    hipError_t _e = (expr);                                                        \
    if (_e != hipSuccess) {                                                        \
      tg_set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(_e)); \
      return 1;                                                                    \
    }                                                                              \
  } while (0)

#define TG_REQUIRE(cond, ...)                                                      \
  do {                                                                             \
    if (!(cond)) {                                                                 \
      tg_set_error(__VA_ARGS__);                                                   \
      return 2;                                                                    \
    }                                                                              \
  } while (0)

#define TG_REQUIRE_INIT() TG_REQUIRE(g_tg.ready, "tg_init() has not been called")
#define TG_REQUIRE_CANONICAL(m) \
  TG_REQUIRE(!(m) || !(m)->rowcnt, "%s: loose-row intermediate of a PtAP stage; call tg_csr_compact first", __func__)

#define TG_LAUNCH_CHECK() TG_CHECK_HIP(hipGetLastError())

struct tg_vec_s {
  int64_t n = 0;
  double *d = nullptr;
};

struct tg_sell_s;   // sliced, pattern-compressed copy for repeated products (tg_sell.hip)

struct tg_csr_s {
  int64_t nrows = 0, ncols = 0, nnz = 0;
  int64_t *rowptr = nullptr;   // device, nrows+1
  int32_t *col = nullptr;      // device, nnz
  double *val = nullptr;       // device, nnz
  // "loose rows" (outputs of intermediate PtAP stages only): row r occupies col/val
  // [rowptr[r], rowptr[r] + rowcnt[r]), rows lie in arbitrary order with gaps between them (the
  // order in which the producing kernel reserved space), nnz counts the col/val entries in use
  // incl. the gaps.  Accepted by tg_ptap_kron*, tg_csr_vstack, tg_csr_compact, tg_csr_download.
  int32_t *rowcnt = nullptr;   // device, nrows; nullptr = canonical CSR
  // "view" over the entry arrays of several blocks (tg_csr_vstack_view): loose rows whose starts
  // are element offsets relative to col / val of the FIRST block (they may point into the other
  // blocks' allocations, hence separate offsets for val); col/val are borrowed, not owned
  int64_t *rowptr_val = nullptr;   // device, nrows+1; nullptr = rowptr serves both arrays
  bool view = false;
  // SpMV plan (CSR-stream row blocks), built lazily
  int32_t *rowblocks = nullptr;  // device, nblocks+1 row indices
  int64_t nblocks = 0;
  int32_t max_row_nnz = 0;
  int spmv_cap = 0;              // LDS products per workgroup of the stream plan
  int spmv_mode = 0;             // 0 = not planned, 1 = stream (LDS), 2 = vector (wave/row)
  // snapshot of the VALUES in the sliced layout of tg_sell.hip: built by the Krylov solvers for the
  // duration of a solve, or on request (tg_spmv_sell) -- then the caller re-requests it after
  // changing values
  tg_sell_s *sell = nullptr;
  // Certificate of the sparsity pattern, set by the kernels of this library that write a matrix with a pattern known in
  // closed form (tg_kron_sum_csr: Kronecker product of the 1-D patterns it was given): pattern_tag = tg_pattern_hash of
  // those 1-D patterns, pattern_row0 = first global row held.  0 = no certificate (matrices from the host, products,
  // stacks).  Consumers that would otherwise verify the pattern entry by entry (tensor-pattern PtAP) compare tags.
  uint64_t pattern_tag = 0;
  int64_t pattern_row0 = 0;
  int sell_state = 0;            // 0 = not tried, 1 = in use, -1 = declined
  // half-storage product (tg_symgrid.hip): 1 = the copy of THESE values was compared with the CSR product once and agreed
  // (later solves with the same matrix do not compare again), -1 = it did not (not symmetric: no further attempts), 0 = not
  // compared yet.  Reset by every function that changes values.
  int sym_verified = 0;
  // diagonal of a square row block (entry (r, row0 + r)) recorded by the kernel that wrote the values (the z pass of
  // the tensor-pattern PtAP): the Jacobi set-up of the Krylov solvers then needs no pass over the matrix.  Dropped
  // by every function that changes values afterwards.
  double *diag_cache = nullptr;
  int64_t diag_rows = 0;         // rows of diag_cache that are valid (from row 0)
};

int tg_dmalloc_bytes(void **p, size_t bytes);   // caching allocator (tg_core.hip)
void tg_dfree(void *p);
int tg_pool_take_largest(size_t min_bytes, void **p, size_t *bytes);   // 1 = none; nothing is allocated
template <typename T>
static inline int tg_dmalloc(T **p, int64_t count) {
  if (count <= 0) count = 1;
  return tg_dmalloc_bytes((void **)p, (size_t)count * sizeof(T));
}
#define TG_TRY(expr)          \
  do {                        \
    int _rc = (expr);         \
    if (_rc) return _rc;      \
  } while (0)

static inline int64_t tg_cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

// device-wide in-place exclusive scan of int64 (n elements; out[n] gets the total if
// total_slot != nullptr it is also copied to host) -- tg_core.hip
int tg_exclusive_scan_i64(int64_t *d, int64_t n, int64_t *host_total);
// deterministic reduction of `n` partial doubles (device) into out_dev[0..k) sums of k
// interleaved streams -- tg_core.hip
int tg_csr_alloc(int64_t nrows, int64_t ncols, int64_t nnz, tg_csr_s **out);
struct tg_csr_builder_s {      // incremental vstack into one allocation (tg_extract.hip)
  tg_csr_s *m = nullptr;
  int64_t rows_done = 0, nnz_done = 0, cap = 0;
};
int tg_csr_builder_reserve(tg_csr_builder_s *b, int64_t nrows, int64_t nnz);   // room for one more block (may grow)
int tg_csr_compact_impl(tg_csr_s *in, tg_csr_s **out);   // loose rows -> canonical CSR (tg_ptap_box.hip)
int tg_spmv_plan(tg_csr_s *a);
// y = A x with x addressed by column index: x_shifted[col]; [cmin, cmax] = columns x_shifted may be read at
int tg_spmv_raw(tg_csr_s *a, const double *x_shifted, int64_t cmin, int64_t cmax, double *y);
// multiplicative hash (FNV prime, word-wise) over the 1-D CSR patterns (rows, columns per direction, row pointers, column indices) of a Kronecker-product
// pattern with `col_offset` added to every column; never 0
static inline uint64_t tg_pattern_hash(int d, const int64_t *nrows, const int64_t *ncols, const int32_t *const *rowptr,
                                       const int32_t *const *col, int64_t col_offset) {
  uint64_t h = 1469598103934665603ull;
  auto mix = [&](uint64_t v) {       // (one multiply per word: this runs on tables of ~10^4 entries at every assembly call)
    h = (h ^ v) * 1099511628211ull;
    h ^= h >> 29;
  };
  mix((uint64_t)d);
  mix((uint64_t)col_offset);
  for (int k = 0; k < d; k++) {
    mix((uint64_t)nrows[k]);
    mix((uint64_t)ncols[k]);
    for (int64_t r = 0; r <= nrows[k]; r++) mix((uint64_t)(uint32_t)rowptr[k][r]);
    for (int64_t q = 0; q < rowptr[k][nrows[k]]; q++) mix((uint64_t)(uint32_t)col[k][q]);
  }
  return h ? h : 1;
}

// wave-per-row Gustavson PtAP (tg_ptap_wave.hip): table sizes / lane groups of the two stages and, after the first numeric
// pass, the row pointer of K
struct tg_gw_plan {
  bool usable = false;
  int ts_am = 0, ts_k = 0;        // slots of a wave's table: rows of A M / rows of K
  int lg_m = 6, lg_am = 6;        // log2 of the lanes that walk one operand row: rows of M / rows of A M
  int max_am = 0, max_k = 0;
  double mean_am = 0.0, mean_k = 0.0;
  int64_t *k_rowptr = nullptr;    // device, nrows + 1 (valid when k_nnz >= 0)
  int64_t k_nnz = -1;
};
int tg_ptap_wave_plan(tg_csr_s *a, tg_csr_s *m, int64_t m_row0, tg_csr_s *mt, int max_k, double mean_k, tg_gw_plan *plan);
void tg_ptap_wave_plan_free(tg_gw_plan *plan);
int tg_ptap_wave_numeric(tg_gw_plan *plan, tg_csr_s *a, int64_t a_row0, tg_csr_s *m, int64_t m_row0, tg_csr_s *mt,
                         int64_t mt_row0, const uint8_t *mask, double diag, tg_csr_s **k_out);

int tg_sell_plan(tg_csr_s *a);          // tg_sell.hip
void tg_sell_drop(tg_csr_s *a);
void tg_sell_cache_clear(void);
void tg_kron_cache_clear(void);
void tg_asm_cache_clear(void);
void tg_kron_pattern_only(void);
int tg_h2d_staged(void *dst, const void *src, size_t bytes);   // stream-ordered upload of a small host table, no wait
int tg_sell_spmv(tg_csr_s *a, const double *x_shifted, int64_t cmin, int64_t cmax, double *y);
int64_t tg_sell_slice_rows(void);
int tg_sell_spmv_rows(tg_csr_s *a, const double *x_shifted, int64_t cmin, int64_t cmax, double *y, int64_t r0,
                      int64_t r1, const double *gate, double gate_tol);

// banded Cholesky for symmetric positive definite systems (tg_chol.hip): *done = 1 when it solved K x = b
int tg_chol_try(tg_csr_s *k, int kl, int ku, const double *b, double *x, int *done);
// half-storage product for symmetric box-stencil matrices on a 3-D grid (tg_symgrid.hip)
struct tg_symgrid_s;
int tg_symgrid_build(tg_csr_s *a, int64_t row0, int verify, tg_symgrid_s **out);   // *out = nullptr: declined
void tg_symgrid_free(tg_symgrid_s *s);
// part 0: the whole product; 1: what needs no halo of x (may run beside the exchange); 2: the rest
int tg_symgrid_spmv(tg_symgrid_s *s, tg_csr_s *a, const double *x_shifted, int64_t cmin, int64_t cmax, double *y, int part,
                    const double *gate, double gate_tol);
int tg_symgrid_chunks(const tg_symgrid_s *s);
void tg_symgrid_info(const tg_symgrid_s *s, int64_t *val_bytes, int64_t *stage_bytes);

int tg_csr_sort_rows(tg_csr_s *m);
int tg_csr_transpose_block(tg_csr_s *m, int64_t row_base, int64_t out_ncols, tg_csr_s **out);
// persistent single-kernel Krylov loop for small systems (tg_krylov_small.hip)
bool tg_cg_persistent_applies(const tg_csr_s *k);
int tg_cg_persistent(tg_csr_s *k, tg_vec_s *b, tg_vec_s *x, int pc, double rtol, double atol, int maxit, int nonzero_guess,
                     int *iters, double *resnorm, int *status);
int tg_gmres_persistent(tg_csr_s *k, tg_vec_s *b, tg_vec_s *x, int pc, double rtol, double atol, int maxit, int restart,
                        int nonzero_guess, int *iters, double *resnorm, int *status);
int tg_bicgstab_persistent(tg_csr_s *k, tg_vec_s *b, tg_vec_s *x, int pc, double rtol, double atol, int maxit, int nonzero_guess,
                           int *iters, double *resnorm, int *status);
int tg_pcg_cheb_persistent(tg_csr_s *k, tg_vec_s *b, tg_vec_s *x, int m, double theta, double delta, double rtol, double atol,
                           int maxit, int nonzero_guess, int *iters, double *resnorm, int *status);
int tg_build_dof_mask(const int32_t *dofs, int64_t n, int64_t ndofs_total, uint8_t **mask_out);
int64_t tg_spmv_num_partials(tg_csr_s *a);

// XCD-aware logical block id: hardware dispatches block b to XCD b % 8; give each XCD a
// contiguous range of logical blocks so neighbouring rows share one L2.
__device__ __forceinline__ int64_t tg_xcd_block(int64_t b, int64_t nb) {
  const int64_t per = (nb + 7) >> 3;
  return (b & 7) * per + (b >> 3);
}

// ---- small host/device helpers shared by the translation units ----
static inline int tg_grid_1d(int64_t n, int block) {
  int64_t g = tg_cdiv(n, block);
  int64_t cap = (int64_t)g_tg.num_cu * 8;
  if (g > cap) g = cap;
  if (g < 1) g = 1;
  return (int)g;
}

__device__ __forceinline__ double tg_wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  return v;
}

// block-wide sum for 256 threads; result valid in thread 0
__device__ __forceinline__ double tg_block_sum256(double v, double *lds4) {
  v = tg_wave_sum(v);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (lane == 0) lds4[w] = v;
  __syncthreads();
  double r = 0.0;
  if (threadIdx.x == 0) r = (lds4[0] + lds4[1]) + (lds4[2] + lds4[3]);
  __syncthreads();
  return r;
}

// ---- order-independent accumulation (the general PtAP kernels): a sum whose terms arrive in an order that differs from
// run to run (LDS atomics) is made bit-reproducible by adding INTEGERS.  With B >= the sum of the |terms| of any
// accumulator of the row, every term is rounded to the grid 2^e, e = exponent(B) - 61, and the 62-bit integers are added
// with 64-bit integer atomics: exact, hence independent of the order.  The grid lies 8 bits below the last bit of a
// term as large as B / 2, so terms within 2^-9 of the bound are not rounded at all and the rounding of the smaller ones
// (B 2^-62 each) stays below what a floating-point sum of the same terms may lose when its partial sums come near B.
struct tg_fix_t {
  double inv;     // 2^-e
  double scale;   // 2^e
  int fp;         // 1: this row accumulates floating-point numbers instead (see tg_fix_choose)
};
__device__ __forceinline__ tg_fix_t tg_fix_make(double bound) {
  tg_fix_t f;
  int e = ((bound > 0.0 && bound <= 1.7e308) ? ilogb(bound) : 0) - 61;
  e = max(e, -1000);                               // (2^-e must be a finite double)
  f.inv = ldexp(1.0, -e);
  f.scale = ldexp(1.0, e);
  f.fp = 0;
  return f;
}
// The grid belongs to the ROW of K: an entry keeps 62 bits relative to the largest sum of magnitudes of its row.  That is
// as good as floating point when the operand rows of the row are of one scale, and not when they are not -- a penalty or
// contact term of 1e12 in one FE row would leave the ordinary entries of the K rows it touches with seven digits.  So a
// row whose operand rows differ by more than 2^16 in their largest entries (or hold Inf / NaN) accumulates in floating
// point as before round 3 (accurate per entry, last bits dependent on the order); `mode`: 0 this rule, 1 always
// integers, 2 always floating point (TIGAR_PTAP_ACCUM=auto|int|float).
__device__ __forceinline__ tg_fix_t tg_fix_choose(double bound, double rowmax_lo, double rowmax_hi, int mode) {
  tg_fix_t f = tg_fix_make(bound);
  const bool finite = bound <= 1.7e308;            // (false for NaN as well)
  const bool one_scale = !(rowmax_hi > 65536.0 * rowmax_lo);
  f.fp = mode == 2 || !finite || (mode == 0 && !one_scale);
  return f;
}
// what is added to the accumulator (integer or the bits of the floating-point term), the addition itself, and the way back
__device__ __forceinline__ unsigned long long tg_fix(double v, const tg_fix_t &f) {
  return f.fp ? (unsigned long long)__double_as_longlong(v) : (unsigned long long)__double2ll_rn(v * f.inv);
}
__device__ __forceinline__ void tg_fix_add(unsigned long long *slot, unsigned long long t, const tg_fix_t &f) {
  if (f.fp)
    unsafeAtomicAdd(reinterpret_cast<double *>(slot), __longlong_as_double((long long)t));
  else
    atomicAdd(slot, t);
}
__device__ __forceinline__ double tg_unfix(unsigned long long n, const tg_fix_t &f) {
  return f.fp ? __longlong_as_double((long long)n) : __ll2double_rn((long long)n) * f.scale;
}
// smallest / largest of a non-negative quantity over the workgroup (entries < 0 are ignored; `scratch`: blockDim.x doubles)
__device__ __forceinline__ void tg_block_minmax(double lo, double hi, double *scratch, double *out_lo, double *out_hi) {
  const int tid = threadIdx.x;
  scratch[tid] = hi;
  __syncthreads();
  for (int o = (int)blockDim.x >> 1; o > 0; o >>= 1) {
    if (tid < o) scratch[tid] = fmax(scratch[tid], scratch[tid + o]);
    __syncthreads();
  }
  *out_hi = scratch[0];
  __syncthreads();
  scratch[tid] = lo;
  __syncthreads();
  for (int o = (int)blockDim.x >> 1; o > 0; o >>= 1) {
    if (tid < o) scratch[tid] = fmin(scratch[tid], scratch[tid + o]);
    __syncthreads();
  }
  *out_lo = scratch[0];
  __syncthreads();
}
// sum over the workgroup in a fixed order (tree in LDS; `scratch`: blockDim.x doubles, free before and after)
__device__ __forceinline__ double tg_block_sum_ordered(double x, double *scratch) {
  const int tid = threadIdx.x;
  scratch[tid] = x;
  __syncthreads();
  for (int o = (int)blockDim.x >> 1; o > 0; o >>= 1) {
    if (tid < o) scratch[tid] += scratch[tid + o];
    __syncthreads();
  }
  const double r = scratch[0];
  __syncthreads();
  return r;
}

__device__ __forceinline__ int64_t tg_wave_incl_scan_i64(int64_t v) {
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    int64_t t = __shfl_up(v, o, 64);
    if (lane >= o) v += t;
  }
  return v;
}

// returns exclusive prefix of `v` over the 256-thread block and the block total
__device__ __forceinline__ int64_t tg_block_excl_scan_i64(int64_t v, int64_t *lds5, int64_t *total) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  int64_t inc = tg_wave_incl_scan_i64(v);
  if (lane == 63) lds5[w] = inc;
  __syncthreads();
  int64_t woff = 0;
  for (int k = 0; k < w; k++) woff += lds5[k];
  if (total) *total = lds5[0] + lds5[1] + lds5[2] + lds5[3];
  __syncthreads();
  return woff + inc - v;
}
