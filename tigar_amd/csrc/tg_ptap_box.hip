// extractMatrix for Kronecker-structured extraction operators:  K = P^T A P  with
// P = (x)_k F_k, F_k = M_k (the 1-D extraction matrix) in the contracted directions and the
// identity elsewhere.  A is an ARBITRARY sparse matrix on the tensor index space; only P's
// structure is used: it fixes the factorisation (n0, n1, n2) of row/column indices.
//
// One workgroup per output row (I0,I1,I2):
//   stage 1  T[s] = sum_r w_r A[r, s] over the tensor product of the 1-D supports of
//            F_k^T rows I_k.  The columns s that can occur lie in a small box of the index
//            space (support +- bandwidth of A per direction, the bandwidth is measured, not
//            assumed), so T is a DENSE box in LDS addressed directly: one ds_add_f64 per
//            product, no hashing, no probing.
//   stage 2  the box is contracted with F_k direction by direction (sum factorisation): each
//            output element is a short dot product read from LDS -- no atomics at all.
//   finish   box entries that were structurally touched come out in lexicographic = column
//            order: compaction by ballot/prefix, fused MatZeroRowsColumns, no sort.
// "touched" flags are propagated through the contractions, so the pattern of the result is
// the structural pattern of P^T A P exactly as the general hash kernel produces it.
// If a box would not fit in LDS (or A couples outside the measured bandwidth, impossible by
// construction) the host falls back to the general kernel (tg_ptap.hip).
#include "tg_common.h"
#include <array>
#include <map>
#include <algorithm>
#include <cstring>

struct tg_box_args {
  const int32_t *rowcnt;        // loose rows of an intermediate stage result (nullptr: canonical CSR)
  const int64_t *rowptr_val;    // stacked view: separate row starts for the val array (nullptr: rowptr)
  const int64_t *rowptr;
  const int32_t *col;
  const double *val;
  int64_t row0, nrows;          // rows held by `cur` (global row index of local row 0)
  int d;
  int nin[3], nout[3];          // input / output index-space dimensions
  int contracted[3];
  const int32_t *mrp[3], *mcol[3];   // M_k (nin x nout) CSR, device
  const double *mval[3];
  const int32_t *trp[3], *tcol[3];   // M_k^T (nout x nin) CSR, device
  const double *tval[3];
  const int *blo[3], *bhi[3];   // box bounds (inclusive, input coordinates) per OUTPUT coordinate and direction
  int64_t out_row0, out_nrows, row_stride;
  unsigned mg01, sh01, mg0, sh0; // magic numbers: s / (n0*n1) and s / n0 for 0 <= s < 2^31
  int cap;                      // doubles per LDS buffer
  int cap1;                     // doubles of the second (ping-pong) buffer
  int ctab;                     // total entries of the contraction-table slices
  int coff[3], cstr[3], loff[3]; // per direction: slice offset / entries per output row / list offset
  int nlist;                    // total entries of the support lists
  int g1, lg1;                  // lanes per input row in stage 1
  // 24-bit column decoding relative to the box origin (all multiplies full rate): usable when
  // n0*n1 < 2^23 and n0*n1*max(B2) <= 2^24; quotient = (hi ? mulhi24 : mul24)(e, m) >> sh
  int fast24;
  unsigned m24a, sh24a, hi24a;  // e / (n0*n1)  for 0 <= e < n0*n1*max(B2)
  unsigned m24b, sh24b, hi24b;  // r / n0       for 0 <= r < n0*n1
  unsigned long long *prof;     // optional per-phase cycle counters (TIGAR_BOX_PROF)
  const double *rowmax;         // largest |entry| of every row of `cur` (box kernel, numeric mode: bound of the accumulators)
  int accum_mode;               // 0: integers unless the operand rows differ too much in scale, 1: integers, 2: floating point
};

// "line" kernel (one contracted direction): a wave walks along direction u
struct tg_line_args {
  int a, u;                     // contracted direction, march direction (-1: none, single row per wave)
  int mlen, ulo, uhi;           // rows per wave; range of I_u that intersects the requested rows
  int64_t x0, nx, nchunk;       // cross-section indices [x0, x0+nx) x chunks of mlen along u
  int capx, capy;               // doubles of the accumulator box / of the contracted box
  int64_t grab;                 // output entries a wave reserves at a time
};

enum { TG_BOX_OK = 0, TG_BOX_TOOBIG = 1, TG_BOX_RANGE = 2, TG_BOX_CAP = 3, TG_BOX_OUTSIDE = 4 };
enum { TG_BOXMODE_PROBE = 0, TG_BOXMODE_BUMP = 1 };

// s = q*dsr + r for 0 <= s < 2^31 by multiplication with a precomputed magic number:
// q = umulhi(s, mg) >> sh  (exact: mg = ceil(2^(32+sh) / dsr), sh = ceil(log2 dsr) - 1 when dsr > 1)
__device__ __forceinline__ void tg_divmod(unsigned s, unsigned dsr, unsigned mg, unsigned sh, unsigned *q,
                                          unsigned *r) {
  const unsigned qq = mg ? (__umulhi(s, mg) >> sh) : (s >> sh);   // mg == 0: dsr is a power of two
  *q = qq;
  *r = s - qq * dsr;
}

static void tg_magic(unsigned d, unsigned *mg, unsigned *sh) {
  if ((d & (d - 1)) == 0) {   // power of two (incl. 1)
    unsigned l = 0;
    while ((1u << l) < d) l++;
    *mg = 0;
    *sh = l;
    return;
  }
  unsigned l = 0;
  while ((1ull << l) < d) l++;          // 2^(l-1) < d < 2^l
  const unsigned s = l - 1;
  const unsigned long long num = 1ull << (32 + s);
  *mg = (unsigned)((num + d - 1) / d);  // < 2^32 because d > 2^(l-1)
  *sh = s;
}

// ---- 24-bit arithmetic (v_mul_u32_u24 / v_mul_hi_u32_u24 / v_mad_i32_i24 issue at full rate on
// CDNA; 32-bit integer multiplies at a quarter of it, and the decode of a column index needs six)
__device__ __forceinline__ int tg_mad_i24(int a, int b, int c) {
  int r;
  asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}
__device__ __forceinline__ unsigned tg_mulhi_u24(unsigned a, unsigned b) {
  unsigned r;
  asm("v_mul_hi_u32_u24 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ unsigned tg_mul_u24(unsigned a, unsigned b) {
  unsigned r;
  asm("v_mul_u32_u24 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
// variants taking the wave-uniform factor in a scalar register (no v_mov per use)
__device__ __forceinline__ int tg_mad_i24s(int a, int sb, int c) {
  int r;
  asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(r) : "v"(a), "s"(sb), "v"(c));
  return r;
}
__device__ __forceinline__ unsigned tg_mulhi_u24s(unsigned sa, unsigned b) {
  unsigned r;
  asm("v_mul_hi_u32_u24 %0, %1, %2" : "=v"(r) : "s"(sa), "v"(b));
  return r;
}
__device__ __forceinline__ unsigned tg_mul_u24s(unsigned sa, unsigned b) {
  unsigned r;
  asm("v_mul_u32_u24 %0, %1, %2" : "=v"(r) : "s"(sa), "v"(b));
  return r;
}
__device__ __forceinline__ unsigned tg_div24(unsigned e, unsigned m, unsigned sh, unsigned hi) {
  return (hi ? tg_mulhi_u24(e, m) : tg_mul_u24(e, m)) >> sh;
}

// Magic number for floor(e / dsr), 0 <= e < bound <= 2^24, with 24-bit operands:
// m = ceil(2^k / dsr) with 2^k >= bound*dsr is exact (classic round-up method); either the low
// 32 bits of the 48-bit product suffice (k < 32, bound*m < 2^32) or k >= 32 and the high part is
// used.  The result is verified at every multiple of dsr (both functions are monotone steps).
static bool tg_magic24(unsigned dsr, uint64_t bound, unsigned *m, unsigned *sh, unsigned *hi) {
  if (dsr == 0 || bound > (1ull << 24)) return false;
  int k = 0;
  while ((1ull << k) < bound * (uint64_t)dsr) k++;
  for (int pass = 0; pass < 2; pass++) {
    const int kk = pass == 0 ? k : std::max(k, 32);
    if (kk > 47) return false;
    const uint64_t mm = ((1ull << kk) + dsr - 1) / dsr;
    if (mm >= (1ull << 24)) continue;
    if (pass == 0 && (kk >= 32 || bound * mm >= (1ull << 32))) continue;
    bool ok = true;
    for (uint64_t j = 0; j * dsr < bound && ok; j++) {
      const uint64_t e1 = j * dsr;                       // first value with quotient j
      ok = ((e1 * mm) >> kk) == j;
      if (j > 0) ok = ok && (((e1 - 1) * mm) >> kk) == j - 1;
    }
    if (ok && bound > 0) ok = (((bound - 1) * mm) >> kk) == (bound - 1) / dsr;
    if (!ok) continue;
    *m = (unsigned)mm;
    *hi = kk >= 32 ? 1u : 0u;
    *sh = (unsigned)(kk >= 32 ? kk - 32 : kk);
    return true;
  }
  return false;
}

// Reach of the rows of `cur` per direction and coordinate: for every row coordinate r_k,
// lmax[k][r_k] = max (r_k - s_k), rmax[k][r_k] = max (s_k - r_k) over the row's columns s.
// (arrays of size n0+n1+n2, zero-initialised; used to size the accumulator boxes tightly)
__global__ void __launch_bounds__(256)
    k_box_reach(const int64_t *__restrict__ rowptr, const int32_t *__restrict__ rowcnt, const int32_t *__restrict__ col,
                int64_t nrows, int64_t row0,
                int n0, int n1, int n2, unsigned mg01, unsigned sh01, unsigned mg0, unsigned sh0, int64_t stride,
                int *__restrict__ lmax, int *__restrict__ rmax) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  const int64_t n01 = (int64_t)n0 * n1;
  for (int64_t r = wave * stride; r < nrows; r += nwaves * stride) {
    const int64_t g = r + row0;
    const int r2 = (int)(g / n01);
    const int rem = (int)(g - (int64_t)r2 * n01);
    const int r1 = rem / n0, r0 = rem - r1 * n0;
    int l0 = 0, l1 = 0, l2 = 0, h0 = 0, h1 = 0, h2 = 0;
    const int64_t qa = rowptr[r], qb = rowcnt ? qa + rowcnt[r] : rowptr[r + 1];
    for (int64_t q = qa + lane; q < qb; q += 64) {
      unsigned us2, usm, us1, us0;
      tg_divmod((unsigned)col[q], (unsigned)n01, mg01, sh01, &us2, &usm);
      tg_divmod(usm, (unsigned)n0, mg0, sh0, &us1, &us0);
      const int s2 = (int)us2, s1 = (int)us1, s0 = (int)us0;
      l0 = max(l0, r0 - s0);
      h0 = max(h0, s0 - r0);
      l1 = max(l1, r1 - s1);
      h1 = max(h1, s1 - r1);
      l2 = max(l2, r2 - s2);
      h2 = max(h2, s2 - r2);
    }
    for (int o = 32; o > 0; o >>= 1) {
      l0 = max(l0, __shfl_down(l0, o, 64));
      h0 = max(h0, __shfl_down(h0, o, 64));
      l1 = max(l1, __shfl_down(l1, o, 64));
      h1 = max(h1, __shfl_down(h1, o, 64));
      l2 = max(l2, __shfl_down(l2, o, 64));
      h2 = max(h2, __shfl_down(h2, o, 64));
    }
    if (lane == 0) {
      // most rows do not raise the maxima: test before the atomic
      if (lmax[r0] < l0) atomicMax(&lmax[r0], l0);
      if (rmax[r0] < h0) atomicMax(&rmax[r0], h0);
      if (lmax[n0 + r1] < l1) atomicMax(&lmax[n0 + r1], l1);
      if (rmax[n0 + r1] < h1) atomicMax(&rmax[n0 + r1], h1);
      if (lmax[n0 + n1 + r2] < l2) atomicMax(&lmax[n0 + n1 + r2], l2);
      if (rmax[n0 + n1 + r2] < h2) atomicMax(&rmax[n0 + n1 + r2], h2);
    }
  }
}

#define TG_BOX_UNTOUCHED 0x8000000000000000ull   // bit pattern of -0.0 (line kernel: untouched slot)
#define TG_BOX_UNROLL 4
#define TG_BOX_MAXLIST 96     // longest 1-D support list (entries of one row of F_k^T)
#define TG_BOX_MAXD 48        // most output indices per direction reachable from one box

// LDS carve (dynamic): buf0[cap] f64 | buf1[cap] f64 | pre_start[NT] i64 | pre_w[NT] f64 |
//                      lw[3][MAXLIST] f64 | cw[3][ctab] f64 | pre_len[NT] i32 | misc[32] i32 |
//                      la[3][MAXLIST] i32 | cptr[3][MAXD+1] i32 | ca[3][ctab] i32 | fl0[cap] u8 | fl1[cap] u8
// The per-row critical path is a chain of dependent global loads, so everything the row needs
// from the small 1-D tables is fetched in two cooperative rounds at the start (support lists,
// then operand-row descriptors + the table slices of the contraction stage) and kept in LDS.
template <int MODE, int NT>
__global__ void __launch_bounds__(NT)
    k_ptap_box(tg_box_args P, int64_t *__restrict__ row_cnt, int64_t *__restrict__ row_off,
               int32_t *__restrict__ k_col, double *__restrict__ k_val, unsigned long long *__restrict__ cursor,
               int64_t capacity, const uint8_t *__restrict__ mask, double diag, int *__restrict__ status,
               int *__restrict__ maxima) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  double *buf0 = reinterpret_cast<double *>(smem);
  double *buf1 = buf0 + P.cap;
  int64_t *pre_start = reinterpret_cast<int64_t *>(buf1 + P.cap1);
  double *pre_w = reinterpret_cast<double *>(pre_start + NT);
  double *lw = pre_w + NT;                       // support-list weights  [nlist]
  double *cw = lw + P.nlist;                     // contraction-table values [ctab]
  int *pre_len = reinterpret_cast<int *>(cw + P.ctab);
  int *misc = pre_len + NT;
  int *la = misc + 32;                           // support-list indices [nlist]
  int *cptr = la + P.nlist;                      // entries per output row [3][MAXD]
  int *ca = cptr + 3 * TG_BOX_MAXD;              // contraction-table input indices [ctab]
  uint8_t *fl0 = reinterpret_cast<uint8_t *>(ca + P.ctab);
  uint8_t *fl1 = fl0 + P.cap;

  const int tid = threadIdx.x;
  const int64_t L = tg_xcd_block(blockIdx.x, P.out_nrows);
  if (L >= P.out_nrows) return;
  const int64_t li = L * P.row_stride;
  const int64_t R = P.out_row0 + li;            // global output row
  int I[3];
  {
    const int64_t m01 = (int64_t)P.nout[0] * P.nout[1];
    I[2] = (int)(R / m01);
    const int rem = (int)(R - (int64_t)I[2] * m01);
    I[1] = rem / P.nout[0];
    I[0] = rem - I[1] * P.nout[0];
  }
  // ---- round 1: support lists of the contracted directions -> LDS
  int e0[3], len[3];
#pragma unroll
  for (int k = 0; k < 3; k++) {
    if (k < P.d && P.contracted[k]) {
      e0[k] = P.trp[k][I[k]];
      len[k] = min(P.trp[k][I[k] + 1] - e0[k], TG_BOX_MAXLIST);
    } else {
      e0[k] = 0;
      len[k] = 1;
    }
  }
  if (tid < 32) misc[tid] = (tid == 8 || tid == 10 || tid == 12) ? 0x7fffffff : ((tid == 9 || tid == 11 || tid == 13) ? -1 : 0);
#pragma unroll
  for (int k = 0; k < 3; k++) {
    if (k < P.d && P.contracted[k]) {
      for (int e = tid; e < len[k]; e += NT) {
        la[P.loff[k] + e] = P.tcol[k][e0[k] + e];
        lw[P.loff[k] + e] = P.tval[k][e0[k] + e];
      }
    }
  }
  __syncthreads();
  int lo[3], hi[3], bo[3], B[3];
#pragma unroll
  for (int k = 0; k < 3; k++) {
    if (k < P.d && P.contracted[k]) {
      lo[k] = len[k] > 0 ? la[P.loff[k]] : 0;
      hi[k] = len[k] > 0 ? la[P.loff[k] + len[k] - 1] : 0;
    } else
      lo[k] = hi[k] = (k < P.d) ? I[k] : 0;
    if (k < P.d) {
      bo[k] = P.blo[k][I[k]];
      B[k] = P.bhi[k][I[k]] - bo[k] + 1;
    } else {
      bo[k] = 0;
      B[k] = 1;
    }
  }
  int nbox = B[0] * B[1] * B[2];
  if (nbox > P.cap) {  // uniform
    if (tid == 0) atomicMax(status, TG_BOX_TOOBIG);
    return;
  }
  // ---- round 2 (all issued together): output ranges of the contraction (first/last column of
  // the rows [bo, bo+B) of M_k), descriptors of the first chunk of operand rows, box clearing
  for (int k = 0; k < P.d; k++) {
    if (!P.contracted[k]) continue;
    for (int a = tid; a < B[k]; a += NT) {
      const int p0 = P.mrp[k][bo[k] + a], p1 = P.mrp[k][bo[k] + a + 1];
      if (p1 > p0) {
        atomicMin(&misc[8 + 2 * k], P.mcol[k][p0]);
        atomicMax(&misc[9 + 2 * k], P.mcol[k][p1 - 1]);
      }
    }
  }
  for (int s = tid; s < nbox; s += NT) {
    buf0[s] = 0.0;
    fl0[s] = 0;
  }
  const int ncombo = len[0] * len[1] * len[2];
  const int64_t n01 = (int64_t)P.nin[0] * P.nin[1];
  bool range = false, outside = false, toobig = false;
  auto stage_rows = [&](int c0) {
    const int c = c0 + tid;
    if (c < ncombo) {
      const int q0 = c % len[0];
      const int q12 = c / len[0];
      const int q1 = q12 % len[1];
      const int q2 = q12 / len[1];
      const int qq[3] = {q0, q1, q2};
      int r[3];
      double w = 1.0;
#pragma unroll
      for (int k = 0; k < 3; k++) {
        if (k < P.d && P.contracted[k]) {
          r[k] = la[P.loff[k] + qq[k]];
          w *= lw[P.loff[k] + qq[k]];
        } else
          r[k] = lo[k];
      }
      const int64_t lr = (int64_t)r[0] + (int64_t)P.nin[0] * r[1] + n01 * r[2] - P.row0;
      if (lr < 0 || lr >= P.nrows) {
        range = true;
        pre_len[tid] = 0;
        pre_start[tid] = 0;
      } else {
        const int64_t s0 = P.rowptr[lr];
        pre_start[tid] = s0;
        pre_len[tid] = P.rowcnt ? P.rowcnt[lr] : (int)(P.rowptr[lr + 1] - s0);
      }
      pre_w[tid] = w;
    }
  };
  // bound of every accumulator of the box: sum over the operand rows of |weight| * (largest |entry| of the row); the
  // scatter adds integers on the grid derived from it (tg_fix, tg_common.h: bit-reproducible whatever the order)
  tg_fix_t fx = tg_fix_make(1.0);
  if (MODE != TG_BOXMODE_PROBE) {
    double bsum = 0.0, rlo = 1.7e308, rhi = 0.0;
    for (int c = tid; c < ncombo; c += NT) {
      const int q0 = c % len[0];
      const int q12 = c / len[0];
      const int q1 = q12 % len[1];
      const int q2 = q12 / len[1];
      const int qq[3] = {q0, q1, q2};
      int r[3];
      double w = 1.0;
#pragma unroll
      for (int k = 0; k < 3; k++) {
        if (k < P.d && P.contracted[k]) {
          r[k] = la[P.loff[k] + qq[k]];
          w *= lw[P.loff[k] + qq[k]];
        } else
          r[k] = lo[k];
      }
      const int64_t lr = (int64_t)r[0] + (int64_t)P.nin[0] * r[1] + n01 * r[2] - P.row0;
      if (lr >= 0 && lr < P.nrows) {
        const double rm = P.rowmax[lr];
        bsum += fabs(w) * rm;
        if (w != 0.0 && rm != 0.0) {
          rlo = fmin(rlo, rm);
          rhi = fmax(rhi, rm);
        }
      }
    }
    const double b1 = tg_block_sum_ordered(bsum, pre_w);
    tg_block_minmax(rlo, rhi, pre_w, &rlo, &rhi);
    fx = tg_fix_choose(b1, rlo, rhi, P.accum_mode);
  }
  stage_rows(0);
  __syncthreads();
  // table slices of the contraction stage: rows ilo..ihi of F_k^T, clipped to the box -> LDS
  // (issued now, consumed after stage 1)
  int ilo[3], D[3];
#pragma unroll
  for (int k = 0; k < 3; k++) {
    ilo[k] = misc[8 + 2 * k];
    const int ih = misc[9 + 2 * k];
    D[k] = (k < P.d && P.contracted[k] && ih >= ilo[k]) ? ih - ilo[k] + 1 : 0;
    if (D[k] > TG_BOX_MAXD) toobig = true;   // cannot happen: the host sized the tables from the same data
  }
  if (!toobig) {
    // one (direction, output row) pair per thread, spread over the workgroup
    int k = -1, q = 0;
    {
      int t = tid;
#pragma unroll
      for (int kk = 0; kk < 3; kk++) {
        if (k < 0 && kk < P.d && P.contracted[kk]) {
          if (t < D[kk]) {
            k = kk;
            q = t;
          } else
            t -= D[kk];
        }
      }
    }
    if (k >= 0) {
      // thread q copies the entries of row ilo+q that fall into the box; slot q*stride
      const int iout = ilo[k] + q;
      const int t0 = P.trp[k][iout], t1 = P.trp[k][iout + 1];
      const int stride = P.cstr[k];
      int n = 0;
      for (int t = t0; t < t1 && n < stride; t++) {
        const int a = P.tcol[k][t] - bo[k];
        if ((unsigned)a < (unsigned)B[k]) {
          ca[P.coff[k] + q * stride + n] = a;
          cw[P.coff[k] + q * stride + n] = P.tval[k][t];
          n++;
        }
      }
      cptr[k * TG_BOX_MAXD + q] = n;
    }
  }

  // ---- stage 1: scatter the weighted input rows into the dense box
  for (int c0 = 0; c0 < ncombo; c0 += NT) {
    if (c0 > 0) {
      stage_rows(c0);
      __syncthreads();
    }
    const int nch = min(NT, ncombo - c0);
    const int sub = tid & (P.g1 - 1);
    const int grp = tid >> P.lg1;
    const int ngrp = NT >> P.lg1;
    for (int le = grp; le < nch; le += ngrp) {
      const int64_t start = pre_start[le];
      const int ln = pre_len[le];
      const double w = pre_w[le];
      for (int o = sub; o < ln; o += P.g1 * TG_BOX_UNROLL) {
        int32_t cc[TG_BOX_UNROLL];
        double vv[TG_BOX_UNROLL];
#pragma unroll
        for (int u = 0; u < TG_BOX_UNROLL; u++) {
          const int oo = min(o + u * P.g1, ln - 1);
          cc[u] = P.col[start + oo];
          vv[u] = (MODE == TG_BOXMODE_PROBE) ? 0.0 : P.val[start + oo];   // (probe: pattern only; `start` may be a col-only offset)
        }
#pragma unroll
        for (int u = 0; u < TG_BOX_UNROLL; u++) {
          if (o + u * P.g1 < ln) {
            unsigned s2, sm, s1, s0;
            tg_divmod((unsigned)cc[u], (unsigned)n01, P.mg01, P.sh01, &s2, &sm);
            tg_divmod(sm, (unsigned)P.nin[0], P.mg0, P.sh0, &s1, &s0);
            const int x0 = (int)s0 - bo[0], x1 = (int)s1 - bo[1], x2 = (int)s2 - bo[2];
            if ((unsigned)x0 < (unsigned)B[0] && (unsigned)x1 < (unsigned)B[1] && (unsigned)x2 < (unsigned)B[2]) {
              const int slot = x0 + B[0] * (x1 + B[1] * x2);
              if (MODE != TG_BOXMODE_PROBE) tg_fix_add(reinterpret_cast<unsigned long long *>(buf0) + slot, tg_fix(w * vv[u], fx), fx);
              fl0[slot] = 1;
            } else
              outside = true;
          }
        }
      }
    }
    __syncthreads();
  }
  if (range) atomicMax(status, TG_BOX_RANGE);
  if (outside) atomicMax(status, TG_BOX_OUTSIDE);
  if (toobig) {
    if (tid == 0) atomicMax(status, TG_BOX_TOOBIG);
    return;
  }
  if (MODE != TG_BOXMODE_PROBE && !fx.fp) {   // the integers of the box back to floating point
    for (int s2 = tid; s2 < nbox; s2 += NT) buf0[s2] = tg_unfix(reinterpret_cast<const unsigned long long *>(buf0)[s2], fx);
    __syncthreads();
  }

  // ---- stage 2: contract the box with F_k, one direction after the other (LDS only, no atomics)
  double *src = buf0, *dst = buf1;
  uint8_t *fs = fl0, *fd = fl1;
  int org[3] = {bo[0], bo[1], bo[2]};
  for (int k = 0; k < P.d; k++) {
    if (!P.contracted[k]) continue;      // uniform
    int nB[3] = {B[0], B[1], B[2]};
    nB[k] = D[k];
    const int nnew = nB[0] * nB[1] * nB[2];
    if (nnew > ((dst == buf0) ? P.cap : P.cap1)) {
      if (tid == 0) atomicMax(status, TG_BOX_TOOBIG);
      return;
    }
    for (int o = tid; o < nnew; o += NT) {
      const int y0 = o % nB[0];
      const int y12 = o / nB[0];
      const int y1 = y12 % nB[1];
      const int y2 = y12 / nB[1];
      int y[3] = {y0, y1, y2};
      const int q = y[k];
      const int n = cptr[k * TG_BOX_MAXD + q];
      const int sk = (k == 0) ? 1 : (k == 1 ? B[0] : B[0] * B[1]);   // stride of direction k in src
      y[k] = 0;
      const int base = y[0] + B[0] * (y[1] + B[1] * y[2]);
      double acc = 0.0;
      uint8_t tch = 0;
      for (int t = 0; t < n; t++) {
        const int si = base + sk * ca[P.coff[k] + q * P.cstr[k] + t];
        acc += src[si] * cw[P.coff[k] + q * P.cstr[k] + t];
        tch |= fs[si];
      }
      dst[o] = acc;
      fd[o] = tch;
    }
    __syncthreads();
    B[k] = D[k];
    org[k] = ilo[k];
    double *tp = src;
    src = dst;
    dst = tp;
    uint8_t *tf = fs;
    fs = fd;
    fd = tf;
  }
  nbox = B[0] * B[1] * B[2];

  // ---- count, place and write the touched entries in column order
  int cnt_local = 0;
  for (int s = tid; s < nbox; s += NT) cnt_local += fs[s] ? 1 : 0;
  for (int o = 32; o > 0; o >>= 1) cnt_local += __shfl_down(cnt_local, o, 64);
  if ((tid & 63) == 0 && cnt_local) atomicAdd(&misc[0], cnt_local);
  __syncthreads();
  const int nK = misc[0];
  if (MODE == TG_BOXMODE_PROBE) {
    if (tid == 0) {
      row_cnt[li] = nK;
      atomicMax(&maxima[0], nbox);
      atomicMax(&maxima[1], nK);
    }
    return;
  }
  if (tid == 0) {
    const unsigned long long o = atomicAdd(cursor, (unsigned long long)nK);
    pre_start[0] = (int64_t)o;
    row_cnt[li] = nK;
    row_off[li] = (int64_t)o;
  }
  __syncthreads();
  const int64_t out0 = pre_start[0];
  if (out0 + nK > capacity) {
    if (tid == 0) atomicMax(status, TG_BOX_CAP);
    return;
  }
  const bool mrow = mask ? (mask[R] != 0) : false;
  const int64_t m0 = P.nout[0], m01 = (int64_t)P.nout[0] * P.nout[1];
  int base = 0;
  for (int s0 = 0; s0 < nbox; s0 += NT) {
    const int s = s0 + tid;
    const bool occ = (s < nbox) && fs[s];
    const unsigned long long bm = __ballot(occ);
    const int wv = tid >> 6, ln = tid & 63;
    if (ln == 0) misc[16 + wv] = __popcll(bm);
    __syncthreads();
    int woff = 0, chunk_total = 0;
#pragma unroll
    for (int q = 0; q < NT / 64; q++) {
      if (q < wv) woff += misc[16 + q];
      chunk_total += misc[16 + q];
    }
    if (occ) {
      const unsigned long long below = (ln == 0) ? 0ull : (~0ull >> (64 - ln));
      const int rank = base + woff + __popcll(bm & below);
      const int x0 = s % B[0];
      const int x12 = s / B[0];
      const int x1 = x12 % B[1];
      const int x2 = x12 / B[1];
      const int64_t c = (int64_t)(org[0] + x0) + m0 * (org[1] + x1) + m01 * (org[2] + x2);
      double v = src[s];
      if (mask && (mrow || mask[c])) v = (mrow && c == R) ? diag : 0.0;
      k_col[out0 + rank] = (int32_t)c;
      k_val[out0 + rank] = v;
    }
    base += chunk_total;
    __syncthreads();
  }
}

__global__ void __launch_bounds__(256)
    k_box_reorder(const int64_t *__restrict__ rowptr, const int64_t *__restrict__ tmp_off,
                  const int64_t *__restrict__ tmp_off_val,
                  const int32_t *__restrict__ tcol, const double *__restrict__ tval, int64_t nrows,
                  int32_t *__restrict__ col, double *__restrict__ val) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t r = wave; r < nrows; r += nwaves) {
    const int64_t dst = rowptr[r], n = rowptr[r + 1] - dst, src = tmp_off[r];
    const int64_t srcv = tmp_off_val ? tmp_off_val[r] : src;
    for (int64_t q = lane; q < n; q += 64) {
      col[dst + q] = tcol[src + q];
      val[dst + q] = tval[srcv + q];
    }
  }
}


// ------------------------------------------------------------------------------------------------
// "Line" kernel: ONE contracted direction a, one WAVE per run of output rows.
//
// The workgroup-per-row kernel above is bound by latency, not by LDS atomics or bandwidth
// (measured: dropping the atomics or half the operand rows changes nothing, the time is inversely
// proportional to the resident workgroups; ~10 barriers and ~8 dependent global round trips per
// row).  When only one direction is contracted a row has little work (|supp| operand rows), so the
// fixed cost dominates.  Here a wave owns a run of consecutive output rows along an uncontracted
// direction u: support list, contraction matrix and box extents of direction a are set up once per
// run, there are no barriers at all, and per output row
//   * the rowptr pairs of the operand rows were fetched during the previous row,
//   * the entries of the operand rows are requested TG_LINE_NB wave-wide loads at a time before the
//     first one is consumed (registers are the staging buffer),
//   * columns are decoded with 24-bit multiplies relative to the box origin, one ds_add_f64 each,
//   * the box is contracted out of registers against the dense (D x B_a) slice of F_a (LDS broadcast
//     reads) and stored over the head of its own line (no second box), "touched" travels as a bit mask,
//   * output space comes from a wave-private chunk of the temporary (one global atomic per
//     TG grab, not per row); rows are put in final order by k_box_reorder as before.
#define TG_LINE_NB 4

__device__ __forceinline__ int64_t tg_readlane_i64(int64_t v, int l) {
  const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(unsigned long long)v, l);
  const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)((unsigned long long)v >> 32), l);
  return (int64_t)(((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ double tg_readlane_f64(double v, int l) {
  return __longlong_as_double(tg_readlane_i64(__double_as_longlong(v), l));
}

// a wave-uniform pointer, forced into scalar registers (under register pressure the compiler keeps
// uniform 64-bit values in VGPRs, which would turn every buffer load into a waterfall loop)
__device__ __forceinline__ void *tg_uniform_ptr(const void *p) {
  const unsigned long long v = (unsigned long long)p;
  const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)v);
  const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(v >> 32));
  return (void *)(((unsigned long long)hi << 32) | lo);
}

template <int BAMAX, int DMAXT, int A, int U, bool HI>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(BAMAX <= 20 ? 5 : 4)))
    k_ptap_line(tg_box_args P, tg_line_args Q, int64_t *__restrict__ row_cnt, int64_t *__restrict__ row_off,
                int32_t *__restrict__ k_col, double *__restrict__ k_val, unsigned long long *__restrict__ cursor,
                int64_t capacity, const uint8_t *__restrict__ mask, double diag, int *__restrict__ status) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  double *X = reinterpret_cast<double *>(smem);          // accumulator box [capx]
  double *Fd = X + Q.capx;                               // dense slice of F_a^T: [DMAXT][BAMAX]
  unsigned *fmask = reinterpret_cast<unsigned *>(Fd + DMAXT * BAMAX);   // [DMAXT] structural masks, bit (Ba-1-x)

  const int lane = threadIdx.x;
  constexpr int a = A, u = U;                     // contracted direction, direction of the run (-1: none)
#ifdef TG_LINE_PROF
  unsigned long long tstamp = 0;
  if (P.prof) tstamp = __builtin_readcyclecounter();
  auto lap = [&](int slot) {
    if (P.prof) {
      const unsigned long long now = __builtin_readcyclecounter();
      if (lane == 0) atomicAdd(&P.prof[slot], now - tstamp);
      tstamp = now;
    }
  };
#else
  auto lap = [](int) {};
#endif
  const int64_t m01 = (int64_t)P.nout[0] * P.nout[1];
  const int64_t n01 = (int64_t)P.nin[0] * P.nin[1];
  // ---- the run of output rows of this wave
  int I[3];
  int Iu = 0, nsteps = 1;
  int64_t Rfirst, rstep = 0;
  if (u < 0) {
    const int64_t L = tg_xcd_block(blockIdx.x, P.out_nrows);
    if (L >= P.out_nrows) return;
    Rfirst = P.out_row0 + L;
    I[2] = (int)(Rfirst / m01);
    const int rem = (int)(Rfirst - (int64_t)I[2] * m01);
    I[1] = rem / P.nout[0];
    I[0] = rem - I[1] * P.nout[0];
  } else {
    const int64_t nb = Q.nx * Q.nchunk;
    const int64_t b = tg_xcd_block(blockIdx.x, nb);
    if (b >= nb) return;
    const int64_t c = b / Q.nx;
    const int64_t x = Q.x0 + (b - c * Q.nx);
    const int iu0 = Q.ulo + (int)c * Q.mlen;
    int last = min(Q.mlen, Q.uhi + 1 - iu0);
    I[0] = I[1] = I[2] = 0;
    if (u == 0) {
      I[1] = (int)(x % P.nout[1]);
      I[2] = (int)(x / P.nout[1]);
      I[0] = iu0;
      rstep = 1;
    } else if (u == 1) {
      I[0] = (int)(x % P.nout[0]);
      I[2] = (int)(x / P.nout[0]);
      I[1] = iu0;
      rstep = P.nout[0];
    } else {
      I[0] = (int)(x % P.nout[0]);
      I[1] = (int)(x / P.nout[0]);
      I[2] = iu0;
      rstep = m01;
    }
    const int64_t R0 = (int64_t)I[0] + (int64_t)P.nout[0] * I[1] + m01 * I[2];
    int first = 0;
    if (R0 < P.out_row0) first = (int)((P.out_row0 - R0 + rstep - 1) / rstep);
    const int64_t rend = P.out_row0 + P.out_nrows;
    if (R0 + (int64_t)(last - 1) * rstep >= rend) last = (R0 < rend) ? (int)((rend - 1 - R0) / rstep) + 1 : 0;
    if (first >= last) return;
    Iu = iu0 + first;
    Rfirst = R0 + (int64_t)first * rstep;
    nsteps = last - first;
  }
  const int Ia = a == 0 ? I[0] : (a == 1 ? I[1] : I[2]);
  const int32_t *trp = a == 0 ? P.trp[0] : (a == 1 ? P.trp[1] : P.trp[2]);
  const int32_t *tcol = a == 0 ? P.tcol[0] : (a == 1 ? P.tcol[1] : P.tcol[2]);
  const double *tval = a == 0 ? P.tval[0] : (a == 1 ? P.tval[1] : P.tval[2]);
  const int32_t *mrp = a == 0 ? P.mrp[0] : (a == 1 ? P.mrp[1] : P.mrp[2]);
  const int32_t *mcol = a == 0 ? P.mcol[0] : (a == 1 ? P.mcol[1] : P.mcol[2]);
  const int *ublo = u == 0 ? P.blo[0] : (u == 1 ? P.blo[1] : P.blo[2]);
  const int *ubhi = u == 0 ? P.bhi[0] : (u == 1 ? P.bhi[1] : P.bhi[2]);
  // ---- support list of row Ia of F_a^T: lane j holds entry j
  const int e0 = trp[Ia];
  const int len = trp[Ia + 1] - e0;             // <= 64 (host)
  int la = 0;
  double lw = 0.0;
  if (lane < len) {
    la = tcol[e0 + lane];
    lw = tval[e0 + lane];
  }
  int bo[3], B[3];
#pragma unroll
  for (int k = 0; k < 3; k++) {
    if (k < P.d) {
      bo[k] = P.blo[k][I[k]];
      B[k] = P.bhi[k][I[k]] - bo[k] + 1;
    } else {
      bo[k] = 0;
      B[k] = 1;
    }
  }
  const int boa = a == 0 ? bo[0] : (a == 1 ? bo[1] : bo[2]);
  const int Ba = a == 0 ? B[0] : (a == 1 ? B[1] : B[2]);
  // ---- output indices reachable from the box along a: first/last column of rows [boa, boa+Ba) of M_a
  int ilo = 0x7fffffff, ihi = -1;
  for (int x = lane; x < Ba; x += 64) {
    const int p0 = mrp[boa + x], p1 = mrp[boa + x + 1];
    if (p1 > p0) {
      ilo = min(ilo, mcol[p0]);
      ihi = max(ihi, mcol[p1 - 1]);
    }
  }
  for (int o = 32; o > 0; o >>= 1) {
    ilo = min(ilo, __shfl_xor(ilo, o, 64));
    ihi = max(ihi, __shfl_xor(ihi, o, 64));
  }
  ilo = __builtin_amdgcn_readfirstlane(ilo);
  ihi = __builtin_amdgcn_readfirstlane(ihi);
  const int D = ihi >= ilo ? ihi - ilo + 1 : 0;
  // (D > Ba: the contracted line is stored over the head of its own line of the box, which then has to be at least as
  // long.  The host compares the maxima only; a box that ends inside an element of a direction with repeated knots --
  // a coupling added by hand reaches there -- sees more functions than it has nodes.  Declined: general kernels.)
  if (D > DMAXT || Ba > BAMAX || D > Ba) {
    if (lane == 0) atomicMax(status, TG_BOX_TOOBIG);
    return;
  }
  // ---- dense slice Fd[q][x] = F_a^T[ilo+q, boa+x], structural mask per q
  for (int s = lane; s < DMAXT * BAMAX; s += 64) Fd[s] = 0.0;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  if (lane < D) {
    unsigned bits = 0;
    for (int t = trp[ilo + lane]; t < trp[ilo + lane + 1]; t++) {
      const int x = tcol[t] - boa;
      if ((unsigned)x < (unsigned)Ba) {
        Fd[lane * BAMAX + x] = tval[t];
        bits |= 1u << (Ba - 1 - x);
      }
    }
    fmask[lane] = bits;
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");

  const int negn01 = -(int)n01, negn0 = -P.nin[0];
  bool range = false, outside = false;
  int64_t chunk_pos = 0, chunk_end = 0;         // wave-private piece of the temporary
  // rowptr pair of operand row `lane` for the output row with coordinate iu along u
  int64_t ns0 = 0, ns1 = 0, nsv = 0;
  auto fetch = [&](int iu) {
    ns0 = ns1 = nsv = 0;
    if (lane < len) {
      const int r0 = a == 0 ? la : (u == 0 ? iu : I[0]);
      const int r1 = a == 1 ? la : (u == 1 ? iu : I[1]);
      const int r2 = a == 2 ? la : (u == 2 ? iu : I[2]);
      const int64_t lr = (int64_t)r0 + (int64_t)P.nin[0] * r1 + n01 * r2 - P.row0;
      if (lr < 0 || lr >= P.nrows)
        range = true;
      else {
        ns0 = P.rowptr[lr];
        ns1 = P.rowcnt ? ns0 + P.rowcnt[lr] : P.rowptr[lr + 1];
        nsv = P.rowptr_val ? P.rowptr_val[lr] : ns0;
      }
    }
  };
  // box extent along u of step `lane` of the run (runs have at most 64 rows)
  int ub0 = 0, ub1 = 0;
  if (u >= 0 && lane < nsteps) {
    ub0 = ublo[Iu + lane];
    ub1 = ubhi[Iu + lane];
  }
  fetch(u >= 0 ? Iu : 0);
  // ---- stream of (operand row, 64-entry pass) items, TG_LINE_NB at a time: all loads of a batch
  // are issued (buffer loads, scalar base, no address VGPRs) before the first item is consumed, and
  // the uniform per-item state the consumer needs (entry count, weight of the operand row) is read
  // out of the lanes at issue time, so the consumer does not walk the row list a second time
  // (that replay cost ~10 % of the kernel; batches of 2/3/4/6/8: 35.5/34.5/33.3/33.3/33.0 ms on the
  // x stage of 96^3 p=3).
  // (Issuing batch i+1 before consuming batch i, and the next row's first batch before this row's
  // contraction, was tried: the second register set costs half the resident waves and was 1.8x
  // slower -- the kernel is bound by instruction issue once enough waves are resident.)
  // operand rows of the output row being streamed: byte addresses of their first col / val entries
  // (computed per lane once per output row; the stream below only reads them out of the lanes, so an
  // item costs no scalar address arithmetic -- the scalar unit is the busiest one in this kernel)
  int64_t s0v = (int64_t)(uintptr_t)(P.col + ns0);
  int64_t s0val = (int64_t)(uintptr_t)(P.val + nsv);
  int lnv = (int)min((int64_t)0x7fffffff, ns1 - ns0);
  int cj = 0, cp = 0;                             // uniform cursor of the loads: operand row, entry offset of the pass
  const int lastj = max(len, 1) - 1;
  // (branch-free: items past the end of the row list reload the last row's first entries and carry
  // an entry count of 0)
  auto issue = [&](int32_t (&cc)[TG_LINE_NB], double (&vv)[TG_LINE_NB], int (&rem)[TG_LINE_NB], double (&ww)[TG_LINE_NB]) {
#pragma unroll
    for (int i = 0; i < TG_LINE_NB; i++) {
      const bool act = cj < len;                  // uniform
      const int jc = min(cj, lastj);
      const int64_t s = tg_readlane_i64(s0v, jc);
      const int64_t sv = tg_readlane_i64(s0val, jc);
      const int l = __builtin_amdgcn_readlane(lnv, jc);
      const int off = act ? cp : 0;
      rem[i] = act ? l - off : 0;                  // entries of this item (uniform)
      ww[i] = tg_readlane_f64(lw, jc);             // weight of its operand row (uniform)
      // entries past the end of the row are loaded (the arrays are padded) and masked.
      // Buffer loads: the per-item base is a scalar resource, the only address VGPR is lane*4 / lane*8,
      // so a batch needs no address registers and never waits on the other batch's destinations.
      const __amdgpu_buffer_rsrc_t rc =
          __builtin_amdgcn_make_buffer_rsrc((void *)(uintptr_t)s, 0, 0x7fffffff, 0x00020000);
      const __amdgpu_buffer_rsrc_t rv =
          __builtin_amdgcn_make_buffer_rsrc((void *)(uintptr_t)sv, 0, 0x7fffffff, 0x00020000);
      // (the pass offset goes into the scalar offset field of the load)
      cc[i] = __builtin_amdgcn_raw_buffer_load_b32(rc, lane * 4, off << 2, 0);
      vv[i] = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(rv, lane * 8, off << 3, 0));
      const bool endrow = cp + 64 >= l;
      cp = act ? (endrow ? 0 : cp + 64) : cp;
      cj = (act && endrow) ? cj + 1 : cj;
    }
  };
  lap(0);

  for (int t = 0; t < nsteps; t++) {
    const int64_t R = Rfirst + (int64_t)t * rstep;
    const int64_t li = R - P.out_row0;
    if (t + 1 < nsteps) fetch(Iu + t + 1);        // rowptr pairs of the next output row
    const bool mrow = mask ? (mask[R] != 0) : false;
    if (u >= 0) {
      const int b0 = __builtin_amdgcn_readlane(ub0, t), b1 = __builtin_amdgcn_readlane(ub1, t);
#pragma unroll
      for (int k = 0; k < 3; k++)
        if (k == u) {
          bo[k] = b0;
          B[k] = b1 - b0 + 1;
        }
    }
    const int nbox = B[0] * B[1] * B[2];
    if (nbox > Q.capx) {
      if (lane == 0) atomicMax(status, TG_BOX_TOOBIG);
      return;
    }
    {
      unsigned long long *x64 = reinterpret_cast<unsigned long long *>(X);
      for (int s = lane; s < nbox; s += 64) x64[s] = TG_BOX_UNTOUCHED;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");

    // ---- scatter
    {
      const unsigned cref = (unsigned)((int64_t)bo[0] + (int64_t)P.nin[0] * bo[1] + n01 * bo[2]);
      const unsigned emax = (unsigned)min((int64_t)0xffffffffll, n01 * B[2]);
      const int B01 = B[0] * B[1];
      const unsigned m24a = P.m24a, m24b = P.m24b;
      // total shift of the 48-bit product (HI: the high word is shifted by the remainder)
      const unsigned sh24a = HI ? P.sh24a : (P.hi24a ? P.sh24a + 32 : P.sh24a);
      const unsigned sh24b = HI ? P.sh24b : (P.hi24b ? P.sh24b + 32 : P.sh24b);
      // column -> box slot with 24-bit multiplies (the host only selects this kernel when they apply)
      auto add = [&](int32_t c, double v, double w, bool live) {
        const unsigned e = (unsigned)c - cref;
        unsigned x2, x1;
        if (HI) {
          x2 = tg_mulhi_u24s(m24a, e) >> sh24a;
        } else {
          x2 = (unsigned)((((unsigned long long)tg_mulhi_u24s(m24a, e) << 32) | tg_mul_u24s(m24a, e)) >> sh24a);
        }
        const int rem = tg_mad_i24s((int)x2, negn01, (int)e);
        if (HI) {
          x1 = tg_mulhi_u24s(m24b, (unsigned)rem) >> sh24b;
        } else {
          x1 = (unsigned)((((unsigned long long)tg_mulhi_u24s(m24b, (unsigned)rem) << 32) | tg_mul_u24s(m24b, (unsigned)rem)) >> sh24b);
        }
        const int x0 = tg_mad_i24s((int)x1, negn0, rem);
        const bool in = e < emax && (unsigned)x0 < (unsigned)B[0] && x1 < (unsigned)B[1];
        const int slot = tg_mad_i24s((int)x2, B01, tg_mad_i24s((int)x1, B[0], x0));
        if (live && in) unsafeAtomicAdd(&X[slot], fma(w, v, 0.0));
        outside |= live && !in;
      };
      cj = 0;
      cp = 0;
      while (cj < len) {
        int32_t cc[TG_LINE_NB];
        double vv[TG_LINE_NB], ww[TG_LINE_NB];
        int rem[TG_LINE_NB];
        issue(cc, vv, rem, ww);
#pragma unroll
        for (int i = 0; i < TG_LINE_NB; i++)
          if (rem[i] > 0) add(cc[i], vv[i], ww[i], lane < rem[i]);
      }
    }
    if (t + 1 < nsteps) {
      s0v = (int64_t)(uintptr_t)(P.col + ns0);
      s0val = (int64_t)(uintptr_t)(P.val + nsv);
      lnv = (int)min((int64_t)0x7fffffff, ns1 - ns0);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    lap(1);

    // ---- contraction along a out of registers: lane <-> one line of the box along a
    int Bc[3] = {B[0], B[1], B[2]};
    int org[3] = {bo[0], bo[1], bo[2]};
#pragma unroll
    for (int k = 0; k < 3; k++)
      if (k == a) {
        Bc[k] = D;
        org[k] = ilo;
      }
    const int nlines = nbox / Ba;               // product of the other two extents
    const int nY = nlines * D;
    const int sa = a == 0 ? 1 : (a == 1 ? B[0] : B[0] * B[1]);      // stride along a in X
    // The contracted line overwrites the head of its own line of X (D <= Ba, host-checked): each lane
    // has its whole line in registers before it stores, lines are disjoint, so no second box is needed.
    int mycnt = 0;
    const float invB0 = 1.0f / (float)B[0];
    for (int o0 = 0; o0 < nlines; o0 += 64) {
      const int o = o0 + lane;
      const bool act = o < nlines;
      int xb;                                     // offset of line o in X
      if (a == 0)
        xb = o * Ba;
      else if (a == 1) {
        const int x2 = (int)(((float)o + 0.5f) * invB0), x0 = o - x2 * B[0];
        xb = x0 + B[0] * B[1] * x2;
      } else
        xb = o;
      double xr[BAMAX];
      unsigned tb = 0;                            // touched flags, bit (Ba-1-x) <-> x (shifted in)
#pragma unroll
      for (int x = 0; x < BAMAX; x++) {
        xr[x] = -0.0;
        if (x < Ba) {                             // uniform
          if (act) xr[x] = X[xb + x * sa];
          tb = tb + tb + (((unsigned long long)__double_as_longlong(xr[x]) != TG_BOX_UNTOUCHED) ? 1u : 0u);
        }
      }
#pragma unroll 1
      for (int q = 0; q < D; q++) {
        double acc = 0.0;
#pragma unroll
        for (int x = 0; x < BAMAX; x++)
          if (x < Ba) acc = fma(xr[x], Fd[q * BAMAX + x], acc);   // untouched operands add +-0.0
        const bool tch = (tb & fmask[q]) != 0;
        if (act) {
          X[xb + q * sa] = tch ? acc : -0.0;
          mycnt += tch ? 1 : 0;
        }
      }
    }
    for (int o = 32; o > 0; o >>= 1) mycnt += __shfl_xor(mycnt, o, 64);
    const int nK = __builtin_amdgcn_readfirstlane(mycnt);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    lap(2);

    // ---- reserve, write: slots of the contracted box in lexicographic = column order
    const unsigned long long *xbits = reinterpret_cast<const unsigned long long *>(X);
    if (chunk_pos + nK > chunk_end) {             // uniform
      unsigned long long o = 0;
      // the rows of a run have similar lengths: reserve for the next few rows of the run (what is
      // left of a chunk at the end of the run is lost, so the look-ahead is short: reserving for the
      // whole run from its first row wasted 70 % of the temporary when the run length was a multiple
      // of the period of the row lengths and every run started on a long row)
      const int64_t grab = max((int64_t)nK, min(Q.grab, (int64_t)nK * min(nsteps - t, 8) * 9 / 8 + 32));
      if (lane == 0) o = atomicAdd(cursor, (unsigned long long)grab);
      chunk_pos = tg_readlane_i64((int64_t)o, 0);
      chunk_end = chunk_pos + grab;
      if (chunk_end > capacity) {
        if (lane == 0) atomicMax(status, TG_BOX_CAP);
        return;
      }
    }
    if (lane == 0) {
      row_cnt[li] = nK;
      row_off[li] = chunk_pos;
    }
    const int64_t m0 = P.nout[0];
    const float inv0 = 1.0f / (float)Bc[0], inv1 = 1.0f / (float)Bc[1];
    int64_t pos = chunk_pos;
    for (int s0 = 0; s0 < nY; s0 += 64) {
      const int s = s0 + lane;
      // (exact: s < 2^16 and s * Bc < 2^21, far inside float precision at the half-integer offsets)
      const int y12 = (int)(((float)s + 0.5f) * inv0);
      const int y0 = s - y12 * Bc[0];
      const int y2 = (int)(((float)y12 + 0.5f) * inv1);
      const int y1 = y12 - y2 * Bc[1];
      const int xi = y0 + B[0] * (y1 + B[1] * y2);
      const bool occ = s < nY && xbits[xi] != TG_BOX_UNTOUCHED;
      const unsigned long long bm = __ballot(occ);
      if (occ) {
        const unsigned long long below = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
        const int64_t dst = pos + __popcll(bm & below);
        const int64_t c = (int64_t)(org[0] + y0) + m0 * (org[1] + y1) + m01 * (org[2] + y2);
        double v = X[xi];
        if (mask && (mrow || mask[c])) v = (mrow && c == R) ? diag : 0.0;
        k_col[dst] = (int32_t)c;
        k_val[dst] = v;
      }
      pos += __popcll(bm);
    }
    chunk_pos += nK;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    lap(3);
  }
  if (range) atomicMax(status, TG_BOX_RANGE);
  if (outside) atomicMax(status, TG_BOX_OUTSIDE);
}

__global__ void k_i64_to_i32(const int64_t *__restrict__ a, int32_t *__restrict__ b, int64_t n) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) b[i] = (int32_t)a[i];
}
__global__ void k_shift_i64(int64_t *__restrict__ dst, const int64_t *__restrict__ src, int64_t n, int64_t shift) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) dst[i] = src[i] + shift;
}
__global__ void k_i32_to_i64(const int32_t *__restrict__ a, int64_t *__restrict__ b, int64_t n) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) b[i] = a[i];
}

// loose rows -> canonical CSR (scan of the row lengths + the row-reorder copy)
int tg_csr_compact_impl(tg_csr_s *in, tg_csr_s **out) {
  TG_REQUIRE_INIT();
  TG_REQUIRE(in && out && in->rowcnt, "tg_csr_compact: not a loose-row matrix");
  int64_t *cnt = nullptr;
  TG_TRY(tg_dmalloc(&cnt, in->nrows + 1));
  if (in->nrows > 0)
    hipLaunchKernelGGL(k_i32_to_i64, dim3(tg_grid_1d(in->nrows, 256)), dim3(256), 0, g_tg.stream, in->rowcnt, cnt, in->nrows);
  int64_t nnz = 0;
  int rc = tg_exclusive_scan_i64(cnt, in->nrows, &nnz);
  tg_csr_s *k = nullptr;
  if (!rc) rc = tg_csr_alloc(in->nrows, in->ncols, nnz, &k);
  if (!rc) {
    hipMemcpyAsync(k->rowptr, cnt, (size_t)(in->nrows + 1) * sizeof(int64_t), hipMemcpyDeviceToDevice, g_tg.stream);
    if (in->nrows > 0) {
      const unsigned rg = (unsigned)std::min<int64_t>(tg_cdiv(in->nrows, 4), (int64_t)g_tg.num_cu * 16);
      hipLaunchKernelGGL(k_box_reorder, dim3(rg), dim3(256), 0, g_tg.stream, k->rowptr, in->rowptr,
                         (const int64_t *)in->rowptr_val, in->col, in->val, in->nrows, k->col, k->val);
    }
    if (hipStreamSynchronize(g_tg.stream) != hipSuccess || hipGetLastError() != hipSuccess) {
      tg_set_error("tg_csr_compact: kernel failed");
      rc = 1;
    }
  }
  tg_dfree(cnt);
  if (rc) {
    if (k) tg_csr_destroy(k);
    return rc;
  }
  *out = k;
  return 0;
}

extern "C" int tg_csr_compact(tg_csr_t in, tg_csr_t *out) { return tg_csr_compact_impl(in, out); }

// largest |entry| of every row of the operand of a stage (canonical CSR, loose rows, or a stacked view): wave per row
__global__ void __launch_bounds__(256)
    k_box_row_absmax(const int64_t *__restrict__ rowptr, const int64_t *__restrict__ rowptr_val, const int32_t *__restrict__ rowcnt,
                     const double *__restrict__ val, int64_t nrows, double *__restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t r = wave; r < nrows; r += nwaves) {
    const int64_t len = rowcnt ? (int64_t)rowcnt[r] : rowptr[r + 1] - rowptr[r];
    const int64_t s0 = rowptr_val ? rowptr_val[r] : rowptr[r];
    double m = 0.0;
    for (int64_t q = lane; q < len; q += 64) {
      const double a = fabs(val[s0 + q]);
      m = (a > m || a != a) ? a : m;
    }
    for (int o = 32; o > 0; o >>= 1) {
      const double b = __shfl_xor(m, o);
      m = (b > m || b != b) ? b : m;
    }
    if (lane == 0) out[r] = m;
  }
}

static size_t tg_box_lds(int cap, int cap1, int ctab, int nlist, int nt) {
  size_t b = ((size_t)cap + cap1) * 8 + (size_t)nt * 16 + (size_t)nlist * 8 + (size_t)ctab * 8;   // f64 / i64 part
  b += (size_t)nt * 4 + 128 + (size_t)nlist * 4 + 3 * TG_BOX_MAXD * 4 + (size_t)ctab * 4;
  b += (size_t)cap * 2 + 64;
  return b;
}

// returns 0 ok, 100 = "use the general kernel" (box too large / not applicable), 101 = an entry fell
// outside a box sized from SAMPLED reach data (retry with the exact reach), other = error
static int tg_ptap_kron_impl(tg_csr_t cur, int64_t cur_row0, int d, const int64_t *dims_in, const tg_kron1d_t *fac,
                             int64_t out_row0, int64_t out_row1, const int32_t *zero_dofs, int64_t nzero, double diag,
                             int64_t reach_stride, bool loose_out, tg_csr_builder_s *dest, tg_csr_t *out) {
  TG_REQUIRE_INIT();
  TG_REQUIRE(cur && d >= 1 && d <= 3 && dims_in && fac && (out || dest) && out_row1 >= out_row0, "bad arguments to tg_ptap_kron");
  static bool lim = false;
  if (!lim) {
    hipFuncSetAttribute((const void *)k_ptap_box<TG_BOXMODE_PROBE, 256>, hipFuncAttributeMaxDynamicSharedMemorySize,
                        160 * 1024);
    hipFuncSetAttribute((const void *)k_ptap_box<TG_BOXMODE_BUMP, 256>, hipFuncAttributeMaxDynamicSharedMemorySize,
                        160 * 1024);
    hipFuncSetAttribute((const void *)k_ptap_box<TG_BOXMODE_PROBE, 64>, hipFuncAttributeMaxDynamicSharedMemorySize,
                        160 * 1024);
    hipFuncSetAttribute((const void *)k_ptap_box<TG_BOXMODE_BUMP, 64>, hipFuncAttributeMaxDynamicSharedMemorySize,
                        160 * 1024);
    lim = true;
  }
  tg_box_args P;
  memset(&P, 0, sizeof(P));
  {
    const char *am = getenv("TIGAR_PTAP_ACCUM");
    P.accum_mode = am && !strcmp(am, "int") ? 1 : am && !strcmp(am, "float") ? 2 : 0;
  }
  P.rowptr = cur->rowptr;
  P.rowcnt = cur->rowcnt;
  P.rowptr_val = cur->rowptr_val;
  P.col = cur->col;
  P.val = cur->val;
  P.row0 = cur_row0;
  P.nrows = cur->nrows;
  P.d = d;
  int64_t nin_total = 1, nout_total = 1;
  std::vector<void *> dev;
  int rc = 0;
  int maxlen[3] = {1, 1, 1};
  for (int k = 0; k < 3; k++) {
    P.nin[k] = P.nout[k] = 1;
    P.contracted[k] = 0;
  }
  for (int k = 0; k < d && !rc; k++) {
    TG_REQUIRE(dims_in[k] >= 1 && dims_in[k] < (1ll << 31), "bad input dimension %d", k);
    P.nin[k] = (int)dims_in[k];
    P.nout[k] = P.nin[k];
    if (fac[k].rowptr) {
      const tg_kron1d_t &F = fac[k];
      TG_REQUIRE(F.n == dims_in[k] && F.m >= 1 && F.col && F.val && F.t_rowptr && F.t_col && F.t_val,
                 "bad 1-D factor %d", k);
      P.contracted[k] = 1;
      P.nout[k] = (int)F.m;
      const int64_t nnz1 = F.rowptr[F.n];
      int32_t *a = nullptr, *b = nullptr, *c = nullptr, *e = nullptr;
      double *v = nullptr, *tv = nullptr;
      rc = tg_dmalloc(&a, F.n + 1) || tg_dmalloc(&b, nnz1) || tg_dmalloc(&v, nnz1) || tg_dmalloc(&c, F.m + 1) ||
           tg_dmalloc(&e, nnz1) || tg_dmalloc(&tv, nnz1);
      dev.insert(dev.end(), {(void *)a, (void *)b, (void *)v, (void *)c, (void *)e, (void *)tv});
      if (rc) break;
      hipMemcpyAsync(a, F.rowptr, (F.n + 1) * sizeof(int32_t), hipMemcpyHostToDevice, g_tg.stream);
      hipMemcpyAsync(b, F.col, nnz1 * sizeof(int32_t), hipMemcpyHostToDevice, g_tg.stream);
      hipMemcpyAsync(v, F.val, nnz1 * sizeof(double), hipMemcpyHostToDevice, g_tg.stream);
      hipMemcpyAsync(c, F.t_rowptr, (F.m + 1) * sizeof(int32_t), hipMemcpyHostToDevice, g_tg.stream);
      hipMemcpyAsync(e, F.t_col, nnz1 * sizeof(int32_t), hipMemcpyHostToDevice, g_tg.stream);
      hipMemcpyAsync(tv, F.t_val, nnz1 * sizeof(double), hipMemcpyHostToDevice, g_tg.stream);
      P.mrp[k] = a;
      P.mcol[k] = b;
      P.mval[k] = v;
      P.trp[k] = c;
      P.tcol[k] = e;
      P.tval[k] = tv;
      for (int64_t i = 0; i < F.m; i++) {
        const int p0 = F.t_rowptr[i], p1 = F.t_rowptr[i + 1];
        if (p1 > p0) maxlen[k] = std::max(maxlen[k], F.t_col[p1 - 1] - F.t_col[p0] + 1);
      }
    }
    nin_total *= P.nin[k];
    nout_total *= P.nout[k];
  }
  auto cleanup = [&]() {
    hipStreamSynchronize(g_tg.stream);
    for (void *p : dev) tg_dfree(p);
  };
  if (rc) {
    cleanup();
    return rc;
  }
  if (!(out_row1 <= nout_total && cur->ncols == nin_total)) {
    cleanup();
    tg_set_error("tg_ptap_kron: index spaces do not match the operands (cols %lld vs %lld)", (long long)cur->ncols,
                 (long long)nin_total);
    return 2;
  }
  // measured reach of cur's rows per direction and coordinate -> tight accumulator boxes
  int *status = (int *)g_tg.scratch;     // [0] status, [1..2] maxima
  hipMemsetAsync(status, 0, 8 * sizeof(int), g_tg.stream);
  const int ntot = P.nin[0] + P.nin[1] + P.nin[2];
  int *reach = nullptr;                  // lmax[ntot] | rmax[ntot]
  rc = tg_dmalloc(&reach, 2 * (int64_t)ntot);
  if (rc) {
    cleanup();
    return rc;
  }
  dev.push_back(reach);
  hipMemsetAsync(reach, 0, 2 * (size_t)ntot * sizeof(int), g_tg.stream);
  if ((int64_t)P.nin[0] * P.nin[1] >= (1ll << 31) || nin_total >= (1ll << 31)) {
    cleanup();
    tg_set_error("tg_ptap_kron: index space too large for 32-bit decomposition");
    return 100;
  }
  tg_magic((unsigned)((int64_t)P.nin[0] * P.nin[1]), &P.mg01, &P.sh01);
  tg_magic((unsigned)P.nin[0], &P.mg0, &P.sh0);
  if (cur->nrows > 0)
    hipLaunchKernelGGL(k_box_reach, dim3((unsigned)std::min<int64_t>(tg_cdiv(cur->nrows, 4), (int64_t)g_tg.num_cu * 32)),
                       dim3(256), 0, g_tg.stream, cur->rowptr, (const int32_t *)cur->rowcnt, cur->col, cur->nrows, cur_row0, P.nin[0],
                       P.nin[1],
                       P.nin[2], P.mg01, P.sh01, P.mg0, P.sh0, reach_stride, reach, reach + ntot);
  std::vector<int> hreach(2 * (size_t)ntot);
  hipMemcpyAsync(hreach.data(), reach, 2 * (size_t)ntot * sizeof(int), hipMemcpyDeviceToHost, g_tg.stream);
  hipStreamSynchronize(g_tg.stream);
  // box bounds per output coordinate: hull over the 1-D support of [a - lmax[a], a + rmax[a]]
  int hw[3] = {0, 0, 0};                 // largest box extent beyond the support (only for reporting)
  int64_t boxmax = 1;
  int maxBk[3] = {1, 1, 1};
  std::vector<int> hb;                   // blo | bhi for the three directions, concatenated
  size_t boff[3];
  {
    int coord0 = 0;
    for (int k = 0; k < 3; k++) {
      boff[k] = hb.size();
      const int nk = P.nin[k];
      const int mk = P.nout[k];
      const int *L = hreach.data() + coord0, *Rr = hreach.data() + ntot + coord0;
      std::vector<int> lo(mk), hi(mk);
      for (int i = 0; i < mk; i++) {
        int l = 0x7fffffff, h = -1;
        if (k < d && P.contracted[k]) {
          const tg_kron1d_t &F = fac[k];
          for (int t = F.t_rowptr[i]; t < F.t_rowptr[i + 1]; t++) {
            const int a = F.t_col[t];
            l = std::min(l, a - L[a]);
            h = std::max(h, a + Rr[a]);
          }
          if (h < l) {
            l = 0;
            h = 0;
          }
        } else {
          l = i - L[i];
          h = i + Rr[i];
        }
        l = std::max(l, 0);
        h = std::min(h, nk - 1);
        lo[i] = l;
        hi[i] = h;
        maxBk[k] = std::max(maxBk[k], h - l + 1);
      }
      hb.insert(hb.end(), lo.begin(), lo.end());
      hb.insert(hb.end(), hi.begin(), hi.end());
      coord0 += nk;
      boxmax *= maxBk[k];
      hw[k] = maxBk[k] - maxlen[k];
    }
  }
  int *dbox = nullptr;
  rc = tg_dmalloc(&dbox, (int64_t)hb.size());
  if (rc) {
    cleanup();
    return rc;
  }
  dev.push_back(dbox);
  hipMemcpyAsync(dbox, hb.data(), hb.size() * sizeof(int), hipMemcpyHostToDevice, g_tg.stream);
  for (int k = 0; k < 3; k++) {
    P.blo[k] = dbox + boff[k];
    P.bhi[k] = dbox + boff[k] + P.nout[k];
  }
  int cap = (int)std::min<int64_t>(boxmax, 1 << 20);
  cap = (cap + 7) & ~7;
  // per direction: largest box, most output indices reachable from a box, table/list layout
  int maxB[3], Dmax[3] = {1, 1, 1};
  int ctab = 0, nlist = 0, maxl = 1;
  for (int k = 0; k < 3; k++) {
    maxB[k] = maxBk[k];
    P.coff[k] = ctab;
    P.cstr[k] = 0;
    P.loff[k] = nlist;
    if (k < d && P.contracted[k]) {
      const tg_kron1d_t &F = fac[k];
      Dmax[k] = 1;
      for (int64_t a0 = 0; a0 + 1 <= F.n; a0++) {
        const int64_t a1 = std::min<int64_t>(F.n, a0 + maxB[k]) - 1;
        int c0 = 0x7fffffff, c1 = -1;
        // first / last column over the rows of the window (rows are sorted; scan ends only)
        for (int64_t a = a0; a <= a1; a++)
          if (F.rowptr[a + 1] > F.rowptr[a]) {
            c0 = std::min(c0, F.col[F.rowptr[a]]);
            break;
          }
        for (int64_t a = a1; a >= a0; a--)
          if (F.rowptr[a + 1] > F.rowptr[a]) {
            c1 = std::max(c1, F.col[F.rowptr[a + 1] - 1]);
            break;
          }
        if (c1 >= c0) Dmax[k] = std::max(Dmax[k], c1 - c0 + 1);
      }
      maxl = std::max(maxl, maxlen[k]);
      int rowmax = 1;     // longest row of F^T (entries)
      for (int64_t i = 0; i < F.m; i++) rowmax = std::max(rowmax, F.t_rowptr[i + 1] - F.t_rowptr[i]);
      P.cstr[k] = rowmax;
      ctab += Dmax[k] * rowmax;
      nlist += rowmax;
    }
  }
  ctab = (ctab + 7) & ~7;
  nlist = (nlist + 7) & ~7;
  // second buffer: sizes after the 1st and 3rd contraction
  int64_t sz[4];
  int nc = 0;
  {
    int64_t cur[3] = {maxB[0], maxB[1], maxB[2]};
    for (int k = 0; k < d; k++)
      if (P.contracted[k]) {
        cur[k] = Dmax[k];
        sz[nc++] = cur[0] * cur[1] * cur[2];
      }
  }
  int64_t c1need = 8;
  for (int q = 0; q < nc; q += 2) c1need = std::max(c1need, sz[q]);
  for (int q = 1; q < nc; q += 2) cap = (int)std::max<int64_t>(cap, sz[q]);
  const int cap1 = (int)((c1need + 7) & ~7);
  // The "touched" flags of both buffers are carved as cap bytes each.  Without repeated knots a window of B nodes reaches
  // at most B functions, so the box after a contraction (cap1) is no larger than before it (cap).  With repeated knots it
  // can be: a window that ends inside an element of a C^0 direction sees ALL functions of that element (a coupling added
  // by hand widens the windows beyond whole elements) -- then the flags of the second buffer ran past the end of the LDS
  // allocation, the writes were dropped and the tail of the row came out as untouched: rows of K short by their last
  // entries (found by the random runs of round 6; present since the kernel was written).
  if (cap1 > cap) cap = cap1;
  P.ctab = ctab;
  P.nlist = nlist;
  P.cap1 = cap1;
  bool too_many = false;
  for (int k = 0; k < 3; k++) too_many |= Dmax[k] > TG_BOX_MAXD;
  // workgroup size: one wave per output row when a row has little work (few operand rows),
  // a full 256-thread group otherwise
  int64_t maxcombo = 1;
  for (int k = 0; k < 3; k++) maxcombo *= (k < d && P.contracted[k]) ? std::max(1, P.cstr[k]) : 1;
  int nt = 256;   // (a wave-per-row variant, nt = 64, measured slower: rows in flight are LDS-bound either way)
  (void)maxcombo;
  if (getenv("TIGAR_BOX_NT")) nt = atoi(getenv("TIGAR_BOX_NT")) == 64 ? 64 : 256;
  // the table slices need one thread per (direction, output index)
  {
    int need = 0;
    for (int k = 0; k < 3; k++) need += (k < d && P.contracted[k]) ? Dmax[k] : 0;
    if (need > nt) nt = 256;
  }
  const size_t lds = tg_box_lds(cap, cap1, ctab, nlist, nt);
  if (boxmax > (1 << 20) || lds > 150 * 1024 || maxl > TG_BOX_MAXLIST || too_many) {
    cleanup();
    tg_set_error("tg_ptap_kron: accumulator box (%lld entries) does not fit in LDS", (long long)boxmax);
    return 100;
  }
  P.cap = cap;
  tg_magic((unsigned)((int64_t)P.nin[0] * P.nin[1]), &P.mg01, &P.sh01);
  tg_magic((unsigned)P.nin[0], &P.mg0, &P.sh0);
  {
    const uint64_t n01u = (uint64_t)P.nin[0] * P.nin[1];
    P.fast24 = 0;
    if (!getenv("TIGAR_BOX_SLOWDECODE") && n01u < (1u << 23) && n01u * (uint64_t)maxBk[2] <= (1u << 24) &&
        (int64_t)maxBk[0] * maxBk[1] < (1 << 23) &&
        tg_magic24((unsigned)n01u, n01u * (uint64_t)maxBk[2], &P.m24a, &P.sh24a, &P.hi24a) &&
        tg_magic24((unsigned)P.nin[0], n01u, &P.m24b, &P.sh24b, &P.hi24b))
      P.fast24 = 1;
  }
  unsigned long long *prof = nullptr;
  if (getenv("TIGAR_BOX_PROF") && !tg_dmalloc(&prof, 8)) {
    dev.push_back(prof);
    hipMemsetAsync(prof, 0, 8 * sizeof(unsigned long long), g_tg.stream);
    P.prof = prof;
  }
  // one contracted direction: wave-per-run "line" kernel
  tg_line_args Q;
  memset(&Q, 0, sizeof(Q));
  int line_variant = 0;          // 0: no, 1: <20,10>, 2: <32,16>
  size_t line_lds = 0;
  {
    int ncon = 0, a = -1;
    for (int k = 0; k < d; k++)
      if (P.contracted[k]) {
        ncon++;
        a = k;
      }
    const char *env = getenv("TIGAR_PTAP_LINE");
    if (ncon == 1 && !(env && atoi(env) == 0) && P.cstr[a] <= 64 && boxmax <= (1 << 16) && P.fast24) {
      if (maxBk[a] <= 20 && Dmax[a] <= 10)
        line_variant = 1;
      else if (maxBk[a] <= 32 && Dmax[a] <= 16)
        line_variant = 2;
      Q.a = a;
      Q.capx = (int)((boxmax + 7) & ~7ll);
      Q.capy = 0;                                  // (contracted in place)
      if (Dmax[a] > maxBk[a]) line_variant = 0;
      const int bam = line_variant == 1 ? 20 : 32, dmt = line_variant == 1 ? 10 : 16;
      line_lds = ((size_t)Q.capx + (size_t)bam * dmt + dmt) * 8;
      if (line_lds > 60 * 1024) line_variant = 0;
      // direction of the runs: the first uncontracted one; instantiated (a, u) pairs only
      int u = -1;
      for (int k = 0; k < d && u < 0; k++)
        if (!P.contracted[k] && P.nout[k] > 1) u = k;
      if (!((a == 0 && u == 1) || (a >= 1 && u == 0))) u = -1;
      if (u < 0 && a != 0) line_variant = 0;
      Q.u = u;
    }
  }
  if (cur->rowptr_val && !line_variant) {   // the box kernel reads col and val through one row start
    cleanup();
    return 102;
  }
  if ((int64_t)P.nin[0] * P.nin[1] >= (1ll << 31) || nin_total >= (1ll << 31)) {
    cleanup();
    tg_set_error("tg_ptap_kron: index space too large for 32-bit decomposition");
    return 100;
  }
  const double avg = cur->nrows ? (double)cur->nnz / (double)cur->nrows : 1.0;
  {
    const char *s = getenv("TIGAR_BOX_G1");
    int g = s ? atoi(s) : 4;
    if (!s) while (g < 64 && g * 2 <= avg * 0.44) g <<= 1;
    P.g1 = g;
    P.lg1 = 0;
    while ((1 << P.lg1) < g) P.lg1++;
  }
  P.out_row0 = out_row0;
  P.out_nrows = out_row1 - out_row0;
  P.row_stride = 1;
  const int64_t nrows = P.out_nrows;
  tg_csr_s *k = nullptr;
  uint8_t *mask = nullptr;
  int64_t *cnt = nullptr, *off = nullptr;
  unsigned long long *cursor = nullptr;
  int32_t *tcol = nullptr;
  double *tval = nullptr;
  if (nrows == 0) {
    rc = tg_csr_alloc(0, nout_total, 0, &k);
    if (!rc) hipMemsetAsync(k->rowptr, 0, sizeof(int64_t), g_tg.stream);
  } else {
    if (nzero > 0) rc = tg_build_dof_mask(zero_dofs, nzero, nout_total, &mask);
    if (!rc) rc = tg_dmalloc(&cnt, nrows + 1) || tg_dmalloc(&off, nrows + 1) || tg_dmalloc(&cursor, 1);
    double mean_k = 0.0;
    int hmax[3] = {0, 0, 0};
    // Capacity of the temporary: from the last product with the same signature (stage, index spaces in
    // the first two directions, kernel variant) if there was one -- entries actually used per row, chunk
    // slack included, largest value seen -- else from a probe of 512 sample rows.  A stage of a
    // streamed assembly is called once per sub-slab with the same signature; probe + scan + host round
    // trip cost ~0.8 ms of the ~9 ms a stage takes at 256^3 p=3.  An estimate that turns out too small
    // is caught by the kernel (TG_BOX_CAP) and retried with a larger temporary as before.
    const std::array<int, 8> cap_key = {d, P.contracted[0] | (P.contracted[1] << 1) | (P.contracted[2] << 2), P.nin[0],
                                        P.nin[1], P.nout[0], P.nout[1], line_variant, loose_out ? 1 : (dest ? 2 : 0)};
    // (first: largest use per row seen, second: capacity per row of the last allocation that was large
    // enough -- re-used as it is, so that the same sub-slab asks for the same block in every step: a
    // capacity recomputed from the use would land in another size class than the probe-sized
    // allocation of the first step and cost a round of hipMalloc in the second)
    struct tg_cap_entry {
      double used_pr = 0.0, cap_pr = 0.0;
      int hmax2 = 0;
    };
    static std::map<std::array<int, 8>, tg_cap_entry> cap_cache;
    static const bool force_probe = getenv("TIGAR_PTAP_PROBE") && atoi(getenv("TIGAR_PTAP_PROBE")) == 1;
    const auto cached = cap_cache.find(cap_key);
    const bool have_cached = !force_probe && cached != cap_cache.end();
    if (!rc && have_cached) {
      mean_k = cached->second.used_pr;
      hmax[2] = cached->second.hmax2;
    }
    if (!rc && !have_cached) {
      // probe a sample of rows: mean row length -> capacity of the temporary
      tg_box_args S = P;
      S.prof = nullptr;
      const int64_t nsample = std::min<int64_t>(nrows, 512);
      S.out_nrows = nsample;
      S.row_stride = std::max<int64_t>(1, nrows / nsample);
      hipMemsetAsync(cnt, 0, (size_t)(nrows + 1) * sizeof(int64_t), g_tg.stream);
      hipMemsetAsync(status, 0, 3 * sizeof(int), g_tg.stream);
      if (nt == 64)
        hipLaunchKernelGGL((k_ptap_box<TG_BOXMODE_PROBE, 64>), dim3((unsigned)(tg_cdiv(nsample, 8) * 8)), dim3(64), lds,
                           g_tg.stream, S, cnt, (int64_t *)nullptr, (int32_t *)nullptr, (double *)nullptr,
                           (unsigned long long *)nullptr, (int64_t)0, (const uint8_t *)nullptr, 0.0, status,
                           status + 1);
      else
        hipLaunchKernelGGL((k_ptap_box<TG_BOXMODE_PROBE, 256>), dim3((unsigned)(tg_cdiv(nsample, 8) * 8)), dim3(256),
                           lds, g_tg.stream, S, cnt, (int64_t *)nullptr, (int32_t *)nullptr, (double *)nullptr,
                           (unsigned long long *)nullptr, (int64_t)0, (const uint8_t *)nullptr, 0.0, status,
                           status + 1);
      hipMemcpyAsync(hmax, status, 3 * sizeof(int), hipMemcpyDeviceToHost, g_tg.stream);
      int64_t total = 0;
      rc = tg_exclusive_scan_i64(cnt, nrows, &total);
      hipStreamSynchronize(g_tg.stream);
      if (!rc && hmax[0] == TG_BOX_TOOBIG) rc = 100;
      if (!rc && hmax[0] == TG_BOX_OUTSIDE) rc = 101;
      if (!rc && hmax[0] == TG_BOX_RANGE) {
        tg_set_error("tg_ptap_kron: the row block does not cover the rows referenced (slab halo too small)");
        rc = 3;
      }
      mean_k = (double)total / (double)nsample;
    }
    unsigned long long used_final = 0;
    int64_t capacity = (int64_t)(mean_k * 1.05 * (double)nrows) + hmax[2] + 1024;
    if (line_variant && !have_cached) capacity = (int64_t)(capacity * 1.3) + nrows / 8 * 64;   // slack of the wave-private chunks
    // (measured use incl. the slack of the wave-private chunks, which varies with the order in which the
    // waves reserve: +20 %.  With +10 % a stage of 128^3 p=2 overflowed now and then, and a retry means
    // a new size class from hipMalloc: 0.19 instead of 0.035 s)
    if (have_cached) capacity = (int64_t)(cached->second.cap_pr * (double)nrows) + 65536;
    if (!rc && !line_variant) {        // row maxima of the operand: bounds of the box kernel's integer accumulation
      double *rowmax = nullptr;
      rc = tg_dmalloc(&rowmax, std::max<int64_t>(cur->nrows, 1));
      if (!rc) {
        dev.push_back(rowmax);
        if (cur->nrows > 0)
          hipLaunchKernelGGL(k_box_row_absmax, dim3((unsigned)std::min<int64_t>(tg_cdiv(cur->nrows, 4), (int64_t)g_tg.num_cu * 16)),
                             dim3(256), 0, g_tg.stream, cur->rowptr, (const int64_t *)cur->rowptr_val, (const int32_t *)cur->rowcnt,
                             cur->val, cur->nrows, rowmax);
        P.rowmax = rowmax;
      }
    }
    for (int attempt = 0; attempt < 6 && !rc; attempt++) {
      rc = tg_dmalloc(&tcol, capacity + TG_CSR_PAD) || tg_dmalloc(&tval, capacity + TG_CSR_PAD);
      if (rc) break;
      hipMemsetAsync(status, 0, 3 * sizeof(int), g_tg.stream);
      hipMemsetAsync(cursor, 0, sizeof(unsigned long long), g_tg.stream);
      hipMemsetAsync(cnt, 0, (size_t)(nrows + 1) * sizeof(int64_t), g_tg.stream);
      if (line_variant) {
        // runs along the first uncontracted direction; output space in wave-private chunks
        const int u = Q.u;
        int64_t nwaves = nrows;
        if (u >= 0) {
          const int64_t m0 = P.nout[0], m1 = P.nout[1], m01 = m0 * m1;
          const int64_t ra = out_row0, rb = out_row1 - 1;
          int64_t ulo, uhi, x0, nx;
          if (u == 2) {
            ulo = ra / m01;
            uhi = rb / m01;
            x0 = 0;
            nx = m01;
          } else if (u == 1) {
            ulo = 0;
            uhi = m1 - 1;
            x0 = m0 * (ra / m01);
            nx = m0 * (rb / m01 - ra / m01 + 1);
          } else {
            ulo = 0;
            uhi = m0 - 1;
            x0 = ra / m0;
            nx = rb / m0 - x0 + 1;
          }
          const int64_t span = uhi - ulo + 1;
          // rows per wave: 20 measured best at 96^3 and 256^3 p=3 (x/y/z stage at 96^3: 32 rows 30.9 / 15.8 / 8.5 ms,
          // 24: 30.2 / 15.1 / 8.1, 20: 29.0 / 13.2 / 7.3, 16: 29.6 / 14.2 / 7.3, 12: 29.4 / 13.7 / 7.0)
          int64_t T = std::min<int64_t>(getenv("TIGAR_LINE_RUN") ? std::max(1, atoi(getenv("TIGAR_LINE_RUN"))) : 20, span);
          while (T > 1 && nx * tg_cdiv(span, T) < (int64_t)g_tg.num_cu * 64) T = (T + 1) / 2;
          Q.mlen = (int)T;
          Q.ulo = (int)ulo;
          Q.uhi = (int)uhi;
          Q.x0 = x0;
          Q.nx = nx;
          Q.nchunk = tg_cdiv(span, T);
          nwaves = nx * Q.nchunk;
        }
        Q.grab = getenv("TIGAR_LINE_GRAB") ? atoll(getenv("TIGAR_LINE_GRAB")) : 8192;
        const unsigned grid = (unsigned)(tg_cdiv(nwaves, 8) * 8);
#define TG_LINE_LAUNCH(BA, DM, AA, UU)                                                                                 \
  do {                                                                                                                  \
    if (line_hi)                                                                                                        \
      hipLaunchKernelGGL((k_ptap_line<BA, DM, AA, UU, true>), dim3(grid), dim3(64), line_lds, g_tg.stream, P, Q, cnt,  \
                         off, tcol, tval, cursor, capacity, (const uint8_t *)mask, diag, status);                      \
    else                                                                                                                \
      hipLaunchKernelGGL((k_ptap_line<BA, DM, AA, UU, false>), dim3(grid), dim3(64), line_lds, g_tg.stream, P, Q, cnt, \
                         off, tcol, tval, cursor, capacity, (const uint8_t *)mask, diag, status);                      \
  } while (0)
#define TG_LINE_DISPATCH(BA, DM)                                  \
  do {                                                            \
    if (Q.a == 0 && u == 1) TG_LINE_LAUNCH(BA, DM, 0, 1);         \
    else if (Q.a == 1 && u == 0) TG_LINE_LAUNCH(BA, DM, 1, 0);    \
    else if (Q.a == 2 && u == 0) TG_LINE_LAUNCH(BA, DM, 2, 0);    \
    else TG_LINE_LAUNCH(BA, DM, 0, -1);                           \
  } while (0)
        const bool line_hi = P.hi24a && P.hi24b;
        if (line_variant == 1)
          TG_LINE_DISPATCH(20, 10);
        else
          TG_LINE_DISPATCH(32, 16);
      } else if (!P.rowmax) {
        tg_set_error("tg_ptap_kron: internal error (no row maxima for the box kernel)");
        rc = 1;
        break;
      } else if (nt == 64)
        hipLaunchKernelGGL((k_ptap_box<TG_BOXMODE_BUMP, 64>), dim3((unsigned)(tg_cdiv(nrows, 8) * 8)), dim3(64), lds,
                           g_tg.stream, P, cnt, off, tcol, tval, cursor, capacity, (const uint8_t *)mask, diag, status,
                           status + 1);
      else
        hipLaunchKernelGGL((k_ptap_box<TG_BOXMODE_BUMP, 256>), dim3((unsigned)(tg_cdiv(nrows, 8) * 8)), dim3(256), lds,
                           g_tg.stream, P, cnt, off, tcol, tval, cursor, capacity, (const uint8_t *)mask, diag, status,
                           status + 1);
      int h = 0;
      unsigned long long used = 0;
      hipMemcpyAsync(&h, status, sizeof(int), hipMemcpyDeviceToHost, g_tg.stream);
      hipMemcpyAsync(&used, cursor, sizeof(used), hipMemcpyDeviceToHost, g_tg.stream);
      if (hipStreamSynchronize(g_tg.stream) != hipSuccess || hipGetLastError() != hipSuccess) {
        tg_set_error("tg_ptap_kron: kernel failed to run (LDS %zu B)", lds);
        rc = 1;
        break;
      }
      if (prof) {
        unsigned long long hp[8];
        hipMemcpy(hp, prof, sizeof(hp), hipMemcpyDeviceToHost);
        const double tot = (double)(hp[0] + hp[1] + hp[2] + hp[3]) + 1e-30;
        fprintf(stderr, "[tigar] %s phases (%lld rows, fast24=%d): setup %.1f%%  scatter %.1f%%  contract %.1f%%  write %.1f%%\n",
                line_variant ? "line" : "box", (long long)nrows, P.fast24, 100.0 * hp[0] / tot, 100.0 * hp[1] / tot,
                100.0 * hp[2] / tot, 100.0 * hp[3] / tot);
      }
      used_final = used;
      if (h == TG_BOX_OK) {
        if (nrows > 0) {
          tg_cap_entry &c = cap_cache[cap_key];
          c.used_pr = std::max(c.used_pr, (double)used / (double)nrows);
          c.hmax2 = std::max(c.hmax2, hmax[2]);
          const double cp = (double)std::max<int64_t>(capacity - 65536, 0) / (double)nrows;
          // fixed once it has worked (a capacity that follows the largest use seen creeps upwards over
          // the first steps and every change is a round of hipMalloc); only an overflow moves it
          if (c.cap_pr == 0.0 || attempt > 0) c.cap_pr = cp;
        }
        break;
      }
      tg_dfree(tcol);
      tg_dfree(tval);
      tcol = nullptr;
      tval = nullptr;
      if (h == TG_BOX_CAP) {
        if (getenv("TIGAR_TRACE"))
          fprintf(stderr, "[tigar] ptap temporary too small: capacity %lld, used %llu, rows %lld, mean row %.1f, run %d -> retry\n",
                  (long long)capacity, used, (long long)nrows, mean_k, Q.mlen);
        if (nrows > 0) {
          tg_cap_entry &c = cap_cache[cap_key];      // what was reserved before the kernel gave up is a lower bound
          c.used_pr = std::max(c.used_pr, (double)used / (double)nrows);
          c.hmax2 = std::max(c.hmax2, hmax[2]);
        }
        capacity = std::max<int64_t>((int64_t)used + 1024, capacity * 2);
        continue;
      }
      if (h == TG_BOX_TOOBIG) {
        rc = 100;
        break;
      }
      if (h == TG_BOX_OUTSIDE) {
        rc = 101;
        break;
      }
      tg_set_error("tg_ptap_kron: kernel status %d", h);
      rc = h == TG_BOX_RANGE ? 3 : 4;
    }
    if (!rc && loose_out) {
      // intermediate stage: hand the temporary over as it is (rows where the kernel put them, lengths
      // in rowcnt) -- the consumers (next stage, vstack) read (start, length) pairs, so the
      // row-reorder copy and the scan are skipped
      k = new tg_csr_s();
      k->nrows = nrows;
      k->ncols = nout_total;
      k->nnz = (int64_t)used_final;
      k->rowptr = off;
      k->col = tcol;
      k->val = tval;
      rc = tg_dmalloc(&k->rowcnt, nrows);
      if (!rc) {
        hipLaunchKernelGGL(k_i64_to_i32, dim3(tg_grid_1d(nrows, 256)), dim3(256), 0, g_tg.stream, cnt, k->rowcnt, nrows);
        const int64_t endm = (int64_t)used_final;
        hipMemcpyAsync(off + nrows, &endm, sizeof(int64_t), hipMemcpyHostToDevice, g_tg.stream);
        hipStreamSynchronize(g_tg.stream);
        off = nullptr;      // owned by k now
        tcol = nullptr;
        tval = nullptr;
      } else {
        delete k;
        k = nullptr;
      }
    } else if (!rc && dest) {
      // last stage of a slab: rows go straight into the slab-wise builder of K (no block of its own,
      // no second copy by tg_csr_builder_append)
      int64_t nnz = 0;
      rc = tg_exclusive_scan_i64(cnt, nrows, &nnz);
      if (!rc && dest->m->ncols != nout_total) {
        tg_set_error("tg_ptap_kron_append: the builder has %lld columns, the product %lld", (long long)dest->m->ncols,
                     (long long)nout_total);
        rc = 2;
      }
      if (!rc) rc = tg_csr_builder_reserve(dest, nrows, nnz);
      if (!rc) {
        int64_t *rp = dest->m->rowptr + dest->rows_done;       // rp[0] == nnz_done already
        hipLaunchKernelGGL(k_shift_i64, dim3(tg_grid_1d(nrows, 256)), dim3(256), 0, g_tg.stream, rp + 1, cnt + 1, nrows,
                           dest->nnz_done);
        const unsigned rg = (unsigned)std::min<int64_t>(tg_cdiv(nrows, 4), (int64_t)g_tg.num_cu * 16);
        hipLaunchKernelGGL(k_box_reorder, dim3(rg), dim3(256), 0, g_tg.stream, rp, off, (const int64_t *)nullptr, tcol, tval,
                           nrows, dest->m->col, dest->m->val);
        if (hipGetLastError() != hipSuccess) {
          tg_set_error("tg_ptap_kron_append: reorder launch failed");
          rc = 1;
        } else {
          dest->rows_done += nrows;
          dest->nnz_done += nnz;
        }
      }
    } else if (!rc) {
      int64_t nnz = 0;
      rc = tg_exclusive_scan_i64(cnt, nrows, &nnz);
      if (!rc) rc = tg_csr_alloc(nrows, nout_total, nnz, &k);
      if (!rc) {
        hipMemcpyAsync(k->rowptr, cnt, (size_t)(nrows + 1) * sizeof(int64_t), hipMemcpyDeviceToDevice, g_tg.stream);
        const unsigned rg = (unsigned)std::min<int64_t>(tg_cdiv(nrows, 4), (int64_t)g_tg.num_cu * 16);
        hipLaunchKernelGGL(k_box_reorder, dim3(rg), dim3(256), 0, g_tg.stream, k->rowptr, off, (const int64_t *)nullptr,
                           tcol, tval, nrows, k->col, k->val);
        if (hipGetLastError() != hipSuccess) {
          tg_set_error("tg_ptap_kron: reorder launch failed");
          rc = 1;
        }
      }
    }
  }
  hipStreamSynchronize(g_tg.stream);
  tg_dfree(cnt);
  tg_dfree(off);
  tg_dfree(cursor);
  tg_dfree(tcol);
  tg_dfree(tval);
  tg_dfree(mask);
  cleanup();
  if (rc) {
    if (k) tg_csr_destroy(k);
    return rc;
  }
  if (out)
    *out = k;
  else if (k)
    tg_csr_destroy(k);
  return 0;
}

static int tg_ptap_kron_any(tg_csr_t cur, int64_t cur_row0, int d, const int64_t *dims_in, const tg_kron1d_t *fac,
                            int64_t out_row0, int64_t out_row1, const int32_t *zero_dofs, int64_t nzero, double diag,
                            bool loose_out, tg_csr_builder_s *dest, tg_csr_t *out) {
  TG_REQUIRE(cur && dims_in, "bad arguments to tg_ptap_kron");
  // The accumulator boxes are sized from the reach of cur's rows.  Scanning every entry of cur costs
  // a noticeable fraction of the product, so the reach is first measured on every `stride`-th row
  // (stride coprime to the fastest dimension: every coordinate of every direction is still visited);
  // the kernel flags any entry that falls outside a box, in which case the exact reach is used.
  int64_t stride = 1;
  if (cur->nrows > (1 << 20) && !getenv("TIGAR_BOX_EXACT_REACH")) {
    const int64_t cand[5] = {29, 31, 37, 41, 43};
    for (int c = 0; c < 5; c++)
      if (dims_in[0] % cand[c] != 0) {
        stride = cand[c];
        break;
      }
  }
  if (getenv("TIGAR_BOX_REACH_STRIDE")) stride = std::max(1, atoi(getenv("TIGAR_BOX_REACH_STRIDE")));
  int rc = tg_ptap_kron_impl(cur, cur_row0, d, dims_in, fac, out_row0, out_row1, zero_dofs, nzero, diag, stride, loose_out, dest, out);
  if (rc == 102) {   // stacked view into a kernel that cannot read it: compact once, then as usual
    tg_csr_s *flat = nullptr;
    TG_TRY(tg_csr_compact_impl(cur, &flat));
    rc = tg_ptap_kron_any(flat, cur_row0, d, dims_in, fac, out_row0, out_row1, zero_dofs, nzero, diag, loose_out, dest, out);
    tg_csr_destroy(flat);
    return rc;
  }
  if (rc == 101 && stride > 1)
    rc = tg_ptap_kron_impl(cur, cur_row0, d, dims_in, fac, out_row0, out_row1, zero_dofs, nzero, diag, 1, loose_out, dest, out);
  if (rc == 101) {
    // also with the exact reach: the support of an output row is not one interval per direction (periodic wrap) --
    // not a product for the box kernels; the caller takes the general ones (100 = declined, nothing was written)
    tg_set_error("tg_ptap_kron: an entry fell outside its accumulator box");
    rc = 100;
  }
  return rc;
}

extern "C" int tg_ptap_kron(tg_csr_t cur, int64_t cur_row0, int d, const int64_t *dims_in, const tg_kron1d_t *fac,
                            int64_t out_row0, int64_t out_row1, const int32_t *zero_dofs, int64_t nzero, double diag,
                            tg_csr_t *out) {
  return tg_ptap_kron_any(cur, cur_row0, d, dims_in, fac, out_row0, out_row1, zero_dofs, nzero, diag, false, nullptr, out);
}

// intermediate stage of a direction-by-direction product: no boundary conditions, result in the
// loose-row form (consumed by the next tg_ptap_kron* call, tg_csr_vstack or tg_csr_compact)
extern "C" int tg_ptap_kron_stage(tg_csr_t cur, int64_t cur_row0, int d, const int64_t *dims_in, const tg_kron1d_t *fac,
                                  int64_t out_row0, int64_t out_row1, tg_csr_t *out) {
  return tg_ptap_kron_any(cur, cur_row0, d, dims_in, fac, out_row0, out_row1, nullptr, 0, 1.0,
                          !getenv("TIGAR_PTAP_NOLOOSE"), nullptr, out);
}

// last stage of a slab with the result appended to a slab-wise builder of K (the rows must be the next
// rows the builder expects); returns 100 like tg_ptap_kron when the kernel declines (nothing appended)
extern "C" int tg_ptap_kron_append(tg_csr_t cur, int64_t cur_row0, int d, const int64_t *dims_in, const tg_kron1d_t *fac,
                                   int64_t out_row0, int64_t out_row1, const int32_t *zero_dofs, int64_t nzero, double diag,
                                   tg_csr_builder_t dest) {
  TG_REQUIRE(dest && dest->m, "null builder");
  return tg_ptap_kron_any(cur, cur_row0, d, dims_in, fac, out_row0, out_row1, zero_dofs, nzero, diag, false, dest, nullptr);
}
