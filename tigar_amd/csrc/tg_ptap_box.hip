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
#include <algorithm>

struct tg_box_args {
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

// Reach of the rows of `cur` per direction and coordinate: for every row coordinate r_k,
// lmax[k][r_k] = max (r_k - s_k), rmax[k][r_k] = max (s_k - r_k) over the row's columns s.
// (arrays of size n0+n1+n2, zero-initialised; used to size the accumulator boxes tightly)
__global__ void __launch_bounds__(256)
    k_box_reach(const int64_t *__restrict__ rowptr, const int32_t *__restrict__ col, int64_t nrows, int64_t row0,
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
    for (int64_t q = rowptr[r] + lane; q < rowptr[r + 1]; q += 64) {
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
        pre_len[tid] = (int)(P.rowptr[lr + 1] - s0);
      }
      pre_w[tid] = w;
    }
  };
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
          vv[u] = P.val[start + oo];
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
              if (MODE != TG_BOXMODE_PROBE) unsafeAtomicAdd(&buf0[slot], w * vv[u]);
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
                  const int32_t *__restrict__ tcol, const double *__restrict__ tval, int64_t nrows,
                  int32_t *__restrict__ col, double *__restrict__ val) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t r = wave; r < nrows; r += nwaves) {
    const int64_t dst = rowptr[r], n = rowptr[r + 1] - dst, src = tmp_off[r];
    for (int64_t q = lane; q < n; q += 64) {
      col[dst + q] = tcol[src + q];
      val[dst + q] = tval[src + q];
    }
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
                             int64_t reach_stride, tg_csr_t *out) {
  TG_REQUIRE_INIT();
  TG_REQUIRE(cur && d >= 1 && d <= 3 && dims_in && fac && out && out_row1 >= out_row0, "bad arguments to tg_ptap_kron");
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
  P.rowptr = cur->rowptr;
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
                       dim3(256), 0, g_tg.stream, cur->rowptr, cur->col, cur->nrows, cur_row0, P.nin[0], P.nin[1],
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
    if (!rc) {
      // probe a sample of rows: mean row length -> capacity of the temporary
      tg_box_args S = P;
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
    int64_t capacity = (int64_t)(mean_k * 1.05 * (double)nrows) + hmax[2] + 1024;
    for (int attempt = 0; attempt < 6 && !rc; attempt++) {
      rc = tg_dmalloc(&tcol, capacity + TG_CSR_PAD) || tg_dmalloc(&tval, capacity + TG_CSR_PAD);
      if (rc) break;
      hipMemsetAsync(status, 0, 3 * sizeof(int), g_tg.stream);
      hipMemsetAsync(cursor, 0, sizeof(unsigned long long), g_tg.stream);
      hipMemsetAsync(cnt, 0, (size_t)(nrows + 1) * sizeof(int64_t), g_tg.stream);
      if (nt == 64)
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
      if (h == TG_BOX_OK) break;
      tg_dfree(tcol);
      tg_dfree(tval);
      tcol = nullptr;
      tval = nullptr;
      if (h == TG_BOX_CAP) {
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
    if (!rc) {
      int64_t nnz = 0;
      rc = tg_exclusive_scan_i64(cnt, nrows, &nnz);
      if (!rc) rc = tg_csr_alloc(nrows, nout_total, nnz, &k);
      if (!rc) {
        hipMemcpyAsync(k->rowptr, cnt, (size_t)(nrows + 1) * sizeof(int64_t), hipMemcpyDeviceToDevice, g_tg.stream);
        const unsigned rg = (unsigned)std::min<int64_t>(tg_cdiv(nrows, 4), (int64_t)g_tg.num_cu * 16);
        hipLaunchKernelGGL(k_box_reorder, dim3(rg), dim3(256), 0, g_tg.stream, k->rowptr, off, tcol, tval, nrows, k->col,
                           k->val);
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
  *out = k;
  return 0;
}

extern "C" int tg_ptap_kron(tg_csr_t cur, int64_t cur_row0, int d, const int64_t *dims_in, const tg_kron1d_t *fac,
                            int64_t out_row0, int64_t out_row1, const int32_t *zero_dofs, int64_t nzero, double diag,
                            tg_csr_t *out) {
  TG_REQUIRE(cur && dims_in, "bad arguments to tg_ptap_kron");
  // The accumulator boxes are sized from the reach of cur's rows.  Scanning every entry of cur costs
  // a noticeable fraction of the product, so the reach is first measured on every `stride`-th row
  // (stride coprime to the fastest dimension: every coordinate of every direction is still visited);
  // the kernel flags any entry that falls outside a box, in which case the exact reach is used.
  int64_t stride = 1;
  if (cur->nrows > (1 << 20) && !getenv("TIGAR_BOX_EXACT_REACH")) {
    const int64_t cand[5] = {7, 11, 13, 17, 19};
    for (int c = 0; c < 5; c++)
      if (dims_in[0] % cand[c] != 0) {
        stride = cand[c];
        break;
      }
  }
  if (getenv("TIGAR_BOX_REACH_STRIDE")) stride = std::max(1, atoi(getenv("TIGAR_BOX_REACH_STRIDE")));
  int rc = tg_ptap_kron_impl(cur, cur_row0, d, dims_in, fac, out_row0, out_row1, zero_dofs, nzero, diag, stride, out);
  if (rc == 101 && stride > 1)
    rc = tg_ptap_kron_impl(cur, cur_row0, d, dims_in, fac, out_row0, out_row1, zero_dofs, nzero, diag, 1, out);
  if (rc == 101) {
    tg_set_error("tg_ptap_kron: an entry fell outside its accumulator box");
    rc = 4;
  }
  return rc;
}
