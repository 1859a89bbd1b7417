// extractMatrix: K = M^T A M  (tIGAr/common.py:1176-1204, PETSc MatPtAP [ext]).
//
// Row-wise fused triple product, one workgroup per K row i, both Gustavson stages kept in
// LDS so the intermediate (M^T A or A M) never touches HBM:
//
//   stage 1:  T[s]  = sum_{r in row i of M^T}  M^T[i,r] * A[r,s]      (LDS hash table 1)
//   stage 2:  K[i,j]= sum_{s in T}             T[s]     * M[s,j]      (LDS hash table 2)
//   finish :  rank-sort the <= (2p+1)^d entries of table 2 by column, apply the optional
//             fused MatZeroRowsColumns, write the CSR row.
//
// The kernel is latency-sensitive (three dependent global loads per operand row), so the
// row descriptors (start, length, scale) of a chunk of operand rows are first staged into LDS
// by all threads at once, and groups of G lanes then walk the operand rows with U-way
// unrolled 4+8-byte loads before touching the hash tables.
//
// One numeric pass, no separate symbolic pass: in BUMP mode every workgroup reserves its
// row's space in a temporary array with one atomicAdd, row counts are scanned afterwards and
// rows are copied into CSR order (costs 2x nnz(K) of traffic instead of a second traversal of
// A and M).  Once a plan knows the row pointer (second call with the same pattern, e.g. a
// Newton loop), rows are PLACED directly.
// Inputs are row blocks (z-slabs) with GLOBAL column indices: local row = global - row0.
#include "tg_common.h"
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstring>

struct tg_ptap_s {
  int64_t nrows = 0;        // K rows in this block
  int64_t ncols = 0;        // = M.ncols
  int64_t nnz = -1;         // known after the first numeric pass
  int64_t *rowptr = nullptr;  // device, nrows+1 (valid when nnz >= 0)
  int64_t a_row0 = 0, m_row0 = 0, mt_row0 = 0;
  int ts1 = 0, ts2 = 0, g1 = 0, g2 = 0;
  int max_t = 0, max_k = 0;
  double mean_k = 0.0;
  tg_gw_plan wave;          // the wave-per-row kernels (tg_ptap_wave.hip) take the product when their tables fit
};

struct tg_ptap_args {
  const int64_t *mt_rowptr;
  const int32_t *mt_col;
  const double *mt_val;
  const int64_t *a_rowptr;
  const int32_t *a_col;
  const double *a_val;
  const int64_t *m_rowptr;
  const int32_t *m_col;
  const double *m_val;
  int64_t a_row0, a_nrows, m_row0, m_nrows, mt_row0;
  int64_t nrows;       // K rows to compute
  int64_t row_stride;  // probing: row = idx * row_stride
  int ts1, ts2, lg1, lg2, g1, g2;
  const double *a_rowmax, *m_rowmax;   // largest |entry| of every row of A / M (numeric modes: bounds of the accumulators)
  int accum_mode;                      // 0: integers unless the operand rows differ too much in scale, 1: integers, 2: floating point
};

enum { TG_PTAP_OK = 0, TG_PTAP_OVF1 = 1, TG_PTAP_OVF2 = 2, TG_PTAP_RANGE = 3, TG_PTAP_CAP = 4 };
enum { TG_MODE_PROBE = 0, TG_MODE_BUMP = 1, TG_MODE_PLACED = 2 };

#define TG_PTAP_UNROLL 4

__device__ __forceinline__ unsigned tg_hash(int32_t key, int lg) {
  return ((unsigned)key * 2654435761u) >> (32 - lg);
}

// Bucketed hash tables: 4 keys per 16-byte bucket, one ds_read_b128 fetches a whole bucket.
// With the load factor kept below ~0.4 a lookup of a key that is already present -- the
// common case by a factor of 20-40 -- finishes in the first bucket: one LDS read, four
// compares, one ds_add_f64.  Inserts CAS the first empty position of the bucket; a full
// bucket without the key spills into the next bucket.  key < 0 = nothing to do.
// Returns false on table overflow.
typedef int tg_i4v __attribute__((ext_vector_type(4)));

// The values are accumulated as integers (tg_fix, tg_common.h): `vals` holds 64-bit integers until the table is compacted.
template <bool NUMERIC, int U>
__device__ __forceinline__ bool tg_hash_insert_n(int32_t *keys, unsigned long long *vals, int ts, int lg, const int32_t *key,
                                                 const unsigned long long *v, const tg_fix_t &fx) {
  const int nb_mask = (ts >> 2) - 1;
  // ---- fast round: the U home buckets are fetched back to back (independent LDS reads), then
  // matched; a key that sits in its home bucket -- the overwhelmingly common case -- costs one
  // read, four compares and one ds_add_f64 with a single exposed LDS latency for all U.
  int b[U];
  tg_i4v kk[U];
  bool pending[U];
#pragma unroll
  for (int u = 0; u < U; u++) {
    b[u] = (int)(tg_hash(key[u], lg) >> 2);
    kk[u] = *reinterpret_cast<const tg_i4v *>(keys + 4 * b[u]);
  }
  bool any_pending = false;
#pragma unroll
  for (int u = 0; u < U; u++) {
    const int32_t k = key[u];
    const int pos = (kk[u].x == k) ? 0 : (kk[u].y == k) ? 1 : (kk[u].z == k) ? 2 : (kk[u].w == k) ? 3 : -1;
    const bool hit = (k >= 0) && (pos >= 0);
    if (NUMERIC && hit) tg_fix_add(&vals[4 * b[u] + pos], v[u], fx);
    pending[u] = (k >= 0) && !hit;
    any_pending |= pending[u];
  }
  if (!__any(any_pending)) return true;
  // ---- slow rounds: inserts and bucket spills
  bool ok = true;
#pragma unroll
  for (int u = 0; u < U; u++) {
    const int32_t k = key[u];
    int bb = b[u];
    bool pend = pending[u];
    int rounds = 0;
    while (__any(pend)) {
      const tg_i4v q = *reinterpret_cast<const tg_i4v *>(keys + 4 * bb);
      int pos = (q.x == k) ? 0 : (q.y == k) ? 1 : (q.z == k) ? 2 : (q.w == k) ? 3 : -1;
      if (pend && pos < 0) {
        const int e = (q.x == -1) ? 0 : (q.y == -1) ? 1 : (q.z == -1) ? 2 : (q.w == -1) ? 3 : -1;
        if (e >= 0) {
          const int32_t old = atomicCAS(&keys[4 * bb + e], -1, k);
          if (old == -1 || old == k) pos = e;   // else: somebody else took it, re-read this bucket
        } else {
          bb = (bb + 1) & nb_mask;             // bucket full of other keys
        }
      }
      const bool hit = pend && pos >= 0;
      if (NUMERIC && hit) tg_fix_add(&vals[4 * bb + pos], v[u], fx);
      pend = pend && !hit;
      if (++rounds > ts) {  // wave-uniform
        ok = false;
        break;
      }
    }
  }
  return ok;
}

// walks operand rows [0,nch) described in LDS (start,len,scale) with groups of G lanes
template <bool NUMERIC>
__device__ __forceinline__ bool tg_accumulate_rows(int nch, const int64_t *pre_start, const int *pre_len,
                                                   const double *pre_w, const int32_t *__restrict__ col,
                                                   const double *__restrict__ val, int G, int lg, int32_t *keys,
                                                   unsigned long long *vals, int ts, int lgts, const tg_fix_t &fx) {
  const int tid = threadIdx.x;
  const int sub = tid & (G - 1);
  const int grp = tid >> lg;
  const int ngrp = (int)blockDim.x >> lg;
  bool ok = true;
  for (int le = grp; le < nch; le += ngrp) {
    const int64_t start = pre_start[le];
    const int len = pre_len[le];
    const double w = NUMERIC ? pre_w[le] : 0.0;
    for (int o = sub; o < len; o += G * TG_PTAP_UNROLL) {
      int32_t c[TG_PTAP_UNROLL];
      unsigned long long v[TG_PTAP_UNROLL];
#pragma unroll
      for (int u = 0; u < TG_PTAP_UNROLL; u++) {
        // unconditional (clamped) loads: no branches in the load phase
        const int oo = o + u * G;
        const int oc = min(oo, len - 1);
        const int32_t cc = col[start + oc];
        c[u] = (oo < len) ? cc : -2;
        v[u] = NUMERIC ? tg_fix(w * val[start + oc], fx) : 0ull;
      }
      ok &= tg_hash_insert_n<NUMERIC, TG_PTAP_UNROLL>(keys, vals, ts, lgts, c, v, fx);
    }
  }
  return ok;
}

// LDS carve (dynamic, 16-byte aligned offsets), NT = threads per workgroup:
//   keys1[ts1] | keys2[ts2] | cnt[4] | pre_len[NT]  (int32)
//   pre_start[NT] (int64) | pre_w[NT] | vals1[ts1] | vals2[ts2]  (f64)
template <int MODE, int NT>
__global__ void __launch_bounds__(NT)
    k_ptap(tg_ptap_args P, int64_t *__restrict__ row_cnt, int64_t *__restrict__ row_off, int32_t *__restrict__ k_col,
           double *__restrict__ k_val, unsigned long long *__restrict__ cursor, int64_t capacity,
           const uint8_t *__restrict__ mask, double diag, int *__restrict__ status, int *__restrict__ maxima) {
  constexpr bool NUMERIC = MODE != TG_MODE_PROBE;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  int32_t *keys1 = reinterpret_cast<int32_t *>(smem);
  int32_t *keys2 = keys1 + P.ts1;
  int32_t *cnt = keys2 + P.ts2;
  int *pre_len = cnt + 4;
  int64_t *pre_start = reinterpret_cast<int64_t *>(pre_len + NT);
  double *pre_w = reinterpret_cast<double *>(pre_start + NT);
  unsigned long long *vals1 = reinterpret_cast<unsigned long long *>(pre_w + NT);
  unsigned long long *vals2 = vals1 + (NUMERIC ? P.ts1 : 0);

  const int tid = threadIdx.x;
  const int64_t L = tg_xcd_block(blockIdx.x, P.nrows);
  if (L >= P.nrows) return;
  const int64_t li = L * P.row_stride;  // local K / M^T row
  const int lgts1 = 32 - __clz(P.ts1 - 1), lgts2 = 32 - __clz(P.ts2 - 1);

  for (int s = tid; s < P.ts1; s += NT) {
    keys1[s] = -1;
    if (NUMERIC) vals1[s] = 0ull;
  }
  for (int s = tid; s < P.ts2; s += NT) {
    keys2[s] = -1;
    if (NUMERIC) vals2[s] = 0ull;
  }
  if (tid < 4) cnt[tid] = 0;
  __syncthreads();

  bool ovf1 = false, ovf2 = false, range = false;
  // ---- stage 1: T = (row i of M^T) * A
  int row_fp = 0;                      // this row accumulates in floating point (both stages)
  {
    const int64_t e0 = P.mt_rowptr[li], e1 = P.mt_rowptr[li + 1];
    // bound of every entry of T: sum over the operand rows of |weight| * (largest |entry| of the row)
    tg_fix_t fx = tg_fix_make(1.0);
    if (NUMERIC) {
      double bsum = 0.0, rlo = 1.7e308, rhi = 0.0;
      for (int64_t e = e0 + tid; e < e1; e += NT) {
        const int64_t ra = (int64_t)P.mt_col[e] - P.a_row0;
        if (ra >= 0 && ra < P.a_nrows) {
          const double rm = P.a_rowmax[ra];
          bsum += fabs(P.mt_val[e]) * rm;
          if (P.mt_val[e] != 0.0 && rm != 0.0) {   // (a NaN fails both comparisons below and reaches the sum)
            rlo = fmin(rlo, rm);
            rhi = fmax(rhi, rm);
          }
        }
      }
      const double b1 = tg_block_sum_ordered(bsum, pre_w);
      tg_block_minmax(rlo, rhi, pre_w, &rlo, &rhi);
      fx = tg_fix_choose(b1, rlo, rhi, P.accum_mode);
    }
    for (int64_t c0 = e0; c0 < e1; c0 += NT) {
      const int64_t e = c0 + tid;
      if (e < e1) {
        const int64_t ra = (int64_t)P.mt_col[e] - P.a_row0;
        if (ra < 0 || ra >= P.a_nrows) {
          range = true;
          pre_len[tid] = 0;
          pre_start[tid] = 0;
        } else {
          const int64_t s0 = P.a_rowptr[ra];
          pre_start[tid] = s0;
          pre_len[tid] = (int)(P.a_rowptr[ra + 1] - s0);
        }
        if (NUMERIC) pre_w[tid] = P.mt_val[e];
      }
      __syncthreads();
      const int nch = (int)min((int64_t)NT, e1 - c0);
      if (!tg_accumulate_rows<NUMERIC>(nch, pre_start, pre_len, pre_w, P.a_col, P.a_val, P.g1, P.lg1, keys1, vals1,
                                       P.ts1, lgts1, fx))
        ovf1 = true;
      __syncthreads();
    }
    // the integers of table 1 back to floating point (in place; a slot is read and written by one thread)
    if (NUMERIC)
      for (int s = tid; s < P.ts1; s += NT) vals1[s] = (unsigned long long)__double_as_longlong(tg_unfix(vals1[s], fx));
    __syncthreads();
    row_fp = fx.fp;
  }

  // ---- compact the occupied (key, value) pairs of table 1 to the front of its own storage
  // (each chunk is read into registers before anything is written at or below it)
  for (int c0 = 0; c0 < P.ts1; c0 += NT) {
    const int slot = c0 + tid;
    const int32_t key = (slot < P.ts1) ? keys1[slot] : -1;
    const double tv = (NUMERIC && slot < P.ts1) ? __longlong_as_double((long long)vals1[slot]) : 0.0;
    __syncthreads();
    const bool occ = key != -1;
    const unsigned long long m = __ballot(occ);
    int base = 0;
    if ((tid & 63) == 0 && m) base = atomicAdd(&cnt[0], __popcll(m));
    base = __shfl(base, 0, 64);
    if (occ) {
      const unsigned long long below = ((tid & 63) == 0) ? 0ull : (~0ull >> (64 - (tid & 63)));
      const int pos = base + __popcll(m & below);
      keys1[pos] = key;
      if (NUMERIC) vals1[pos] = (unsigned long long)__double_as_longlong(tv);
    }
    __syncthreads();
  }
  const int nT = cnt[0];

  // ---- stage 2: K row = T * M; bound of its entries: sum over T of |T_c| * (largest |entry| of row c of M)
  tg_fix_t fx2 = tg_fix_make(1.0);
  if (NUMERIC) {
    double bsum = 0.0;
    for (int e = tid; e < nT; e += NT) {
      const int64_t sm = (int64_t)keys1[e] - P.m_row0;
      if (sm >= 0 && sm < P.m_nrows) bsum += fabs(__longlong_as_double((long long)vals1[e])) * P.m_rowmax[sm];
    }
    const double b2 = tg_block_sum_ordered(bsum, pre_w);
    fx2 = tg_fix_choose(b2, 1.0, 1.0, row_fp ? 2 : 1);
  }
  for (int c0 = 0; c0 < nT; c0 += NT) {
    const int e = c0 + tid;
    if (e < nT) {
      const int64_t sm = (int64_t)keys1[e] - P.m_row0;
      if (sm < 0 || sm >= P.m_nrows) {
        range = true;
        pre_len[tid] = 0;
        pre_start[tid] = 0;
      } else {
        const int64_t s0 = P.m_rowptr[sm];
        pre_start[tid] = s0;
        pre_len[tid] = (int)(P.m_rowptr[sm + 1] - s0);
      }
      if (NUMERIC) pre_w[tid] = __longlong_as_double((long long)vals1[e]);
    }
    __syncthreads();
    if (!tg_accumulate_rows<NUMERIC>(min(NT, nT - c0), pre_start, pre_len, pre_w, P.m_col, P.m_val, P.g2, P.lg2,
                                     keys2, vals2, P.ts2, lgts2, fx2))
      ovf2 = true;
    __syncthreads();
  }
  if (ovf1) atomicMax(status, TG_PTAP_OVF1);
  if (ovf2) atomicMax(status, TG_PTAP_OVF2);
  if (range) atomicMax(status, TG_PTAP_RANGE);

  // ---- compact table 2 into (ckey, cval) living in table-1 storage (no longer needed)
  int32_t *ckey = keys1;   // capacity ts1 >= ts2 (host guarantees)
  double *cval = reinterpret_cast<double *>(vals1);
  for (int s0 = 0; s0 < P.ts2; s0 += NT) {
    const int s = s0 + tid;
    const bool occ = (s < P.ts2) && keys2[s] != -1;
    const unsigned long long m = __ballot(occ);
    int base = 0;
    if ((tid & 63) == 0 && m) base = atomicAdd(&cnt[1], __popcll(m));
    base = __shfl(base, 0, 64);
    if (NUMERIC && occ) {
      const unsigned long long below = ((tid & 63) == 0) ? 0ull : (~0ull >> (64 - (tid & 63)));
      const int pos = base + __popcll(m & below);
      ckey[pos] = keys2[s];
      cval[pos] = tg_unfix(vals2[s], fx2);
    }
  }
  __syncthreads();
  const int nK = cnt[1];
  if (MODE == TG_MODE_PROBE) {
    if (tid == 0) {
      row_cnt[li] = nK;
      atomicMax(&maxima[0], nT);
      atomicMax(&maxima[1], nK);
    }
    return;
  }

  // ---- where does this row go?
  int64_t out0;
  if (MODE == TG_MODE_PLACED) {
    out0 = row_off[li];
    if (row_off[li + 1] - out0 != nK) {  // pattern changed since the plan was made
      if (tid == 0) atomicMax(status, TG_PTAP_CAP);
      return;
    }
  } else {
    if (tid == 0) {
      const unsigned long long o = atomicAdd(cursor, (unsigned long long)nK);
      pre_start[0] = (int64_t)o;
      row_cnt[li] = nK;
      row_off[li] = (int64_t)o;
    }
    __syncthreads();
    out0 = pre_start[0];
    if (out0 + nK > capacity) {
      if (tid == 0) atomicMax(status, TG_PTAP_CAP);
      return;
    }
  }

  // ---- rank sort by column and write the row (fused MatZeroRowsColumns)
  const int64_t gi = li + P.mt_row0;  // global K row
  const bool mrow = mask ? (mask[gi] != 0) : false;
  for (int e = tid; e < nK; e += NT) {
    const int32_t key = ckey[e];
    int rank = 0;
    for (int f = 0; f < nK; f++) rank += (ckey[f] < key) ? 1 : 0;
    double v = cval[e];
    if (mask && (mrow || mask[key])) v = (mrow && key == gi) ? diag : 0.0;
    k_col[out0 + rank] = key;
    k_val[out0 + rank] = v;
  }
}

// largest |entry| of every row (wave per row): the bounds of the integer accumulation
__global__ void __launch_bounds__(256)
    k_row_absmax(const int64_t *__restrict__ rowptr, const double *__restrict__ val, int64_t nrows, double *__restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t r = wave; r < nrows; r += nwaves) {
    double m = 0.0;
    for (int64_t q = rowptr[r] + lane; q < rowptr[r + 1]; q += 64) {
      const double a = fabs(val[q]);
      m = (a > m || a != a) ? a : m;              // (a NaN surfaces)
    }
    for (int o = 32; o > 0; o >>= 1) {
      const double b = __shfl_xor(m, o);
      m = (b > m || b != b) ? b : m;
    }
    if (lane == 0) out[r] = m;
  }
}

// copies bump-allocated rows into CSR order: wave per row
__global__ void __launch_bounds__(256)
    k_ptap_reorder(const int64_t *__restrict__ rowptr, const int64_t *__restrict__ tmp_off,
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

static int tg_pow2_ge(int64_t v) {
  int p = 1;
  while (p < v) p <<= 1;
  return p;
}
static int tg_lg(int v) {
  int l = 0;
  while ((1 << l) < v) l++;
  return l;
}
static int tg_env_int(const char *name, int dflt) {
  const char *s = getenv(name);
  return s ? atoi(s) : dflt;
}
// lanes per operand row: largest power of two <= 0.44 * mean row length, in [4,64]
static int tg_group_for(double avg) {
  const double target = avg * 0.44;
  int g = 4;
  while (g < 64 && g * 2 <= target) g <<= 1;
  return g;
}

static size_t tg_ptap_lds_bytes(int ts1, int ts2, bool numeric, int nt) {
  size_t b = (size_t)ts1 * 4 + (size_t)ts2 * 4 + 16 + (size_t)nt * 4;  // int32 part
  b += (size_t)nt * 8 * 2;                                             // pre_start, pre_w
  if (numeric) b += (size_t)ts1 * 8 + (size_t)ts2 * 8;
  return b;
}

// workgroup size: small tables -> many 256-thread groups per CU; big tables -> fewer, wider
// groups so that a CU still holds >= 16 waves
static int tg_ptap_threads(int ts1, int ts2, bool numeric) {
  const int forced = tg_env_int("TIGAR_PTAP_NT", 0);
  if (forced == 256 || forced == 512 || forced == 1024) return forced;
  // smallest workgroup that keeps >= 16 waves resident per CU (wider groups pay more per-row
  // synchronisation; fewer waves expose LDS / memory latency)
  for (int nt = 256; nt <= 1024; nt *= 2) {
    const size_t lds = tg_ptap_lds_bytes(ts1, ts2, numeric, nt);
    const int blocks = (int)std::min<size_t>(8, (160 * 1024) / lds);
    if (blocks * nt >= 1024 || nt == 1024) return nt;
  }
  return 1024;
}

// resident waves per CU under that choice
static int tg_ptap_waves(int ts1, int ts2) {
  const int nt = tg_ptap_threads(ts1, ts2, true);
  const size_t lds = tg_ptap_lds_bytes(ts1, ts2, true, nt);
  if (lds > 160 * 1024) return 0;
  const int blocks = (int)std::min<size_t>(2048 / nt, (160 * 1024) / lds);
  return blocks * nt / 64;
}

template <int MODE>
static void tg_ptap_launch(int nt, unsigned grid, size_t lds, const tg_ptap_args &P, int64_t *row_cnt,
                           int64_t *row_off, int32_t *k_col, double *k_val, unsigned long long *cursor,
                           int64_t capacity, const uint8_t *mask, double diag, int *status) {
  if (nt == 256)
    hipLaunchKernelGGL((k_ptap<MODE, 256>), dim3(grid), dim3(256), lds, g_tg.stream, P, row_cnt, row_off, k_col, k_val,
                       cursor, capacity, mask, diag, status, status + 1);
  else if (nt == 512)
    hipLaunchKernelGGL((k_ptap<MODE, 512>), dim3(grid), dim3(512), lds, g_tg.stream, P, row_cnt, row_off, k_col, k_val,
                       cursor, capacity, mask, diag, status, status + 1);
  else
    hipLaunchKernelGGL((k_ptap<MODE, 1024>), dim3(grid), dim3(1024), lds, g_tg.stream, P, row_cnt, row_off, k_col,
                       k_val, cursor, capacity, mask, diag, status, status + 1);
}

static void tg_fill_args(tg_ptap_args &P, tg_csr_s *a, int64_t a_row0, tg_csr_s *m, int64_t m_row0, tg_csr_s *mt,
                         int64_t mt_row0) {
  P.mt_rowptr = mt->rowptr;
  P.mt_col = mt->col;
  P.mt_val = mt->val;
  P.a_rowptr = a->rowptr;
  P.a_col = a->col;
  P.a_val = a->val;
  P.m_rowptr = m->rowptr;
  P.m_col = m->col;
  P.m_val = m->val;
  P.a_row0 = a_row0;
  P.a_nrows = a->nrows;
  P.m_row0 = m_row0;
  P.m_nrows = m->nrows;
  P.mt_row0 = mt_row0;
  P.nrows = mt->nrows;
  P.row_stride = 1;
  P.a_rowmax = nullptr;
  P.m_rowmax = nullptr;
  const char *am = getenv("TIGAR_PTAP_ACCUM");
  P.accum_mode = am && !strcmp(am, "int") ? 1 : am && !strcmp(am, "float") ? 2 : 0;
}

static int P_accum_mode_env() {
  const char *am = getenv("TIGAR_PTAP_ACCUM");
  return am && !strcmp(am, "int") ? 1 : am && !strcmp(am, "float") ? 2 : 0;
}

static int tg_status_error(int st) {
  if (st == TG_PTAP_RANGE) {
    tg_set_error("PtAP: a row block does not cover the rows referenced (slab halo too small)");
    return 3;
  }
  return 0;
}

static void tg_ptap_set_lds_limits() {
  static bool done = false;
  if (done) return;
#define TG_SETLIM(MODE, NT) \
  hipFuncSetAttribute((const void *)k_ptap<MODE, NT>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)
  TG_SETLIM(TG_MODE_PROBE, 256);
  TG_SETLIM(TG_MODE_PROBE, 512);
  TG_SETLIM(TG_MODE_PROBE, 1024);
  TG_SETLIM(TG_MODE_BUMP, 256);
  TG_SETLIM(TG_MODE_BUMP, 512);
  TG_SETLIM(TG_MODE_BUMP, 1024);
  TG_SETLIM(TG_MODE_PLACED, 256);
  TG_SETLIM(TG_MODE_PLACED, 512);
  TG_SETLIM(TG_MODE_PLACED, 1024);
#undef TG_SETLIM
  done = true;
}

// "symbolic" phase: probes a sample of rows to size the LDS hash tables and to estimate
// nnz(K); the pattern itself is produced by the (single) numeric traversal.
extern "C" int tg_ptap_symbolic(tg_csr_t a, int64_t a_row0, tg_csr_t m, int64_t m_row0, tg_csr_t mt,
                                int64_t mt_row0, tg_ptap_t *plan_out) {
  TG_REQUIRE_CANONICAL(a);
  TG_REQUIRE_CANONICAL(m);
  TG_REQUIRE_CANONICAL(mt);
  TG_REQUIRE_INIT();
  TG_REQUIRE(a && m && mt && plan_out, "null argument to tg_ptap_symbolic");
  TG_REQUIRE(mt->nrows <= m->ncols, "PtAP: M^T block has more rows than M has columns");
  tg_ptap_set_lds_limits();
  tg_ptap_s *plan = new tg_ptap_s();
  plan->nrows = mt->nrows;
  plan->ncols = m->ncols;
  plan->a_row0 = a_row0;
  plan->m_row0 = m_row0;
  plan->mt_row0 = mt_row0;
  plan->g1 = tg_env_int("TIGAR_PTAP_G1", tg_group_for(a->nrows ? (double)a->nnz / (double)a->nrows : 1.0));
  plan->g2 = tg_env_int("TIGAR_PTAP_G2", tg_group_for(m->nrows ? (double)m->nnz / (double)m->nrows : 1.0));
  int rc = 0;
  int h[3] = {0, 0, 0};
  if (plan->nrows > 0) {
    int *status = (int *)g_tg.scratch;  // [0] status, [1..2] maxima
    tg_ptap_args S;
    tg_fill_args(S, a, a_row0, m, m_row0, mt, mt_row0);
    S.g1 = plan->g1;
    S.g2 = plan->g2;
    S.lg1 = tg_lg(S.g1);
    S.lg2 = tg_lg(S.g2);
    const int64_t nsample = std::min<int64_t>(plan->nrows, 512);
    S.nrows = nsample;
    S.row_stride = std::max<int64_t>(1, plan->nrows / nsample);
    S.ts1 = 16384;
    S.ts2 = 4096;
    int64_t *cnt = nullptr;
    rc = tg_dmalloc(&cnt, plan->nrows + 1);
    if (!rc) {
      hipMemsetAsync(cnt, 0, (size_t)(plan->nrows + 1) * sizeof(int64_t), g_tg.stream);
      hipMemsetAsync(status, 0, 3 * sizeof(int), g_tg.stream);
      const size_t lds = tg_ptap_lds_bytes(S.ts1, S.ts2, false, 1024);
      tg_ptap_launch<TG_MODE_PROBE>(1024, (unsigned)(tg_cdiv(nsample, 8) * 8), lds, S, cnt, nullptr, nullptr, nullptr,
                                    nullptr, 0, nullptr, 0.0, status);
      if (hipGetLastError() != hipSuccess) {
        tg_set_error("PtAP probe launch failed");
        rc = 1;
      } else {
        hipMemcpyAsync(h, status, 3 * sizeof(int), hipMemcpyDeviceToHost, g_tg.stream);
        // mean row length over the sample = sum(cnt)/nsample
        int64_t total = 0;
        rc = tg_exclusive_scan_i64(cnt, plan->nrows, &total);
        hipStreamSynchronize(g_tg.stream);
        if (!rc) rc = tg_status_error(h[0]);
        if (!rc && h[0] != TG_PTAP_OK) {
          tg_set_error("PtAP: a K row needs more than %d / %d LDS hash slots (intermediate / result)", S.ts1, S.ts2);
          rc = 4;
        }
        plan->mean_k = (double)total / (double)nsample;
      }
      tg_dfree(cnt);
    }
    plan->max_t = h[1];
    plan->max_k = h[2];
    // table sizes: power of two >= load_inv * entries (4-key buckets tolerate load factors ~0.6)
    const double li1 = getenv("TIGAR_PTAP_LOADINV1") ? atof(getenv("TIGAR_PTAP_LOADINV1")) : 2.0;
    const double li2 = getenv("TIGAR_PTAP_LOADINV2") ? atof(getenv("TIGAR_PTAP_LOADINV2")) : 2.5;
    plan->ts2 = std::max(64, tg_pow2_ge((int64_t)(h[2] * li2) + 8));
    plan->ts1 = std::max(plan->ts2, std::max(64, tg_pow2_ge((int64_t)(h[1] * li1) + 8)));
    if (!getenv("TIGAR_PTAP_LOADINV1")) {
      auto waves = [&](int t1) { return tg_ptap_waves(t1, plan->ts2); };
      const int small = std::max(plan->ts2, std::max(64, tg_pow2_ge((int64_t)(h[1] * 1.3) + 8)));
      // a denser table costs more probing: accept it only where occupancy is the problem
      if (small < plan->ts1 && waves(plan->ts1) < 24 && waves(small) > waves(plan->ts1)) plan->ts1 = small;
    }
  }
  if (!rc && plan->nrows > 0) rc = tg_ptap_wave_plan(a, m, m_row0, mt, plan->max_k, plan->mean_k, &plan->wave);
  if (rc) {
    delete plan;
    return rc;
  }
  *plan_out = plan;
  return 0;
}

extern "C" int tg_ptap_numeric(tg_ptap_t plan, tg_csr_t a, tg_csr_t m, tg_csr_t mt, const int32_t *zero_dofs,
                               int64_t nzero, double diag, tg_csr_t *k_out) {
  TG_REQUIRE_CANONICAL(a);
  TG_REQUIRE_INIT();
  TG_REQUIRE(plan && a && m && mt && k_out, "null argument to tg_ptap_numeric");
  TG_REQUIRE(mt->nrows == plan->nrows && m->ncols == plan->ncols, "PtAP plan does not match the operands");
  tg_ptap_set_lds_limits();
  int rc = 0;
  uint8_t *mask = nullptr;
  if (nzero > 0) TG_TRY(tg_build_dof_mask(zero_dofs, nzero, plan->ncols, &mask));
  tg_csr_s *k = nullptr;
  // (TIGAR_PTAP_ACCUM=int asks for the integer grid of the workgroup kernel: the wave kernels add floating-point numbers)
  if (plan->wave.usable && plan->nrows > 0 && P_accum_mode_env() != 1) {
    // two Gustavson products, one wave per row (tg_ptap_wave.hip); 100 = declined, the workgroup kernel below takes over
    rc = tg_ptap_wave_numeric(&plan->wave, a, plan->a_row0, m, plan->m_row0, mt, plan->mt_row0, mask, diag, &k);
    if (rc != 100) {
      tg_dfree(mask);
      if (!rc) *k_out = k;
      return rc;
    }
    plan->wave.usable = false;
    rc = 0;
    k = nullptr;
  }
  int *status = (int *)g_tg.scratch;
  tg_ptap_args P;
  tg_fill_args(P, a, plan->a_row0, m, plan->m_row0, mt, plan->mt_row0);
  P.g1 = plan->g1;
  P.g2 = plan->g2;
  P.lg1 = tg_lg(P.g1);
  P.lg2 = tg_lg(P.g2);
  const unsigned grid = (unsigned)(tg_cdiv(std::max<int64_t>(plan->nrows, 1), 8) * 8);
  double *a_rowmax = nullptr, *m_rowmax = nullptr;
  if (plan->nrows > 0) {
    rc = tg_dmalloc(&a_rowmax, std::max<int64_t>(a->nrows, 1)) || tg_dmalloc(&m_rowmax, std::max<int64_t>(m->nrows, 1));
    if (rc) {
      tg_dfree(a_rowmax);
      tg_dfree(mask);
      return rc;
    }
    if (a->nrows > 0)
      hipLaunchKernelGGL(k_row_absmax, dim3((unsigned)std::min<int64_t>(tg_cdiv(a->nrows, 4), (int64_t)g_tg.num_cu * 16)), dim3(256),
                         0, g_tg.stream, a->rowptr, a->val, a->nrows, a_rowmax);
    if (m->nrows > 0)
      hipLaunchKernelGGL(k_row_absmax, dim3((unsigned)std::min<int64_t>(tg_cdiv(m->nrows, 4), (int64_t)g_tg.num_cu * 16)), dim3(256),
                         0, g_tg.stream, m->rowptr, m->val, m->nrows, m_rowmax);
  }
  P.a_rowmax = a_rowmax;
  P.m_rowmax = m_rowmax;

  if (plan->nrows == 0) {
    rc = tg_csr_alloc(0, plan->ncols, 0, &k);
    if (!rc) hipMemsetAsync(k->rowptr, 0, sizeof(int64_t), g_tg.stream);
  } else if (plan->nnz >= 0) {
    // ---- pattern known: rows are placed directly
    rc = tg_csr_alloc(plan->nrows, plan->ncols, plan->nnz, &k);
    if (!rc) {
      hipMemcpyAsync(k->rowptr, plan->rowptr, (size_t)(plan->nrows + 1) * sizeof(int64_t), hipMemcpyDeviceToDevice,
                     g_tg.stream);
      P.ts1 = plan->ts1;
      P.ts2 = plan->ts2;
      hipMemsetAsync(status, 0, 3 * sizeof(int), g_tg.stream);
      const int nt = tg_ptap_threads(P.ts1, P.ts2, true);
      const size_t lds = tg_ptap_lds_bytes(P.ts1, P.ts2, true, nt);
      tg_ptap_launch<TG_MODE_PLACED>(nt, grid, lds, P, nullptr, k->rowptr, k->col, k->val, nullptr, 0, mask, diag,
                                     status);
      int h = 0;
      hipMemcpyAsync(&h, status, sizeof(int), hipMemcpyDeviceToHost, g_tg.stream);
      if (hipStreamSynchronize(g_tg.stream) != hipSuccess || hipGetLastError() != hipSuccess) {
        tg_set_error("PtAP numeric (placed) failed to run");
        rc = 1;
      } else if (h != TG_PTAP_OK) {
        rc = tg_status_error(h);
        if (!rc) {
          tg_set_error("PtAP numeric: operands no longer match the plan's pattern (status %d)", h);
          rc = 4;
        }
      }
    }
  } else {
    // ---- first numeric pass: bump-allocate rows, scan counts, copy into CSR order
    int64_t *cnt = nullptr, *off = nullptr;
    unsigned long long *cursor = nullptr;
    int32_t *tcol = nullptr;
    double *tval = nullptr;
    int ts1 = plan->ts1, ts2 = plan->ts2;
    int64_t capacity = (int64_t)(plan->mean_k * 1.05 * (double)plan->nrows) + plan->max_k + 1024;
    rc = tg_dmalloc(&cnt, plan->nrows + 1) || tg_dmalloc(&off, plan->nrows + 1) || tg_dmalloc(&cursor, 1);
    // (tables and capacity start from the symbolic pass's SAMPLE of rows; a matrix with a few rows far longer than the rest
    //  needs several rounds: double at first, quadruple from the fourth round on)
    const int max_attempts = 20;
    for (int attempt = 0; attempt < max_attempts && !rc; attempt++) {
      rc = tg_dmalloc(&tcol, capacity + TG_CSR_PAD) || tg_dmalloc(&tval, capacity + TG_CSR_PAD);
      if (rc) break;
      P.ts1 = std::max(ts1, ts2);
      P.ts2 = ts2;
      hipMemsetAsync(status, 0, 3 * sizeof(int), g_tg.stream);
      hipMemsetAsync(cursor, 0, sizeof(unsigned long long), g_tg.stream);
      hipMemsetAsync(cnt, 0, (size_t)(plan->nrows + 1) * sizeof(int64_t), g_tg.stream);
      const int nt = tg_ptap_threads(P.ts1, P.ts2, true);
      const size_t lds = tg_ptap_lds_bytes(P.ts1, P.ts2, true, nt);
      if (lds > 160 * 1024) {
        tg_set_error("PtAP: K row too dense for the LDS tables (%d / %d slots)", P.ts1, P.ts2);
        rc = 4;
        break;
      }
      tg_ptap_launch<TG_MODE_BUMP>(nt, grid, lds, P, cnt, off, tcol, tval, cursor, capacity, mask, diag, status);
      int h = 0;
      unsigned long long used = 0;
      hipMemcpyAsync(&h, status, sizeof(int), hipMemcpyDeviceToHost, g_tg.stream);
      hipMemcpyAsync(&used, cursor, sizeof(used), hipMemcpyDeviceToHost, g_tg.stream);
      if (hipStreamSynchronize(g_tg.stream) != hipSuccess || hipGetLastError() != hipSuccess) {
        tg_set_error("PtAP numeric failed to run (LDS %zu B)", lds);
        rc = 1;
        break;
      }
      if ((rc = tg_status_error(h))) break;
      if (h == TG_PTAP_OK) break;
      // grow whatever overflowed and retry
      tg_dfree(tcol);
      tg_dfree(tval);
      tcol = nullptr;
      tval = nullptr;
      if (h == TG_PTAP_CAP) capacity = std::max<int64_t>((int64_t)used + 1024, capacity * 2);
      if (h == TG_PTAP_OVF1) ts1 *= attempt >= 3 ? 4 : 2;
      if (h == TG_PTAP_OVF2) ts2 *= attempt >= 3 ? 4 : 2;
      if (attempt == max_attempts - 1) {
        tg_set_error("PtAP numeric: could not size tables / output (status %d)", h);
        rc = 4;
      }
    }
    if (!rc) {
      int64_t nnz = 0;
      rc = tg_exclusive_scan_i64(cnt, plan->nrows, &nnz);
      if (!rc) rc = tg_csr_alloc(plan->nrows, plan->ncols, nnz, &k);
      if (!rc) {
        hipMemcpyAsync(k->rowptr, cnt, (size_t)(plan->nrows + 1) * sizeof(int64_t), hipMemcpyDeviceToDevice,
                       g_tg.stream);
        const unsigned rg = (unsigned)std::min<int64_t>(tg_cdiv(plan->nrows, 4), (int64_t)g_tg.num_cu * 16);
        hipLaunchKernelGGL(k_ptap_reorder, dim3(rg), dim3(256), 0, g_tg.stream, k->rowptr, off, tcol, tval,
                           plan->nrows, k->col, k->val);
        if (hipGetLastError() != hipSuccess) {
          tg_set_error("PtAP reorder launch failed");
          rc = 1;
        }
        // remember the pattern for later calls with the same operands' structure
        plan->rowptr = cnt;
        cnt = nullptr;
        plan->nnz = nnz;
        plan->ts1 = std::max(ts1, ts2);
        plan->ts2 = ts2;
      }
    }
    hipStreamSynchronize(g_tg.stream);
    tg_dfree(cnt);
    tg_dfree(off);
    tg_dfree(cursor);
    tg_dfree(tcol);
    tg_dfree(tval);
  }
  hipStreamSynchronize(g_tg.stream);
  tg_dfree(mask);
  tg_dfree(a_rowmax);
  tg_dfree(m_rowmax);
  if (rc) {
    if (k) tg_csr_destroy(k);
    return rc;
  }
  *k_out = k;
  return 0;
}

extern "C" int tg_ptap_destroy(tg_ptap_t plan) {
  if (!plan) return 0;
  if (g_tg.ready) hipStreamSynchronize(g_tg.stream);
  tg_dfree(plan->rowptr);
  tg_ptap_wave_plan_free(&plan->wave);
  delete plan;
  return 0;
}
