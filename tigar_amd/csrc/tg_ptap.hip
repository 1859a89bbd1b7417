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
// symbolic pass = same traversal on indices only (row counts -> rowptr, table sizes).
// Inputs are row blocks (z-slabs) with GLOBAL column indices: local row = global - row0.
#include "tg_common.h"
#include <hip/hip_runtime.h>
#include <algorithm>

int tg_build_dof_mask(const int32_t *dofs, int64_t n, int64_t ndofs_total, uint8_t **mask_out);

struct tg_ptap_s {
  int64_t nrows = 0;        // K rows in this block
  int64_t ncols = 0;        // = M.ncols
  int64_t nnz = 0;
  int64_t *rowptr = nullptr;  // device, nrows+1
  int64_t a_row0 = 0, m_row0 = 0, mt_row0 = 0;
  int ts1 = 0, ts2 = 0, g1 = 0, g2 = 0;
  int max_t = 0, max_k = 0;
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
  int64_t row_stride;  // symbolic sampling: row = idx * row_stride
  int ts1, ts2, lg1, lg2, g1, g2;
};

enum { TG_PTAP_OK = 0, TG_PTAP_OVF1 = 1, TG_PTAP_OVF2 = 2, TG_PTAP_RANGE = 3 };

__device__ __forceinline__ unsigned tg_hash(int32_t key, int lg) {
  return ((unsigned)key * 2654435761u) >> (32 - lg);
}

// returns false on table overflow
template <bool NUMERIC>
__device__ __forceinline__ bool tg_hash_insert(int32_t *keys, double *vals, int ts, int lg, int32_t key, double v) {
  unsigned slot = tg_hash(key, lg);
  for (int probe = 0; probe < ts; probe++) {
    int32_t cur = keys[slot];
    if (cur != key) {
      if (cur != -1) {
        slot = (slot + 1) & (ts - 1);
        continue;
      }
      cur = atomicCAS(&keys[slot], -1, key);
      if (cur != -1 && cur != key) {
        slot = (slot + 1) & (ts - 1);
        continue;
      }
    }
    if (NUMERIC) unsafeAtomicAdd(&vals[slot], v);
    return true;
  }
  return false;
}

// LDS carve (dynamic, 16-byte aligned offsets):
//   keys1[ts1] int32 | list1[ts1] int32 | keys2[ts2] int32 | cnt[4] int32 | vals1[ts1] f64 | vals2[ts2] f64
template <bool NUMERIC>
__global__ void __launch_bounds__(256)
    k_ptap(tg_ptap_args P, int64_t *__restrict__ k_rowptr_or_cnt, int32_t *__restrict__ k_col,
           double *__restrict__ k_val, const uint8_t *__restrict__ mask, double diag, int *__restrict__ status,
           int *__restrict__ maxima) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  int32_t *keys1 = reinterpret_cast<int32_t *>(smem);
  int32_t *list1 = keys1 + P.ts1;
  int32_t *keys2 = list1 + P.ts1;
  int32_t *cnt = keys2 + P.ts2;
  double *vals1 = reinterpret_cast<double *>(cnt + 4);
  double *vals2 = vals1 + (NUMERIC ? P.ts1 : 0);

  const int tid = threadIdx.x;
  const int64_t nb = (int64_t)gridDim.x;
  const int64_t L = tg_xcd_block(blockIdx.x, P.nrows);
  (void)nb;
  if (L >= P.nrows) return;
  const int64_t li = L * P.row_stride;  // local K / M^T row

  for (int s = tid; s < P.ts1; s += 256) {
    keys1[s] = -1;
    if (NUMERIC) vals1[s] = 0.0;
  }
  for (int s = tid; s < P.ts2; s += 256) {
    keys2[s] = -1;
    if (NUMERIC) vals2[s] = 0.0;
  }
  if (tid < 4) cnt[tid] = 0;
  __syncthreads();

  // ---- stage 1: T = (row i of M^T) * A
  {
    const int64_t e0 = P.mt_rowptr[li], e1 = P.mt_rowptr[li + 1];
    const int sub = tid & (P.g1 - 1);
    const int grp = tid >> P.lg1;
    const int ngrp = 256 >> P.lg1;
    bool ovf = false, range = false;
    for (int64_t e = e0 + grp; e < e1; e += ngrp) {
      const int64_t ra = (int64_t)P.mt_col[e] - P.a_row0;
      if (ra < 0 || ra >= P.a_nrows) {
        range = true;
        continue;
      }
      const double w = NUMERIC ? P.mt_val[e] : 0.0;
      const int64_t q1 = P.a_rowptr[ra + 1];
      for (int64_t q = P.a_rowptr[ra] + sub; q < q1; q += P.g1) {
        const double v = NUMERIC ? w * P.a_val[q] : 0.0;
        if (!tg_hash_insert<NUMERIC>(keys1, vals1, P.ts1, 32 - __clz(P.ts1 - 1), P.a_col[q], v)) ovf = true;
      }
    }
    if (ovf) atomicMax(status, TG_PTAP_OVF1);
    if (range) atomicMax(status, TG_PTAP_RANGE);
  }
  __syncthreads();

  // ---- compact occupied slots of table 1
  for (int s0 = 0; s0 < P.ts1; s0 += 256) {
    const int s = s0 + tid;
    const bool occ = (s < P.ts1) && keys1[s] != -1;
    const unsigned long long m = __ballot(occ);
    int base = 0;
    if ((tid & 63) == 0 && m) base = atomicAdd(&cnt[0], __popcll(m));
    base = __shfl(base, 0, 64);
    if (occ) {
      const unsigned long long below = ((tid & 63) == 0) ? 0ull : (~0ull >> (64 - (tid & 63)));
      list1[base + __popcll(m & below)] = s;
    }
  }
  __syncthreads();
  const int nT = cnt[0];

  // ---- stage 2: K row = T * M
  {
    const int sub = tid & (P.g2 - 1);
    const int grp = tid >> P.lg2;
    const int ngrp = 256 >> P.lg2;
    bool ovf = false, range = false;
    const int lgts2 = 32 - __clz(P.ts2 - 1);
    for (int e = grp; e < nT; e += ngrp) {
      const int slot = list1[e];
      const int64_t sm = (int64_t)keys1[slot] - P.m_row0;
      if (sm < 0 || sm >= P.m_nrows) {
        range = true;
        continue;
      }
      const double w = NUMERIC ? vals1[slot] : 0.0;
      const int64_t q1 = P.m_rowptr[sm + 1];
      for (int64_t q = P.m_rowptr[sm] + sub; q < q1; q += P.g2) {
        const double v = NUMERIC ? w * P.m_val[q] : 0.0;
        if (!tg_hash_insert<NUMERIC>(keys2, vals2, P.ts2, lgts2, P.m_col[q], v)) ovf = true;
      }
    }
    if (ovf) atomicMax(status, TG_PTAP_OVF2);
    if (range) atomicMax(status, TG_PTAP_RANGE);
  }
  __syncthreads();

  // ---- compact table 2 into (ckey, cval) living in table-1 storage (no longer needed)
  int32_t *ckey = keys1;   // capacity ts1 >= ts2 (host guarantees)
  double *cval = vals1;
  for (int s0 = 0; s0 < P.ts2; s0 += 256) {
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
      cval[pos] = vals2[s];
    }
  }
  __syncthreads();
  const int nK = cnt[1];
  if (!NUMERIC) {
    if (tid == 0) {
      k_rowptr_or_cnt[li] = nK;
      atomicMax(&maxima[0], nT);
      atomicMax(&maxima[1], nK);
    }
    return;
  }

  // ---- rank sort by column and write the CSR row (fused MatZeroRowsColumns)
  const int64_t out0 = k_rowptr_or_cnt[li];
  const int64_t gi = li + P.mt_row0;  // global K row
  const bool mrow = mask ? (mask[gi] != 0) : false;
  for (int e = tid; e < nK; e += 256) {
    const int32_t key = ckey[e];
    int rank = 0;
    for (int f = 0; f < nK; f++) rank += (ckey[f] < key) ? 1 : 0;
    double v = cval[e];
    if (mask && (mrow || mask[key])) v = (mrow && key == gi) ? diag : 0.0;
    k_col[out0 + rank] = key;
    k_val[out0 + rank] = v;
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
static int tg_group_for(double avg) {
  if (avg >= 48) return 64;
  if (avg >= 24) return 32;
  if (avg >= 12) return 16;
  return 8;
}

static size_t tg_ptap_lds_bytes(int ts1, int ts2, bool numeric) {
  size_t b = (size_t)ts1 * 4 * 2 + (size_t)ts2 * 4 + 16;
  if (numeric) b += (size_t)ts1 * 8 + (size_t)ts2 * 8;
  return b;
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
}

static int tg_status_error(int st) {
  if (st == TG_PTAP_RANGE) {
    tg_set_error("PtAP: a row block does not cover the rows referenced (slab halo too small)");
    return 3;
  }
  return 0;
}

extern "C" int tg_ptap_symbolic(tg_csr_t a, int64_t a_row0, tg_csr_t m, int64_t m_row0, tg_csr_t mt,
                                int64_t mt_row0, tg_ptap_t *plan_out) {
  TG_REQUIRE_INIT();
  TG_REQUIRE(a && m && mt && plan_out, "null argument to tg_ptap_symbolic");
  TG_REQUIRE(mt->nrows <= m->ncols, "PtAP: M^T block has more rows than M has columns");
  tg_ptap_s *plan = new tg_ptap_s();
  plan->nrows = mt->nrows;
  plan->ncols = m->ncols;
  plan->a_row0 = a_row0;
  plan->m_row0 = m_row0;
  plan->mt_row0 = mt_row0;
  plan->g1 = tg_group_for(a->nrows ? (double)a->nnz / (double)a->nrows : 1.0);
  plan->g2 = tg_group_for(m->nrows ? (double)m->nnz / (double)m->nrows : 1.0);
  if (tg_dmalloc(&plan->rowptr, plan->nrows + 1)) {
    delete plan;
    return 1;
  }
  hipMemsetAsync(plan->rowptr, 0, (size_t)(plan->nrows + 1) * sizeof(int64_t), g_tg.stream);
  int *status = (int *)g_tg.scratch;  // [0] status, [1..2] maxima
  int rc = 0;
  tg_ptap_args P;
  tg_fill_args(P, a, a_row0, m, m_row0, mt, mt_row0);
  P.g1 = plan->g1;
  P.g2 = plan->g2;
  P.lg1 = tg_lg(P.g1);
  P.lg2 = tg_lg(P.g2);
  int h[3] = {0, 0, 0};
  if (plan->nrows > 0) {
    // probe a sample of rows with large tables to size the real ones
    int ts1 = 16384, ts2 = 4096;
    const int64_t nsample = std::min<int64_t>(plan->nrows, 256);
    int64_t *scratch_cnt = nullptr;
    rc = tg_dmalloc(&scratch_cnt, plan->nrows + 1);
    for (int attempt = 0; attempt < 2 && !rc; attempt++) {
      hipMemsetAsync(status, 0, 3 * sizeof(int), g_tg.stream);
      tg_ptap_args S = P;
      S.nrows = nsample;
      S.row_stride = std::max<int64_t>(1, plan->nrows / nsample);
      S.ts1 = ts1;
      S.ts2 = ts2;
      const size_t lds = tg_ptap_lds_bytes(ts1, ts2, false);
      hipFuncSetAttribute((const void *)k_ptap<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      hipLaunchKernelGGL((k_ptap<false>), dim3((unsigned)(tg_cdiv(nsample, 8) * 8)), dim3(256), lds, g_tg.stream, S,
                         scratch_cnt, (int32_t *)nullptr, (double *)nullptr, (const uint8_t *)nullptr, 0.0, status,
                         status + 1);
      if (hipGetLastError() != hipSuccess) {
        tg_set_error("PtAP probe launch failed");
        rc = 1;
        break;
      }
      hipMemcpyAsync(h, status, 3 * sizeof(int), hipMemcpyDeviceToHost, g_tg.stream);
      hipStreamSynchronize(g_tg.stream);
      if ((rc = tg_status_error(h[0]))) break;
      if (h[0] == TG_PTAP_OK) break;
      tg_set_error("PtAP: a K row needs more than %d / %d LDS hash slots (intermediate / result)", ts1, ts2);
      rc = 4;
    }
    hipFree(scratch_cnt);
    // full pass, growing the tables if a non-sampled row overflows
    int cur1 = std::max(64, tg_pow2_ge((int64_t)h[1] * 3 / 2 + 8));
    int cur2 = std::max(64, tg_pow2_ge((int64_t)h[2] * 3 / 2 + 8));
    while (!rc) {
      if (cur1 < cur2) cur1 = cur2;
      hipMemsetAsync(status, 0, 3 * sizeof(int), g_tg.stream);
      P.ts1 = cur1;
      P.ts2 = cur2;
      const size_t lds = tg_ptap_lds_bytes(cur1, cur2, false);
      hipLaunchKernelGGL((k_ptap<false>), dim3((unsigned)(tg_cdiv(plan->nrows, 8) * 8)), dim3(256), lds, g_tg.stream, P,
                         plan->rowptr, (int32_t *)nullptr, (double *)nullptr, (const uint8_t *)nullptr, 0.0, status,
                         status + 1);
      if (hipGetLastError() != hipSuccess) {
        tg_set_error("PtAP symbolic launch failed (LDS %zu B)", lds);
        rc = 1;
        break;
      }
      hipMemcpyAsync(h, status, 3 * sizeof(int), hipMemcpyDeviceToHost, g_tg.stream);
      hipStreamSynchronize(g_tg.stream);
      if ((rc = tg_status_error(h[0]))) break;
      if (h[0] == TG_PTAP_OK) break;
      if (h[0] == TG_PTAP_OVF1) cur1 *= 2;
      if (h[0] == TG_PTAP_OVF2) cur2 *= 2;
      if (tg_ptap_lds_bytes(std::max(cur1, cur2), cur2, true) > 160 * 1024) {
        tg_set_error("PtAP: K row too dense for the LDS tables (%d / %d slots)", cur1, cur2);
        rc = 4;
      }
    }
    plan->max_t = h[1];
    plan->max_k = h[2];
    plan->ts1 = cur1;
    plan->ts2 = cur2;
  }
  if (!rc) rc = tg_exclusive_scan_i64(plan->rowptr, plan->nrows, &plan->nnz);
  if (rc) {
    hipFree(plan->rowptr);
    delete plan;
    return rc;
  }
  *plan_out = plan;
  return 0;
}

extern "C" int tg_ptap_numeric(tg_ptap_t plan, tg_csr_t a, tg_csr_t m, tg_csr_t mt, const int32_t *zero_dofs,
                               int64_t nzero, double diag, tg_csr_t *k_out) {
  TG_REQUIRE_INIT();
  TG_REQUIRE(plan && a && m && mt && k_out, "null argument to tg_ptap_numeric");
  TG_REQUIRE(mt->nrows == plan->nrows && m->ncols == plan->ncols, "PtAP plan does not match the operands");
  tg_csr_s *k = nullptr;
  TG_TRY(tg_csr_alloc(plan->nrows, plan->ncols, plan->nnz, &k));
  TG_CHECK_HIP(hipMemcpyAsync(k->rowptr, plan->rowptr, (size_t)(plan->nrows + 1) * sizeof(int64_t),
                              hipMemcpyDeviceToDevice, g_tg.stream));
  int rc = 0;
  uint8_t *mask = nullptr;
  if (nzero > 0) rc = tg_build_dof_mask(zero_dofs, nzero, plan->ncols, &mask);
  if (!rc && plan->nrows > 0) {
    tg_ptap_args P;
    tg_fill_args(P, a, plan->a_row0, m, plan->m_row0, mt, plan->mt_row0);
    P.g1 = plan->g1;
    P.g2 = plan->g2;
    P.lg1 = tg_lg(P.g1);
    P.lg2 = tg_lg(P.g2);
    P.ts1 = std::max(plan->ts1, plan->ts2);
    P.ts2 = plan->ts2;
    int *status = (int *)g_tg.scratch;
    hipMemsetAsync(status, 0, 3 * sizeof(int), g_tg.stream);
    const size_t lds = tg_ptap_lds_bytes(P.ts1, P.ts2, true);
    hipFuncSetAttribute((const void *)k_ptap<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipLaunchKernelGGL((k_ptap<true>), dim3((unsigned)(tg_cdiv(plan->nrows, 8) * 8)), dim3(256), lds, g_tg.stream, P,
                       k->rowptr, k->col, k->val, (const uint8_t *)mask, diag, status, status + 1);
    if (hipGetLastError() != hipSuccess) {
      tg_set_error("PtAP numeric launch failed (LDS %zu B)", lds);
      rc = 1;
    } else {
      int h = 0;
      hipMemcpyAsync(&h, status, sizeof(int), hipMemcpyDeviceToHost, g_tg.stream);
      hipStreamSynchronize(g_tg.stream);
      if (h != TG_PTAP_OK) {
        rc = tg_status_error(h);
        if (!rc) {
          tg_set_error("PtAP numeric: hash table overflow (%d); plan is stale for these operands", h);
          rc = 4;
        }
      }
    }
  }
  hipStreamSynchronize(g_tg.stream);
  hipFree(mask);
  if (rc) {
    tg_csr_destroy(k);
    return rc;
  }
  *k_out = k;
  return 0;
}

extern "C" int tg_ptap_destroy(tg_ptap_t plan) {
  if (!plan) return 0;
  if (g_tg.ready) hipStreamSynchronize(g_tg.stream);
  hipFree(plan->rowptr);
  delete plan;
  return 0;
}
