// Sliced, pattern-compressed copy of a CSR matrix for repeated products (the K p of the Krylov
// solve, tIGAr/common.py:1255-1258; PETSc KSP on an AIJ matrix in the reference).
//
// K = M^T A M of a tensor-product patch is a stencil matrix stored as general CSR.  The plain CSR
// product (tg_sparse.hip) sits on the practical HBM ceiling of the box at 12 B per entry, and the
// gather x[col] is its second limit, so the only way to a faster product is fewer bytes
// AND a cheaper gather at once.  This plan provides both, from the CSR arrays alone (nothing about
// the origin of the matrix is assumed, matrices without the structure are declined):
//
//   * rows are cut into slices of 64 consecutive rows = one wave, lane i owns row 64 s + i;
//   * the column OFFSETS col - row that occur in a slice are collected into one sorted union pattern
//     U_s; slices with the same set of row patterns share one dictionary entry (a 3-D p=3 patch with
//     17 M rows has a few hundred of them, a few hundred KB in total);
//   * the values are stored slice by slice as a dense [|U_s|][64] block (zero where a row has no entry
//     at that offset: 2 % padding at 256^3 p=3);
//   * the product is  y[r] = sum_k V_s[k][lane] * x[r + U_s[k]]:  the value load is one coalesced
//     512-byte line per k, the x load is 64 CONSECUTIVE doubles (r + const), U_s[k] comes through the
//     scalar unit, there are no column indices, no row pointers, no LDS and no reduction across lanes.
//     Each row is summed sequentially in ascending column order -- the order of PETSc's AIJ MatMult.
//
// Exactness: every entry of the matrix is located in its slice's union by binary search during the
// conversion; an entry that is not found (a hash collision when grouping rows / slices) makes the
// plan fail and the matrix keeps the CSR kernel.  A padded position multiplies a stored 0.0 with a
// (clamped, valid) x: it adds +0.0 to the row sum.
#include "tg_common.h"
#include <algorithm>
#include <memory>
#include <vector>

#ifndef TG_SELL_UNROLL
#define TG_SELL_UNROLL 8
#endif
#ifndef TG_SELL_G
#define TG_SELL_G 2            // positions per group: 2 or 4 (16 or 32 contiguous bytes per lane)
#endif

#define TG_SELL_C 64           // rows per slice
#define TG_SELL_TABLE 16384    // hash slots for slice classes
#define TG_SELL_MAXCLASS 4096  // distinct slice classes accepted
#define TG_SELL_WMAX 2048      // widest union pattern accepted
#define TG_SELL_CAT 8192       // offsets of the distinct rows of a slice that are merged in LDS (power of two)

// What depends on the PATTERN of the matrix only: the slice classes and the sizes of the value blocks.  Shared
// between the copies of matrices with the same pattern and kept in a small cache (below), so that a Newton loop or a
// sequence of solves whose matrices differ in their values only pays for the classification once.
struct tg_sell_shape {
  int64_t nrows = 0, ncols = 0, nnz = 0;   // of the matrix it was built from (the cache key)
  int64_t nslices = 0;
  int64_t padded = 0;            // doubles in val
  int64_t *slice_ptr = nullptr;  // nslices + 1 prefix of the block sizes (doubles)
  int32_t *slice_cls = nullptr;  // class id per slice
  int32_t *cls_w = nullptr;      // width per class
  int32_t *cls_off = nullptr;    // [class][TG_SELL_WMAX] sorted offsets
  int nclasses = 0;
  std::vector<int64_t> hptr;     // slice_ptr on the host
  ~tg_sell_shape() {
    if (!g_tg.ready) return;
    tg_dfree(slice_ptr);
    tg_dfree(slice_cls);
    tg_dfree(cls_w);
    tg_dfree(cls_off);
  }
};

struct tg_sell_s {
  std::shared_ptr<tg_sell_shape> shape;
  int64_t nslices = 0;
  int64_t padded = 0;
  // values: [slice][k][lane] blocks, stored in a few pieces -- idle blocks of the caching allocator's
  // pool first (tens of GB of PtAP temporaries sit there while the solve runs), one fresh allocation
  // for what is left -- so slices are addressed by pointer
  std::vector<void *> pieces;
  int64_t *slice_addr = nullptr; // nslices device addresses of the blocks
  // (of the shape)
  int32_t *slice_cls = nullptr, *cls_w = nullptr, *cls_off = nullptr;
  int nclasses = 0;
};

void tg_sell_free(tg_sell_s *s) {
  if (!s) return;
  for (void *q : s->pieces) tg_dfree(q);
  tg_dfree(s->slice_addr);
  delete s;
}

// the shapes used last (most recent first).  Allocated once and never destroyed: a destructor running at process exit
// would call into the allocator (and HIP) after their own static state may be gone; tg_shutdown empties it explicitly.
static std::vector<std::shared_ptr<tg_sell_shape>> &g_sell_shapes = *new std::vector<std::shared_ptr<tg_sell_shape>>();
#define TG_SELL_SHAPES 3

void tg_sell_cache_clear(void) { g_sell_shapes.clear(); }

__device__ __forceinline__ unsigned long long tg_sell_mix(unsigned long long z) {
  z *= 0x9E3779B97F4A7C15ull;
  z ^= z >> 32;
  z *= 0xD6E8FEB86659FD93ull;
  z ^= z >> 32;
  return z;
}

// ---- 1. one wave per row: hash of (length, offsets in order)
__global__ void __launch_bounds__(256)
    k_sell_rowhash(const int64_t *__restrict__ rowptr, const int32_t *__restrict__ col, int64_t nrows,
                   unsigned long long *__restrict__ rowhash) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t r = wave; r < nrows; r += nwaves) {
    const int64_t a = rowptr[r], b = rowptr[r + 1];
    unsigned long long h = 0;
    for (int64_t q = a + lane; q < b; q += 64)
      h += tg_sell_mix(((unsigned long long)(q - a) << 32) | (unsigned)(col[q] - (int32_t)r));
    for (int o = 32; o > 0; o >>= 1) h += __shfl_xor(h, o, 64);
    h = tg_sell_mix(h + (unsigned long long)(b - a)) | 1ull;
    if (lane == 0) rowhash[r] = h;
  }
}

// ---- 2. one wave per slice: class key = hash of the SET of row hashes of the slice
// ctl[0] = number of classes, ctl[1] = overflow
__global__ void __launch_bounds__(256)
    k_sell_slice_class(const unsigned long long *__restrict__ rowhash, int64_t nrows, int64_t nslices,
                       unsigned long long *__restrict__ keys, int *__restrict__ rep, int32_t *__restrict__ slot_of_slice,
                       int *__restrict__ ctl) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t s = wave; s < nslices; s += nwaves) {
    const int64_t r = s * TG_SELL_C + lane;
    const unsigned long long h = r < nrows ? rowhash[r] : 0ull;
    // set semantics: a hash counts once (lane is the first one holding it)
    bool first = h != 0ull;
    for (int j = 1; j < 64; j++) {
      const unsigned long long o = __shfl(h, (lane + 64 - j) & 63, 64);
      if (lane >= j && o == h) first = false;
    }
    unsigned long long k = first ? tg_sell_mix(h) : 0ull;   // commutative combination of the set
    for (int o = 32; o > 0; o >>= 1) k += __shfl_xor(k, o, 64);
    k = tg_sell_mix(k) | 1ull;
    if (lane == 0) {
      unsigned slot = (unsigned)(k >> 17) & (TG_SELL_TABLE - 1);
      int probes = 0;
      for (;;) {
        unsigned long long cur = __hip_atomic_load(&keys[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (cur == 0ull) {
          cur = atomicCAS(&keys[slot], 0ull, k);
          if (cur == 0ull) {
            if (atomicAdd(&ctl[0], 1) >= TG_SELL_MAXCLASS) atomicExch(&ctl[1], 1);
            cur = k;
          }
        }
        if (cur == k) break;
        slot = (slot + 1) & (TG_SELL_TABLE - 1);
        if (++probes >= TG_SELL_TABLE) {
          atomicExch(&ctl[1], 1);
          break;
        }
      }
      if ((int)s < __hip_atomic_load(&rep[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMin(&rep[slot], (int)s);
      slot_of_slice[s] = (int32_t)slot;
    }
  }
}

// ---- 3. dense class ids
__global__ void k_sell_class_ids(const unsigned long long *__restrict__ keys, const int *__restrict__ rep,
                                 int *__restrict__ id_of_slot, int *__restrict__ rep_of_id, int *__restrict__ counter) {
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= TG_SELL_TABLE) return;
  if (keys[s] == 0ull) {
    id_of_slot[s] = -1;
    return;
  }
  const int id = atomicAdd(counter, 1);
  id_of_slot[s] = id;
  rep_of_id[id] = rep[s];
}

// ---- 4. one workgroup per class: union of the offsets of the representative slice's distinct rows
// (concatenate in LDS, bitonic sort, unique) -> cls_off[id][0..w), cls_w[id];  ctl[1] on overflow
__global__ void __launch_bounds__(256)
    k_sell_class_union(const int64_t *__restrict__ rowptr, const int32_t *__restrict__ col, int64_t nrows,
                       const unsigned long long *__restrict__ rowhash, const int *__restrict__ rep_of_id,
                       int32_t *__restrict__ cls_off, int32_t *__restrict__ cls_w, int *__restrict__ ctl) {
  __shared__ int buf[TG_SELL_CAT];
  __shared__ unsigned long long hs[64];
  __shared__ int start[65];
  __shared__ int cnt[257];
  const int tid = threadIdx.x;
  const int id = blockIdx.x;
  const int64_t r0 = (int64_t)rep_of_id[id] * TG_SELL_C;
  if (tid < 64) hs[tid] = (r0 + tid < nrows) ? rowhash[r0 + tid] : 0ull;
  __syncthreads();
  if (tid == 0) {
    int tot = 0;
    for (int i = 0; i < 64; i++) {
      bool first = hs[i] != 0ull;
      for (int j = 0; j < i && first; j++) first = hs[j] != hs[i];
      start[i] = tot;
      if (first) tot += (int)min((int64_t)TG_SELL_CAT + 1, rowptr[r0 + i + 1] - rowptr[r0 + i]);
      else start[i] = -1 - tot;   // negative: not copied
      if (tot > TG_SELL_CAT) tot = TG_SELL_CAT + 1;
    }
    start[64] = tot;
  }
  __syncthreads();
  const int tot = start[64];
  if (tot > TG_SELL_CAT) {
    if (tid == 0) {
      atomicExch(&ctl[1], 1);
      cls_w[id] = 0;
    }
    return;
  }
  int n2 = 1;
  while (n2 < tot) n2 <<= 1;
  for (int i = tid; i < n2; i += 256) buf[i] = 0x7fffffff;
  __syncthreads();
  for (int i = 0; i < 64; i++) {
    const int st = start[i];
    if (st < 0) continue;
    const int64_t a = rowptr[r0 + i], b = rowptr[r0 + i + 1];
    for (int64_t q = a + tid; q < b; q += 256) buf[st + (int)(q - a)] = col[q] - (int32_t)(r0 + i);
  }
  __syncthreads();
  for (int k = 2; k <= n2; k <<= 1)
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int i = tid; i < n2; i += 256) {
        const int l = i ^ j;
        if (l > i) {
          const int x = buf[i], y = buf[l];
          const bool up = (i & k) == 0;
          if ((x > y) == up) {
            buf[i] = y;
            buf[l] = x;
          }
        }
      }
      __syncthreads();
    }
  // unique: thread t owns [t*chunk, (t+1)*chunk)
  const int chunk = (n2 + 255) / 256;
  const int lo = tid * chunk, hi = min(n2, lo + chunk);
  int c = 0;
  for (int i = lo; i < hi; i++)
    if (i < tot && (i == 0 || buf[i] != buf[i - 1])) c++;
  cnt[tid + 1] = c;
  if (tid == 0) cnt[0] = 0;
  __syncthreads();
  if (tid == 0)
    for (int i = 1; i <= 256; i++) cnt[i] += cnt[i - 1];
  __syncthreads();
  const int w = cnt[256];
  if (w > TG_SELL_WMAX) {
    if (tid == 0) {
      atomicExch(&ctl[1], 1);
      cls_w[id] = 0;
    }
    return;
  }
  int o = cnt[tid];
  for (int i = lo; i < hi; i++)
    if (i < tot && (i == 0 || buf[i] != buf[i - 1])) cls_off[(int64_t)id * TG_SELL_WMAX + o++] = buf[i];
  if (tid == 0) cls_w[id] = w;
}

// ---- 5. class id and padded size of every slice
__global__ void k_sell_slice_sizes(const int32_t *__restrict__ slot_of_slice, const int *__restrict__ id_of_slot,
                                   const int32_t *__restrict__ cls_w, int64_t nslices, int32_t *__restrict__ slice_cls,
                                   int64_t *__restrict__ slice_ptr) {
  const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= nslices) return;
  const int id = id_of_slot[slot_of_slice[s]];
  slice_cls[s] = id;
  // positions come in groups of TG_SELL_G (16-byte loads per lane), the last group padded with zeros
  slice_ptr[s] = (int64_t)((cls_w[id] + TG_SELL_G - 1) / TG_SELL_G) * TG_SELL_G * TG_SELL_C;
}

// ---- 6. conversion: one workgroup per slice.  The slice's entries are contiguous in the CSR arrays
// (coalesced reads); every entry is located in the slice's union -- position k of its row if the row
// has the full union (the common case), else by bisection of U[k..w) -- and parked in an LDS tile
// [w][R rows]; the tile then goes out as runs of R consecutive doubles (R = 8 for w = 343: 64-byte
// runs instead of scattered 8-byte writes), zeros included, so the output needs no memset.
#define TG_SELL_TILE 3072   // doubles (24 KB)
__global__ void __launch_bounds__(256)
    k_sell_convert(const int64_t *__restrict__ rowptr, const int32_t *__restrict__ col, const double *__restrict__ val,
                   int64_t nrows, int64_t nslices, const int64_t *__restrict__ slice_addr,
                   const int32_t *__restrict__ slice_cls, const int32_t *__restrict__ cls_w,
                   const int32_t *__restrict__ cls_off, int *__restrict__ fail) {
  __shared__ double tile[TG_SELL_TILE];
  __shared__ int U[TG_SELL_WMAX];
  __shared__ int rs[TG_SELL_C + 1];
  const int tid = threadIdx.x;
  const int64_t s = blockIdx.x;
  const int id = slice_cls[s];
  const int w = cls_w[id];
  const int64_t r0 = s * TG_SELL_C;
  if (w == 0) {     // a class without entries; a slice that has some does not belong to it (shape of another matrix)
    if (tid == 0 && rowptr[min(r0 + (int64_t)TG_SELL_C, nrows)] > rowptr[r0]) atomicExch(fail, 1);
    return;
  }
  for (int k = tid; k < w; k += 256) U[k] = cls_off[(int64_t)id * TG_SELL_WMAX + k];
  const int nr = (int)min((int64_t)TG_SELL_C, nrows - r0);
  const int64_t e0 = rowptr[r0];
  if (tid <= TG_SELL_C) rs[tid] = (int)(rowptr[r0 + min(tid, nr)] - e0);
  int R = TG_SELL_C;                      // rows per tile: power of two, w * R <= TG_SELL_TILE
  while (R > 1 && w * R > TG_SELL_TILE) R >>= 1;
  const int lR = 31 - __builtin_clz(R);
  double *o = reinterpret_cast<double *>(slice_addr[s]);
  bool bad = false;
  for (int g0 = 0; g0 < TG_SELL_C; g0 += R) {
    __syncthreads();                      // (U, rs staged; previous tile written out)
    for (int i = tid; i < w * R; i += 256) tile[i] = 0.0;
    __syncthreads();
    const int t0 = rs[min(g0, nr)], t1 = rs[min(g0 + R, nr)];
    int row = g0;
    // (4 entries per thread and pass: all loads of a pass are issued before the first look-up)
    for (int tb = t0 + tid; tb < t1; tb += 1024) {
      int cc[4];
      double vv[4];
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const int t = min(tb + 256 * j, t1 - 1);
        cc[j] = col[e0 + t];
        vv[j] = val[e0 + t];
      }
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const int t = tb + 256 * j;
        if (t >= t1) break;
        while (t >= rs[row + 1]) row++;
        const int k = t - rs[row];
        const int off = cc[j] - (int32_t)(r0 + row);
        int p = k;
        if (!(p < w && U[p] == off)) {
          int lo = k, hi = w;   // first position with U >= off (the k-th offset of a row is at position >= k)
          while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (U[mid] < off) lo = mid + 1;
            else hi = mid;
          }
          p = lo;
          if (!(p < w && U[p] == off)) {
            bad = true;
            continue;
          }
        }
        tile[(p << lR) + (row - g0)] = vv[j];
      }
    }
    __syncthreads();
    // layout of a slice: [group of TG_SELL_G positions][lane][TG_SELL_G] -- the product kernel loads 16 bytes per
    // lane and instruction; written here as 16-byte pieces (two positions of one lane)
    const int wg = (w + TG_SELL_G - 1) / TG_SELL_G;
    for (int i = tid; i < wg * (TG_SELL_G / 2) * R; i += 256) {
      const int rr = i & (R - 1), q = i >> lR;            // q = group * (G/2) + half
      const int grp = q / (TG_SELL_G / 2), half = q - grp * (TG_SELL_G / 2);
      const int k = grp * TG_SELL_G + 2 * half;
      double2 v2;
      v2.x = (k < w) ? tile[(k << lR) + rr] : 0.0;
      v2.y = (k + 1 < w) ? tile[((k + 1) << lR) + rr] : 0.0;
      reinterpret_cast<double2 *>(o)[((int64_t)grp * TG_SELL_C + g0 + rr) * (TG_SELL_G / 2) + half] = v2;
    }
  }
  if (bad) atomicExch(fail, 1);
}

// ---- the product
// One wave per slice.  U_s[k] is wave-uniform (read through the scalar unit), the x load of a wave is
// 64 consecutive doubles, clamped into the valid column range [cmin, cmax] of x (only padded
// positions -- stored 0.0 -- can fall outside).
__global__ void __launch_bounds__(256)
    k_spmv_sell(const int64_t *__restrict__ slice_addr, const int32_t *__restrict__ slice_cls,
                const int32_t *__restrict__ cls_w, const int32_t *__restrict__ cls_off, const double *__restrict__ x,
                double *__restrict__ y, int64_t nrows, int64_t s_begin, int64_t s_end, int cmin, int cmax,
                const double *__restrict__ gate, double gate_tol) {
  // (a solver that runs ahead of its convergence test gates the products it enqueues on the norm the device has by
  // then: with the iteration frozen, x and therefore y = K x are what they were)
  if (gate && !(*gate > gate_tol)) return;
  const int lane = threadIdx.x & 63;
  const int64_t nwb = (s_end - s_begin + 3) >> 2;              // workgroups of 4 slices (of the range [s_begin, s_end))
  const int64_t Lb = tg_xcd_block(blockIdx.x, nwb);
  const int64_t s = __builtin_amdgcn_readfirstlane((int)(s_begin + Lb * 4 + (threadIdx.x >> 6)));
  if (Lb >= nwb || s >= s_end) return;
  const int id = slice_cls[s];
  const int w = cls_w[id];
  const int32_t *__restrict__ U = cls_off + (int64_t)id * TG_SELL_WMAX;
  typedef double tg_d2 __attribute__((ext_vector_type(2)));
  constexpr int H = TG_SELL_G / 2;                          // 16-byte pieces per group and lane
  const tg_d2 *__restrict__ v = reinterpret_cast<const tg_d2 *>(slice_addr[s]) + (int64_t)lane * H;
  const int r = (int)(s * TG_SELL_C) + lane;
  double sum = 0.0;
  const int wg = (w + TG_SELL_G - 1) / TG_SELL_G;          // groups (the last one zero-padded)
  // the TG_SELL_G positions of a group are adjacent in memory per lane: 16-byte loads; rows still add up in ascending
  // column order (a padded position multiplies a stored 0.0 with a clamped x)
  constexpr int GU = TG_SELL_UNROLL / TG_SELL_G;            // groups per batch of loads
  int g = 0;
  for (; g + GU <= wg; g += GU) {
    tg_d2 vv[GU * H];
    double xx[GU * TG_SELL_G];
#pragma unroll
    for (int j = 0; j < GU; j++) {
#pragma unroll
      for (int h = 0; h < H; h++) vv[j * H + h] = __builtin_nontemporal_load(v + ((int64_t)(g + j) * TG_SELL_C) * H + h);
#pragma unroll
      for (int q = 0; q < TG_SELL_G; q++) {
        const int k = min((g + j) * TG_SELL_G + q, w - 1);
        xx[j * TG_SELL_G + q] = x[min(max(r + U[k], cmin), cmax)];
      }
    }
#pragma unroll
    for (int j = 0; j < GU * H; j++) {
      sum += vv[j].x * xx[2 * j];
      sum += vv[j].y * xx[2 * j + 1];
    }
  }
  for (; g < wg; g++) {
#pragma unroll
    for (int h = 0; h < H; h++) {
      const tg_d2 t = v[((int64_t)g * TG_SELL_C) * H + h];
      const int k0 = min(g * TG_SELL_G + 2 * h, w - 1), k1 = min(g * TG_SELL_G + 2 * h + 1, w - 1);
      sum += t.x * x[min(max(r + U[k0], cmin), cmax)];
      sum += t.y * x[min(max(r + U[k1], cmin), cmax)];
    }
  }
  if (r < nrows) y[r] = sum;
}

int64_t tg_sell_slice_rows(void) { return TG_SELL_C; }

int tg_sell_spmv(tg_csr_s *a, const double *x_shifted, int64_t cmin, int64_t cmax, double *y) {
  return tg_sell_spmv_rows(a, x_shifted, cmin, cmax, y, 0, a->nrows, nullptr, 0.0);
}

// rows [r0, r1) only; r0 a multiple of the slice height (the callers split at slice boundaries).  With `gate` (device
// pointer) the launch leaves y alone unless *gate > gate_tol.
int tg_sell_spmv_rows(tg_csr_s *a, const double *x_shifted, int64_t cmin, int64_t cmax, double *y, int64_t r0,
                      int64_t r1, const double *gate, double gate_tol) {
  tg_sell_s *S = a->sell;
  TG_REQUIRE(r0 >= 0 && r1 <= a->nrows && r0 % TG_SELL_C == 0, "tg_sell_spmv_rows: bad row range");
  if (r1 <= r0) return 0;
  const int64_t s0 = r0 / TG_SELL_C, s1 = std::min<int64_t>(tg_cdiv(r1, TG_SELL_C), S->nslices);
  const int64_t nwb = (s1 - s0 + 3) / 4;
  const unsigned grid = (unsigned)(((nwb + 7) / 8) * 8);
  hipLaunchKernelGGL(k_spmv_sell, dim3(grid), dim3(256), 0, g_tg.stream, S->slice_addr, S->slice_cls, S->cls_w,
                     S->cls_off, x_shifted, y, std::min<int64_t>(r1, a->nrows), s0, s1, (int)cmin, (int)cmax, gate, gate_tol);
  TG_LAUNCH_CHECK();
  return 0;
}

// ---- classification: the shape of the matrix' pattern, or declined
static int tg_sell_classify(tg_csr_s *a, std::shared_ptr<tg_sell_shape> *out, bool *declined_out) {
  const int64_t nrows = a->nrows, nslices = tg_cdiv(nrows, TG_SELL_C);
  int *ctl = (int *)g_tg.scratch;   // [0] classes, [1] overflow, [2] conversion failure, [3] id counter
  unsigned long long *rowhash = nullptr, *keys = nullptr;
  int *rep = nullptr, *ids = nullptr;   // id_of_slot[TABLE] | rep_of_id[MAXCLASS]
  int32_t *slot_of_slice = nullptr;
  std::shared_ptr<tg_sell_shape> S = std::make_shared<tg_sell_shape>();
  int rc = 0;
  bool declined = false;
  auto step = [&](int r) {
    if (!rc && r) rc = r;
    return rc == 0;
  };
  auto hip_ok = [&](hipError_t e) {
    if (!rc && e != hipSuccess) {
      tg_set_error("tg_sell_plan: %s", hipGetErrorString(e));
      rc = 1;
    }
    return rc == 0;
  };
  do {
    const int ctl0[4] = {0, 0, 0, 0};
    if (!hip_ok(hipMemcpyAsync(ctl, ctl0, sizeof(ctl0), hipMemcpyHostToDevice, g_tg.stream))) break;
    if (!step(tg_dmalloc(&rowhash, nrows)) || !step(tg_dmalloc(&keys, TG_SELL_TABLE)) ||
        !step(tg_dmalloc(&rep, TG_SELL_TABLE)) || !step(tg_dmalloc(&ids, TG_SELL_TABLE + TG_SELL_MAXCLASS + 1)) ||
        !step(tg_dmalloc(&slot_of_slice, nslices)))
      break;
    hipMemsetAsync(keys, 0, sizeof(unsigned long long) * TG_SELL_TABLE, g_tg.stream);
    hipMemsetAsync(rep, 0x7f, sizeof(int) * TG_SELL_TABLE, g_tg.stream);
    const unsigned wg = (unsigned)std::min<int64_t>(tg_cdiv(nrows, 4), (int64_t)g_tg.num_cu * 16);
    hipLaunchKernelGGL(k_sell_rowhash, dim3(wg), dim3(256), 0, g_tg.stream, a->rowptr, a->col, nrows, rowhash);
    const unsigned sg = (unsigned)std::min<int64_t>(tg_cdiv(nslices, 4), (int64_t)g_tg.num_cu * 16);
    hipLaunchKernelGGL(k_sell_slice_class, dim3(sg), dim3(256), 0, g_tg.stream, rowhash, nrows, nslices, keys, rep,
                       slot_of_slice, ctl);
    int h[4];
    if (!hip_ok(hipMemcpyAsync(h, ctl, sizeof(h), hipMemcpyDeviceToHost, g_tg.stream)) ||
        !hip_ok(hipStreamSynchronize(g_tg.stream)))
      break;
    const int ncls = h[0];
    if (h[1] || ncls < 1 || ncls > TG_SELL_MAXCLASS) {
      declined = true;
      break;
    }
    int *id_of_slot = ids, *rep_of_id = ids + TG_SELL_TABLE;
    if (!step(tg_dmalloc(&S->cls_w, ncls)) || !step(tg_dmalloc(&S->cls_off, (int64_t)ncls * TG_SELL_WMAX)) ||
        !step(tg_dmalloc(&S->slice_cls, nslices)) || !step(tg_dmalloc(&S->slice_ptr, nslices + 1)))
      break;
    hipLaunchKernelGGL(k_sell_class_ids, dim3(TG_SELL_TABLE / 256), dim3(256), 0, g_tg.stream, keys, rep, id_of_slot,
                       rep_of_id, ctl + 3);
    hipLaunchKernelGGL(k_sell_class_union, dim3(ncls), dim3(256), 0, g_tg.stream, a->rowptr, a->col, nrows, rowhash,
                       rep_of_id, S->cls_off, S->cls_w, ctl);
    hipLaunchKernelGGL(k_sell_slice_sizes, dim3((unsigned)tg_cdiv(nslices, 256)), dim3(256), 0, g_tg.stream,
                       slot_of_slice, id_of_slot, S->cls_w, nslices, S->slice_cls, S->slice_ptr);
    if (!hip_ok(hipGetLastError())) break;
    int64_t padded = 0;
    if (!step(tg_exclusive_scan_i64(S->slice_ptr, nslices, &padded))) break;
    if (!hip_ok(hipMemcpyAsync(h, ctl, sizeof(h), hipMemcpyDeviceToHost, g_tg.stream)) ||
        !hip_ok(hipStreamSynchronize(g_tg.stream)))
      break;
    // worth it only with moderate padding (8 B per stored position against 12 B per entry + gather)
    if (h[1] || padded > a->nnz + a->nnz / 2 + 4096) {
      declined = true;
      break;
    }
    // the prefix of the block sizes comes to the host (8 B per slice): the storage is cut into pieces with it
    S->hptr.resize((size_t)nslices + 1);
    if (!hip_ok(hipMemcpyAsync(S->hptr.data(), S->slice_ptr, sizeof(int64_t) * (size_t)(nslices + 1),
                               hipMemcpyDeviceToHost, g_tg.stream)) ||
        !hip_ok(hipStreamSynchronize(g_tg.stream)))
      break;
    S->nrows = nrows;
    S->ncols = a->ncols;
    S->nnz = a->nnz;
    S->nslices = nslices;
    S->padded = padded;
    S->nclasses = ncls;
  } while (0);
  if (g_tg.ready) hipStreamSynchronize(g_tg.stream);
  tg_dfree(rowhash);
  tg_dfree(keys);
  tg_dfree(rep);
  tg_dfree(ids);
  tg_dfree(slot_of_slice);
  *declined_out = declined;
  if (!rc && !declined) *out = S;
  return rc;
}

// ---- storage and conversion of the values for a given shape.  *mismatch: an entry of the matrix has no place in the
// shape (a shape taken from the cache that belongs to another pattern, or a hash collision of the classification)
static int tg_sell_store(tg_csr_s *a, const std::shared_ptr<tg_sell_shape> &shape, tg_sell_s **out, bool *declined_out,
                         bool *mismatch) {
  const int64_t nrows = a->nrows, nslices = shape->nslices;
  int *ctl = (int *)g_tg.scratch;
  tg_sell_s *S = new tg_sell_s;
  S->shape = shape;
  S->slice_cls = shape->slice_cls;
  S->cls_w = shape->cls_w;
  S->cls_off = shape->cls_off;
  S->nslices = nslices;
  S->padded = shape->padded;
  S->nclasses = shape->nclasses;
  const std::vector<int64_t> &hptr = shape->hptr;
  int rc = 0;
  bool declined = false;
  *mismatch = false;
  auto step = [&](int r) {
    if (!rc && r) rc = r;
    return rc == 0;
  };
  auto hip_ok = [&](hipError_t e) {
    if (!rc && e != hipSuccess) {
      tg_set_error("tg_sell_plan: %s", hipGetErrorString(e));
      rc = 1;
    }
    return rc == 0;
  };
  do {
    // ---- storage: pieces at slice granularity, cut where a block from the pool ends
    std::vector<int64_t> first, base;      // first slice and device address of every piece
    {
      static const bool use_pool = !(getenv("TIGAR_SELL_POOL") && atoi(getenv("TIGAR_SELL_POOL")) == 0);
      int64_t s0 = 0;
      bool failed = false;
      while (s0 < nslices && !failed) {
        const int64_t left = (hptr[(size_t)nslices] - hptr[(size_t)s0]) * 8;
        void *blk = nullptr;
        size_t got = 0;
        // pool blocks of at least 1 GiB (or everything that is left); else one fresh allocation
        if (!use_pool || tg_pool_take_largest((size_t)std::min<int64_t>(left, (int64_t)1 << 30), &blk, &got)) {
          if (tg_dmalloc_bytes(&blk, (size_t)std::max<int64_t>(left, 8))) {
            failed = true;
            break;
          }
          got = (size_t)std::max<int64_t>(left, 8);
        }
        S->pieces.push_back(blk);
        // slices [s0, s1) fit into this block
        const int64_t cap = hptr[(size_t)s0] + (int64_t)(got / 8);
        int64_t s1 = (int64_t)(std::upper_bound(hptr.begin() + s0, hptr.end(), cap) - hptr.begin()) - 1;
        if (s1 <= s0) {
          // the block does not even hold one slice: give it back and allocate the rest freshly
          S->pieces.pop_back();
          tg_dfree(blk);
          if (tg_dmalloc_bytes(&blk, (size_t)left)) {
            failed = true;
            break;
          }
          S->pieces.push_back(blk);
          s1 = nslices;
        }
        first.push_back(s0);
        base.push_back((int64_t)(uintptr_t)blk - hptr[(size_t)s0] * 8);   // address of "offset 0" for this piece
        s0 = s1;
      }
      if (failed) {
        // no room for the copy: not an error, the CSR kernel stays
        rc = 0;
        declined = true;
        break;
      }
    }
    if (!step(tg_dmalloc(&S->slice_addr, nslices))) break;
    {
      // slice_addr[s] = base[piece(s)] + 8 * slice_ptr[s]
      std::vector<int64_t> haddr((size_t)nslices);
      size_t pc = 0;
      for (int64_t sl = 0; sl < nslices; sl++) {
        while (pc + 1 < first.size() && first[pc + 1] <= sl) pc++;
        haddr[(size_t)sl] = base[pc] + hptr[(size_t)sl] * 8;
      }
      if (!hip_ok(hipMemcpyAsync(S->slice_addr, haddr.data(), sizeof(int64_t) * (size_t)nslices, hipMemcpyHostToDevice,
                                 g_tg.stream)) ||
          !hip_ok(hipStreamSynchronize(g_tg.stream)))
        break;
    }
    int h[4] = {0, 0, 0, 0};
    if (!hip_ok(hipMemcpyAsync(ctl, h, sizeof(h), hipMemcpyHostToDevice, g_tg.stream))) break;
    hipLaunchKernelGGL(k_sell_convert, dim3((unsigned)nslices), dim3(256), 0, g_tg.stream, a->rowptr, a->col,
                       a->val, nrows, nslices, S->slice_addr, S->slice_cls, S->cls_w, S->cls_off, ctl + 2);
    if (!hip_ok(hipGetLastError())) break;
    if (!hip_ok(hipMemcpyAsync(h, ctl, sizeof(h), hipMemcpyDeviceToHost, g_tg.stream)) ||
        !hip_ok(hipStreamSynchronize(g_tg.stream)))
      break;
    if (h[2]) {
      *mismatch = true;   // an entry was not found in its slice's union
      declined = true;
      break;
    }
  } while (0);
  if (g_tg.ready) hipStreamSynchronize(g_tg.stream);
  *declined_out = declined;
  if (rc || declined) {
    tg_sell_free(S);
    return rc;
  }
  *out = S;
  return 0;
}

// Builds the plan if the matrix has the structure; a->sell_state: 1 = in use, -1 = declined.
int tg_sell_plan(tg_csr_s *a) {
  if (a->sell_state) return 0;
  TG_REQUIRE_CANONICAL(a);
  a->sell_state = -1;
  static int enabled = getenv("TIGAR_SPMV_SELL") ? atoi(getenv("TIGAR_SPMV_SELL")) : 1;
  const int cache_on = getenv("TIGAR_SELL_CACHE") ? atoi(getenv("TIGAR_SELL_CACHE")) : 1;
  if (!enabled || a->nrows < 1 || a->nnz < 1 || a->nrows >= 0x7fffffffll - 64 || a->ncols >= 0x7fffffffll) return 0;
  tg_sell_s *S = nullptr;
  bool declined = false, mismatch = false;
  // a shape of the same size from an earlier matrix: every entry is located in it during the conversion, so a shape
  // that belongs to another pattern is found out there (and forgotten), never used
  if (cache_on) {
    for (size_t i = 0; i < g_sell_shapes.size() && !S; i++) {
      std::shared_ptr<tg_sell_shape> sh = g_sell_shapes[i];
      if (sh->nrows != a->nrows || sh->ncols != a->ncols || sh->nnz != a->nnz) continue;
      TG_TRY(tg_sell_store(a, sh, &S, &declined, &mismatch));
      if (mismatch) {
        g_sell_shapes.erase(g_sell_shapes.begin() + (long)i);
        i--;
        continue;
      }
      if (declined) return 0;        // (no room for the values)
      g_sell_shapes.erase(g_sell_shapes.begin() + (long)i);
      g_sell_shapes.insert(g_sell_shapes.begin(), sh);
      g_tg.prof_n[TG_PROF_SELL_SHAPE_REUSED] += 1;
    }
  }
  if (!S) {
    std::shared_ptr<tg_sell_shape> sh;
    TG_TRY(tg_sell_classify(a, &sh, &declined));
    if (declined) return 0;
    TG_TRY(tg_sell_store(a, sh, &S, &declined, &mismatch));
    if (declined) return 0;
    if (cache_on) {
      g_sell_shapes.insert(g_sell_shapes.begin(), sh);
      if (g_sell_shapes.size() > TG_SELL_SHAPES) g_sell_shapes.pop_back();
    }
  }
  a->sell = S;
  a->sell_state = 1;
  return 0;
}

void tg_sell_drop(tg_csr_s *a) {
  if (g_tg.ready) hipStreamSynchronize(g_tg.stream);
  tg_sell_free(a->sell);
  a->sell = nullptr;
}

extern "C" int tg_spmv_sell(tg_csr_t a, int enable, int *nclasses, int64_t *padded) {
  TG_REQUIRE_INIT();
  TG_REQUIRE(a, "null argument to tg_spmv_sell");
  if (enable) {
    if (a->sell_state < 0) a->sell_state = 0;   // explicit request: try (again)
    TG_TRY(tg_sell_plan(a));
  } else {
    tg_sell_drop(a);
    a->sell_state = -1;
  }
  if (nclasses) *nclasses = a->sell_state == 1 ? a->sell->nclasses : 0;
  if (padded) *padded = a->sell_state == 1 ? a->sell->padded : 0;
  return 0;
}
