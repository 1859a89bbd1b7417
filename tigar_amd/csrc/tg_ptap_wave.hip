// extractMatrix for arbitrary sparse operands, second generation (round 4): K = M^T (A M) as TWO row-wise Gustavson
// products (PETSc's MatPtAP does the same two-step product [ext]; tIGAr/common.py:1194-1195), one WAVE per output row.
//
//   stage 1:  (A M)_r = sum_s  A[r,s]    * M[s,:]        r = FE rows of the block        (intermediate, in HBM)
//   stage 2:  K_i     = sum_r  M^T[i,r]  * (A M)[r,:]    i = dofs of the block
//
// Why not the fused workgroup-per-row kernel of tg_ptap.hip (round 1-3): its first table holds a row of M^T A -- (3p+1)^d
// keys, 50-100 KB of LDS -- so a CU holds one workgroup, the barriers between the staging and the hashing phases and the
// dependent chain global load -> hash -> LDS read -> LDS atomic are exposed (SQ counters at 48^3 p=3: 70 % of the wave
// cycles waiting, vector ALU 23 % busy, 66 vector instructions per 64 accumulations), and since the waves of a workgroup add
// into one table in an order that differs from run to run it needs integer accumulation to be reproducible.  Here a row
// of either product is small (<= a few hundred keys): the table is PRIVATE to a wave (3-12 KB), a CU holds 13-32 waves
// that never synchronise, the LDS operations of one wave execute in program order and the lanes of one instruction carry
// distinct keys, so plain floating-point ds_add_f64 gives the same bits on every run -- no integer grid, no scale rule.
// One templated row routine serves both stages (outer row -> operand rows -> hash accumulate).
//
// Lanes: groups of G = 2^LG lanes walk one operand row (G chosen from the mean operand row length), U operand rows are
// in flight per group before the table is touched.  The intermediate is written in the loose-row form (space reserved with
// one atomic per row, (start, count) per row); K's rows are rank-sorted, MatZeroRowsColumns applied on the way, and go to
// their CSR place directly when the plan knows the row pointer (second call with the same pattern) or through a
// reserve-scan-reorder pass otherwise.
#include "tg_common.h"
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstring>

enum { GW_COUNT = 0, GW_BUMP = 1, GW_PLACED = 2 };
enum { GW_OK = 0, GW_OVF = 1, GW_RANGE = 3, GW_CAP = 4 };

struct gw_args {
  // outer matrix X (canonical CSR block): row x gives (operand row index, weight) pairs
  const int64_t *x_rowptr;
  const int32_t *x_col;
  const double *x_val;
  int64_t x_nrows;
  int64_t row_stride;   // probing: row = index * row_stride
  // operand matrix Y: row (x_col - y_row0) occupies y_col / y_val [y_start[r], y_start[r] + len), len = y_cnt[r] (loose
  // rows) or y_start[r+1] - y_start[r] (canonical CSR, y_cnt == nullptr)
  // The columns of Y come MIXED (gw_mix: a bijection of the 32-bit column index whose bit fields are the two slots of the
  // cuckoo tables): every entry of an operand matrix is looked up ~100 times, its hash is computed once (k_gw_mix for M,
  // the first stage writes the intermediate's columns mixed); only the rows of K turn them back (gw_unmix).
  const int64_t *y_start;
  const int32_t *y_cnt;
  const uint32_t *y_mix;
  const double *y_val;
  const int64_t *y_start_mix;   // where the row's columns start in y_mix when that differs from y_start (rows that share
                                // one column list: the element matrices of the cell-block product); nullptr = y_start
  int64_t y_row0, y_nrows;
  int ts, lgts;         // slots of a wave's table (power of two)
  int rows_per_wave;
  // order in which the rows are handed to the waves (scheduling only, the result does not depend on it): n0 > 0 = the rows
  // are the points of an n0 x n1 x (x_nrows / n0 n1) lattice, direction 0 fastest, and are visited tile by tile
  // (t0 x t1 x t2 points) so that the rows in flight on an XCD share their operand rows through its L2
  int64_t n0, n1;
  int t0, t1, t2;
  int debug;
  // where a row of the temporary goes (BUMP): out_stride > 0 = row li at li * out_stride (its length must not exceed the
  // stride), 0 = space reserved with an atomic on one counter.  One atomic per row on ONE address is a serial resource
  // of its own: 12 ns each -- 87 ms for the 7.2 M rows of A M at 64^3 elements, as long as the whole product took.
  int64_t out_stride;
};

// index in tile-major order -> lattice point -> row (tiles clipped at the lattice's far faces)
__device__ __forceinline__ int64_t gw_tile_row(const gw_args &P, int64_t idx) {
  const int64_t n0 = P.n0, n1 = P.n1, n2 = P.x_nrows / (n0 * n1);
  const int64_t slab = (int64_t)P.t2 * n0 * n1;
  const int64_t tz = idx / slab;
  int64_t rem = idx - tz * slab;
  const int64_t h = min((int64_t)P.t2, n2 - tz * P.t2);
  const int64_t strip = (int64_t)P.t1 * n0 * h;
  const int64_t ty = rem / strip;
  rem -= ty * strip;
  const int64_t w1 = min((int64_t)P.t1, n1 - ty * P.t1);
  const int64_t blk = (int64_t)P.t0 * w1 * h;
  const int64_t tx = rem / blk;
  rem -= tx * blk;
  const int64_t w0 = min((int64_t)P.t0, n0 - tx * P.t0);
  const int64_t lz = rem / (w0 * w1);
  rem -= lz * w0 * w1;
  const int64_t ly = rem / w0, lx = rem - ly * w0;
  return (tx * P.t0 + lx) + n0 * ((ty * P.t1 + ly) + n1 * (tz * P.t2 + lz));
}

// LDS of one wave: ts slots of 8 bytes, ts / 2 accumulators of 8 bytes, ts / 2 keys of 4 bytes
__host__ __device__ static inline size_t gw_wave_bytes(int ts) { return (size_t)ts * 8 + (size_t)(ts >> 1) * 12; }

__device__ __forceinline__ int64_t gw_readlane_i64(int64_t v, int l) {
  const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(unsigned long long)v, l);
  const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)((unsigned long long)v >> 32), l);
  return (int64_t)(((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ double gw_readlane_f64(double v, int l) {
  return __longlong_as_double(gw_readlane_i64(__double_as_longlong(v), l));
}

// ---- a wave's accumulator table.  Cuckoo hashing with two slots per key: a key lives at T[h1(key)] or T[half + h2(key)],
// nowhere else, so a lookup is two LDS reads and two compares -- no probe loop, whatever the load.  (Linear probing was
// measured first: at any load factor a group of 4 x 64 keys holds a displaced one, every group took the probe loop, and a
// batch cost 330 cycles per SIMD instead of the ~90 its instructions need.)  A slot holds (key, id) as one 64-bit word,
// id = position of the key's accumulator in the dense arrays vals[] / kid[]: evictions move the pair with one ds_wrxchg_rtn_b64
// and the values never move; the finished row is kid[0..n) / vals[0..n) without a scan of the table.
struct gw_tab {
  unsigned long long *T;   // ts slots: [0, half) first choice, [half, ts) second choice; all ones = empty
  double *vals;            // cap accumulators
  uint32_t *kid;           // cap keys (mixed, before any re-seeding), in order of first touch
  int half, lgh, cap;
  unsigned seed;
};
#define GW_EMPTY 0xFFFFFFFFFFFFFFFFull
// operand rows in flight per lane group before the table is touched.  The waves are bound by the latency of their operand
// loads (2 us per group under load), and the tables limit the waves per SIMD (5 with the tables of A M, 2-3 with those of
// K at p = 3): the loads in flight have to come from the depth of a wave's own queue.
#ifndef GW_U1
#define GW_U1 4
#endif
#ifndef GW_U2
#define GW_U2 8
#endif
#define GW_U(FINAL) ((FINAL) ? GW_U2 : GW_U1)
#define GW_MAXIT 48

// Both slots of a key come out of ONE mixed word (two 32-bit multiplies with an xor-shift in between, murmur's finaliser).
// A multiplicative hash alone will not do: the columns of a row are lattice points c0 + dx + n0 dy + n0 n1 dz, a linear hash
// collides as a function of the DIFFERENCE of two keys, so a difference that collides in both tables puts whole chains of
// keys onto the same two slots -- in every row of the matrix alike (measured: 512 slots fail for the 125 columns of a row of
// A M at p = 3, 1024 do not).  The mix is a bijection (odd multipliers, xor-shifts): the tables store and compare the mixed
// word itself, gw_unmix gives the column back.
__host__ __device__ __forceinline__ unsigned gw_mix(unsigned key) {
  unsigned m = key * 0x9E3779B1u;
  m ^= m >> 15;
  m *= 0x85EBCA77u;
  m ^= m >> 13;
  return m;
}
__host__ __device__ __forceinline__ unsigned gw_unmix(unsigned m) {
  m ^= m >> 13;
  m ^= m >> 26;
  m *= 0xB6C92F47u;        // inverse of 0x85EBCA77 mod 2^32
  m ^= m >> 15;
  m ^= m >> 30;
  return m * 0x0E8B2F51u;  // inverse of 0x9E3779B1
}
// a row that does not settle with the plain mix is started over with the keys permuted once more (rare: uniform branch)
template <bool SEEDED>
__device__ __forceinline__ unsigned gw_reseed(unsigned m, unsigned seed) {
  if (!SEEDED) return m;
  m = (m ^ seed) * 0xC2B2AE35u;
  return m ^ (m >> 16);
}
__device__ __forceinline__ unsigned gw_s1(unsigned m, int lgh) { return m >> (32 - lgh); }
__device__ __forceinline__ unsigned gw_s2(unsigned m, int lgh, int half) { return (unsigned)half + ((m >> (32 - 2 * lgh)) & (unsigned)(half - 1)); }
// (an empty slot is all ones: if it matches a key that happens to be 0xFFFFFFFF the id comes out negative = not found)
__device__ __forceinline__ int gw_find(unsigned long long e1, unsigned long long e2, unsigned key) {
  return ((unsigned)e1 == key) ? (int)(e1 >> 32) : ((unsigned)e2 == key) ? (int)(e2 >> 32) : -1;
}

// the lanes with `pend` carry keys that the group's fast round did not find: look again (an earlier batch of the group
// may have brought the key in), insert what is still missing, add.  Lanes of one lane group carry distinct keys; with
// several groups per instruction (LG < 6) the groups take turns.  Returns false when the table is full / a chain does
// not end (the host retries with a larger table).
template <int LG, bool SEEDED>
__device__ __forceinline__ bool gw_insert_new(const gw_tab &t, int &n, unsigned m0, double v, bool pend) {
  constexpr int NG = 64 >> LG;
  const int lane = threadIdx.x & 63, grp = lane >> LG;
  const unsigned key = gw_reseed<SEEDED>(m0, t.seed);
  const unsigned s1 = gw_s1(key, t.lgh), s2 = gw_s2(key, t.lgh, t.half);
  for (int g = 0; g < NG; g++) {
    const bool mine = pend && (NG == 1 || grp == g);
    if (!__any(mine)) continue;
    int id = -1;
    if (mine) id = gw_find(t.T[s1], t.T[s2], key);
    const bool need = mine && id < 0;
    const unsigned long long m = __ballot(need);
    if (need) id = n + __popcll(m & ((1ull << lane) - 1ull));
    n += __popcll(m);
    if (n > t.cap) return false;                     // wave-uniform
    bool active = need;
    unsigned long long cur = ((unsigned long long)(unsigned)id << 32) | key;
    unsigned slot = s1;
    if (need) t.kid[id] = m0;                         // (vals[id] is zero since the row was started)
    for (int it = 0; it < GW_MAXIT && __any(active); it++) {
      if (active) {
        const unsigned long long old = atomicExch(&t.T[slot], cur);
        if (old == GW_EMPTY) {
          active = false;
        } else {                                      // evicted pair: on to its other slot
          const unsigned a = gw_s1((unsigned)old, t.lgh);
          slot = (slot == a) ? gw_s2((unsigned)old, t.lgh, t.half) : a;
          cur = old;
        }
      }
    }
    if (__any(active)) return false;
    if (mine) unsafeAtomicAdd(&t.vals[id], v);
  }
  return true;
}

// U keys per lane: the 2 U slots are read back to back, then matched; a key that is present -- the common case by a factor
// of 50-300 -- costs two LDS reads, two compares and one ds_add_f64.
template <int U, int LG, bool SEEDED>
__device__ __forceinline__ bool gw_accum(const gw_tab &t, int &n, const unsigned *m0, const bool *valid, const double *v) {
  unsigned long long e1[U], e2[U];
  unsigned key[U];
#pragma unroll
  for (int u = 0; u < U; u++) {
    key[u] = gw_reseed<SEEDED>(m0[u], t.seed);
    e1[u] = t.T[gw_s1(key[u], t.lgh)];
    e2[u] = t.T[gw_s2(key[u], t.lgh, t.half)];
  }
  bool pending[U];
  bool any_pending = false;
#pragma unroll
  for (int u = 0; u < U; u++) {
    const int id = gw_find(e1[u], e2[u], key[u]);
    const bool hit = valid[u] && id >= 0;
    if (hit) unsafeAtomicAdd(&t.vals[id], v[u]);
    pending[u] = valid[u] && id < 0;
    any_pending |= pending[u];
  }
  if (!__any(any_pending)) return true;
  bool ok = true;
#pragma unroll
  for (int u = 0; u < U; u++)
    if (__any(pending[u])) ok = ok && gw_insert_new<LG, SEEDED>(t, n, m0[u], v[u], pending[u]);
  return ok;
}

// accumulates output row `xr` into the wave's table
template <int LG, int U, bool SEEDED, bool SHARED>
__device__ __forceinline__ void gw_row(const gw_args &P, int64_t xr, const gw_tab &t, int &n, bool &ovf, bool &range) {
  constexpr int G = 1 << LG, NG = 64 >> LG;
  const int lane = threadIdx.x & 63, sub = lane & (G - 1), grp = lane >> LG;
  // (the row is the wave's: bounds and counts are told to the compiler as scalars)
  const int64_t e0 = gw_readlane_i64(P.x_rowptr[xr], 0), e1 = gw_readlane_i64(P.x_rowptr[xr + 1], 0);
  for (int64_t c0 = e0; c0 < e1; c0 += 64) {
    const int nb = __builtin_amdgcn_readfirstlane((int)min((int64_t)64, e1 - c0));
    int64_t st = 0, stm = 0;
    int ln = 0;
    double w = 0.0;
    if (lane < nb) {
      const int64_t yr = (int64_t)P.x_col[c0 + lane] - P.y_row0;
      w = P.x_val[c0 + lane];
      if (yr < 0 || yr >= P.y_nrows) {
        range = true;
      } else {
        st = P.y_start[yr];
        stm = SHARED ? P.y_start_mix[yr] : st;
        ln = P.y_cnt ? P.y_cnt[yr] : (int)(P.y_start[yr + 1] - st);
      }
    }
    for (int j = 0; j < nb; j += U * NG) {
      unsigned m0[U];
      bool valid[U];
      double v[U];
      int lnu[U];
      int64_t stu[U], smu[U];
      double wu[U];
      int longest = 0;
#pragma unroll
      for (int u = 0; u < U; u++) {
        // (every lane takes part in the cross-lane reads; what a lane beyond the chunk fetched is dropped afterwards)
        int jj, lraw;
        if (LG == 6) {
          jj = j + u;                                   // wave-uniform: the row's start travels in scalar registers
          const int src = min(jj, 63);
          stu[u] = gw_readlane_i64(st, src);
          smu[u] = SHARED ? gw_readlane_i64(stm, src) : stu[u];
          lraw = __builtin_amdgcn_readlane(ln, src);
          wu[u] = gw_readlane_f64(w, src);
        } else {
          jj = j + u * NG + grp;
          const int src = min(jj, 63);
          stu[u] = __shfl(st, src, 64);
          smu[u] = SHARED ? __shfl(stm, src, 64) : stu[u];
          lraw = __shfl(ln, src, 64);
          wu[u] = __shfl(w, src, 64);
        }
        lnu[u] = jj < nb ? lraw : 0;
        longest = max(longest, lnu[u]);
        // unconditional loads, no clamp: a lane beyond the end of its row reads the entries that follow (the arrays are
        // padded by TG_CSR_PAD) and is dropped by `valid`
        const uint32_t *pm = P.y_mix + smu[u];
        const double *pv = P.y_val + stu[u];
        m0[u] = pm[sub];
        v[u] = wu[u] * pv[sub];
        valid[u] = sub < lnu[u];
      }
      if (!gw_accum<U, LG, SEEDED>(t, n, m0, valid, v)) ovf = true;
      if (__any(longest > G)) {                         // operand rows longer than a lane group: the rest, one chunk at a time
#pragma unroll
        for (int u = 0; u < U; u++) {
          for (int o = G + sub; __any(o - sub < lnu[u]); o += G) {
            unsigned m1[1];
            bool valid1[1];
            double v1[1];
            const int oc = min(o, max(lnu[u] - 1, 0));
            m1[0] = P.y_mix[smu[u] + oc];
            v1[0] = wu[u] * P.y_val[stu[u] + oc];
            valid1[0] = o < lnu[u];
            if (!gw_accum<1, LG, SEEDED>(t, n, m1, valid1, v1)) ovf = true;
          }
        }
      }
      if (ovf) return;                                  // (wave-uniform: the flags come out of ballots)
    }
  }
}

// mixed column indices of an operand matrix (one pass: 4 bytes read, 4 written per entry)
__global__ void __launch_bounds__(256) k_gw_mix(const int32_t *__restrict__ col, int64_t n, uint32_t *__restrict__ out) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < n; q += stride) out[q] = gw_mix((unsigned)col[q]);
}

// FINAL = false: rows of the intermediate in the loose-row form (out_off / out_cnt per row, entries in table order).
// FINAL = true : rows of K, rank-sorted by column, MatZeroRowsColumns fused; BUMP reserves space in the temporary and records
//                (row_cnt, row_off) for the scan + reorder pass, PLACED writes at row_off[li] and checks the length.
// COUNT: nothing is written; maxima[0] = longest row, sum[0] += lengths (row sample of the plan).
template <int MODE, bool FINAL, int LG, bool SHARED = false>
__global__ void __launch_bounds__(256)
    k_gw(gw_args P, int64_t *__restrict__ out_off, int32_t *__restrict__ out_cnt, int64_t *__restrict__ row_cnt,
         uint32_t *__restrict__ ocol, double *__restrict__ oval, unsigned long long *__restrict__ cursor, int64_t capacity,
         const uint8_t *__restrict__ mask, double diag, int64_t out_row0, int *__restrict__ status,
         int *__restrict__ maxima, unsigned long long *__restrict__ sum) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  gw_tab t;
  t.half = P.ts >> 1;
  t.lgh = P.lgts - 1;
  t.cap = P.ts >> 1;
  t.seed = 0u;
  char *mine = smem + (size_t)wave * gw_wave_bytes(P.ts);
  t.T = reinterpret_cast<unsigned long long *>(mine);
  t.vals = reinterpret_cast<double *>(t.T + P.ts);
  t.kid = reinterpret_cast<uint32_t *>(t.vals + t.cap);
  const int64_t rpb = 4 * (int64_t)P.rows_per_wave;
  const int64_t nblk = (P.x_nrows + rpb - 1) / rpb;
  const int64_t L = tg_xcd_block(blockIdx.x, nblk);
  if (L >= nblk) return;
  bool ovf = false, range = false, cap = false;
  for (int jr = 0; jr < P.rows_per_wave; jr++) {
    const int64_t idx = L * rpb + 4 * (int64_t)jr + wave;     // the four waves of a workgroup work on neighbouring rows
    if (idx >= P.x_nrows) break;
    const int64_t li = P.n0 > 0 ? gw_tile_row(P, idx) : idx * P.row_stride;
    // a row whose keys do not settle in the table (a cuckoo chain that does not end: ~1e-4 of the rows at a third of the
    // slots in use) is started over with the keys permuted once more (every row begins with the plain mix)
    int n = 0;
    bool row_ovf = true;
    for (int attempt = 0; attempt < 4 && row_ovf; attempt++) {
      if (attempt) t.seed = t.seed * 0x2545F491u + 0x9E3779B9u;
      for (int s = lane; s < P.ts; s += 64) t.T[s] = GW_EMPTY;
      for (int s = lane; s < t.cap; s += 64) t.vals[s] = 0.0;
      n = 0;
      row_ovf = false;
      if (attempt == 0)
        gw_row<LG, GW_U(FINAL), false, SHARED>(P, li, t, n, row_ovf, range);
      else
        gw_row<LG, GW_U(FINAL), true, SHARED>(P, li, t, n, row_ovf, range);
      if (n > t.cap) break;                    // more keys than accumulators: another seed does not help
    }
    if (row_ovf) {
      ovf = true;
      continue;
    }
    if (MODE == GW_COUNT) {
      if (lane == 0) {
        atomicMax(&maxima[0], n);
        atomicAdd(sum, (unsigned long long)n);
      }
      continue;
    }
    // ---- where does the row go?
    int64_t o0;
    if (MODE == GW_PLACED) {
      o0 = out_off[li];
      if (out_off[li + 1] - o0 != n) {       // pattern changed since the plan was made
        cap = true;
        continue;
      }
    } else {
      if (P.out_stride > 0) {
        o0 = li * P.out_stride;
        if (n > P.out_stride) {              // (the host doubles the stride)
          cap = true;
          if (lane == 0) atomicMax(&maxima[0], n);
          continue;
        }
      } else {
        unsigned long long o = 0;
        if (lane == 0) o = atomicAdd(cursor, (unsigned long long)n);
        o0 = gw_readlane_i64((int64_t)o, 0);
      }
      if (FINAL) {
        if (lane == 0) {
          row_cnt[li] = n;
          out_off[li] = o0;
        }
      } else if (lane == 0) {
        out_off[li] = o0;
        out_cnt[li] = n;
      }
      if (o0 + n > capacity) {
        cap = true;
        continue;
      }
    }
    if (!FINAL) {
      for (int e = lane; e < n; e += 64) {            // (columns stay mixed: the second stage looks them up as they are)
        ocol[o0 + e] = t.kid[e];
        oval[o0 + e] = t.vals[e];
      }
    } else {
      const int64_t gi = li + out_row0;       // global K row
      const bool mrow = mask ? (mask[gi] != 0) : false;
      for (int e = lane; e < n; e += 64) t.kid[e] = gw_unmix(t.kid[e]);      // back to column indices
      for (int e = lane; e < n; e += 64) {
        const int32_t key = (int32_t)t.kid[e];
        int rank = 0;
        for (int f = 0; f < n; f++) rank += ((int32_t)t.kid[f] < key) ? 1 : 0;
        double x = t.vals[e];
        if (mask && (mrow || mask[key])) x = (mrow && key == gi) ? diag : 0.0;
        ocol[o0 + rank] = (uint32_t)key;
        oval[o0 + rank] = x;
      }
    }
  }
  if (lane == 0) {
    if (ovf) atomicMax(status, GW_OVF);
    if (range) atomicMax(status, GW_RANGE);
    if (cap) atomicMax(status, GW_CAP);
  }
}

// copies reserved rows into CSR order: wave per row
__global__ void __launch_bounds__(256)
    k_gw_reorder(const int64_t *__restrict__ rowptr, const int64_t *__restrict__ tmp_off, const int32_t *__restrict__ tcol,
                 const double *__restrict__ tval, int64_t nrows, int32_t *__restrict__ col, double *__restrict__ val) {
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

static int gw_lg(int v) {
  int l = 0;
  while ((1 << l) < v) l++;
  return l;
}
static int gw_pow2_ge(int64_t v) {
  int p = 1;
  while (p < v) p <<= 1;
  return p;
}
// lanes per operand row from its mean length: the smallest power of two >= 0.75 * mean, in [8, 64]
static int gw_lg_group(double mean) {
  int g = 8;
  while (g < 64 && g < 0.75 * mean) g <<= 1;
  return gw_lg(g);
}
static int gw_env_int(const char *name, int dflt) {
  const char *s = getenv(name);
  return s ? atoi(s) : dflt;
}

template <int MODE, bool FINAL>
static void gw_launch(int lg, unsigned grid, size_t lds, const gw_args &P, int64_t *out_off, int32_t *out_cnt,
                      int64_t *row_cnt, uint32_t *ocol, double *oval, unsigned long long *cursor, int64_t capacity,
                      const uint8_t *mask, double diag, int64_t out_row0, int *status, unsigned long long *sum) {
#define GW_GO(LGV)                                                                                                    \
  hipLaunchKernelGGL((k_gw<MODE, FINAL, LGV>), dim3(grid), dim3(256), lds, g_tg.stream, P, out_off, out_cnt, row_cnt, \
                     ocol, oval, cursor, capacity, mask, diag, out_row0, status, status + 1, sum)
  switch (lg) {
    case 3: GW_GO(3); break;
    case 4: GW_GO(4); break;
    case 5: GW_GO(5); break;
    default: GW_GO(6); break;
  }
#undef GW_GO
}

static void gw_set_lds_limits() {
  static bool done = false;
  if (done) return;
#define GW_LIM(MODE, FINAL, LGV) \
  hipFuncSetAttribute((const void *)k_gw<MODE, FINAL, LGV>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)
#define GW_LIM4(MODE, FINAL) \
  GW_LIM(MODE, FINAL, 3);    \
  GW_LIM(MODE, FINAL, 4);    \
  GW_LIM(MODE, FINAL, 5);    \
  GW_LIM(MODE, FINAL, 6)
  GW_LIM4(GW_COUNT, false);
  GW_LIM4(GW_BUMP, false);
  GW_LIM4(GW_BUMP, true);
  GW_LIM4(GW_PLACED, true);
#undef GW_LIM4
#undef GW_LIM
  done = true;
}

// lattice hint "n0,n1,t0,t1,t2" (experiments: TIGAR_PTAP_WAVE_TILE1 / _TILE2); ignored unless the rows are whole planes
static void gw_tile_hint(gw_args &P, const char *env) {
  P.n0 = P.n1 = 0;
  P.t0 = P.t1 = P.t2 = 1;
  const char *s = getenv(env);
  long long n0 = 0, n1 = 0;
  int t0 = 8, t1 = 8, t2 = 8;
  if (!s || sscanf(s, "%lld,%lld,%d,%d,%d", &n0, &n1, &t0, &t1, &t2) < 2) return;
  if (n0 <= 0 || n1 <= 0 || t0 <= 0 || t1 <= 0 || t2 <= 0 || P.x_nrows % (n0 * n1) != 0) return;
  P.n0 = n0;
  P.n1 = n1;
  P.t0 = t0;
  P.t1 = t1;
  P.t2 = t2;
}

// mixed copy of a matrix' column indices (padded like the column array itself)
static int gw_mix_columns(tg_csr_s *m, uint32_t **out) {
  TG_TRY(tg_dmalloc(out, m->nnz + TG_CSR_PAD));
  if (m->nnz > 0)
    hipLaunchKernelGGL(k_gw_mix, dim3((unsigned)std::min<int64_t>(tg_cdiv(m->nnz, 256), (int64_t)g_tg.num_cu * 32)), dim3(256), 0,
                       g_tg.stream, m->col, m->nnz, *out);
  hipMemsetAsync(*out + m->nnz, 0, TG_CSR_PAD * sizeof(uint32_t), g_tg.stream);
  TG_LAUNCH_CHECK();
  return 0;
}

static void gw_fill_stage1(gw_args &P, tg_csr_s *a, tg_csr_s *m, const uint32_t *m_mix, int64_t m_row0) {
  P.n0 = P.n1 = 0;
  P.t0 = P.t1 = P.t2 = 1;
  P.debug = getenv("TIGAR_PTAP_WAVE_DEBUG") ? atoi(getenv("TIGAR_PTAP_WAVE_DEBUG")) : 0;
  P.out_stride = 0;
  P.y_start_mix = nullptr;
  P.x_rowptr = a->rowptr;
  P.x_col = a->col;
  P.x_val = a->val;
  P.x_nrows = a->nrows;
  P.row_stride = 1;
  P.y_start = m->rowptr;
  P.y_cnt = nullptr;
  P.y_mix = m_mix;
  P.y_val = m->val;
  P.y_row0 = m_row0;
  P.y_nrows = m->nrows;
}

static unsigned gw_grid(int64_t nrows, int rows_per_wave) {
  const int64_t nblk = tg_cdiv(std::max<int64_t>(nrows, 1), 4 * (int64_t)rows_per_wave);
  return (unsigned)(tg_cdiv(nblk, 8) * 8);
}

// sizes the tables of stage 1 from a sample of A's rows; the tables of stage 2 come from the caller's probe of K's rows
static int g_gw_prefer = 0;
extern "C" int tg_ptap_prefer(int kernels) {
  const int old = g_gw_prefer;
  if (kernels >= 0 && kernels <= 2) g_gw_prefer = kernels;
  return old;
}

int tg_ptap_wave_plan(tg_csr_s *a, tg_csr_s *m, int64_t m_row0, tg_csr_s *mt, int max_k, double mean_k, tg_gw_plan *plan) {
  plan->usable = false;
  // Where the wave kernels win (MI355X, profiles/r4_general_ptap.md): operand rows that fill a wave -- 3-D patches of
  // degree >= 3 held resident (48^3 ... 96^3 elements: 1.5-1.8 x).  Short rows (2-D, p = 2, T-spline cells: 12-27 entries)
  // leave most lanes of a row-per-wave walk idle and the fused kernel is up to 2 x faster there; so is it when the product
  // is streamed in row blocks whose operand rows overlap (every block recomputes the rows of A M in its halo).
  // TIGAR_PTAP_WAVE=1/0 and tg_ptap_prefer() override the rule.
  const int forced = gw_env_int("TIGAR_PTAP_WAVE", -1);
  const int pref = forced == 1 ? 1 : forced == 0 ? 2 : g_gw_prefer;
  if (pref == 2) return 0;
  const double mean_m = (double)m->nnz / (double)std::max<int64_t>(m->nrows, 1);
  if (pref == 0 && mean_m < 40.0) return 0;
  if (a->nrows <= 0 || mt->nrows <= 0 || a->nnz <= 0 || m->nnz <= 0) return 0;
  gw_set_lds_limits();
  int *status = (int *)g_tg.scratch;           // [0] status, [1] maximum, [2..3] sum
  unsigned long long *sum = (unsigned long long *)(status + 2);
  uint32_t *m_mix = nullptr;
  TG_TRY(gw_mix_columns(m, &m_mix));
  gw_args S;
  gw_fill_stage1(S, a, m, m_mix, m_row0);
  const int64_t nsample = std::min<int64_t>(a->nrows, 4096);
  S.x_nrows = nsample;
  S.row_stride = std::max<int64_t>(1, a->nrows / nsample);
  S.ts = 2048;                                 // 4 waves x 2048 x 14 B = 112 KB
  S.lgts = 11;
  S.rows_per_wave = 4;
  plan->lg_m = gw_env_int("TIGAR_PTAP_WAVE_LG1", gw_lg_group((double)m->nnz / (double)std::max<int64_t>(m->nrows, 1)));
  TG_CHECK_HIP(hipMemsetAsync(status, 0, 4 * sizeof(int), g_tg.stream));
  gw_launch<GW_COUNT, false>(plan->lg_m, gw_grid(nsample, S.rows_per_wave), 4 * gw_wave_bytes(S.ts), S, nullptr, nullptr, nullptr,
                             nullptr, nullptr, nullptr, 0, nullptr, 0.0, 0, status, sum);
  int h[4] = {0, 0, 0, 0};
  const hipError_t e1 = hipGetLastError();
  const hipError_t e2 = hipMemcpyAsync(h, status, sizeof(h), hipMemcpyDeviceToHost, g_tg.stream);
  const hipError_t e3 = hipStreamSynchronize(g_tg.stream);
  tg_dfree(m_mix);
  TG_CHECK_HIP(e1);
  TG_CHECK_HIP(e2);
  TG_CHECK_HIP(e3);
  if (h[0] == GW_RANGE) {
    tg_set_error("PtAP: a row block does not cover the rows referenced (slab halo too small)");
    return 3;
  }
  if (getenv("TIGAR_PTAP_WAVE_DEBUG")) fprintf(stderr, "[gw] probe status %d max %d sum %d lg_m %d\n", h[0], h[1], h[2], plan->lg_m);
  if (h[0] != GW_OK) return 0;                 // a row of A M fills the probe's table: the workgroup kernel's business
  unsigned long long total;
  memcpy(&total, &h[2], sizeof(total));
  plan->max_am = h[1];
  plan->mean_am = (double)total / (double)nsample;
  plan->max_k = max_k;
  plan->mean_k = mean_k;
  const double load_inv = getenv("TIGAR_PTAP_WAVE_LOADINV") ? atof(getenv("TIGAR_PTAP_WAVE_LOADINV")) : 2.5;
  plan->ts_am = std::max(64, gw_pow2_ge((int64_t)(plan->max_am * load_inv) + 4));
  plan->ts_k = std::max(64, gw_pow2_ge((int64_t)(max_k * load_inv) + 4));
  plan->lg_am = gw_env_int("TIGAR_PTAP_WAVE_LG2", gw_lg_group(plan->mean_am));
  if (getenv("TIGAR_PTAP_WAVE_DEBUG"))
    fprintf(stderr, "[gw] max_am %d mean_am %.1f max_k %d mean_k %.1f ts %d / %d lg %d / %d\n", plan->max_am, plan->mean_am, max_k,
            mean_k, plan->ts_am, plan->ts_k, plan->lg_m, plan->lg_am);
  // (the four tables of a workgroup have to fit the 160 KB of a CU: 2048 slots each)
  plan->usable = 4 * gw_wave_bytes(plan->ts_am) <= 160 * 1024 && 4 * gw_wave_bytes(plan->ts_k) <= 160 * 1024;
  return 0;
}

void tg_ptap_wave_plan_free(tg_gw_plan *plan) {
  tg_dfree(plan->k_rowptr);
  plan->k_rowptr = nullptr;
  plan->k_nnz = -1;
}

// Returns 0 with *k_out set, 100 when the kernels decline (tables beyond their limits after the retries: the caller falls
// back to the workgroup kernel), another value on error.
int tg_ptap_wave_numeric(tg_gw_plan *plan, tg_csr_s *a, int64_t a_row0, tg_csr_s *m, int64_t m_row0, tg_csr_s *mt,
                         int64_t mt_row0, const uint8_t *mask, double diag, tg_csr_s **k_out) {
  gw_set_lds_limits();
  int *status = (int *)g_tg.scratch;
  unsigned long long *sum = (unsigned long long *)(status + 2);
  int rc = 0;
  int64_t *am_off = nullptr;
  int32_t *am_cnt = nullptr;
  uint32_t *am_col = nullptr, *m_mix = nullptr;     // (am_col: the intermediate's columns, mixed)
  double *am_val = nullptr;
  unsigned long long *cursor = nullptr;
  int64_t *cnt = nullptr, *off = nullptr;
  int32_t *tcol = nullptr;
  double *tval = nullptr;
  tg_csr_s *k = nullptr;
  auto cleanup = [&]() {
    tg_dfree(am_off);
    tg_dfree(am_cnt);
    tg_dfree(am_col);
    tg_dfree(m_mix);
    tg_dfree(am_val);
    tg_dfree(cursor);
    tg_dfree(cnt);
    tg_dfree(off);
    tg_dfree(tcol);
    tg_dfree(tval);
  };
  rc = tg_dmalloc(&am_off, a->nrows + 1) || tg_dmalloc(&am_cnt, a->nrows + 1) || tg_dmalloc(&cursor, 2);
  if (rc) {
    cleanup();
    return rc;
  }
  // ---- stage 1: the intermediate A M in loose rows
  rc = gw_mix_columns(m, &m_mix);
  if (rc) {
    cleanup();
    return rc;
  }
  gw_args P1;
  gw_fill_stage1(P1, a, m, m_mix, m_row0);
  P1.rows_per_wave = gw_env_int("TIGAR_PTAP_WAVE_RPW1", 8);
  gw_tile_hint(P1, "TIGAR_PTAP_WAVE_TILE1");
  // rows of the temporary at a fixed stride (no atomics) unless the longest row is far above the mean
  int64_t stride1 = (plan->max_am <= 2.0 * plan->mean_am + 32.0 && gw_env_int("TIGAR_PTAP_WAVE_STRIDED", 1)) ? ((plan->max_am + 7) & ~7) : 0;
  int64_t cap1 = stride1 ? stride1 * a->nrows : (int64_t)(plan->mean_am * 1.08 * (double)a->nrows) + plan->max_am + 4096;
  bool done = false;
  for (int attempt = 0; attempt < 6 && !done; attempt++) {
    rc = tg_dmalloc(&am_col, cap1 + TG_CSR_PAD) || tg_dmalloc(&am_val, cap1 + TG_CSR_PAD);
    if (rc) break;
    P1.ts = plan->ts_am;
    P1.lgts = gw_lg(plan->ts_am);
    P1.out_stride = stride1;
    const size_t lds = 4 * gw_wave_bytes(P1.ts);
    if (lds > 160 * 1024) {
      rc = 100;
      break;
    }
    hipMemsetAsync(status, 0, 4 * sizeof(int), g_tg.stream);
    hipMemsetAsync(cursor, 0, 2 * sizeof(unsigned long long), g_tg.stream);
    gw_launch<GW_BUMP, false>(plan->lg_m, gw_grid(a->nrows, P1.rows_per_wave), lds, P1, am_off, am_cnt, nullptr, am_col, am_val,
                              cursor, cap1, nullptr, 0.0, 0, status, sum);
    int h = 0;
    unsigned long long used = 0;
    hipMemcpyAsync(&h, status, sizeof(int), hipMemcpyDeviceToHost, g_tg.stream);
    hipMemcpyAsync(&used, cursor, sizeof(used), hipMemcpyDeviceToHost, g_tg.stream);
    if (hipStreamSynchronize(g_tg.stream) != hipSuccess || hipGetLastError() != hipSuccess) {
      tg_set_error("PtAP (wave kernels): stage A M failed to run (LDS %zu B)", lds);
      rc = 1;
      break;
    }
    if (h == GW_RANGE) {
      tg_set_error("PtAP: a row block does not cover the rows referenced (slab halo too small)");
      rc = 3;
      break;
    }
    if (getenv("TIGAR_PTAP_WAVE_DEBUG"))
      fprintf(stderr, "[gw] stage 1 attempt %d: status %d used %llu cap %lld ts %d\n", attempt, h, used, (long long)cap1, plan->ts_am);
    if (h == GW_OK) {
      done = true;
      plan->mean_am = std::max(plan->mean_am, (double)used / (double)a->nrows);
      break;
    }
    tg_dfree(am_col);
    tg_dfree(am_val);
    am_col = nullptr;
    am_val = nullptr;
    if (h == GW_OVF) {
      plan->ts_am *= 2;
      if (!stride1) cap1 = std::max<int64_t>(cap1, (int64_t)used + 4096);
    } else if (stride1) {
      int hm = 0;
      hipMemcpy(&hm, status + 1, sizeof(int), hipMemcpyDeviceToHost);
      plan->max_am = std::max(plan->max_am, hm);
      stride1 = (std::max<int64_t>(hm, stride1 + stride1 / 4) + 7) & ~(int64_t)7;
      cap1 = stride1 * a->nrows;
    } else {
      cap1 = std::max<int64_t>((int64_t)used + 4096, cap1 + cap1 / 2);
    }
  }
  if (!rc && !done) rc = 100;
  if (rc) {
    cleanup();
    return rc;
  }
  tg_dfree(m_mix);
  m_mix = nullptr;
  // ---- stage 2: K = M^T (A M), operand rows = the loose rows of the intermediate (FE row r -> r - a_row0)
  gw_args P2;
  P2.x_rowptr = mt->rowptr;
  P2.x_col = mt->col;
  P2.x_val = mt->val;
  P2.x_nrows = mt->nrows;
  P2.row_stride = 1;
  P2.y_start_mix = nullptr;
  P2.y_start = am_off;
  P2.y_cnt = am_cnt;
  P2.y_mix = am_col;
  P2.y_val = am_val;
  P2.y_row0 = a_row0;
  P2.y_nrows = a->nrows;
  P2.debug = P1.debug & ~1;
  P2.out_stride = 0;
  P2.rows_per_wave = gw_env_int("TIGAR_PTAP_WAVE_RPW2", 2);
  gw_tile_hint(P2, "TIGAR_PTAP_WAVE_TILE2");
  const int64_t nrows = mt->nrows;
  if (plan->k_nnz >= 0) {
    // ---- pattern known: rows are placed directly
    rc = tg_csr_alloc(nrows, m->ncols, plan->k_nnz, &k);
    if (!rc) {
      hipMemcpyAsync(k->rowptr, plan->k_rowptr, (size_t)(nrows + 1) * sizeof(int64_t), hipMemcpyDeviceToDevice, g_tg.stream);
      P2.ts = plan->ts_k;
      P2.lgts = gw_lg(plan->ts_k);
      hipMemsetAsync(status, 0, 4 * sizeof(int), g_tg.stream);
      gw_launch<GW_PLACED, true>(plan->lg_am, gw_grid(nrows, P2.rows_per_wave), 4 * gw_wave_bytes(P2.ts), P2, k->rowptr, nullptr,
                                 nullptr, (uint32_t *)k->col, k->val, nullptr, 0, mask, diag, mt_row0, status, sum);
      int h = 0;
      hipMemcpyAsync(&h, status, sizeof(int), hipMemcpyDeviceToHost, g_tg.stream);
      if (hipStreamSynchronize(g_tg.stream) != hipSuccess || hipGetLastError() != hipSuccess) {
        tg_set_error("PtAP (wave kernels): numeric pass (placed) failed to run");
        rc = 1;
      } else if (h == GW_RANGE) {
        tg_set_error("PtAP: a row block does not cover the rows referenced (slab halo too small)");
        rc = 3;
      } else if (h != GW_OK) {
        tg_set_error("PtAP numeric: operands no longer match the plan's pattern (status %d)", h);
        rc = 4;
      }
    }
  } else {
    int64_t stride2 = (plan->max_k <= 2.0 * plan->mean_k + 32.0 && gw_env_int("TIGAR_PTAP_WAVE_STRIDED", 1)) ? ((plan->max_k + 7) & ~7) : 0;
    int64_t cap2 = stride2 ? stride2 * nrows : (int64_t)(plan->mean_k * 1.05 * (double)nrows) + plan->max_k + 1024;
    rc = tg_dmalloc(&cnt, nrows + 1) || tg_dmalloc(&off, nrows + 1);
    done = false;
    for (int attempt = 0; attempt < 6 && !rc && !done; attempt++) {
      rc = tg_dmalloc(&tcol, cap2 + TG_CSR_PAD) || tg_dmalloc(&tval, cap2 + TG_CSR_PAD);
      if (rc) break;
      P2.ts = plan->ts_k;
      P2.lgts = gw_lg(plan->ts_k);
      P2.out_stride = stride2;
      const size_t lds = 4 * gw_wave_bytes(P2.ts);
      if (lds > 160 * 1024) {
        rc = 100;
        break;
      }
      hipMemsetAsync(status, 0, 4 * sizeof(int), g_tg.stream);
      hipMemsetAsync(cursor, 0, 2 * sizeof(unsigned long long), g_tg.stream);
      hipMemsetAsync(cnt, 0, (size_t)(nrows + 1) * sizeof(int64_t), g_tg.stream);
      gw_launch<GW_BUMP, true>(plan->lg_am, gw_grid(nrows, P2.rows_per_wave), lds, P2, off, nullptr, cnt, (uint32_t *)tcol, tval, cursor, cap2,
                               mask, diag, mt_row0, status, sum);
      int h = 0;
      unsigned long long used = 0;
      hipMemcpyAsync(&h, status, sizeof(int), hipMemcpyDeviceToHost, g_tg.stream);
      hipMemcpyAsync(&used, cursor, sizeof(used), hipMemcpyDeviceToHost, g_tg.stream);
      if (hipStreamSynchronize(g_tg.stream) != hipSuccess || hipGetLastError() != hipSuccess) {
        tg_set_error("PtAP (wave kernels): stage M^T (A M) failed to run (LDS %zu B)", lds);
        rc = 1;
        break;
      }
      if (h == GW_RANGE) {
        tg_set_error("PtAP: a row block does not cover the rows referenced (slab halo too small)");
        rc = 3;
        break;
      }
      if (getenv("TIGAR_PTAP_WAVE_DEBUG"))
        fprintf(stderr, "[gw] stage 2 attempt %d: status %d used %llu cap %lld ts %d\n", attempt, h, used, (long long)cap2, plan->ts_k);
      if (h == GW_OK) {
        done = true;
        break;
      }
      tg_dfree(tcol);
      tg_dfree(tval);
      tcol = nullptr;
      tval = nullptr;
      if (h == GW_OVF) {
        plan->ts_k *= 2;
        if (!stride2) cap2 = std::max<int64_t>(cap2, (int64_t)used + 1024);
      } else if (stride2) {
        int hm = 0;
        hipMemcpy(&hm, status + 1, sizeof(int), hipMemcpyDeviceToHost);
        plan->max_k = std::max(plan->max_k, hm);
        stride2 = (std::max<int64_t>(hm, stride2 + stride2 / 4) + 7) & ~(int64_t)7;
        cap2 = stride2 * nrows;
      } else {
        cap2 = std::max<int64_t>((int64_t)used + 1024, cap2 + cap2 / 2);
      }
    }
    if (!rc && !done) rc = 100;
    if (!rc) {
      int64_t nnz = 0;
      rc = tg_exclusive_scan_i64(cnt, nrows, &nnz);
      if (!rc) rc = tg_csr_alloc(nrows, m->ncols, nnz, &k);
      if (!rc) {
        hipMemcpyAsync(k->rowptr, cnt, (size_t)(nrows + 1) * sizeof(int64_t), hipMemcpyDeviceToDevice, g_tg.stream);
        const unsigned rg = (unsigned)std::min<int64_t>(tg_cdiv(nrows, 4), (int64_t)g_tg.num_cu * 16);
        hipLaunchKernelGGL(k_gw_reorder, dim3(std::max(1u, rg)), dim3(256), 0, g_tg.stream, k->rowptr, off, tcol, tval, nrows,
                           k->col, k->val);
        if (hipGetLastError() != hipSuccess) {
          tg_set_error("PtAP reorder launch failed");
          rc = 1;
        }
        // remember the pattern for later calls with the same operands' structure
        tg_dfree(plan->k_rowptr);
        plan->k_rowptr = cnt;
        cnt = nullptr;
        plan->k_nnz = nnz;
      }
    }
  }
  hipStreamSynchronize(g_tg.stream);
  cleanup();
  if (rc) {
    if (k) tg_csr_destroy(k);
    return rc;
  }
  *k_out = k;
  return 0;
}


// ------------------------------------------------------------------------------------------------------------------
// Cell-block product (round 4): K = M^T A M when the FE space is CELL-LOCAL -- every cell carries its own b nodes, numbered
// cell after cell, as the meshes of disconnected cells the reference builds for Rhino T-splines and multi-patch B-splines
// (tIGAr/RhinoTSplines.py:195-240, tIGAr/BSplines.py:800-860), so that a matrix assembled on it is block diagonal with one
// dense b x b block per cell.  Then
//
//     K = sum_c  S_c^T ( M_c^T A_c M_c ) S_c ,      M_c = the rows of M of cell c as a dense b x nf_c block over the
//                                                   cell's own list of functions, S_c = that list as a selection
//
// -- the "supernodal" form of the product: the multiply-adds run on dense little blocks without any look-up
// (k_cell_element: one wave per cell, E_c = M_c^T (A_c M_c) out of LDS), and only the nf_c^2 entries of an element
// matrix are merged into K by look-up: the second Gustavson stage of the wave kernels above with the rows of the element
// matrices as operand rows (weight 1), each row of K gathering the rows (c, q) of the cells that contain its function.
// 12 x fewer accumulations by look-up than the row-wise product on the T-spline benchmark.
// The plan (cell function lists, dense M_c, incidence) depends on M only: built once per extraction operator on the host
// (tigar_amd/cellptap.py), kept here on the device.  A is verified to be block diagonal with dense blocks (row lengths
// and first / last column of every row: rows are sorted) -- status 100 otherwise, the caller takes the general kernels.
struct tg_cellplan_s {
  int64_t ncell = 0, ncols = 0;
  int b = 0, nfmax = 0;
  double *md = nullptr;          // [ncell][b][nfmax] dense rows of M per cell (zero where a node has no entry)
  uint32_t *flmix = nullptr;     // [ncell][nfmax] the cell's functions, mixed (gw_mix), padded
  int32_t *nf = nullptr;         // [ncell]
  int64_t *e_start = nullptr, *e_start_mix = nullptr;   // per row (c, q) of the element matrices: where its values / columns start
  int32_t *e_cnt = nullptr;
  tg_csr_s *inc = nullptr;       // incidence: row i -> rows (c, q) of the element matrices with function i (borrowed)
  int max_k = 0;
  double mean_k = 0.0;
  tg_gw_plan gw;
  // known after the first product: the columns of K and, for every entry (c, q, r) of the element matrices, its PLACE in its
  // row of K (the slot of column fl[c][r] in row fl[c][q]; rows of at most TG_CELL_ROWCAP - 1 entries) -- the merge then needs
  // no look-up
  int32_t *k_col = nullptr;
  uint16_t *slot = nullptr;      // [ncell * nfmax][nfmax]
};

#define TG_CELL_ROWCAP 512      // accumulators per row of K in the merge by places (3-D p = 3: 343 entries per row)
#define TG_CELL_NOSLOT 0xffffu
// slot[(c, q)][r] = position of column fl[c][r] in row fl[c][q] of K (binary search in the sorted row), 0xffff = not an entry
__global__ void __launch_bounds__(256) k_cell_slots(const uint32_t *__restrict__ flmix, const int32_t *__restrict__ nfc, int64_t ncell,
                                                    int nfmax, const int64_t *__restrict__ krowptr, const int32_t *__restrict__ kcol,
                                                    uint16_t *__restrict__ slot, int *__restrict__ bad) {
  const int64_t total = ncell * (int64_t)nfmax * nfmax, stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += stride) {
    const int r = (int)(t % nfmax);
    const int64_t cq = t / nfmax, c = cq / nfmax;
    const int q = (int)(cq - c * nfmax), nf = nfc[c];
    uint16_t sl = TG_CELL_NOSLOT;
    if (q < nf && r < nf) {
      const int64_t i = gw_unmix(flmix[c * nfmax + q]);
      const int32_t j = (int32_t)gw_unmix(flmix[c * nfmax + r]);
      int64_t lo = krowptr[i], hi = krowptr[i + 1];
      const int64_t a = lo;
      while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (kcol[mid] < j) lo = mid + 1; else hi = mid;
      }
      if (lo < krowptr[i + 1] && kcol[lo] == j && lo - a < TG_CELL_ROWCAP) sl = (uint16_t)(lo - a);
      else atomicOr(bad, 1);
    }
    slot[t] = sl;
  }
}

// K rows from the element matrices by PLACES: one wave per row of K; the 64 / LPR groups of lanes take the incident rows
// (c, q) in turn and add their nf values into the group's own accumulators (places within one row are distinct: plain
// read-modify-write), the groups' accumulators are added in a fixed order; MatZeroRowsColumns on the way out.
template <int LGR>
__global__ void __launch_bounds__(256)
    k_cell_merge(const int64_t *__restrict__ irowptr, const int32_t *__restrict__ icol, const double *__restrict__ eval,
                 const uint16_t *__restrict__ slot, const int32_t *__restrict__ nfc, int nfmax, int64_t nrows,
                 const int64_t *__restrict__ krowptr, const int32_t *__restrict__ kcol, const uint8_t *__restrict__ mask, double diag,
                 double *__restrict__ kval) {
  constexpr int LPR = 1 << LGR, NG = 64 >> LGR;
  constexpr int CAP = NG == 1 ? TG_CELL_ROWCAP : 256;     // (64 KB of static LDS: the long rows belong to the 64-function cells)
  __shared__ double acc_all[4][NG][CAP];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, grp = lane >> LGR, sub = lane & (LPR - 1);
  double(*acc)[CAP] = acc_all[wave];
  const int64_t wstride = (int64_t)gridDim.x * 4;
  for (int64_t i = (int64_t)blockIdx.x * 4 + wave; i < nrows; i += wstride) {
    const int64_t k0 = krowptr[i];
    const int n = (int)(krowptr[i + 1] - k0);
    for (int e = lane; e < NG * CAP; e += 64) acc[0][e] = 0.0;       // (rows of one wave: no barrier needed, LDS ops are in order)
    for (int64_t x = irowptr[i] + grp; x < irowptr[i + 1]; x += NG) {
      const int64_t cq = icol[x];
      const int nf = nfc[cq / nfmax];
      for (int r = sub; r < nf; r += LPR) {
        const uint16_t sl = slot[cq * nfmax + r];
        if (sl != TG_CELL_NOSLOT) acc[grp][sl] += eval[cq * nfmax + r];
      }
    }
    const bool mrow = mask && mask[i];
    for (int e = lane; e < n; e += 64) {
      double v = acc[0][e];
#pragma unroll
      for (int g = 1; g < NG; g++) v += acc[g][e];
      if (mask) {
        const int32_t c = kcol[k0 + e];
        if (mrow || mask[c]) v = (mrow && c == i) ? diag : 0.0;
      }
      kval[k0 + e] = v;
    }
  }
}

// one wave per cell: E_c = M_c^T (A_c M_c).  Lanes: groups of G = 2^LGF lanes <-> the function index f, 64 / G rows at once.
template <int LGF>
__global__ void __launch_bounds__(256)
    k_cell_element(const double *__restrict__ aval, const int64_t *__restrict__ rstart, const double *__restrict__ md,
                   const int32_t *__restrict__ nfc, int64_t ncell, int b, int nfmax, double *__restrict__ eval) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int G = 1 << LGF, NG = 64 >> LGF;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, sub = lane & (G - 1), grp = lane >> LGF;
  const size_t per_wave = (size_t)b * b + 2 * (size_t)b * nfmax;
  double *As = reinterpret_cast<double *>(smem) + wave * per_wave;
  double *Ms = As + (size_t)b * b, *Ts = Ms + (size_t)b * nfmax;
  const int64_t c = (int64_t)blockIdx.x * 4 + wave;
  if (c >= ncell) return;
  const int nf = nfc[c];
  const double *ac = aval + c * (int64_t)b * b, *mc = md + c * (int64_t)b * nfmax;
  if (rstart) {        // (the cell's rows hold other entries besides their block: where the block's b values start, row by row)
    for (int s = lane; s < b * b; s += 64) As[s] = aval[rstart[c * b + s / b] + s % b];
  } else {
    for (int s = lane; s < b * b; s += 64) As[s] = ac[s];
  }
  for (int s = lane; s < b * nfmax; s += 64) Ms[s] = mc[s];
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  // T = A_c M_c : [b][nf]
  for (int r0 = 0; r0 < b; r0 += NG) {
    const int r = r0 + grp;
    double acc = 0.0;
    if (r < b && sub < nf)
      for (int q = 0; q < b; q++) acc = fma(As[r * b + q], Ms[q * nfmax + sub], acc);
    if (r < b && sub < nfmax) Ts[r * nfmax + sub] = acc;
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  // E = M_c^T T : [nf][nf], row (c, q) at (c * nfmax + q) * nfmax
  double *ec = eval + c * (int64_t)nfmax * nfmax;
  for (int q0 = 0; q0 < nf; q0 += NG) {
    const int q = q0 + grp;
    double acc = 0.0;
    if (q < nf && sub < nf) {
      for (int r = 0; r < b; r++) acc = fma(Ms[r * nfmax + q], Ts[r * nfmax + sub], acc);
      ec[(int64_t)q * nfmax + sub] = acc;
    }
  }
}

// every row of A: b entries, the first at column (row / b) * b, the last at + b - 1 (rows are sorted: the block, dense)
// Cells too large for four of them in the LDS of a workgroup (3-D p = 3: 64 nodes, 64 functions): ONE cell per workgroup,
// thread (ti, tj) of 16 x 16 holds a 4 x 4 tile of T = A_c M_c and then of E = M_c^T T in registers -- 8 LDS reads per 16
// multiply-adds instead of 2 per 1.  A_c and T share their LDS (T is written when A_c is no longer read).
__global__ void __launch_bounds__(256)
    k_cell_element_big(const double *__restrict__ aval, const double *__restrict__ md, const int32_t *__restrict__ nfc,
                       int64_t ncell, int b, int nfmax, double *__restrict__ eval) {
  constexpr int LA = 65;
  __shared__ double As[64 * LA];     // A_c [row][q], then T [r][s] with row length 64
  __shared__ double Ms[64 * 64];     // M_c [node][function], zero-padded to 64 x 64
  const int tid = threadIdx.x, ti = tid >> 4, tj = tid & 15;
  for (int64_t c = blockIdx.x; c < ncell; c += gridDim.x) {
    const int nf = nfc[c];
    const double *ac = aval + c * (int64_t)b * b, *mc = md + c * (int64_t)b * nfmax;
    __syncthreads();
    for (int t = tid; t < 64 * 64; t += 256) {
      const int i = t >> 6, j = t & 63;
      As[i * LA + j] = (i < b && j < b) ? ac[i * b + j] : 0.0;
      Ms[t] = (i < b && j < nf) ? mc[i * nfmax + j] : 0.0;
    }
    __syncthreads();
    double acc[4][4];
#pragma unroll
    for (int u = 0; u < 4; u++)
#pragma unroll
      for (int v = 0; v < 4; v++) acc[u][v] = 0.0;
    for (int q = 0; q < b; q++) {
      double a4[4], m4[4];
#pragma unroll
      for (int u = 0; u < 4; u++) a4[u] = As[(4 * ti + u) * LA + q];
#pragma unroll
      for (int v = 0; v < 4; v++) m4[v] = Ms[q * 64 + 4 * tj + v];
#pragma unroll
      for (int u = 0; u < 4; u++)
#pragma unroll
        for (int v = 0; v < 4; v++) acc[u][v] = fma(a4[u], m4[v], acc[u][v]);
    }
    __syncthreads();                       // (everyone is done with A_c)
#pragma unroll
    for (int u = 0; u < 4; u++)
#pragma unroll
      for (int v = 0; v < 4; v++) {
        As[(4 * ti + u) * 64 + 4 * tj + v] = acc[u][v];
        acc[u][v] = 0.0;
      }
    __syncthreads();
    for (int r = 0; r < b; r++) {
      double q4[4], t4[4];
#pragma unroll
      for (int u = 0; u < 4; u++) q4[u] = Ms[r * 64 + 4 * ti + u];
#pragma unroll
      for (int v = 0; v < 4; v++) t4[v] = As[r * 64 + 4 * tj + v];
#pragma unroll
      for (int u = 0; u < 4; u++)
#pragma unroll
        for (int v = 0; v < 4; v++) acc[u][v] = fma(q4[u], t4[v], acc[u][v]);
    }
    double *ec = eval + c * (int64_t)nfmax * nfmax;
#pragma unroll
    for (int u = 0; u < 4; u++)
#pragma unroll
      for (int v = 0; v < 4; v++) {
        const int q = 4 * ti + u, sidx = 4 * tj + v;
        if (q < nf && sidx < nf) ec[(int64_t)q * nfmax + sidx] = acc[u][v];
      }
  }
}

__global__ void __launch_bounds__(256) k_cell_check(const int64_t *__restrict__ rowptr, const int32_t *__restrict__ col, int64_t nrows,
                                                  int b, int *__restrict__ bad) {
  int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  bool wrong = false;
  for (; r < nrows; r += stride) {
    const int64_t a = rowptr[r], e = rowptr[r + 1];
    const int64_t c0 = (r / b) * b;
    if (a != r * b || e - a != b || col[a] != c0 || col[e - 1] != c0 + b - 1) wrong = true;
  }
  if (wrong) atomicMax(bad, 1);
}

// md_host == nullptr: the dense rows of M are filled in on the device afterwards (tg_cellplan_create_from_rows)
static int tg_cellplan_create_common(int64_t ncell, int b, int nfmax, int64_t ncols, const double *md_host, const int32_t *fl_host,
                                     const int32_t *nf_host, tg_csr_t incidence, int max_k, double mean_k, tg_cellplan_t *out) {
  TG_REQUIRE_INIT();
  TG_REQUIRE(ncell > 0 && b >= 1 && b <= 64 && nfmax >= 1 && nfmax <= 64 && fl_host && nf_host && incidence && out,
             "bad arguments to tg_cellplan_create");
  TG_REQUIRE(incidence->nrows == ncols && incidence->ncols == ncell * nfmax, "tg_cellplan_create: incidence of the wrong shape");
  tg_cellplan_s *pl = new tg_cellplan_s();
  pl->ncell = ncell;
  pl->ncols = ncols;
  pl->b = b;
  pl->nfmax = nfmax;
  pl->inc = incidence;
  pl->max_k = max_k;
  pl->mean_k = mean_k;
  const int64_t nrowsE = ncell * nfmax;
  std::vector<uint32_t> mix((size_t)nrowsE);
  std::vector<int64_t> st((size_t)nrowsE), stm((size_t)nrowsE);
  std::vector<int32_t> cnt((size_t)nrowsE);
  for (int64_t c = 0; c < ncell; c++)
    for (int q = 0; q < nfmax; q++) {
      const int64_t k = c * nfmax + q;
      mix[(size_t)k] = gw_mix((unsigned)(q < nf_host[c] ? fl_host[k] : 0));
      st[(size_t)k] = k * nfmax;
      stm[(size_t)k] = c * nfmax;
      cnt[(size_t)k] = q < nf_host[c] ? nf_host[c] : 0;
    }
  int rc = tg_dmalloc(&pl->md, ncell * (int64_t)b * nfmax) || tg_dmalloc(&pl->flmix, nrowsE + TG_CSR_PAD) || tg_dmalloc(&pl->nf, ncell) ||
           tg_dmalloc(&pl->e_start, nrowsE) || tg_dmalloc(&pl->e_start_mix, nrowsE) || tg_dmalloc(&pl->e_cnt, nrowsE);
  if (!rc) {
    if (md_host)
      hipMemcpyAsync(pl->md, md_host, (size_t)(ncell * (int64_t)b * nfmax) * sizeof(double), hipMemcpyHostToDevice, g_tg.stream);
    else
      hipMemsetAsync(pl->md, 0, (size_t)(ncell * (int64_t)b * nfmax) * sizeof(double), g_tg.stream);
    hipMemsetAsync(pl->flmix, 0, (size_t)(nrowsE + TG_CSR_PAD) * sizeof(uint32_t), g_tg.stream);
    hipMemcpyAsync(pl->flmix, mix.data(), mix.size() * sizeof(uint32_t), hipMemcpyHostToDevice, g_tg.stream);
    hipMemcpyAsync(pl->nf, nf_host, (size_t)ncell * sizeof(int32_t), hipMemcpyHostToDevice, g_tg.stream);
    hipMemcpyAsync(pl->e_start, st.data(), st.size() * sizeof(int64_t), hipMemcpyHostToDevice, g_tg.stream);
    hipMemcpyAsync(pl->e_start_mix, stm.data(), stm.size() * sizeof(int64_t), hipMemcpyHostToDevice, g_tg.stream);
    hipMemcpyAsync(pl->e_cnt, cnt.data(), cnt.size() * sizeof(int32_t), hipMemcpyHostToDevice, g_tg.stream);
    if (hipStreamSynchronize(g_tg.stream) != hipSuccess) rc = 1;
  }
  if (rc) {
    tg_cellplan_destroy(pl);
    return rc;
  }
  const double load_inv = 2.5;
  pl->gw.ts_k = std::max(64, gw_pow2_ge((int64_t)(max_k * load_inv) + 4));
  pl->gw.lg_am = gw_lg_group((double)nfmax);
  pl->gw.max_k = max_k;
  pl->gw.mean_k = mean_k;
  *out = pl;
  return 0;
}

extern "C" int tg_cellplan_create(int64_t ncell, int b, int nfmax, int64_t ncols, const double *md_host, const int32_t *fl_host,
                                  const int32_t *nf_host, tg_csr_t incidence, int max_k, double mean_k, tg_cellplan_t *out) {
  TG_REQUIRE(md_host, "bad arguments to tg_cellplan_create");
  return tg_cellplan_create_common(ncell, b, nfmax, ncols, md_host, fl_host, nf_host, incidence, max_k, mean_k, out);
}

extern "C" int tg_cellplan_destroy(tg_cellplan_t pl) {
  if (!pl) return 0;
  if (g_tg.ready) hipStreamSynchronize(g_tg.stream);
  tg_dfree(pl->md);
  tg_dfree(pl->flmix);
  tg_dfree(pl->nf);
  tg_dfree(pl->e_start);
  tg_dfree(pl->e_start_mix);
  tg_dfree(pl->e_cnt);
  tg_dfree(pl->k_col);
  tg_dfree(pl->slot);
  tg_ptap_wave_plan_free(&pl->gw);
  delete pl;
  return 0;
}

template <int MODE>
static void gw_launch_shared(int lg, unsigned grid, size_t lds, const gw_args &P, int64_t *out_off, int64_t *row_cnt, uint32_t *ocol,
                             double *oval, unsigned long long *cursor, int64_t capacity, const uint8_t *mask, double diag, int *status,
                             unsigned long long *sum) {
#define GW_GOS(LGV)                                                                                                                  \
  do {                                                                                                                               \
    hipFuncSetAttribute((const void *)k_gw<MODE, true, LGV, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);          \
    hipLaunchKernelGGL((k_gw<MODE, true, LGV, true>), dim3(grid), dim3(256), lds, g_tg.stream, P, out_off, (int32_t *)nullptr, row_cnt, \
                       ocol, oval, cursor, capacity, mask, diag, (int64_t)0, status, status + 1, sum);                               \
  } while (0)
  switch (lg) {
    case 3: GW_GOS(3); break;
    case 4: GW_GOS(4); break;
    case 5: GW_GOS(5); break;
    default: GW_GOS(6); break;
  }
#undef GW_GOS
}

// rstart (device, one entry per FE row, or null): where the b values of the row's own cell block start in a->val -- for a
// matrix that holds other entries besides its dense cell blocks (tg_cellplan_ptap_extras); null: a is verified to consist of
// the blocks alone
// trusted: a->val IS the array of dense blocks [ncell][b][b] (tg_elemsplit_ptap: written by this library), nothing else of
// `a` is looked at
static int tg_cellplan_ptap_impl(tg_cellplan_t pl, tg_csr_t a, const int64_t *rstart, const int32_t *zero_dofs, int64_t nzero,
                                 double diag, tg_csr_t *k_out, bool trusted = false) {
  const int64_t nfe = pl->ncell * pl->b;
  if (!trusted && (a->nrows != nfe || a->ncols != nfe || (!rstart && a->nnz != nfe * pl->b))) return 100;
  int *status = (int *)g_tg.scratch;
  unsigned long long *sum = (unsigned long long *)(status + 2);
  if (!rstart && !trusted) {
    int hbad = 0;
    hipMemsetAsync(status, 0, 4 * sizeof(int), g_tg.stream);
    hipLaunchKernelGGL(k_cell_check, dim3((unsigned)std::min<int64_t>(tg_cdiv(nfe, 256), (int64_t)g_tg.num_cu * 16)), dim3(256), 0,
                       g_tg.stream, a->rowptr, a->col, nfe, pl->b, status);
    TG_LAUNCH_CHECK();
    TG_CHECK_HIP(hipMemcpyAsync(&hbad, status, sizeof(int), hipMemcpyDeviceToHost, g_tg.stream));
    TG_CHECK_HIP(hipStreamSynchronize(g_tg.stream));
    if (hbad) return 100;                     // not block diagonal with dense b x b blocks: the general kernels
  }
  uint8_t *mask = nullptr;
  if (nzero > 0) TG_TRY(tg_build_dof_mask(zero_dofs, nzero, pl->ncols, &mask));
  double *eval = nullptr;
  int64_t *cnt = nullptr, *off = nullptr;
  int32_t *tcol = nullptr;
  double *tval = nullptr;
  unsigned long long *cursor = nullptr;
  tg_csr_s *k = nullptr;
  auto cleanup = [&]() {
    tg_dfree(eval);
    tg_dfree(cnt);
    tg_dfree(off);
    tg_dfree(tcol);
    tg_dfree(tval);
    tg_dfree(cursor);
    tg_dfree(mask);
  };
  const int64_t nrowsE = pl->ncell * pl->nfmax;
  int rc = tg_dmalloc(&eval, nrowsE * pl->nfmax + TG_CSR_PAD) || tg_dmalloc(&cursor, 2);
  if (rc) {
    cleanup();
    return rc;
  }
  // ---- element matrices
  {
    const size_t lds = 4 * ((size_t)pl->b * pl->b + 2 * (size_t)pl->b * pl->nfmax) * sizeof(double);
    const unsigned grid = (unsigned)tg_cdiv(pl->ncell, 4);
    const int lgf = gw_lg(pl->nfmax <= 8 ? 8 : pl->nfmax <= 16 ? 16 : pl->nfmax <= 32 ? 32 : 64);
#define CELL_GO(LGV)                                                                                                    \
  do {                                                                                                                  \
    hipFuncSetAttribute((const void *)k_cell_element<LGV>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);      \
    hipLaunchKernelGGL((k_cell_element<LGV>), dim3(grid), dim3(256), lds, g_tg.stream, a->val, rstart, pl->md, pl->nf,   \
                       pl->ncell, pl->b, pl->nfmax, eval);                                                              \
  } while (0)
    if (lds > 160 * 1024 && rstart) {
      cleanup();
      return 100;
    }
    if (lds > 160 * 1024) {
      hipLaunchKernelGGL(k_cell_element_big, dim3((unsigned)std::min<int64_t>(pl->ncell, (int64_t)g_tg.num_cu * 64)), dim3(256), 0,
                         g_tg.stream, a->val, pl->md, pl->nf, pl->ncell, pl->b, pl->nfmax, eval);
    } else
    switch (lgf) {
      case 3: CELL_GO(3); break;
      case 4: CELL_GO(4); break;
      case 5: CELL_GO(5); break;
      default: CELL_GO(6); break;
    }
#undef CELL_GO
    if (hipGetLastError() != hipSuccess) {
      tg_set_error("cell-block PtAP: the element kernel failed to launch");
      cleanup();
      return 1;
    }
  }
  // ---- K rows: the second Gustavson stage over the rows of the element matrices
  gw_set_lds_limits();
  gw_args P2;
  memset(&P2, 0, sizeof(P2));
  P2.x_rowptr = pl->inc->rowptr;
  P2.x_col = pl->inc->col;
  P2.x_val = pl->inc->val;
  P2.x_nrows = pl->inc->nrows;
  P2.row_stride = 1;
  P2.y_start = pl->e_start;
  P2.y_start_mix = pl->e_start_mix;
  P2.y_cnt = pl->e_cnt;
  P2.y_mix = pl->flmix;
  P2.y_val = eval;
  P2.y_row0 = 0;
  P2.y_nrows = nrowsE;
  P2.rows_per_wave = 2;
  P2.t0 = P2.t1 = P2.t2 = 1;
  const int64_t nrows = pl->ncols;
  tg_gw_plan *plan = &pl->gw;
  // values of K by places (k_cell_merge) into an allocated k with its row pointer and columns set
  auto merge_by_places = [&](tg_csr_s *kk) -> int {
    const unsigned grid = (unsigned)std::min<int64_t>(tg_cdiv(nrows, 4), (int64_t)g_tg.num_cu * 32);
#define CELL_MERGE(LGV)                                                                                                         \
  hipLaunchKernelGGL((k_cell_merge<LGV>), dim3(std::max(1u, grid)), dim3(256), 0, g_tg.stream, pl->inc->rowptr, pl->inc->col, eval, \
                     pl->slot, pl->nf, pl->nfmax, nrows, kk->rowptr, kk->col, mask, diag, kk->val)
    if (pl->nfmax <= 16) CELL_MERGE(4);
    else if (pl->nfmax <= 32) CELL_MERGE(5);
    else CELL_MERGE(6);
#undef CELL_MERGE
    if (hipGetLastError() != hipSuccess) {
      tg_set_error("cell-block PtAP: the merge kernel failed to launch");
      return 1;
    }
    return 0;
  };
  if (plan->k_nnz >= 0 && pl->slot && pl->k_col && !getenv("TIGAR_CELL_MERGE_HASH")) {
    // ---- pattern and places known: no look-up at all
    rc = tg_csr_alloc(nrows, pl->ncols, plan->k_nnz, &k);
    if (!rc) {
      hipMemcpyAsync(k->rowptr, plan->k_rowptr, (size_t)(nrows + 1) * sizeof(int64_t), hipMemcpyDeviceToDevice, g_tg.stream);
      hipMemcpyAsync(k->col, pl->k_col, (size_t)plan->k_nnz * sizeof(int32_t), hipMemcpyDeviceToDevice, g_tg.stream);
      rc = merge_by_places(k);
      if (!rc && hipStreamSynchronize(g_tg.stream) != hipSuccess) {
        tg_set_error("cell-block PtAP: merge by places failed to run");
        rc = 1;
      }
    }
  } else if (plan->k_nnz >= 0) {
    rc = tg_csr_alloc(nrows, pl->ncols, plan->k_nnz, &k);
    if (!rc) {
      hipMemcpyAsync(k->rowptr, plan->k_rowptr, (size_t)(nrows + 1) * sizeof(int64_t), hipMemcpyDeviceToDevice, g_tg.stream);
      P2.ts = plan->ts_k;
      P2.lgts = gw_lg(plan->ts_k);
      hipMemsetAsync(status, 0, 4 * sizeof(int), g_tg.stream);
      gw_launch_shared<GW_PLACED>(plan->lg_am, gw_grid(nrows, P2.rows_per_wave), 4 * gw_wave_bytes(P2.ts), P2, k->rowptr, nullptr,
                                  (uint32_t *)k->col, k->val, nullptr, 0, mask, diag, status, sum);
      int h = 0;
      hipMemcpyAsync(&h, status, sizeof(int), hipMemcpyDeviceToHost, g_tg.stream);
      if (hipStreamSynchronize(g_tg.stream) != hipSuccess || hipGetLastError() != hipSuccess) {
        tg_set_error("cell-block PtAP: numeric pass (placed) failed to run");
        rc = 1;
      } else if (h != GW_OK) {
        tg_set_error("cell-block PtAP: the operands no longer match the plan's pattern (status %d)", h);
        rc = 4;
      }
    }
  } else {
    int64_t stride2 = (plan->max_k + 7) & ~7;
    int64_t cap2 = stride2 * nrows;
    rc = tg_dmalloc(&cnt, nrows + 1) || tg_dmalloc(&off, nrows + 1);
    bool done = false;
    for (int attempt = 0; attempt < 6 && !rc && !done; attempt++) {
      rc = tg_dmalloc(&tcol, cap2 + TG_CSR_PAD) || tg_dmalloc(&tval, cap2 + TG_CSR_PAD);
      if (rc) break;
      P2.ts = plan->ts_k;
      P2.lgts = gw_lg(plan->ts_k);
      P2.out_stride = stride2;
      const size_t lds = 4 * gw_wave_bytes(P2.ts);
      if (lds > 160 * 1024) {
        rc = 100;
        break;
      }
      hipMemsetAsync(status, 0, 4 * sizeof(int), g_tg.stream);
      hipMemsetAsync(cnt, 0, (size_t)(nrows + 1) * sizeof(int64_t), g_tg.stream);
      gw_launch_shared<GW_BUMP>(plan->lg_am, gw_grid(nrows, P2.rows_per_wave), lds, P2, off, cnt, (uint32_t *)tcol, tval, cursor, cap2,
                                mask, diag, status, sum);
      int h[2] = {0, 0};
      hipMemcpyAsync(h, status, sizeof(h), hipMemcpyDeviceToHost, g_tg.stream);
      if (hipStreamSynchronize(g_tg.stream) != hipSuccess || hipGetLastError() != hipSuccess) {
        tg_set_error("cell-block PtAP: the gather stage failed to run (LDS %zu B)", lds);
        rc = 1;
        break;
      }
      if (h[0] == GW_OK) {
        done = true;
        break;
      }
      tg_dfree(tcol);
      tg_dfree(tval);
      tcol = nullptr;
      tval = nullptr;
      if (h[0] == GW_OVF)
        plan->ts_k *= 2;
      else if (h[0] == GW_CAP) {
        plan->max_k = std::max(plan->max_k, h[1]);
        stride2 = (std::max<int64_t>(h[1], stride2 + stride2 / 4) + 7) & ~(int64_t)7;
        cap2 = stride2 * nrows;
      } else {
        tg_set_error("cell-block PtAP: kernel status %d", h[0]);
        rc = 4;
      }
    }
    if (!rc && !done) rc = 100;
    if (!rc) {
      int64_t nnz = 0;
      rc = tg_exclusive_scan_i64(cnt, nrows, &nnz);
      if (!rc) rc = tg_csr_alloc(nrows, pl->ncols, nnz, &k);
      if (!rc) {
        hipMemcpyAsync(k->rowptr, cnt, (size_t)(nrows + 1) * sizeof(int64_t), hipMemcpyDeviceToDevice, g_tg.stream);
        const unsigned rg = (unsigned)std::min<int64_t>(tg_cdiv(nrows, 4), (int64_t)g_tg.num_cu * 16);
        hipLaunchKernelGGL(k_gw_reorder, dim3(std::max(1u, rg)), dim3(256), 0, g_tg.stream, k->rowptr, off, tcol, tval, nrows, k->col,
                           k->val);
        tg_dfree(plan->k_rowptr);
        plan->k_rowptr = cnt;
        cnt = nullptr;
        plan->k_nnz = nnz;
        // the places of the element entries in their rows of K, for all later products -- and for THIS one: its values
        // are formed again by places, so that every product on the plan adds in the same order (bit for bit the same K)
        if (plan->max_k < (pl->nfmax > 32 ? TG_CELL_ROWCAP : 256) && nnz < 0x7fffffffll * 4 && !getenv("TIGAR_CELL_MERGE_HASH")) {
          tg_dfree(pl->k_col);
          tg_dfree(pl->slot);
          pl->k_col = nullptr;
          pl->slot = nullptr;
          int *bad = status + 3;
          int hbad2 = 0;
          if (!tg_dmalloc(&pl->k_col, nnz + TG_CSR_PAD) && !tg_dmalloc(&pl->slot, nrowsE * pl->nfmax + 16)) {
            hipMemcpyAsync(pl->k_col, k->col, (size_t)nnz * sizeof(int32_t), hipMemcpyDeviceToDevice, g_tg.stream);
            hipMemsetAsync(bad, 0, sizeof(int), g_tg.stream);
            hipLaunchKernelGGL(k_cell_slots, dim3((unsigned)std::min<int64_t>(tg_cdiv(nrowsE * pl->nfmax, 256), (int64_t)g_tg.num_cu * 32)),
                               dim3(256), 0, g_tg.stream, pl->flmix, pl->nf, pl->ncell, pl->nfmax, k->rowptr, k->col, pl->slot, bad);
            hipMemcpyAsync(&hbad2, bad, sizeof(int), hipMemcpyDeviceToHost, g_tg.stream);
            if (hipStreamSynchronize(g_tg.stream) != hipSuccess || hipGetLastError() != hipSuccess) hbad2 = 1;
          } else
            hbad2 = 1;
          if (hbad2) {                       // (a row of more than 255 entries, or no memory: the look-up merge stays)
            tg_dfree(pl->k_col);
            tg_dfree(pl->slot);
            pl->k_col = nullptr;
            pl->slot = nullptr;
          } else
            rc = merge_by_places(k);
        }
      }
    }
  }
  hipStreamSynchronize(g_tg.stream);
  cleanup();
  if (rc) {
    if (k) tg_csr_destroy(k);
    return rc;
  }
  *k_out = k;
  return 0;
}

extern "C" int tg_cellplan_ptap(tg_cellplan_t pl, tg_csr_t a, const int32_t *zero_dofs, int64_t nzero, double diag, tg_csr_t *k_out) {
  TG_REQUIRE_INIT();
  TG_REQUIRE(pl && a && k_out, "null argument to tg_cellplan_ptap");
  TG_REQUIRE_CANONICAL(a);
  return tg_cellplan_ptap_impl(pl, a, nullptr, zero_dofs, nzero, diag, k_out);
}

// ---- cell blocks PLUS couplings outside them (demos/kl-shell-svk/reef-knot.py:455-467: contact terms added by hand) --------
// per row: how many entries lie outside its cell block, and where the block's b entries start (rows are sorted: they are
// consecutive); bad: a row without all b entries of its block
__global__ void __launch_bounds__(256)
    k_cellx_count(const int64_t *__restrict__ rowptr, const int32_t *__restrict__ col, int b, int64_t n, int64_t *__restrict__ len_r,
                  int64_t *__restrict__ rstart, int *__restrict__ bad) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t r = wave; r < n; r += nwaves) {
    const int64_t a = rowptr[r], e = rowptr[r + 1];
    const int64_t c0 = (r / b) * b;
    int in = 0, before = 0;
    for (int64_t q = a + lane; q < e; q += 64) {
      const int32_t c = col[q];
      in += (c >= c0 && c < c0 + b) ? 1 : 0;
      before += c < c0 ? 1 : 0;
    }
    for (int o = 32; o > 0; o >>= 1) {
      in += __shfl_xor(in, o, 64);
      before += __shfl_xor(before, o, 64);
    }
    if (lane == 0) {
      len_r[r] = (e - a) - in;
      rstart[r] = a + before;
      if (in != b) atomicOr(bad, 1);
    }
  }
}
__global__ void __launch_bounds__(256)
    k_cellx_rest(const int64_t *__restrict__ rowptr, const int32_t *__restrict__ col, const double *__restrict__ val, int b,
                 const int64_t *__restrict__ orowptr, int64_t n, int32_t *__restrict__ ocol, double *__restrict__ oval) {
  // (thread per row: the rows with anything to copy are few and hold few entries)
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < n; r += stride) {
    int64_t o = orowptr[r];
    if (orowptr[r + 1] == o) continue;
    const int64_t c0 = (r / b) * b;
    for (int64_t q = rowptr[r]; q < rowptr[r + 1]; q++)
      if (col[q] < c0 || col[q] >= c0 + b) {
        ocol[o] = col[q];
        oval[o] = val[q];
        o++;
      }
  }
}

extern "C" int tg_cellplan_ptap_extras(tg_cellplan_t pl, tg_csr_t a, tg_csr_t *k_out, tg_csr_t *r_out) {
  TG_REQUIRE_INIT();
  TG_REQUIRE(pl && a && k_out && r_out, "null argument to tg_cellplan_ptap_extras");
  TG_REQUIRE_CANONICAL(a);
  const int64_t n = pl->ncell * pl->b;
  if (a->nrows != n || a->ncols != n) return 100;
  int64_t *len_r = nullptr, *rstart = nullptr;
  int *bad = nullptr;
  tg_csr_s *mr = nullptr;
  int rc = tg_dmalloc(&len_r, n + 1) || tg_dmalloc(&rstart, n) || tg_dmalloc(&bad, 4);
  int h_bad = 0;
  int64_t tot_r = 0;
  auto cleanup = [&]() {
    tg_dfree(len_r);
    tg_dfree(rstart);
    tg_dfree(bad);
  };
  if (!rc) {
    const unsigned grid = (unsigned)std::max<int64_t>(1, std::min<int64_t>(tg_cdiv(n, 4), (int64_t)g_tg.num_cu * 16));
    hipMemsetAsync(bad, 0, sizeof(int), g_tg.stream);
    hipLaunchKernelGGL(k_cellx_count, dim3(grid), dim3(256), 0, g_tg.stream, a->rowptr, a->col, pl->b, n, len_r, rstart, bad);
    if (hipGetLastError() != hipSuccess) rc = 1;
    if (!rc && hipMemcpyAsync(&h_bad, bad, sizeof(int), hipMemcpyDeviceToHost, g_tg.stream) != hipSuccess) rc = 1;
  }
  if (!rc) rc = tg_exclusive_scan_i64(len_r, n, &tot_r);          // (synchronises: h_bad is valid afterwards)
  if (!rc && h_bad) {
    cleanup();
    return 100;                                                    // a row lacks entries of its own cell block
  }
  if (!rc) rc = tg_csr_alloc(n, n, tot_r, &mr);
  if (!rc) {
    if (hipMemcpyAsync(mr->rowptr, len_r, (size_t)(n + 1) * sizeof(int64_t), hipMemcpyDeviceToDevice, g_tg.stream) != hipSuccess) rc = 1;
    if (!rc && tot_r > 0) {
      hipLaunchKernelGGL(k_cellx_rest, dim3((unsigned)std::min<int64_t>(tg_cdiv(n, 256), (int64_t)g_tg.num_cu * 16)), dim3(256), 0,
                         g_tg.stream, a->rowptr, a->col, a->val, pl->b, mr->rowptr, n, mr->col, mr->val);
      if (hipGetLastError() != hipSuccess) rc = 1;
    }
  }
  tg_csr_t k = nullptr;
  if (!rc) rc = tg_cellplan_ptap_impl(pl, a, rstart, nullptr, 0, 1.0, &k);
  if (hipStreamSynchronize(g_tg.stream) != hipSuccess && !rc) rc = 1;
  cleanup();
  if (rc) {
    if (mr) tg_csr_destroy(mr);
    if (rc != 100) tg_set_error("tg_cellplan_ptap_extras failed");
    return rc;
  }
  mr->nnz = tot_r;
  *k_out = k;
  *r_out = mr;
  return 0;
}

// the rows of a matrix that hold entries, ascending: rows_host[0 .. min(*count, cap)) (two-call protocol: cap = 0 asks for
// the count); used to restrict a product with a nearly empty operand to the rows that matter
__global__ void __launch_bounds__(256) k_row_flags(const int64_t *__restrict__ rowptr, int64_t n, int64_t *__restrict__ flag) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < n; r += stride) flag[r] = rowptr[r + 1] > rowptr[r] ? 1 : 0;
}
__global__ void __launch_bounds__(256) k_row_compact(const int64_t *__restrict__ pos, int64_t n, int64_t *__restrict__ out) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < n; r += stride)
    if (pos[r + 1] > pos[r]) out[pos[r]] = r;
}
extern "C" int tg_csr_nonempty_rows(tg_csr_t a, int64_t cap, int64_t *rows_host, int64_t *count) {
  TG_REQUIRE_INIT();
  TG_REQUIRE(a && count && (rows_host || cap == 0), "bad arguments to tg_csr_nonempty_rows");
  TG_REQUIRE_CANONICAL(a);
  const int64_t n = a->nrows;
  int64_t *flag = nullptr, *out = nullptr;
  int64_t total = 0;
  int rc = tg_dmalloc(&flag, n + 1);
  const unsigned grid = (unsigned)std::max<int64_t>(1, std::min<int64_t>(tg_cdiv(n, 256), (int64_t)g_tg.num_cu * 16));
  if (!rc && n > 0) {
    hipLaunchKernelGGL(k_row_flags, dim3(grid), dim3(256), 0, g_tg.stream, a->rowptr, n, flag);
    if (hipGetLastError() != hipSuccess) rc = 1;
  }
  if (!rc) rc = tg_exclusive_scan_i64(flag, n, &total);
  *count = total;
  if (!rc && cap > 0 && total > 0) {
    const int64_t m = std::min(cap, total);
    rc = tg_dmalloc(&out, total);
    if (!rc) {
      hipLaunchKernelGGL(k_row_compact, dim3(grid), dim3(256), 0, g_tg.stream, flag, n, out);
      if (hipGetLastError() != hipSuccess ||
          hipMemcpyAsync(rows_host, out, (size_t)m * sizeof(int64_t), hipMemcpyDeviceToHost, g_tg.stream) != hipSuccess)
        rc = 1;
    }
  }
  if (hipStreamSynchronize(g_tg.stream) != hipSuccess) rc = 1;
  tg_dfree(flag);
  tg_dfree(out);
  if (rc) {
    tg_set_error("tg_csr_nonempty_rows failed");
    return 1;
  }
  return 0;
}

// ------------------------------------------------------------------------------------------------------------------
// Fold by places (round 4): K = R^T K_u R for a 0/1 matrix R with ONE entry per row -- the identification of the wrapped
// functions of a periodic patch after the tensor line walks ran on the unwrapped space (tigar_amd/kronptap.py:
// KronExtraction.unwrapped / fold; tIGAr/BSplines.py:204-212, 310-319: `% ncp` in getNodes).  The first product on a pattern
// goes through the general kernels (R as the extraction operator) and gives the pattern of K; the plan then stores for every
// entry of K_u its PLACE in the row of K it is added to (16 bits), and every later product -- the pattern of K_u is the
// closed-form tensor pattern, the same at every call -- is one wave per row of K adding its source rows into LDS accumulators
// at the stored places (distinct within a source row: plain read-modify-write; the source rows one after the other): no
// look-up, one pass over K_u, a fixed order of additions.
struct tg_foldplan_s {
  int64_t u_nrows = 0, u_ncols = 0, u_nnz = 0, u_row0 = 0;
  unsigned long long u_sum = 0;        // checksum of K_u's pattern
  int64_t nrows = 0, ncols = 0, nnz = 0, rt_row0 = 0;
  int max_k = 0;
  int64_t *k_rowptr = nullptr;
  int32_t *k_col = nullptr;
  uint16_t *place = nullptr;           // [u_nnz]
  tg_csr_s *rt = nullptr;              // (borrowed) rows of R^T = rows of K: the K_u rows that are added
};

__global__ void __launch_bounds__(256) k_fold_checksum(const int64_t *__restrict__ rowptr, const int32_t *__restrict__ col, int64_t nrows,
                                                       int64_t nnz, unsigned long long *__restrict__ out) {
  unsigned long long s = 0;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < nnz; e += stride)
    s += (unsigned long long)(unsigned)col[e] * (unsigned long long)(2 * e + 1);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i <= nrows; i += stride)
    s += (unsigned long long)rowptr[i] * 0x9E3779B97F4A7C15ull + (unsigned long long)i;
  for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
  if ((threadIdx.x & 63) == 0) atomicAdd(out, s);
}

// place of every entry of the K_u rows named by R^T: one wave per row of K, its source rows in turn
__global__ void __launch_bounds__(256)
    k_fold_places(const int64_t *__restrict__ trowptr, const int32_t *__restrict__ tcol, int64_t nrows, int64_t u_row0,
                  const int64_t *__restrict__ urowptr, const int32_t *__restrict__ ucol, const int32_t *__restrict__ map,
                  const int64_t *__restrict__ krowptr, const int32_t *__restrict__ kcol, uint16_t *__restrict__ place, int *__restrict__ bad) {
  // (two entries of ONE source row with the same place -- columns g and g + n of an unwrapped row that are both stored, which
  //  the tensor pattern of a patch with at least 2p+1 elements per periodic direction never has -- would make the plain
  //  read-modify-write of k_fold_apply lose an addend: found here with a bitmap of the row of K, no plan then)
  __shared__ unsigned seen_all[4][2048];
  const int lane = threadIdx.x & 63;
  unsigned *seen = seen_all[threadIdx.x >> 6];
  const int64_t wstride = (int64_t)gridDim.x * 4;
  for (int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); i < nrows; i += wstride) {
    const int64_t k0 = krowptr[i], k1 = krowptr[i + 1];
    const int words = (int)((k1 - k0 + 31) >> 5);
    for (int64_t x = trowptr[i]; x < trowptr[i + 1]; x++) {
      const int64_t g = (int64_t)tcol[x] - u_row0;
      for (int w = lane; w < words; w += 64) seen[w] = 0u;
      for (int64_t e = urowptr[g] + lane; e < urowptr[g + 1]; e += 64) {
        const int32_t j = map[ucol[e]];
        int64_t lo = k0, hi = k1;
        while (lo < hi) {
          const int64_t mid = (lo + hi) >> 1;
          if (kcol[mid] < j) lo = mid + 1; else hi = mid;
        }
        if (lo < k1 && kcol[lo] == j && lo - k0 < 65535) {
          const unsigned pos = (unsigned)(lo - k0);
          place[e] = (uint16_t)pos;
          if (atomicOr(&seen[pos >> 5], 1u << (pos & 31)) & (1u << (pos & 31))) atomicOr(bad, 1);
        } else {
          place[e] = 0xffff;
          atomicOr(bad, 1);
        }
      }
    }
  }
}

__global__ void __launch_bounds__(256)
    k_fold_apply(const int64_t *__restrict__ trowptr, const int32_t *__restrict__ tcol, int64_t nrows, int64_t u_row0, int64_t row0,
                 const int64_t *__restrict__ urowptr, const double *__restrict__ uval, const uint16_t *__restrict__ place,
                 const int64_t *__restrict__ krowptr, const int32_t *__restrict__ kcol, int max_k, const uint8_t *__restrict__ mask,
                 double diag, double *__restrict__ kval) {
  extern __shared__ double fold_acc[];
  const int lane = threadIdx.x & 63;
  double *acc = fold_acc + (size_t)(threadIdx.x >> 6) * max_k;
  const int64_t wstride = (int64_t)gridDim.x * 4;
  for (int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); i < nrows; i += wstride) {
    const int64_t k0 = krowptr[i];
    const int n = (int)(krowptr[i + 1] - k0);
    for (int e = lane; e < n; e += 64) acc[e] = 0.0;
    for (int64_t x = trowptr[i]; x < trowptr[i + 1]; x++) {
      const int64_t g = (int64_t)tcol[x] - u_row0;
      for (int64_t e = urowptr[g] + lane; e < urowptr[g + 1]; e += 64) acc[place[e]] += uval[e];
    }
    const int64_t R = row0 + i;
    const bool mrow = mask && mask[R];
    for (int e = lane; e < n; e += 64) {
      double v = acc[e];
      if (mask) {
        const int32_t c = kcol[k0 + e];
        if (mrow || mask[c]) v = (mrow && c == R) ? diag : 0.0;
      }
      kval[k0 + e] = v;
    }
  }
}

static int fold_checksum(tg_csr_s *ku, unsigned long long *out) {
  unsigned long long *d = (unsigned long long *)(g_tg.scratch + 64);
  TG_CHECK_HIP(hipMemsetAsync(d, 0, sizeof(unsigned long long), g_tg.stream));
  hipLaunchKernelGGL(k_fold_checksum, dim3((unsigned)std::min<int64_t>(tg_cdiv(std::max<int64_t>(ku->nnz, 1), 256), (int64_t)g_tg.num_cu * 32)),
                     dim3(256), 0, g_tg.stream, ku->rowptr, ku->col, ku->nrows, ku->nnz, d);
  TG_LAUNCH_CHECK();
  TG_CHECK_HIP(hipMemcpyAsync(out, d, sizeof(unsigned long long), hipMemcpyDeviceToHost, g_tg.stream));
  TG_CHECK_HIP(hipStreamSynchronize(g_tg.stream));
  return 0;
}

extern "C" int tg_foldplan_destroy(tg_foldplan_t pl) {
  if (!pl) return 0;
  if (g_tg.ready) hipStreamSynchronize(g_tg.stream);
  tg_dfree(pl->k_rowptr);
  tg_dfree(pl->k_col);
  tg_dfree(pl->place);
  delete pl;
  return 0;
}

// k: the product R^T K_u R as the general kernels gave it (its PATTERN is taken); returns 100 when a row of K holds 65 535
// entries or more (no plan: keep using the general kernels)
extern "C" int tg_foldplan_create(tg_csr_t ku, int64_t ku_row0, tg_csr_t r, tg_csr_t rt, int64_t rt_row0, tg_csr_t k, tg_foldplan_t *out) {
  TG_REQUIRE_INIT();
  TG_REQUIRE(ku && r && rt && k && out, "null argument to tg_foldplan_create");
  TG_REQUIRE(r->nnz == r->nrows && r->nrows == ku->ncols && rt->ncols == r->nrows && k->nrows == rt->nrows && k->ncols == r->ncols,
             "tg_foldplan_create: operands do not fit");
  TG_REQUIRE_CANONICAL(ku);
  TG_REQUIRE_CANONICAL(k);
  tg_foldplan_s *pl = new tg_foldplan_s();
  pl->u_nrows = ku->nrows;
  pl->u_ncols = ku->ncols;
  pl->u_nnz = ku->nnz;
  pl->u_row0 = ku_row0;
  pl->nrows = k->nrows;
  pl->ncols = k->ncols;
  pl->nnz = k->nnz;
  pl->rt_row0 = rt_row0;
  pl->rt = rt;
  int rc = tg_spmv_plan(k);
  pl->max_k = std::max(1, k->max_row_nnz);
  if (!rc && (pl->max_k >= 65535 || 4 * (size_t)pl->max_k * sizeof(double) > 160 * 1024)) rc = 100;
  if (!rc) rc = tg_dmalloc(&pl->k_rowptr, k->nrows + 1) || tg_dmalloc(&pl->k_col, k->nnz + TG_CSR_PAD) || tg_dmalloc(&pl->place, ku->nnz + 8);
  if (!rc) rc = fold_checksum(ku, &pl->u_sum);
  if (!rc) {
    hipMemcpyAsync(pl->k_rowptr, k->rowptr, (size_t)(k->nrows + 1) * sizeof(int64_t), hipMemcpyDeviceToDevice, g_tg.stream);
    hipMemcpyAsync(pl->k_col, k->col, (size_t)k->nnz * sizeof(int32_t), hipMemcpyDeviceToDevice, g_tg.stream);
    int *bad = (int *)g_tg.scratch;
    hipMemsetAsync(bad, 0, sizeof(int), g_tg.stream);
    hipLaunchKernelGGL(k_fold_places, dim3((unsigned)std::min<int64_t>(tg_cdiv(std::max<int64_t>(k->nrows, 1), 4), (int64_t)g_tg.num_cu * 32)),
                       dim3(256), 0, g_tg.stream, rt->rowptr, rt->col, k->nrows, ku_row0, ku->rowptr, ku->col, r->col, pl->k_rowptr, pl->k_col,
                       pl->place, bad);
    int hbad = 0;
    hipMemcpyAsync(&hbad, bad, sizeof(int), hipMemcpyDeviceToHost, g_tg.stream);
    if (hipStreamSynchronize(g_tg.stream) != hipSuccess || hipGetLastError() != hipSuccess) {
      tg_set_error("tg_foldplan_create: kernels failed");
      rc = 1;
    } else if (hbad)
      rc = 100;                              // an entry of K_u without a place in K: not the product's pattern
  }
  if (rc) {
    tg_foldplan_destroy(pl);
    return rc;
  }
  *out = pl;
  return 0;
}

// K = R^T K_u R on the plan's patterns, MatZeroRowsColumns fused; 100 = K_u has another pattern than the plan's
extern "C" int tg_foldplan_apply(tg_foldplan_t pl, tg_csr_t ku, const int32_t *zero_dofs, int64_t nzero, double diag, tg_csr_t *k_out) {
  TG_REQUIRE_INIT();
  TG_REQUIRE(pl && ku && k_out, "null argument to tg_foldplan_apply");
  if (ku->nrows != pl->u_nrows || ku->ncols != pl->u_ncols || ku->nnz != pl->u_nnz || ku->rowcnt) return 100;
  unsigned long long sum = 0;
  TG_TRY(fold_checksum(ku, &sum));
  if (sum != pl->u_sum) return 100;
  uint8_t *mask = nullptr;
  if (nzero > 0) TG_TRY(tg_build_dof_mask(zero_dofs, nzero, pl->ncols, &mask));
  tg_csr_s *k = nullptr;
  int rc = tg_csr_alloc(pl->nrows, pl->ncols, pl->nnz, &k);
  if (!rc) {
    hipMemcpyAsync(k->rowptr, pl->k_rowptr, (size_t)(pl->nrows + 1) * sizeof(int64_t), hipMemcpyDeviceToDevice, g_tg.stream);
    hipMemcpyAsync(k->col, pl->k_col, (size_t)pl->nnz * sizeof(int32_t), hipMemcpyDeviceToDevice, g_tg.stream);
    const size_t lds = 4 * (size_t)pl->max_k * sizeof(double);
    hipFuncSetAttribute((const void *)k_fold_apply, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipLaunchKernelGGL(k_fold_apply, dim3((unsigned)std::min<int64_t>(tg_cdiv(std::max<int64_t>(pl->nrows, 1), 4), (int64_t)g_tg.num_cu * 64)),
                       dim3(256), lds, g_tg.stream, pl->rt->rowptr, pl->rt->col, pl->nrows, pl->u_row0, pl->rt_row0, ku->rowptr, ku->val,
                       pl->place, k->rowptr, k->col, pl->max_k, mask, diag, k->val);
    if (hipGetLastError() != hipSuccess || hipStreamSynchronize(g_tg.stream) != hipSuccess) {
      tg_set_error("tg_foldplan_apply: the kernel failed");
      rc = 1;
    }
  }
  tg_dfree(mask);
  if (rc) {
    if (k) tg_csr_destroy(k);
    return rc;
  }
  *k_out = k;
  return 0;
}
