// extractMatrix (tIGAr/common.py:1176-1204: A.PtAP(M) + MatZeroRowsColumns) for ARBITRARY sparse operands on a CONNECTED FE
// mesh, third generation (round 6): the element split, everything on the device, in chunks of cells.
//
// A matrix assembled on a mesh whose cells share nodes is a sum of element matrices; ANY splitting A = sum_c R_c^T A_c R_c of
// its entries over the cells that hold both nodes gives
//
//     K = M^T A M = sum_c (R_c M)^T A_c (R_c M)
//
// -- dense little triple products without a look-up, merged into K by stored places.  What is used besides the CSR arrays of
// A and M is the cells' node lists (dolfin's V.dofmap().cell_dofs(c); tg_cells_*), nothing about a lattice.  An entry of A whose
// nodes share no cell (a contact / penalty coupling added by hand, demos/kl-shell-svk/reef-knot.py:455-467) makes the product
// return status 100: the caller takes the row-wise kernels (tg_ptap_*).
//
// Round 5 built the plan on the host (numpy unique / argsort / scipy products: 2.7 s at 64^3 p = 3, M downloaded) and split A
// cell by cell (every cell scanned all entries of its nodes' rows: 36 ms).  Here:
//
//   k_el_node_*   node -> (cell, position) lists, ascending by cell           (the transposed dofmap)
//   k_el_fl       function list of a cell = union of the columns of its nodes' rows of M: a 64-way merge of sorted rows, one
//                 wave per cell, the minimum of the 64 heads by DPP
//   k_el_inc_*    function -> rows (c, q) of the element matrices that hold it, ascending
//   k_el_scatter  one pass over the ENTRIES of A (one wave per row): the lowest listed cell that holds both nodes owns the
//                 entry; its value goes straight to its place in the cell's dense block -- no map is stored, the pass costs
//                 about what reading a map would
//   k_el_dense    E_c = M_c^T (A_c M_c), one cell per workgroup, register tiles; M_c gathered from the CSR rows of M into LDS
//                 by the kernel itself (no dense copy of M per cell in HBM), E_c written over A_c
//   k_el_rowsym   the pattern of a row of K = union of the function lists of the cells that hold its function: the same
//                 merge with the lists staged in LDS; the rank of every list entry in the union IS its place in the row
//   k_el_merge    values of K by places: no look-up, fixed order of additions (K is bit-reproducible)
//
// Chunks: the listed cells are in ascending PRIORITY order, [own0, own1) of them are computed here, the others only take part
// in the ownership rule (cells of a neighbouring chunk that share nodes with this one).  K of a chunk holds the rows of its
// own cells' functions; the caller adds the chunks (rows touched by one chunk only are final).  A chunk sees the rows of A
// and M of its own cells' nodes ([a_row0, a_row0 + nrows)); every entry of the rows [check0, check1) must have a listed
// common cell (rows whose cells are all listed) -- the others are some other chunk's to check.
#include "tg_common.h"
#include <algorithm>

#define EL_FLS 128            // stride of a cell's function list: at most 128 functions per cell
#define EL_INF 0x7fffffff
#define EL_POSBITS 7          // node -> cell list entries: (cell << 7) | position of the node in the cell
#define EL_MAXNODES 128
#define EL_MAXROW 1024        // longest row of K (3-D p = 4: 729)
#define EL_BLOCK 4096         // entries of the temporary a wave reserves at a time (one atomic per block, not per row)

struct tg_cells_s {
  int64_t ncell = 0;
  int b = 0;
  int32_t *nodes = nullptr;   // device [ncell][b]
};

struct tg_elemplan_s {
  tg_cells_s *cells = nullptr;   // borrowed
  tg_csr_s *m = nullptr;         // borrowed: rows [m_row0, m_row0 + m->nrows) of M, global columns
  int64_t m_row0 = 0;
  int64_t own0 = 0, own1 = 0;
  int b = 0, S = 0, nfmax = 0, ninc_max = 0;
  int64_t node0 = 0, nnode = 0;
  int64_t *nptr = nullptr;       // [nnode + 1]
  int32_t *nlist = nullptr;      // (cell << 7 | pos), ascending per node
  int32_t *nfix = nullptr;       // [nnode][8] the same as a table, when no node lies in more than 8 cells
  int32_t *fl = nullptr, *nf = nullptr;   // own cells: [nown][EL_FLS], [nown]
  uint8_t *mpos = nullptr;                // [nown][b][64]: position in the cell's list of the first 64 entries of every node's row
                                          // of M, entry e at byte (e % 4) * 16 + e / 4 (the 16 entries a thread gathers are adjacent)
  int64_t dof0 = 0, dof1 = 0;
  int64_t *iptr = nullptr;       // [ndof + 1]
  int32_t *ient = nullptr;       // c_own * S + q, ascending per function
  // after the first product
  int64_t k_nnz = -1;
  int max_k = 0;
  int64_t *k_rowptr = nullptr;
  int32_t *k_col = nullptr;
  uint16_t *slot = nullptr;      // [nown * S][S]
};

#define EL_WAVE_SYNC()                                   \
  do {                                                   \
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); \
    __builtin_amdgcn_wave_barrier();                     \
  } while (0)

// Atomics of many waves on ONE address are a serial resource (about 12 ns each at the L2, whatever the value): statistics are
// read first and updated only when that changes them.
__device__ __forceinline__ void el_stat_min(int *p, int v) {
  if (v < __builtin_nontemporal_load(p)) atomicMin(p, v);
}
__device__ __forceinline__ void el_stat_max(int *p, int v) {
  if (v > __builtin_nontemporal_load(p)) atomicMax(p, v);
}

// minimum over the 64 lanes, in every lane (all lanes must be active): butterflies inside the rows of 16 lanes by DPP (xor 1,
// xor 2, half-row mirror, row mirror), then the four rows through scalar registers
__device__ __forceinline__ int el_wave_min(int v) {
  v = min(v, __builtin_amdgcn_update_dpp(v, v, 0xB1, 0xf, 0xf, false));     // quad_perm [1,0,3,2]
  v = min(v, __builtin_amdgcn_update_dpp(v, v, 0x4E, 0xf, 0xf, false));     // quad_perm [2,3,0,1]
  v = min(v, __builtin_amdgcn_update_dpp(v, v, 0x141, 0xf, 0xf, false));    // row_half_mirror
  v = min(v, __builtin_amdgcn_update_dpp(v, v, 0x140, 0xf, 0xf, false));    // row_mirror
  const int a = __builtin_amdgcn_readlane(v, 0), b = __builtin_amdgcn_readlane(v, 16), c = __builtin_amdgcn_readlane(v, 32),
            d = __builtin_amdgcn_readlane(v, 48);
  return min(min(a, b), min(c, d));
}

// The merge both symbolic kernels run: sorted lists staged in LDS (list x at L + x * LD), lane `lane` walks list `lane` (n0
// entries) and, with TWO, list lane + 64 (n1 entries).  Every step takes the smallest head m (the next element of the union)
// and stores it in U[k]; with PLACES the lanes whose head it is store the step number in place of the list entry (its rank in
// the union).  A lane keeps a WINDOW of its next eight entries in registers, refilled every eight steps (a lane moves on at
// most once per step): one LDS round trip per eight steps instead of one per step on the chain min -> compare -> min
// (measured: 800 cycles per step with the head read from LDS in the step).  Returns the number of steps (= size of the union),
// -1 when that exceeds ucap.  All lanes must be active.
template <bool PLACES, bool TWO>
__device__ __forceinline__ int el_merge_staged(int32_t *L, int LD, int lane, int n0, int n1, int32_t *U, int ucap) {
  int32_t *L0 = L + lane * LD, *L1 = L + (lane + 64) * LD;
  int c0 = 0, c1 = 0, k = 0;
  for (;;) {
    int w[8], v[8];
#pragma unroll
    for (int j = 0; j < 8; j++) {
      const int t = L0[max(min(c0 + j, n0 - 1), 0)];
      w[j] = c0 + j < n0 ? t : EL_INF;
    }
    if (TWO) {
#pragma unroll
      for (int j = 0; j < 8; j++) {
        const int t = L1[max(min(c1 + j, n1 - 1), 0)];
        v[j] = c1 + j < n1 ? t : EL_INF;
      }
    }
    // (opaque to the compiler: it otherwise proves w[j] == L0[c0 + j], carries seven values around the loop and reads the
    //  eighth inside the step -- the LDS round trip back on the chain)
    asm volatile("" : "+v"(w[0]), "+v"(w[1]), "+v"(w[2]), "+v"(w[3]), "+v"(w[4]), "+v"(w[5]), "+v"(w[6]), "+v"(w[7]));
    if (TWO) asm volatile("" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]));
#pragma unroll
    for (int it = 0; it < 8; it++) {
      const int m = el_wave_min(TWO ? min(w[0], v[0]) : w[0]);
      if (m == EL_INF) return k;
      if (k >= ucap) return -1;
      const bool a0 = w[0] == m;
      if (PLACES && a0) L0[c0] = k;
      c0 += a0 ? 1 : 0;
#pragma unroll
      for (int j = 0; j < 7; j++) w[j] = a0 ? w[j + 1] : w[j];
      if (TWO) {
        const bool a1 = v[0] == m;
        if (PLACES && a1) L1[c1] = k;
        c1 += a1 ? 1 : 0;
#pragma unroll
        for (int j = 0; j < 7; j++) v[j] = a1 ? v[j + 1] : v[j];
      }
      U[k] = m;            // (every lane, one address, one value)
      k++;
    }
  }
}

// ---- the cells' node lists ------------------------------------------------------------------------------------------
extern "C" int tg_cells_from_host(const int32_t *nodes_host, int64_t ncell, int b, tg_cells_t *out) {
  TG_REQUIRE_INIT();
  TG_REQUIRE(nodes_host && out && ncell > 0 && b >= 1 && b <= EL_MAXNODES, "bad arguments to tg_cells_from_host");
  tg_cells_s *c = new tg_cells_s();
  c->ncell = ncell, c->b = b;
  if (tg_dmalloc(&c->nodes, ncell * b)) {
    delete c;
    return 1;
  }
  if (hipMemcpyAsync(c->nodes, nodes_host, (size_t)(ncell * b) * sizeof(int32_t), hipMemcpyHostToDevice, g_tg.stream) != hipSuccess ||
      hipStreamSynchronize(g_tg.stream) != hipSuccess) {
    tg_set_error("tg_cells_from_host: upload failed");
    tg_dfree(c->nodes);
    delete c;
    return 1;
  }
  *out = c;
  return 0;
}

struct el_grid_args {
  int d, p;
  int64_t nn[3];      // nodes per direction
  int64_t lo[3], ne[3];   // the box of elements: first element and count per direction
};
__global__ void __launch_bounds__(256) k_cells_grid(el_grid_args g, int b, int64_t total, int32_t *__restrict__ out) {
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < total; t += stride) {
    int64_t c = t / b;
    int a = (int)(t - c * b);
    int64_t node = 0, st = 1;
    for (int k = 0; k < g.d; k++) {
      const int64_t e = g.lo[k] + c % g.ne[k];
      c /= g.ne[k];
      const int ak = a % (g.p + 1);
      a /= (g.p + 1);
      node += (e * g.p + ak) * st;
      st *= g.nn[k];
    }
    out[t] = (int32_t)node;
  }
}

/* The dofmap of a continuous Q_p space on a tensor-product grid of nodes (this package's FE-side stand-in, TensorNodeGrid:
 * nodes_per_dir[k] = p * elements + 1 nodes per direction, direction 0 fastest): the cells of the box elem_lo[k] <= e_k <
 * elem_hi[k], direction 0 fastest, local nodes direction 0 fastest. */
extern "C" int tg_cells_from_grid(int d, const int64_t *nodes_per_dir, int degree, const int64_t *elem_lo, const int64_t *elem_hi,
                                  tg_cells_t *out) {
  TG_REQUIRE_INIT();
  TG_REQUIRE(d >= 1 && d <= 3 && nodes_per_dir && elem_lo && elem_hi && out && degree >= 1, "bad arguments to tg_cells_from_grid");
  el_grid_args g;
  memset(&g, 0, sizeof(g));
  g.d = d, g.p = degree;
  int64_t ncell = 1, nnodes = 1;
  int b = 1;
  for (int k = 0; k < d; k++) {
    TG_REQUIRE(elem_lo[k] >= 0 && elem_hi[k] > elem_lo[k] && elem_hi[k] * degree + 1 <= nodes_per_dir[k],
               "tg_cells_from_grid: the box of elements leaves the grid");
    g.nn[k] = nodes_per_dir[k], g.lo[k] = elem_lo[k], g.ne[k] = elem_hi[k] - elem_lo[k];
    ncell *= g.ne[k];
    nnodes *= nodes_per_dir[k];
    b *= degree + 1;
    TG_REQUIRE(b <= EL_MAXNODES, "tg_cells_from_grid: more than %d nodes per cell", EL_MAXNODES);
  }
  TG_REQUIRE(nnodes < 0x7fffffffll, "tg_cells_from_grid: node numbers beyond 32 bits");
  tg_cells_s *c = new tg_cells_s();
  c->ncell = ncell, c->b = b;
  if (tg_dmalloc(&c->nodes, ncell * b)) {
    delete c;
    return 1;
  }
  hipLaunchKernelGGL(k_cells_grid, dim3(tg_grid_1d(ncell * b, 256)), dim3(256), 0, g_tg.stream, g, b, ncell * b, c->nodes);
  if (hipGetLastError() != hipSuccess) {
    tg_set_error("tg_cells_from_grid: launch failed");
    tg_dfree(c->nodes);
    delete c;
    return 1;
  }
  *out = c;
  return 0;
}

extern "C" int tg_cells_dims(tg_cells_t c, int64_t *ncell, int *b) {
  TG_REQUIRE(c, "null cells");
  if (ncell) *ncell = c->ncell;
  if (b) *b = c->b;
  return 0;
}

extern "C" int tg_cells_download(tg_cells_t c, int32_t *nodes_host) {
  TG_REQUIRE_INIT();
  TG_REQUIRE(c && nodes_host, "null argument to tg_cells_download");
  TG_CHECK_HIP(hipMemcpyAsync(nodes_host, c->nodes, (size_t)(c->ncell * c->b) * sizeof(int32_t), hipMemcpyDeviceToHost, g_tg.stream));
  TG_CHECK_HIP(hipStreamSynchronize(g_tg.stream));
  return 0;
}

extern "C" int tg_cells_destroy(tg_cells_t c) {
  if (!c) return 0;
  if (g_tg.ready) {
    hipStreamSynchronize(g_tg.stream);
    tg_dfree(c->nodes);
  }
  delete c;
  return 0;
}

// ---- node -> cells ---------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_el_minmax(const int32_t *__restrict__ v, int64_t n, int *__restrict__ mm) {
  const int64_t stride = (int64_t)gridDim.x * 256;
  int lo = EL_INF, hi = -1;
  for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < n; t += stride) {
    const int x = v[t];
    lo = min(lo, x);
    hi = max(hi, x);
  }
  for (int o = 32; o > 0; o >>= 1) {
    lo = min(lo, __shfl_xor(lo, o, 64));
    hi = max(hi, __shfl_xor(hi, o, 64));
  }
  if ((threadIdx.x & 63) == 0) {
    el_stat_min(mm, lo);
    el_stat_max(mm + 1, hi);
  }
}

__global__ void __launch_bounds__(256)
    k_el_node_count(const int32_t *__restrict__ cn, int64_t total, int64_t node0, int64_t *__restrict__ cnt) {
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < total; t += stride)
    atomicAdd((unsigned long long *)(cnt + (cn[t] - node0)), 1ull);
}

__global__ void __launch_bounds__(256) k_el_node_fill(const int32_t *__restrict__ cn, int64_t total, int b, int64_t node0,
                                                      const int64_t *__restrict__ nptr, int32_t *__restrict__ cur,
                                                      int32_t *__restrict__ nlist) {
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < total; t += stride) {
    const int64_t n = cn[t] - node0, c = t / b;
    const int pos = (int)(t - c * b);
    const int q = atomicAdd(cur + n, 1);
    nlist[nptr[n] + q] = (int32_t)((c << EL_POSBITS) | pos);
  }
}

// ascending per node (insertion sort by one thread: a node belongs to a handful of cells); bad: a node twice in one cell
__global__ void __launch_bounds__(256)
    k_el_node_sort(const int64_t *__restrict__ nptr, int64_t nnode, int32_t *__restrict__ nlist, int *__restrict__ bad) {
  const int64_t stride = (int64_t)gridDim.x * 256;
  int longest = 0;
  for (int64_t n = (int64_t)blockIdx.x * 256 + threadIdx.x; n < nnode; n += stride) {
    const int64_t a = nptr[n], e = nptr[n + 1];
    longest = max(longest, (int)(e - a));
    for (int64_t i = a + 1; i < e; i++) {
      const int32_t v = nlist[i];
      int64_t j = i - 1;
      while (j >= a && nlist[j] > v) {
        nlist[j + 1] = nlist[j];
        j--;
      }
      nlist[j + 1] = v;
    }
    for (int64_t i = a + 1; i < e; i++)
      if ((nlist[i] >> EL_POSBITS) == (nlist[i - 1] >> EL_POSBITS)) atomicOr(bad, 1);
  }
  for (int o = 32; o > 0; o >>= 1) longest = max(longest, __shfl_xor(longest, o, 64));
  if ((threadIdx.x & 63) == 0) el_stat_max(bad - 1, longest);          // (stats[2]: the longest list of a node)
}

// ---- function lists of the own cells: union of the columns of the rows of M of the cell's nodes ------------------------
// one wave per cell: the rows of M of its nodes staged in LDS (coalesced), then el_merge_staged
// stats: [0] smallest function, [1] largest function, [2] longest list, [3] bad (1: a node outside the rows of M, 2: more than
// EL_FLS functions in a cell or in a row of M)
template <bool TWO>
__global__ void __launch_bounds__(256)
    k_el_fl(const int64_t *__restrict__ mrowptr, const int32_t *__restrict__ mcol, int64_t m_row0, int64_t m_nrows,
            const int32_t *__restrict__ cn, int64_t own0, int64_t nown, int b, int LD, int wave_words, int waves,
            int32_t *__restrict__ fl, int32_t *__restrict__ nf, uint8_t *__restrict__ mpos, int *__restrict__ stats, int dbg) {
  extern __shared__ __attribute__((aligned(16))) int32_t el_smem[];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (w >= waves) return;
  int32_t *L = el_smem + (size_t)w * wave_words;
  int32_t *U = L + (size_t)wave_words - EL_FLS;
  const int64_t nw = (int64_t)gridDim.x * waves;
  int s_first = EL_INF, s_last = -1, s_nf = 0;
  for (int64_t c = (int64_t)blockIdx.x * waves + w; c < nown; c += nw) {
    int64_t p0 = 0, p1 = 0;
    int n0 = 0, n1 = 0;
    bool bad = false, longrow = false;
    if (lane < b) {
      const int64_t r = cn[(own0 + c) * b + lane] - m_row0;
      if (r < 0 || r >= m_nrows) bad = true;
      else {
        p0 = mrowptr[r];
        const int64_t len = mrowptr[r + 1] - p0;
        if (len >= LD) longrow = true;
        n0 = (int)len;
      }
    }
    if (TWO && lane + 64 < b) {
      const int64_t r = cn[(own0 + c) * b + lane + 64] - m_row0;
      if (r < 0 || r >= m_nrows) bad = true;
      else {
        p1 = mrowptr[r];
        const int64_t len = mrowptr[r + 1] - p1;
        if (len >= LD) longrow = true;
        n1 = (int)len;
      }
    }
    if (__any(bad || longrow)) {
      if (lane == 0) atomicOr(stats + 3, __any(bad) ? 1 : 2), nf[c] = 0;
      continue;
    }
    EL_WAVE_SYNC();
    for (int xb = 0; xb < b; xb += 16) {        // (sixteen rows in flight: loads first -- the column arrays are padded by
      int32_t ta[16], tb[16];                   //  TG_CSR_PAD entries, reading past a row's end is safe -- then the LDS stores)
      int nx[16];
#pragma unroll
      for (int u = 0; u < 16; u++) {
        const int x = xb + u;
        const int64_t pa = __shfl(p0, x & 63, 64), pb = TWO ? __shfl(p1, x & 63, 64) : 0;
        const int na = __shfl(n0, x & 63, 64), nb = TWO ? __shfl(n1, x & 63, 64) : 0;
        const int64_t px = x < 64 ? pa : pb;
        nx[u] = x >= b ? 0 : (x < 64 ? na : nb);
        ta[u] = mcol[px + lane];
        tb[u] = LD > 65 ? mcol[px + lane + 64] : 0;
      }
#pragma unroll
      for (int u = 0; u < 16; u++) {
        const int x = xb + u;
        if (lane < nx[u]) L[x * LD + lane] = ta[u];
        if (lane + 64 < nx[u]) L[x * LD + lane + 64] = tb[u];
      }
    }
    if (dbg == 1) continue;
    EL_WAVE_SYNC();
    const int k = dbg == 2 ? 0 : el_merge_staged<true, TWO>(L, LD, lane, n0, n1, U, EL_FLS);
    if (k < 0) {
      if (lane == 0) atomicOr(stats + 3, 2), nf[c] = 0;
      continue;
    }
    EL_WAVE_SYNC();
    // the ranks the merge left in place of the rows' entries = their positions in the list (the first 64 of a row; the
    // kernel that gathers M_c looks the others up)
    for (int x = 0; x < b; x++) {
      const int nx = x < 64 ? __shfl(n0, x & 63, 64) : (TWO ? __shfl(n1, x & 63, 64) : 0);
      if (lane < nx) mpos[((c * b + x) << 6) + ((lane & 3) << 4) + (lane >> 2)] = (uint8_t)L[x * LD + lane];
    }
    if (lane < k) fl[c * EL_FLS + lane] = U[lane];
    if (lane + 64 < k) fl[c * EL_FLS + lane + 64] = U[lane + 64];
    if (lane == 0) nf[c] = k;
    if (k > 0) s_first = min(s_first, U[0]), s_last = max(s_last, U[k - 1]), s_nf = max(s_nf, k);
  }
  if (lane == 0 && s_nf > 0) {
    el_stat_min(stats, s_first);
    el_stat_max(stats + 1, s_last);
    el_stat_max(stats + 2, s_nf);
  }
}

__global__ void __launch_bounds__(256) k_el_maxrow(const int64_t *__restrict__ rowptr, int64_t nrows, int *__restrict__ out) {
  const int64_t stride = (int64_t)gridDim.x * 256;
  int m = 0;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nrows; i += stride)
    m = max(m, (int)min(rowptr[i + 1] - rowptr[i], (int64_t)0x7fffffff));
  for (int o = 32; o > 0; o >>= 1) m = max(m, __shfl_xor(m, o, 64));
  if ((threadIdx.x & 63) == 0) el_stat_max(out, m);
}

// ---- function -> rows of the element matrices ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_el_inc_count(const int32_t *__restrict__ fl, const int32_t *__restrict__ nf, int64_t nown,
                                                      int64_t dof0, int64_t *__restrict__ cnt) {
  const int64_t stride = (int64_t)gridDim.x * 256, total = nown * EL_FLS;
  for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < total; t += stride) {
    const int64_t c = t / EL_FLS;
    const int q = (int)(t - c * EL_FLS);
    if (q < nf[c]) atomicAdd((unsigned long long *)(cnt + (fl[t] - dof0)), 1ull);
  }
}
__global__ void __launch_bounds__(256)
    k_el_inc_fill(const int32_t *__restrict__ fl, const int32_t *__restrict__ nf, int64_t nown, int S, int64_t dof0,
                  const int64_t *__restrict__ iptr, int32_t *__restrict__ cur, int32_t *__restrict__ ient) {
  const int64_t stride = (int64_t)gridDim.x * 256, total = nown * EL_FLS;
  for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < total; t += stride) {
    const int64_t c = t / EL_FLS;
    const int q = (int)(t - c * EL_FLS);
    if (q < nf[c]) {
      const int64_t i = fl[t] - dof0;
      const int pos = atomicAdd(cur + i, 1);
      ient[iptr[i] + pos] = (int32_t)(c * S + q);
    }
  }
}
// ascending per function: rank sort by one wave (keys are distinct); stat: longest list
__global__ void __launch_bounds__(256)
    k_el_inc_sort(const int64_t *__restrict__ iptr, int64_t ndof, int32_t *__restrict__ ient, int *__restrict__ stat) {
  __shared__ int32_t keys[4][2 * 64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int64_t nw = (int64_t)gridDim.x * 4;
  int longest = 0;
  for (int64_t i = (int64_t)blockIdx.x * 4 + w; i < ndof; i += nw) {
    const int64_t a = iptr[i];
    const int n = (int)(iptr[i + 1] - a);
    longest = max(longest, n);
    if (n <= 1 || n > 128) continue;
    EL_WAVE_SYNC();
    const int32_t k0 = lane < n ? ient[a + lane] : EL_INF, k1 = lane + 64 < n ? ient[a + lane + 64] : EL_INF;
    keys[w][lane] = k0;
    keys[w][lane + 64] = k1;
    EL_WAVE_SYNC();
    int r0 = 0, r1 = 0;
    for (int j = 0; j < n; j++) {
      const int32_t kj = keys[w][j];
      r0 += kj < k0;
      r1 += kj < k1;
    }
    if (lane < n) ient[a + r0] = k0;
    if (lane + 64 < n) ient[a + r1] = k1;
  }
  for (int o = 32; o > 0; o >>= 1) longest = max(longest, __shfl_xor(longest, o, 64));
  if (lane == 0) el_stat_max(stat, longest);
}

// ---- the splitting: one pass over the entries of A --------------------------------------------------------------------------
// counters: [0] entries put into a block of this chunk, [1] entries that belong to a cell of another chunk, [2] entries of
// checked rows without a listed common cell, [3] bad (a node in more than 64 cells)
__global__ void __launch_bounds__(256)
    k_el_scatter(const int64_t *__restrict__ arowptr, const int32_t *__restrict__ acol, const double *__restrict__ aval,
                 int64_t a_nrows, int64_t a_row0, int64_t node0, int64_t nnode, const int64_t *__restrict__ nptr,
                 const int32_t *__restrict__ nlist, int64_t own0, int64_t own1, int S, int64_t check0, int64_t check1,
                 double *__restrict__ blocks, unsigned long long *__restrict__ counters) {
  __shared__ int32_t rlist[4][64];
  const int lane = threadIdx.x & 63;
  const int64_t nw = (int64_t)gridDim.x * 4;
  unsigned long long n_own = 0, n_foreign = 0, n_unc = 0;
  for (int64_t row = tg_xcd_block(blockIdx.x, gridDim.x) * 4 + (threadIdx.x >> 6); row < a_nrows; row += nw) {
    const int64_t r = a_row0 + row, rn = r - node0;
    const int64_t e0 = arowptr[row], e1 = arowptr[row + 1];
    const bool checked = r >= check0 && r < check1;
    if (rn < 0 || rn >= nnode) {
      if (checked && lane == 0) n_unc += (unsigned long long)(e1 - e0);
      continue;
    }
    const int64_t p0 = nptr[rn];
    const int lr = (int)(nptr[rn + 1] - p0);
    if (lr > 64) {
      if (lane == 0) atomicOr((unsigned int *)(counters + 3), 1u);
      continue;
    }
    const int rl = lane < lr ? nlist[p0 + lane] : -1;
    // the cells of r where divergent code can read them: the first eight in scalar registers (a lane that is switched off
    // does not answer a shuffle), the others (a node in more than eight cells) in LDS
    int rs[8];
#pragma unroll
    for (int x = 0; x < 8; x++) rs[x] = __builtin_amdgcn_readlane(rl, x);
    if (lr > 8) {
      EL_WAVE_SYNC();
      rlist[threadIdx.x >> 6][lane] = rl;
      EL_WAVE_SYNC();
    }
    const int32_t *rl_lds = rlist[threadIdx.x >> 6];
    auto match = [&](int cs) -> int {
      int out = -1;
#pragma unroll
      for (int x = 0; x < 8; x++)
        if (x < lr && (rs[x] >> EL_POSBITS) == cs) out = rs[x];
      for (int x = 8; x < lr; x++)
        if ((rl_lds[x] >> EL_POSBITS) == cs) out = rl_lds[x];
      return out;
    };
    for (int64_t e = e0 + lane; e < e1; e += 64) {
      const int64_t sn = (int64_t)acol[e] - node0;
      int found = -1, fj = 0;
      if (sn >= 0 && sn < nnode) {
        const int64_t q0 = nptr[sn];
        const int ls = (int)(nptr[sn + 1] - q0);
        int sl[8];
#pragma unroll
        for (int k = 0; k < 8; k++) sl[k] = k < ls ? nlist[q0 + k] : -1;
        // the lowest common cell: the lists ascend, the first entry of the s list that is in the r list
#pragma unroll
        for (int k = 0; k < 8; k++) {
          if (found < 0 && sl[k] >= 0) {
            found = match(sl[k] >> EL_POSBITS);
            fj = sl[k] & ((1 << EL_POSBITS) - 1);
          }
        }
        for (int k = 8; k < ls && found < 0; k++) {
          const int sk = nlist[q0 + k];
          found = match(sk >> EL_POSBITS);
          fj = sk & ((1 << EL_POSBITS) - 1);
        }
      }
      if (found >= 0) {
        const int64_t c = found >> EL_POSBITS;
        if (c >= own0 && c < own1) {
          blocks[((c - own0) * S + (found & ((1 << EL_POSBITS) - 1))) * (int64_t)S + fj] = aval[e];
          n_own++;
        } else
          n_foreign++;
      } else if (checked)
        n_unc++;
    }
  }
  for (int o = 32; o > 0; o >>= 1) {
    n_own += __shfl_xor(n_own, o, 64);
    n_foreign += __shfl_xor(n_foreign, o, 64);
    n_unc += __shfl_xor(n_unc, o, 64);
  }
  __shared__ unsigned long long tot[4][3];
  if (lane == 0) tot[threadIdx.x >> 6][0] = n_own, tot[threadIdx.x >> 6][1] = n_foreign, tot[threadIdx.x >> 6][2] = n_unc;
  __syncthreads();
  if (threadIdx.x < 3) {
    const unsigned long long t = tot[0][threadIdx.x] + tot[1][threadIdx.x] + tot[2][threadIdx.x] + tot[3][threadIdx.x];
    if (t) atomicAdd(counters + threadIdx.x, t);
  }
}

// The same pass when no node lies in more than 8 cells (every hexahedral / quadrilateral mesh): the cells of a node sit in a table
// of 8 entries per node (-1 = none), read with two 16-byte loads at an address that follows from the column index alone -- one
// dependent load less per entry than row pointer + list -- and the row's own cells are scalars; two entries per lane in flight.
__global__ void __launch_bounds__(256)
    k_el_scatter8(const int64_t *__restrict__ arowptr, const int32_t *__restrict__ acol, const double *__restrict__ aval,
                  int64_t a_nrows, int64_t a_row0, int64_t node0, int64_t nnode, const int32_t *__restrict__ nfix, int64_t own0,
                  int64_t own1, int S, int64_t check0, int64_t check1, double *__restrict__ blocks,
                  unsigned long long *__restrict__ counters) {
  const int lane = threadIdx.x & 63;
  const int64_t nw = (int64_t)gridDim.x * 4;
  constexpr int PM = (1 << EL_POSBITS) - 1;
  unsigned long long n_own = 0, n_foreign = 0, n_unc = 0;
  for (int64_t row = tg_xcd_block(blockIdx.x, gridDim.x) * 4 + (threadIdx.x >> 6); row < a_nrows; row += nw) {
    const int64_t r = a_row0 + row, rn = r - node0;
    const int64_t e0 = arowptr[row], e1 = arowptr[row + 1];
    const bool checked = r >= check0 && r < check1;
    if (rn < 0 || rn >= nnode) {
      if (checked && lane == 0) n_unc += (unsigned long long)(e1 - e0);
      continue;
    }
    int rs[8];
    {
      const int4 *rp = reinterpret_cast<const int4 *>(nfix + rn * 8);
      const int4 ra = rp[0], rb = rp[1];
      rs[0] = __builtin_amdgcn_readfirstlane(ra.x), rs[1] = __builtin_amdgcn_readfirstlane(ra.y);
      rs[2] = __builtin_amdgcn_readfirstlane(ra.z), rs[3] = __builtin_amdgcn_readfirstlane(ra.w);
      rs[4] = __builtin_amdgcn_readfirstlane(rb.x), rs[5] = __builtin_amdgcn_readfirstlane(rb.y);
      rs[6] = __builtin_amdgcn_readfirstlane(rb.z), rs[7] = __builtin_amdgcn_readfirstlane(rb.w);
    }
    // the lowest common cell of r and s: the first entry of the (ascending) list of s whose cell is one of r's
    auto owner = [&](const int4 &sa, const int4 &sb, int &fj) -> int {
      const int sl[8] = {sa.x, sa.y, sa.z, sa.w, sb.x, sb.y, sb.z, sb.w};
      int found = -1;
#pragma unroll
      for (int k = 0; k < 8; k++) {
        const int cs = sl[k] >> EL_POSBITS;
        int hit = -1;
#pragma unroll
        for (int x = 0; x < 8; x++)
          if (rs[x] >= 0 && (rs[x] >> EL_POSBITS) == cs) hit = rs[x];
        if (found < 0 && sl[k] >= 0 && hit >= 0) found = hit, fj = sl[k] & PM;
      }
      return found;
    };
    auto put = [&](int found, int fj, int64_t e, bool valid) {
      if (!valid) return;
      if (found >= 0) {
        const int64_t c = found >> EL_POSBITS;
        if (c >= own0 && c < own1) {
          blocks[((c - own0) * S + (found & PM)) * (int64_t)S + fj] = aval[e];
          n_own++;
        } else
          n_foreign++;
      } else if (checked)
        n_unc++;
    };
    for (int64_t e = e0 + lane; e < e1; e += 128) {
      const bool v1 = e + 64 < e1;
      const int64_t s0 = (int64_t)acol[e] - node0, s1 = v1 ? (int64_t)acol[e + 64] - node0 : -1;
      const bool in0 = s0 >= 0 && s0 < nnode, in1 = s1 >= 0 && s1 < nnode;
      const int4 *q0 = reinterpret_cast<const int4 *>(nfix + (in0 ? s0 : 0) * 8), *q1 = reinterpret_cast<const int4 *>(nfix + (in1 ? s1 : 0) * 8);
      const int4 a0 = q0[0], b0 = q0[1], a1 = q1[0], b1 = q1[1];
      int j0 = 0, j1 = 0;
      const int f0 = in0 ? owner(a0, b0, j0) : -1, f1 = in1 ? owner(a1, b1, j1) : -1;
      put(f0, j0, e, true);
      put(f1, j1, e + 64, v1);
    }
  }
  for (int o = 32; o > 0; o >>= 1) {
    n_own += __shfl_xor(n_own, o, 64);
    n_foreign += __shfl_xor(n_foreign, o, 64);
    n_unc += __shfl_xor(n_unc, o, 64);
  }
  __shared__ unsigned long long tot[4][3];
  if (lane == 0) tot[threadIdx.x >> 6][0] = n_own, tot[threadIdx.x >> 6][1] = n_foreign, tot[threadIdx.x >> 6][2] = n_unc;
  __syncthreads();
  if (threadIdx.x < 3) {
    const unsigned long long t = tot[0][threadIdx.x] + tot[1][threadIdx.x] + tot[2][threadIdx.x] + tot[3][threadIdx.x];
    if (t) atomicAdd(counters + threadIdx.x, t);
  }
}

// [nnode][8] table of the nodes' cells from the lists (only when no list is longer than 8)
__global__ void __launch_bounds__(256)
    k_el_node_fix(const int64_t *__restrict__ nptr, const int32_t *__restrict__ nlist, int64_t nnode, int32_t *__restrict__ nfix) {
  const int64_t stride = (int64_t)gridDim.x * 256, total = nnode * 8;
  for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < total; t += stride) {
    const int64_t n = t >> 3;
    const int k = (int)(t & 7);
    const int64_t a = nptr[n];
    nfix[t] = a + k < nptr[n + 1] ? nlist[a + k] : -1;
  }
}

// ---- E_c = M_c^T (A_c M_c): one cell per workgroup, thread (ti, tj) of 16 x 16 holds a TS x TS tile ---------------------------
template <int TS>
__global__ void __launch_bounds__(256)
    k_el_dense(double *__restrict__ blocks, int S, int64_t nown, const int32_t *__restrict__ cn, int64_t own0, int b,
               const int64_t *__restrict__ mrowptr, const int32_t *__restrict__ mcol, const double *__restrict__ mval,
               int64_t m_row0, const int32_t *__restrict__ fl, const int32_t *__restrict__ nf) {
  constexpr int W = 16 * TS, LA = W + 1;
  __shared__ double As[W * LA];     // A_c [row][q], then T [r][s] with row length W
  __shared__ double Ms[W * W];      // M_c [node][function], zero-padded
  __shared__ int32_t fls[EL_FLS];
  const int tid = threadIdx.x, ti = tid >> 4, tj = tid & 15;
  for (int64_t c = tg_xcd_block(blockIdx.x, gridDim.x); c < nown; c += gridDim.x) {
    const int nfc = nf[c];
    double *blk = blocks + c * (int64_t)S * S;
    __syncthreads();
    if (tid < EL_FLS) fls[tid] = tid < nfc ? fl[c * EL_FLS + tid] : EL_INF;
    for (int t = tid; t < W * W; t += 256) {
      const int i = t / W, j = t - i * W;
      As[i * LA + j] = (i < S && j < S) ? blk[i * S + j] : 0.0;
      Ms[t] = 0.0;
    }
    __syncthreads();
    for (int i = tid >> 2; i < b; i += 64) {
      const int64_t r = cn[(own0 + c) * b + i] - m_row0;
      for (int64_t e = mrowptr[r] + (tid & 3); e < mrowptr[r + 1]; e += 4) {
        const int32_t col = mcol[e];
        int lo = 0, hi = nfc;
        while (lo < hi) {
          const int mid = (lo + hi) >> 1;
          if (fls[mid] < col) lo = mid + 1; else hi = mid;
        }
        Ms[i * W + lo] = mval[e];        // (the list is the union of these rows: the column is in it)
      }
    }
    __syncthreads();
    double acc[TS][TS];
#pragma unroll
    for (int u = 0; u < TS; u++)
#pragma unroll
      for (int v = 0; v < TS; v++) acc[u][v] = 0.0;
    for (int q = 0; q < b; q++) {
      double a4[TS], m4[TS];
#pragma unroll
      for (int u = 0; u < TS; u++) a4[u] = As[(TS * ti + u) * LA + q];
#pragma unroll
      for (int v = 0; v < TS; v++) m4[v] = Ms[q * W + TS * tj + v];
#pragma unroll
      for (int u = 0; u < TS; u++)
#pragma unroll
        for (int v = 0; v < TS; v++) acc[u][v] = fma(a4[u], m4[v], acc[u][v]);
    }
    __syncthreads();                       // (everyone is done with A_c)
#pragma unroll
    for (int u = 0; u < TS; u++)
#pragma unroll
      for (int v = 0; v < TS; v++) {
        As[(TS * ti + u) * W + TS * tj + v] = acc[u][v];
        acc[u][v] = 0.0;
      }
    __syncthreads();
    for (int r = 0; r < b; r++) {
      double q4[TS], t4[TS];
#pragma unroll
      for (int u = 0; u < TS; u++) q4[u] = Ms[r * W + TS * ti + u];
#pragma unroll
      for (int v = 0; v < TS; v++) t4[v] = As[r * W + TS * tj + v];
#pragma unroll
      for (int u = 0; u < TS; u++)
#pragma unroll
        for (int v = 0; v < TS; v++) acc[u][v] = fma(q4[u], t4[v], acc[u][v]);
    }
#pragma unroll
    for (int u = 0; u < TS; u++)
#pragma unroll
      for (int v = 0; v < TS; v++) {
        const int q = TS * ti + u, sidx = TS * tj + v;
        if (q < nfc && sidx < nfc) blk[(int64_t)q * S + sidx] = acc[u][v];
      }
  }
}

// ---- the same product on the matrix cores: v_mfma_f64_16x16x4_f64 ------------------------------------------------------------------
// The register-tile kernel above is LDS-bound (8 LDS reads per 16 multiply-adds: 2 MB of LDS reads per 64-node cell against the
// 128 B per clock of a CU); a 16 x 16 x 4 matrix instruction takes 128 operand doubles for 1024 multiply-adds.  Operand layout
// (cdna_hip_programming.md, fragment layout: f64 is its own case): A lane l = A[row l % 16][k l / 16], B lane l = B[k l / 16]
// [col l % 16], D lane l, register i = D[row l / 16 + 4 i][col l % 16].  A_c is staged TRANSPOSED ([q][row], stride W + 1) so that
// both operands are read along consecutive lanes; wave w computes the 16 rows [16 w, 16 w + 16) of T = A_c M_c, then of
// E = M_c^T T, all NB column tiles at once.  fp64 matrix and vector instructions share one multiplier array
// (profiles/r5_fp64_rate_valu_vs_mfma.txt): the gain is the LDS traffic, not the peak.
typedef double el_v4d __attribute__((ext_vector_type(4)));
// The global loads of the NEXT cell are issued between the phases of the current one, in the order of their dependences (block
// of A and node numbers before the first product, row pointers of M after it, entries of M before the second): a cell's inputs
// are in registers when its turn comes.  (Without that: 13.7 ms at 64^3 elements p = 3, of which the matrix instructions 3.5.)
template <int NB>
__global__ void __launch_bounds__(256, 2)          // (two workgroups per CU: one loads while the other multiplies)
    k_el_dense_mfma(double *__restrict__ blocks, int S, int64_t nown, const int32_t *__restrict__ cn, int64_t own0, int b,
                    const int64_t *__restrict__ mrowptr, const int32_t *__restrict__ mcol, const double *__restrict__ mval,
                    int64_t m_row0, const int32_t *__restrict__ fl, const int32_t *__restrict__ nf, const uint8_t *__restrict__ mpos) {
  constexpr int W = 16 * NB, LA = W + 1, NA = (W * W + 255) / 256;
  __shared__ double At[W * LA];     // A_c^T [q][row] with row length LA, then T [r][s] with row length W
  __shared__ double Ms[W * W];      // M_c [node][function], zero-padded
  __shared__ int32_t fls[EL_FLS];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, lr = lane & 15, lk = lane >> 4;
  const int gi = tid >> 2, gs = tid & 3;          // the row of M_c this thread gathers (4 threads per row), its phase
  // inputs of a cell in registers
  double areg[NA], mv[16];
  uint4 mp = make_uint4(0, 0, 0, 0);      // the positions of this thread's 16 entries in the cell's list, one byte each
  int64_t c_of_regs = 0;
  int32_t flreg = EL_INF, nfc = 0;
  int64_t e0 = 0, e1 = 0, rnode = -1;
  auto load_a = [&](int64_t c) {
    const double *blk = blocks + c * (int64_t)S * S;
#pragma unroll
    for (int u = 0; u < NA; u++) {
      const int t = tid + 256 * u, i = t / W, j = t - i * W;
      areg[u] = (t < W * W && i < S && j < S) ? blk[i * S + j] : 0.0;
    }
    rnode = gi < b ? (int64_t)cn[(own0 + c) * b + gi] - m_row0 : -1;
    c_of_regs = c;
    nfc = nf[c];
    flreg = tid < EL_FLS ? fl[c * EL_FLS + tid] : EL_INF;
  };
  auto load_ptr = [&]() {
    e0 = rnode >= 0 ? mrowptr[rnode] : 0;
    e1 = rnode >= 0 ? mrowptr[rnode + 1] : 0;
  };
  auto load_m = [&]() {
#pragma unroll
    for (int u = 0; u < 16; u++) {
      const int64_t e = e0 + gs + 4 * u;
      mv[u] = e < e1 ? mval[e] : 0.0;
    }
    if (gi < b) mp = *reinterpret_cast<const uint4 *>(mpos + ((c_of_regs * b + gi) << 6) + (gs << 4));
  };
  int64_t c = tg_xcd_block(blockIdx.x, gridDim.x);
  if (c < nown) {
    load_a(c);
    load_ptr();
    load_m();
  }
  for (; c < nown; c += gridDim.x) {
    const int64_t cnext = c + gridDim.x;
    const int nf_c = nfc;
    double *blk = blocks + c * (int64_t)S * S;
    __syncthreads();
    if (tid < EL_FLS) fls[tid] = tid < nf_c ? flreg : EL_INF;
#pragma unroll
    for (int u = 0; u < NA; u++) {
      const int t = tid + 256 * u, i = t / W, j = t - i * W;
      if (t < W * W) At[j * LA + i] = areg[u], Ms[t] = 0.0;
    }
    __syncthreads();
    if (gi < b) {
      const unsigned pw[4] = {mp.x, mp.y, mp.z, mp.w};
#pragma unroll
      for (int u = 0; u < 16; u++)
        if (e0 + gs + 4 * u < e1) Ms[gi * W + ((pw[u >> 2] >> (8 * (u & 3))) & 0xffu)] = mv[u];
      for (int64_t e = e0 + gs + 64; e < e1; e += 4) {       // (rows of M of more than 64 entries)
        const int32_t col = mcol[e];
        int lo = 0, hi = nf_c;
        while (lo < hi) {
          const int mid = (lo + hi) >> 1;
          if (fls[mid] < col) lo = mid + 1; else hi = mid;
        }
        Ms[gi * W + lo] = mval[e];
      }
    }
    for (int i2 = gi + 64; i2 < b; i2 += 64) {               // (cells of more than 64 nodes do not come here; kept general)
      const int64_t r = cn[(own0 + c) * b + i2] - m_row0;
      for (int64_t e = mrowptr[r] + gs; e < mrowptr[r + 1]; e += 4) {
        const int32_t col = mcol[e];
        int lo = 0, hi = nf_c;
        while (lo < hi) {
          const int mid = (lo + hi) >> 1;
          if (fls[mid] < col) lo = mid + 1; else hi = mid;
        }
        Ms[i2 * W + lo] = mval[e];
      }
    }
    if (cnext < nown) load_a(cnext);
    __syncthreads();
    el_v4d acc[NB];
    if (wave < NB) {
#pragma unroll
      for (int n = 0; n < NB; n++) acc[n] = (el_v4d){0.0, 0.0, 0.0, 0.0};
#pragma unroll 4
      for (int k4 = 0; k4 < W / 4; k4++) {
        const double a = At[(4 * k4 + lk) * LA + 16 * wave + lr];
#pragma unroll
        for (int n = 0; n < NB; n++)
          acc[n] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, Ms[(4 * k4 + lk) * W + 16 * n + lr], acc[n], 0, 0, 0);
      }
    }
    if (cnext < nown) load_ptr();
    __syncthreads();                       // (everyone is done with A_c)
    if (wave < NB) {
#pragma unroll
      for (int n = 0; n < NB; n++)
#pragma unroll
        for (int i = 0; i < 4; i++) At[(16 * wave + lk + 4 * i) * W + 16 * n + lr] = acc[n][i];
    }
    if (cnext < nown) load_m();
    __syncthreads();
    if (wave < NB) {
#pragma unroll
      for (int n = 0; n < NB; n++) acc[n] = (el_v4d){0.0, 0.0, 0.0, 0.0};
#pragma unroll 4
      for (int k4 = 0; k4 < W / 4; k4++) {
        const double a = Ms[(4 * k4 + lk) * W + 16 * wave + lr];          // M_c^T [q][r]
#pragma unroll
        for (int n = 0; n < NB; n++)
          acc[n] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, At[(4 * k4 + lk) * W + 16 * n + lr], acc[n], 0, 0, 0);
      }
#pragma unroll
      for (int n = 0; n < NB; n++)
#pragma unroll
        for (int i = 0; i < 4; i++) {
          const int q = 16 * wave + lk + 4 * i, sidx = 16 * n + lr;
          if (q < nf_c && sidx < nf_c) blk[(int64_t)q * S + sidx] = acc[n][i];
        }
    }
  }
}

// ---- cells of more than 64 nodes (3-D p = 4: 125 nodes, 125 functions): A_c and M_c do not fit the LDS together -------------------
// M_c stays (126 doubles per row), A_c goes through in panels of 16 rows: T_p = A_p M_c (16 x S) through LDS, then
// E += M_p^T T_p with ALL of E in registers (S^2 / 256 = 64 doubles per lane: 16 tiles of 16 x 16 per wave).  One workgroup per
// CU (158 KB of LDS).  S <= 126.
#define EL_BIG_MS 126
__global__ void __launch_bounds__(256)
    k_el_dense_big(double *__restrict__ blocks, int S, int64_t nown, const int32_t *__restrict__ cn, int64_t own0, int b,
                   const int64_t *__restrict__ mrowptr, const int32_t *__restrict__ mcol, const double *__restrict__ mval,
                   int64_t m_row0, const int32_t *__restrict__ fl, const int32_t *__restrict__ nf) {
  extern __shared__ __attribute__((aligned(16))) double el_dsm[];
  double *Ms = el_dsm;                          // [128][EL_BIG_MS]  M_c [node][function], zero-padded
  double *Ap = Ms + 128 * EL_BIG_MS;            // [16][129]         the panel of A_c [row][q]
  double *Tp = Ap + 16 * 129;                   // [16][128]         T_p [row][s]
  int32_t *fls = reinterpret_cast<int32_t *>(Tp);   // (the cell's function list: needed only while M_c is gathered)
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, lr = lane & 15, lk = lane >> 4;
  for (int64_t c = tg_xcd_block(blockIdx.x, gridDim.x); c < nown; c += gridDim.x) {
    const int nfc = nf[c];
    double *blk = blocks + c * (int64_t)S * S;
    __syncthreads();
    if (tid < EL_FLS) fls[tid] = tid < nfc ? fl[c * EL_FLS + tid] : EL_INF;
    for (int t = tid; t < 128 * EL_BIG_MS; t += 256) Ms[t] = 0.0;
    __syncthreads();
    for (int i = tid >> 2; i < b; i += 64) {
      const int64_t r = cn[(own0 + c) * b + i] - m_row0;
      for (int64_t e = mrowptr[r] + (tid & 3); e < mrowptr[r + 1]; e += 4) {
        const int32_t col = mcol[e];
        int lo = 0, hi = nfc;
        while (lo < hi) {
          const int mid = (lo + hi) >> 1;
          if (fls[mid] < col) lo = mid + 1; else hi = mid;
        }
        Ms[i * EL_BIG_MS + lo] = mval[e];
      }
    }
    el_v4d acc[2][8];          // E tiles (q tile 2 wave + u, s tile n)
#pragma unroll
    for (int u = 0; u < 2; u++)
#pragma unroll
      for (int n = 0; n < 8; n++) acc[u][n] = (el_v4d){0.0, 0.0, 0.0, 0.0};
    for (int p0 = 0; p0 < b; p0 += 16) {
      __syncthreads();                       // (M_c is complete / the last panel is done with)
      for (int t = tid; t < 16 * 128; t += 256) {
        const int i = t >> 7, j = t & 127;
        Ap[i * 129 + j] = (p0 + i < S && j < S) ? blk[(int64_t)(p0 + i) * S + j] : 0.0;
      }
      __syncthreads();
      // T_p = A_p M_c: wave w the column tiles 2 w, 2 w + 1
      el_v4d tt[2] = {(el_v4d){0.0, 0.0, 0.0, 0.0}, (el_v4d){0.0, 0.0, 0.0, 0.0}};
#pragma unroll 4
      for (int k4 = 0; k4 < 32; k4++) {
        const double a = Ap[lr * 129 + 4 * k4 + lk];
#pragma unroll
        for (int u = 0; u < 2; u++)
          tt[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, Ms[(4 * k4 + lk) * EL_BIG_MS + 16 * (2 * wave + u) + lr], tt[u], 0, 0, 0);
      }
#pragma unroll
      for (int u = 0; u < 2; u++)
#pragma unroll
        for (int i = 0; i < 4; i++) Tp[(lk + 4 * i) * 128 + 16 * (2 * wave + u) + lr] = tt[u][i];
      __syncthreads();
      // E += M_p^T T_p: the q tiles 2 w, 2 w + 1, all s tiles
#pragma unroll
      for (int k4 = 0; k4 < 4; k4++) {
        double a2[2], b8[8];
#pragma unroll
        for (int u = 0; u < 2; u++) a2[u] = Ms[(p0 + 4 * k4 + lk) * EL_BIG_MS + 16 * (2 * wave + u) + lr];
#pragma unroll
        for (int n = 0; n < 8; n++) b8[n] = Tp[(4 * k4 + lk) * 128 + 16 * n + lr];
#pragma unroll
        for (int u = 0; u < 2; u++)
#pragma unroll
          for (int n = 0; n < 8; n++) acc[u][n] = __builtin_amdgcn_mfma_f64_16x16x4f64(a2[u], b8[n], acc[u][n], 0, 0, 0);
      }
    }
    __syncthreads();                         // (every panel of A_c has been read: E goes over it)
#pragma unroll
    for (int u = 0; u < 2; u++)
#pragma unroll
      for (int n = 0; n < 8; n++)
#pragma unroll
        for (int i = 0; i < 4; i++) {
          const int q = 16 * (2 * wave + u) + lk + 4 * i, sidx = 16 * n + lr;
          if (q < nfc && sidx < nfc) blk[(int64_t)q * S + sidx] = acc[u][n][i];
        }
  }
}

// ---- rows of K: pattern and places ------------------------------------------------------------------------------------------
// One wave per function i (row dof0 + i of K).  The function lists of the cells that hold i are staged in LDS (row x of LD
// words = list of the x-th incident element row); lane x walks list x (and x + 64).  Every step takes the smallest head
// (the next column of the row), the lanes whose head it is store the step number in place of the list entry -- its PLACE in the
// row -- and move on.  mode 0: count only (rows row_phase, row_phase + row_step, ...: the sum goes to *cursor); 1: columns into
// tcol at a reserved offset (off / cnt per row), places into slot.  status: 1 = tcol too small (cnt is complete then), 2 = a row
// longer than EL_MAXROW.
template <bool TWO>
__global__ void __launch_bounds__(256)
    k_el_rowsym(const int64_t *__restrict__ iptr, const int32_t *__restrict__ ient, int64_t ndof, const int32_t *__restrict__ fl,
                const int32_t *__restrict__ nf, int S, int LD, int wave_words, int waves, int mode, int64_t row_step,
                int64_t row_phase, int32_t *__restrict__ tcol, unsigned long long *__restrict__ cursor, int64_t cap,
                int64_t *__restrict__ off, int64_t *__restrict__ cnt, uint16_t *__restrict__ slot, int *__restrict__ status) {
  extern __shared__ __attribute__((aligned(16))) int32_t el_smem[];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (w >= waves) return;
  int32_t *L = el_smem + (size_t)w * wave_words;      // [ninc][LD]
  int32_t *keys = L + (size_t)wave_words - 128 - EL_MAXROW;   // [128]
  int32_t *U = keys + 128;                            // [EL_MAXROW]
  const int64_t nw = (int64_t)gridDim.x * waves;
  unsigned long long counted = 0;
  long long blk_next = 0;
  int blk_left = 0;
  for (int64_t t = (int64_t)blockIdx.x * waves + w;; t += nw) {
    const int64_t i = row_phase + t * row_step;
    if (i >= ndof) break;
    const int64_t x0 = iptr[i];
    const int ninc = (int)(iptr[i + 1] - x0);
    if (ninc == 0) {
      if (mode == 1 && lane == 0) cnt[i] = 0, off[i] = 0;
      continue;
    }
    EL_WAVE_SYNC();
    const int key0 = lane < ninc ? ient[x0 + lane] : -1, key1 = (TWO && lane + 64 < ninc) ? ient[x0 + lane + 64] : -1;
    keys[lane] = key0;
    if (TWO) keys[lane + 64] = key1;
    const int n0 = key0 >= 0 ? nf[key0 / S] : 0, n1 = key1 >= 0 ? nf[key1 / S] : 0;
    // (the loads of the staging loop depend on registers only: several lists are in flight at once)
    for (int xb = 0; xb < ninc; xb += 16) {        // (loads first, sixteen lists in flight; then the LDS stores)
      int32_t ta[16], tb[16];
      int nx[16];
#pragma unroll
      for (int u = 0; u < 16; u++) {
        const int x = xb + u;
        const int ka = __shfl(key0, x & 63, 64), kb = TWO ? __shfl(key1, x & 63, 64) : 0;
        const int na = __shfl(n0, x & 63, 64), nb = TWO ? __shfl(n1, x & 63, 64) : 0;
        const int kx = x < 64 ? ka : kb;
        nx[u] = x >= ninc ? 0 : (x < 64 ? na : nb);
        const int32_t *src = fl + (int64_t)(max(kx, 0) / S) * EL_FLS;
        ta[u] = src[lane];
        tb[u] = LD > 65 ? src[lane + 64] : 0;
      }
#pragma unroll
      for (int u = 0; u < 16; u++) {
        const int x = xb + u;
        if (lane < nx[u]) L[x * LD + lane] = ta[u];
        if (lane + 64 < nx[u]) L[x * LD + lane + 64] = tb[u];
      }
    }
    EL_WAVE_SYNC();
    const int k = mode == 0 ? el_merge_staged<false, TWO>(L, LD, lane, n0, n1, U, EL_MAXROW)
                            : el_merge_staged<true, TWO>(L, LD, lane, n0, n1, U, EL_MAXROW);
    const bool too_long = k < 0;
    if (too_long) {
      if (lane == 0) atomicMax(status, 2);
      if (mode == 1 && lane == 0) cnt[i] = 0, off[i] = 0;
      continue;
    }
    if (mode == 0) {
      counted += (unsigned long long)k;
      continue;
    }
    if (k > blk_left) {                  // (a new block: what is left of the old one stays unused)
      long long nb = 0;
      if (lane == 0) nb = (long long)atomicAdd(cursor, (unsigned long long)EL_BLOCK);
      blk_next = __shfl(nb, 0, 64);
      blk_left = EL_BLOCK;
    }
    const long long o = blk_next;
    blk_next += k;
    blk_left -= k;
    if (lane == 0) cnt[i] = k, off[i] = o;
    if (o + k > cap) {
      if (lane == 0) atomicMax(status, 1);
      continue;
    }
    EL_WAVE_SYNC();
    for (int e = lane; e < k; e += 64) tcol[o + e] = U[e];
    for (int xb = 0; xb < ninc; xb += 4) {
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const int x = xb + u;
        const int ka = __shfl(key0, x & 63, 64), kb = TWO ? __shfl(key1, x & 63, 64) : 0;
        const int na = __shfl(n0, x & 63, 64), nb = TWO ? __shfl(n1, x & 63, 64) : 0;
        const int key = x < 64 ? ka : kb;
        const int n = x >= ninc ? 0 : (x < 64 ? na : nb);
        uint16_t *sl = slot + (int64_t)max(key, 0) * S;
        if (lane < n) sl[lane] = (uint16_t)L[x * LD + lane];
        if (lane + 64 < n) sl[lane + 64] = (uint16_t)L[x * LD + lane + 64];
      }
    }
  }
  if (mode == 0 && lane == 0 && counted) atomicAdd(cursor, counted);
}

// ---- the same by hashing (rows whose function is in at most 64 cells of at most 64 functions: every spline space of degree <= 3)
// The merge above is a chain of as many steps as the row has entries, ~50 instructions each (17 k per row of 343).  Here the 64 x 64
// candidates of a row sit in registers (lane = position in the list, register = list); they are put into a hash SET in LDS
// (one compare-and-swap each, independent of one another), the set is read out, sorted in registers (bitonic network, R keys
// per lane: in-lane stages are min / max, the others one shuffle per key), every sorted key writes its rank next to its table
// slot, and a candidate's place is the rank at the slot it remembered: ~2.8 k instructions per row without a long dependent
// chain.  The result is the same sorted row and the same places (deterministic: nothing depends on the order of insertion).
template <int R>
__device__ __forceinline__ void el_sort_regs(int (&r)[R], int lane) {
#pragma unroll
  for (int k = 2; k <= 64 * R; k <<= 1) {
    const bool asc_lane = ((lane * R) & k) == 0;          // (for k > R: the direction of this lane's keys)
#pragma unroll
    for (int j = k >> 1; j > 0; j >>= 1) {
      if (j < R) {
#pragma unroll
        for (int t = 0; t < R; t++) {
          if (t & j) continue;
          const bool asc = k < R ? ((t & k) == 0) : asc_lane;
          const int a = r[t], b = r[t | j], lo = min(a, b), hi = max(a, b);
          r[t] = asc ? lo : hi;
          r[t | j] = asc ? hi : lo;
        }
      } else {
        const int d = j / R;
        const bool keep_min = (((lane & d) == 0) == asc_lane);
#pragma unroll
        for (int t = 0; t < R; t++) {
          const int pv = __shfl_xor(r[t], d, 64);
          r[t] = keep_min ? min(r[t], pv) : max(r[t], pv);
        }
      }
    }
  }
}

// status: 1 = tcol too small, 2 = a row longer than EL_MAXROW, 3 = a row longer than 64 R (take the next instantiation)
template <int R>
__global__ void __launch_bounds__(256)
    k_el_rowhash(const int64_t *__restrict__ iptr, const int32_t *__restrict__ ient, int64_t ndof, const int32_t *__restrict__ fl,
                 const int32_t *__restrict__ nf, int S, int waves, int mode, int64_t row_step, int64_t row_phase,
                 int32_t *__restrict__ tcol, unsigned long long *__restrict__ cursor, int64_t cap, int64_t *__restrict__ off,
                 int64_t *__restrict__ cnt, uint16_t *__restrict__ slot, int *__restrict__ status) {
  constexpr int T = 256 * R, LGT = R == 8 ? 11 : 12, NU = 64 * R;
  static_assert(R == 8 || R == 16, "k_el_rowhash: 8 or 16 keys per lane");
  extern __shared__ __attribute__((aligned(16))) int32_t el_smem[];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (w >= waves) return;
  int32_t *tab = el_smem + (size_t)w * (T + T / 2 + NU);       // [T] keys, -1 = empty
  uint16_t *place = reinterpret_cast<uint16_t *>(tab + T);     // [T] rank of the key in the slot
  int32_t *U = tab + T + T / 2;                                // [NU]
  const int64_t nw = (int64_t)gridDim.x * waves;
  unsigned long long counted = 0;
  long long blk_next = 0;
  int blk_left = 0;
  for (int64_t t = (int64_t)blockIdx.x * waves + w;; t += nw) {
    const int64_t i = row_phase + t * row_step;
    if (i >= ndof) break;
    const int64_t x0 = iptr[i];
    const int ninc = (int)(iptr[i + 1] - x0);
    if (ninc == 0) {
      if (mode == 1 && lane == 0) cnt[i] = 0, off[i] = 0;
      continue;
    }
    const int key0 = lane < ninc ? ient[x0 + lane] : 0;
    const int n0 = lane < ninc ? nf[key0 / S] : 0;
    // ---- candidates: register x = list x, lane = position
    int kx[64];
#pragma unroll
    for (int x = 0; x < 64; x++) {
      const int c = __shfl(key0, x, 64) / S, nx = __shfl(n0, x, 64);
      kx[x] = lane < nx ? fl[(int64_t)c * EL_FLS + lane] : -1;
    }
    EL_WAVE_SYNC();
    {
      int4 *t4 = reinterpret_cast<int4 *>(tab);
#pragma unroll
      for (int q = 0; q < T / 256; q++) t4[lane + 64 * q] = make_int4(-1, -1, -1, -1);
    }
    EL_WAVE_SYNC();
    // ---- the set: slot of every candidate, two 16-bit indices per register
    unsigned hx[32];
#pragma unroll
    for (int x = 0; x < 64; x++) {
      unsigned h = 0;
      if (x < ninc) {                         // (uniform)
        const int k = kx[x];
        if (k >= 0) {
          h = ((unsigned)k * 0x9E3779B1u) >> (32 - LGT);
          for (;;) {
            const int old = atomicCAS(&tab[h], -1, k);
            if (old == -1 || old == k) break;
            h = (h + 1) & (T - 1);
          }
        }
      }
      if (x & 1) hx[x >> 1] |= h << 16; else hx[x >> 1] = h;
    }
    EL_WAVE_SYNC();
    // ---- read the set out: lane l owns the slots [l T / 64, (l + 1) T / 64)
    int mine = 0;
    {
      const int4 *t4 = reinterpret_cast<const int4 *>(tab + lane * (T / 64));
#pragma unroll
      for (int q = 0; q < T / 256; q++) {
        const int4 v = t4[q];
        mine += (v.x >= 0) + (v.y >= 0) + (v.z >= 0) + (v.w >= 0);
      }
    }
    int incl = mine;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int up = __shfl_up(incl, o, 64);
      if (lane >= o) incl += up;
    }
    const int n = __shfl(incl, 63, 64);
    if (n > NU || n > EL_MAXROW) {
      if (lane == 0) atomicMax(status, n > EL_MAXROW ? 2 : 3);
      if (mode == 1 && lane == 0) cnt[i] = 0, off[i] = 0;
      continue;
    }
    if (mode == 0) {
      counted += (unsigned long long)n;
      continue;
    }
    for (int e = lane; e < NU; e += 64) U[e] = EL_INF;
    EL_WAVE_SYNC();
    {
      int at = incl - mine;
#pragma unroll
      for (int q = 0; q < T / 64; q++) {
        const int v = tab[lane * (T / 64) + q];
        if (v >= 0) U[at++] = v;
      }
    }
    EL_WAVE_SYNC();
    int r[R];
#pragma unroll
    for (int q = 0; q < R; q++) r[q] = U[lane * R + q];
    el_sort_regs<R>(r, lane);
    EL_WAVE_SYNC();
#pragma unroll
    for (int q = 0; q < R; q++) U[lane * R + q] = r[q];
    // ---- the rank of every key, next to its slot
#pragma unroll
    for (int q = 0; q < R; q++) {
      const int k = r[q];
      if (k != EL_INF) {
        unsigned h = ((unsigned)k * 0x9E3779B1u) >> (32 - LGT);
        while (tab[h] != k) h = (h + 1) & (T - 1);
        place[h] = (uint16_t)(lane * R + q);
      }
    }
    EL_WAVE_SYNC();
    // ---- outputs
    if (n > blk_left) {                  // (a new block: what is left of the old one stays unused)
      long long nb = 0;
      if (lane == 0) nb = (long long)atomicAdd(cursor, (unsigned long long)EL_BLOCK);
      blk_next = __shfl(nb, 0, 64);
      blk_left = EL_BLOCK;
    }
    const long long o = blk_next;
    blk_next += n;
    blk_left -= n;
    if (lane == 0) cnt[i] = n, off[i] = o;
    if (o + n > cap) {
      if (lane == 0) atomicMax(status, 1);
      continue;
    }
    for (int e = lane; e < n; e += 64) tcol[o + e] = U[e];
#pragma unroll
    for (int x = 0; x < 64; x++) {
      if (x < ninc) {
        const int key = __shfl(key0, x, 64);
        const unsigned h = (x & 1) ? (hx[x >> 1] >> 16) : (hx[x >> 1] & 0xffffu);
        if (kx[x] >= 0) slot[(int64_t)key * S + lane] = place[h];
      }
    }
  }
  if (mode == 0 && lane == 0 && counted) atomicAdd(cursor, counted);
}

__global__ void __launch_bounds__(256)
    k_el_reorder(const int64_t *__restrict__ rowptr, const int64_t *__restrict__ off, const int32_t *__restrict__ tcol, int64_t nrows,
                 int32_t *__restrict__ col, int *__restrict__ maxrow) {
  const int lane = threadIdx.x & 63;
  const int64_t nw = (int64_t)gridDim.x * 4;
  int longest = 0;
  for (int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); r < nrows; r += nw) {
    const int64_t dst = rowptr[r], n = rowptr[r + 1] - dst, src = off[r];
    longest = max(longest, (int)n);
    for (int64_t q = lane; q < n; q += 64) col[dst + q] = tcol[src + q];
  }
  if (lane == 0 && longest) el_stat_max(maxrow, longest);
}

// ---- values of K by places: one wave per row; the 64 / LPR groups of lanes take the incident element rows in turn and add
// their values into the group's own accumulators (places within one element row are distinct: plain read-modify-write in
// LDS), the groups' sums are added in a fixed order; MatZeroRowsColumns on the way out
template <int LGR>
__global__ void __launch_bounds__(256)
    k_el_merge(const int64_t *__restrict__ iptr, const int32_t *__restrict__ ient, const double *__restrict__ eval,
               const uint16_t *__restrict__ slot, const int32_t *__restrict__ nfc, int S, int64_t nrows, int capk,
               const int64_t *__restrict__ krowptr, const int32_t *__restrict__ kcol, const uint8_t *__restrict__ mask,
               int64_t dof0, double diag, double *__restrict__ kval) {
  constexpr int LPR = 1 << LGR, NG = 64 >> LGR;
  extern __shared__ __attribute__((aligned(16))) double el_acc[];    // [4][NG][capk]
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, grp = lane >> LGR, sub = lane & (LPR - 1);
  double *acc = el_acc + (size_t)wave * NG * capk;
  const int64_t wstride = (int64_t)gridDim.x * 4;
  for (int64_t i = tg_xcd_block(blockIdx.x, gridDim.x) * 4 + wave; i < nrows; i += wstride) {
    const int64_t k0 = krowptr[i];
    const int n = (int)(krowptr[i + 1] - k0);
    if (n == 0) continue;
    for (int e = lane; e < NG * capk; e += 64) acc[e] = 0.0;       // (rows of one wave: LDS operations are in order)
    EL_WAVE_SYNC();
    for (int64_t x = iptr[i] + grp; x < iptr[i + 1]; x += NG) {
      const int64_t key = ient[x];
      const int nf = nfc[key / S];
      for (int r = sub; r < nf; r += LPR) acc[grp * capk + slot[key * S + r]] += eval[key * S + r];
    }
    EL_WAVE_SYNC();
    const int64_t gi = dof0 + i;
    const bool mrow = mask && mask[gi];
    for (int e = lane; e < n; e += 64) {
      double v = acc[e];
#pragma unroll
      for (int g = 1; g < NG; g++) v += acc[g * capk + e];
      if (mask) {
        const int32_t c = kcol[k0 + e];
        if (mrow || mask[c]) v = (mrow && c == gi) ? diag : 0.0;
      }
      kval[k0 + e] = v;
    }
  }
}

// ---- host side ------------------------------------------------------------------------------------------------------------------
// grids of the kernels that renumber their blocks by XCD (tg_xcd_block): a bijection only for multiples of 8
static inline unsigned el_grid8(int64_t g) { return (unsigned)((std::max<int64_t>(g, 1) + 7) & ~(int64_t)7); }
extern "C" int tg_elemplan_destroy(tg_elemplan_t pl) {
  if (!pl) return 0;
  if (g_tg.ready) {
    hipStreamSynchronize(g_tg.stream);
    tg_dfree(pl->nptr);
    tg_dfree(pl->nlist);
    tg_dfree(pl->nfix);
    tg_dfree(pl->fl);
    tg_dfree(pl->nf);
    tg_dfree(pl->mpos);
    tg_dfree(pl->iptr);
    tg_dfree(pl->ient);
    tg_dfree(pl->k_rowptr);
    tg_dfree(pl->k_col);
    tg_dfree(pl->slot);
  }
  delete pl;
  return 0;
}

/* The plan of the element-split product for the cells [own0, own1) of `cells` (all listed cells take part in the ownership
 * rule).  m: the rows [m_row0, m_row0 + rows of m) of M with global columns, holding the rows of every node of the own cells;
 * borrowed, as `cells`, for the plan's lifetime.  100: the cells do not qualify (a cell with more than 128 functions, a
 * function in more than 128 cells). */
extern "C" int tg_elemplan_create(tg_cells_t cells, int64_t own0, int64_t own1, tg_csr_t m, int64_t m_row0, tg_elemplan_t *out) {
  TG_REQUIRE_INIT();
  TG_REQUIRE(cells && m && out && own0 >= 0 && own1 > own0 && own1 <= cells->ncell, "bad arguments to tg_elemplan_create");
  TG_REQUIRE_CANONICAL(m);
  TG_REQUIRE(cells->ncell < (1ll << (31 - EL_POSBITS)), "tg_elemplan_create: more than 16 M cells in one chunk");
  tg_elemplan_s *pl = new tg_elemplan_s();
  pl->cells = cells, pl->m = m, pl->m_row0 = m_row0, pl->own0 = own0, pl->own1 = own1, pl->b = cells->b;
  const int64_t ncell = cells->ncell, nown = own1 - own0, total = ncell * cells->b;
  int *st = (int *)g_tg.scratch;             // [0..7] statistics
  int h[8];
  int32_t *cur = nullptr;
  int rc = 0;
  auto fail = [&](int code) {
    tg_dfree(cur);
    tg_elemplan_destroy(pl);
    return code;
  };
#define EL_SYNC_STATS()                                                                                         \
  (hipMemcpyAsync(h, st, sizeof(h), hipMemcpyDeviceToHost, g_tg.stream) != hipSuccess || hipStreamSynchronize(g_tg.stream) != hipSuccess || \
   hipGetLastError() != hipSuccess)
  // ---- window of node numbers
  {
    const int init[8] = {EL_INF, -1, 0, 0, EL_INF, -1, 0, 0};
    hipMemcpyAsync(st, init, sizeof(init), hipMemcpyHostToDevice, g_tg.stream);
    hipLaunchKernelGGL(k_el_minmax, dim3(tg_grid_1d(total, 256)), dim3(256), 0, g_tg.stream, cells->nodes, total, st);
    if (EL_SYNC_STATS()) {
      tg_set_error("tg_elemplan_create: node window failed to run");
      return fail(1);
    }
    if (h[0] < 0) {
      tg_set_error("tg_elemplan_create: negative node number");
      return fail(2);
    }
    pl->node0 = h[0], pl->nnode = (int64_t)h[1] - h[0] + 1;
  }
  // ---- node -> cells
  rc = tg_dmalloc(&pl->nptr, pl->nnode + 2) || tg_dmalloc(&pl->nlist, total) || tg_dmalloc(&cur, pl->nnode);
  if (rc) return fail(rc);
  hipMemsetAsync(pl->nptr, 0, (size_t)(pl->nnode + 2) * sizeof(int64_t), g_tg.stream);
  hipMemsetAsync(cur, 0, (size_t)pl->nnode * sizeof(int32_t), g_tg.stream);
  hipLaunchKernelGGL(k_el_node_count, dim3(tg_grid_1d(total, 256)), dim3(256), 0, g_tg.stream, cells->nodes, total, pl->node0, pl->nptr);
  rc = tg_exclusive_scan_i64(pl->nptr, pl->nnode, nullptr);
  if (rc) return fail(rc);
  hipLaunchKernelGGL(k_el_node_fill, dim3(tg_grid_1d(total, 256)), dim3(256), 0, g_tg.stream, cells->nodes, total, cells->b, pl->node0,
                     pl->nptr, cur, pl->nlist);
  hipLaunchKernelGGL(k_el_node_sort, dim3(tg_grid_1d(pl->nnode, 256)), dim3(256), 0, g_tg.stream, pl->nptr, pl->nnode, pl->nlist, st + 3);
  tg_dfree(cur);
  cur = nullptr;
  // ---- function lists of the own cells
  rc = tg_dmalloc(&pl->fl, nown * EL_FLS) || tg_dmalloc(&pl->nf, nown) || tg_dmalloc(&pl->mpos, nown * cells->b * 64 + 64);
  if (rc) return fail(rc);
  {
    // longest row of M: the stride of the staged rows (a row of more than EL_FLS entries cannot be part of a cell's list); with
    // it comes the longest node -> cells list (k_el_node_sort): up to 8, the lists are also kept as a table (k_el_scatter8)
    hipMemsetAsync(st + 1, 0, sizeof(int), g_tg.stream);
    hipLaunchKernelGGL(k_el_maxrow, dim3(tg_grid_1d(m->nrows, 256)), dim3(256), 0, g_tg.stream, m->rowptr, m->nrows, st + 1);
    int hmax = 0, hnode = 0;
    if (hipMemcpyAsync(&hmax, st + 1, sizeof(int), hipMemcpyDeviceToHost, g_tg.stream) != hipSuccess ||
        hipMemcpyAsync(&hnode, st + 2, sizeof(int), hipMemcpyDeviceToHost, g_tg.stream) != hipSuccess ||
        hipStreamSynchronize(g_tg.stream) != hipSuccess) {
      tg_set_error("tg_elemplan_create: the row-length kernel failed to run");
      return fail(1);
    }
    if (hnode <= 8 && !getenv("TIGAR_EL_LISTS")) {
      rc = tg_dmalloc(&pl->nfix, pl->nnode * 8 + 16);
      if (rc) return fail(rc);
      hipLaunchKernelGGL(k_el_node_fix, dim3(tg_grid_1d(pl->nnode * 8, 256)), dim3(256), 0, g_tg.stream, pl->nptr, pl->nlist, pl->nnode, pl->nfix);
    }
    const int LD = std::min(hmax, EL_FLS) + 1;
    const bool two = cells->b > 64;
    const int wave_words = (two ? 128 : 64) * LD + EL_FLS;
    const int waves = (size_t)wave_words * 4 > 16 * 1024 ? 1 : 4;
    const size_t lds = (size_t)waves * wave_words * sizeof(int32_t);
    if (lds > 160 * 1024) return fail(100);
    const unsigned grid = (unsigned)std::max<int64_t>(1, std::min<int64_t>(tg_cdiv(nown, waves), (int64_t)g_tg.num_cu * 64));
    if (two) {
      hipFuncSetAttribute((const void *)k_el_fl<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      hipLaunchKernelGGL((k_el_fl<true>), dim3(grid), dim3(64 * waves), lds, g_tg.stream, m->rowptr, m->col, m_row0, m->nrows, cells->nodes, own0,
                         nown, cells->b, LD, wave_words, waves, pl->fl, pl->nf, pl->mpos, st + 4, getenv("TIGAR_EL_DBG") ? atoi(getenv("TIGAR_EL_DBG")) : 0);
    } else {
      hipFuncSetAttribute((const void *)k_el_fl<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      hipLaunchKernelGGL((k_el_fl<false>), dim3(grid), dim3(64 * waves), lds, g_tg.stream, m->rowptr, m->col, m_row0, m->nrows, cells->nodes, own0,
                         nown, cells->b, LD, wave_words, waves, pl->fl, pl->nf, pl->mpos, st + 4, getenv("TIGAR_EL_DBG") ? atoi(getenv("TIGAR_EL_DBG")) : 0);
    }
  }
  if (EL_SYNC_STATS()) {
    tg_set_error("tg_elemplan_create: the function-list kernels failed to run");
    return fail(1);
  }
  if (h[3]) {
    tg_set_error("tg_elemplan_create: a node is listed twice in one cell");
    return fail(2);
  }
  if (h[7] & 1) {
    tg_set_error("tg_elemplan_create: a node of an own cell lies outside the rows of M handed in");
    return fail(2);
  }
  if ((h[7] & 2) || h[6] < 1) return fail(100);          // more than EL_FLS functions in a cell / an operator without entries
  pl->nfmax = h[6];
  pl->S = std::max(pl->b, pl->nfmax);
  pl->dof0 = h[4], pl->dof1 = (int64_t)h[5] + 1;
  TG_REQUIRE(nown * (int64_t)pl->S < 0x7fffffffll, "tg_elemplan_create: too many cells in one chunk");
  // ---- function -> element rows
  const int64_t ndof = pl->dof1 - pl->dof0;
  rc = tg_dmalloc(&pl->iptr, ndof + 2) || tg_dmalloc(&cur, ndof);
  if (rc) return fail(rc);
  hipMemsetAsync(pl->iptr, 0, (size_t)(ndof + 2) * sizeof(int64_t), g_tg.stream);
  hipMemsetAsync(cur, 0, (size_t)ndof * sizeof(int32_t), g_tg.stream);
  hipLaunchKernelGGL(k_el_inc_count, dim3(tg_grid_1d(nown * EL_FLS, 256)), dim3(256), 0, g_tg.stream, pl->fl, pl->nf, nown, pl->dof0, pl->iptr);
  int64_t ninc = 0;
  rc = tg_exclusive_scan_i64(pl->iptr, ndof, &ninc);
  if (!rc) rc = tg_dmalloc(&pl->ient, ninc);
  if (rc) return fail(rc);
  hipMemsetAsync(st, 0, 4 * sizeof(int), g_tg.stream);
  hipLaunchKernelGGL(k_el_inc_fill, dim3(tg_grid_1d(nown * EL_FLS, 256)), dim3(256), 0, g_tg.stream, pl->fl, pl->nf, nown, pl->S, pl->dof0,
                     pl->iptr, cur, pl->ient);
  hipLaunchKernelGGL(k_el_inc_sort, dim3((unsigned)std::min<int64_t>(tg_cdiv(ndof, 4), (int64_t)g_tg.num_cu * 32)), dim3(256), 0, g_tg.stream,
                     pl->iptr, ndof, pl->ient, st);
  if (EL_SYNC_STATS()) {
    tg_set_error("tg_elemplan_create: the incidence kernels failed to run");
    return fail(1);
  }
  tg_dfree(cur);
  cur = nullptr;
  pl->ninc_max = h[0];
  if (pl->ninc_max > 128) return fail(100);
#undef EL_SYNC_STATS
  *out = pl;
  return 0;
}

extern "C" int tg_elemplan_info(tg_elemplan_t pl, int64_t *dof0, int64_t *dof1, int *nfmax, int *ninc_max, int64_t *k_nnz) {
  TG_REQUIRE(pl, "null plan");
  if (dof0) *dof0 = pl->dof0;
  if (dof1) *dof1 = pl->dof1;
  if (nfmax) *nfmax = pl->nfmax;
  if (ninc_max) *ninc_max = pl->ninc_max;
  if (k_nnz) *k_nnz = pl->k_nnz;
  return 0;
}

static int el_symbolic(tg_elemplan_s *pl) {
  const int64_t ndof = pl->dof1 - pl->dof0, nown = pl->own1 - pl->own0;
  // the merge kernel (any cell the plan accepts)
  const int LD = pl->nfmax + 1;
  const bool two = pl->ninc_max > 64;
  const int wave_words = (two ? 128 : 64) * LD + 128 + EL_MAXROW;
  // (one wave per workgroup when a wave's lists take more than 16 KB: more workgroups fit a CU's LDS than waves of one would)
  const int mwaves = (size_t)wave_words * 4 > 16 * 1024 ? 1 : 4;
  const size_t mlds = (size_t)mwaves * wave_words * sizeof(int32_t);
  // the hash kernel: functions in at most 64 cells of at most 64 functions; 8 keys per lane (rows of up to 512 entries), 16 when a
  // row turns out longer (status 3)
  int hashR = (pl->ninc_max <= 64 && pl->nfmax <= 64 && !getenv("TIGAR_EL_MERGE")) ? 8 : 0;
  if (!hashR && mlds > 160 * 1024) return 100;
  hipFuncSetAttribute((const void *)k_el_rowsym<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipFuncSetAttribute((const void *)k_el_rowsym<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipFuncSetAttribute((const void *)k_el_rowhash<8>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipFuncSetAttribute((const void *)k_el_rowhash<16>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  int *status = (int *)g_tg.scratch;
  unsigned long long *cursor = (unsigned long long *)(g_tg.scratch + 2);
  int64_t *off = nullptr, *cnt = nullptr;
  int32_t *tcol = nullptr;
  auto cleanup = [&]() {
    tg_dfree(off);
    tg_dfree(cnt);
    tg_dfree(tcol);
  };
  int rc = tg_dmalloc(&off, ndof + 1) || tg_dmalloc(&cnt, ndof + 2) || tg_dmalloc(&pl->slot, nown * (int64_t)pl->S * pl->S + 16);
  if (rc) {
    cleanup();
    return rc;
  }
  unsigned long long hsum = 0;
  int hstat = 0;
  auto launch = [&](int mode, int64_t step, int64_t phase, int64_t cap, unsigned *grid_out) -> int {
    const int waves = hashR == 8 ? 2 : hashR == 16 ? 1 : mwaves;
    const size_t lds = hashR ? (size_t)waves * (256 * hashR * 3 / 2 + 64 * hashR) * sizeof(int32_t) : mlds;
    const int64_t nrows = tg_cdiv(ndof, step);
    const unsigned grid = (unsigned)std::max<int64_t>(1, std::min<int64_t>(tg_cdiv(nrows, waves), (int64_t)g_tg.num_cu * 32));
    if (grid_out) *grid_out = grid * (unsigned)waves;
    hipMemsetAsync(status, 0, 4 * sizeof(double), g_tg.stream);
    int32_t *tc = mode ? tcol : nullptr;
    int64_t *po = mode ? off : nullptr, *pc = mode ? cnt : nullptr;
    uint16_t *ps = mode ? pl->slot : nullptr;
    if (hashR == 8)
      hipLaunchKernelGGL((k_el_rowhash<8>), dim3(grid), dim3(64 * waves), lds, g_tg.stream, pl->iptr, pl->ient, ndof, pl->fl, pl->nf, pl->S, waves,
                         mode, step, phase, tc, cursor, cap, po, pc, ps, status);
    else if (hashR == 16)
      hipLaunchKernelGGL((k_el_rowhash<16>), dim3(grid), dim3(64 * waves), lds, g_tg.stream, pl->iptr, pl->ient, ndof, pl->fl, pl->nf, pl->S, waves,
                         mode, step, phase, tc, cursor, cap, po, pc, ps, status);
    else if (two)
      hipLaunchKernelGGL((k_el_rowsym<true>), dim3(grid), dim3(64 * waves), lds, g_tg.stream, pl->iptr, pl->ient, ndof, pl->fl, pl->nf, pl->S, LD,
                         wave_words, waves, mode, step, phase, tc, cursor, cap, po, pc, ps, status);
    else
      hipLaunchKernelGGL((k_el_rowsym<false>), dim3(grid), dim3(64 * waves), lds, g_tg.stream, pl->iptr, pl->ient, ndof, pl->fl, pl->nf, pl->S, LD,
                         wave_words, waves, mode, step, phase, tc, cursor, cap, po, pc, ps, status);
    if (hipMemcpyAsync(&hsum, cursor, sizeof(hsum), hipMemcpyDeviceToHost, g_tg.stream) != hipSuccess ||
        hipMemcpyAsync(&hstat, status, sizeof(int), hipMemcpyDeviceToHost, g_tg.stream) != hipSuccess ||
        hipStreamSynchronize(g_tg.stream) != hipSuccess || hipGetLastError() != hipSuccess) {
      tg_set_error("element split: the symbolic pass failed to run (LDS %zu B)", lds);
      return 1;
    }
    return 0;
  };
  // a sample of the rows gives the capacity of the temporary (1/32 of the work).  The waves reserve the temporary in blocks of
  // EL_BLOCK entries: besides the entries themselves room for the unused tail of every block (less than a row each; from the
  // sample's mean row) and for one open block per wave.  A pass that comes out short has reserved exactly what the next one --
  // same rows per wave -- will ask for.
  const int64_t step = ndof >= 8192 ? 32 : 1;
  int64_t cap = 0;
  for (int attempt = 0; attempt < 4 && !rc; attempt++) {
    if (cap == 0) {
      rc = launch(0, step, step / 2, 0, nullptr);
      if (rc) break;
      if (hstat == 3 && hashR == 8) {
        hashR = 16;
        continue;
      }
      if (hstat >= 2) {
        rc = 100;
        break;
      }
      unsigned nwaves_all = 0;
      {
        const int waves = hashR == 8 ? 2 : hashR == 16 ? 1 : mwaves;
        nwaves_all = (unsigned)std::max<int64_t>(1, std::min<int64_t>(tg_cdiv(ndof, waves), (int64_t)g_tg.num_cu * 32)) * (unsigned)waves;
      }
      const int64_t nsampled = std::max<int64_t>(1, tg_cdiv(ndof, step));
      const double mean_row = (double)hsum / (double)nsampled, est = step == 1 ? (double)hsum : (double)hsum * (double)step * 1.04;
      cap = (int64_t)(est * (1.0 + 1.25 * std::min(mean_row, (double)EL_MAXROW) / EL_BLOCK)) + ((int64_t)nwaves_all + 2) * EL_BLOCK;
    }
    rc = tg_dmalloc(&tcol, cap + 16);
    if (rc) break;
    rc = launch(1, 1, 0, cap, nullptr);
    if (rc) break;
    if (hstat == 0) break;
    tg_dfree(tcol);
    tcol = nullptr;
    if (hstat == 3 && hashR == 8) {           // (a row of more than 512 entries the sample did not see)
      hashR = 16;
      cap = 0;
    } else if (hstat == 1) {
      cap = (int64_t)hsum;
    } else {
      rc = 100;
    }
    if (attempt == 3 && !rc) {
      tg_set_error("element split: the symbolic pass did not come to an end");
      rc = 1;
    }
  }
  int64_t nnz = 0;
  if (!rc) rc = tg_exclusive_scan_i64(cnt, ndof, &nnz);
  if (!rc) rc = tg_dmalloc(&pl->k_col, nnz + TG_CSR_PAD);
  if (!rc) {
    int hmax = 0;
    hipMemsetAsync(status, 0, sizeof(int), g_tg.stream);
    hipLaunchKernelGGL(k_el_reorder, dim3((unsigned)std::max<int64_t>(1, std::min<int64_t>(tg_cdiv(ndof, 4), (int64_t)g_tg.num_cu * 16))),
                       dim3(256), 0, g_tg.stream, cnt, off, tcol, ndof, pl->k_col, status);
    if (hipMemcpyAsync(&hmax, status, sizeof(int), hipMemcpyDeviceToHost, g_tg.stream) != hipSuccess ||
        hipStreamSynchronize(g_tg.stream) != hipSuccess || hipGetLastError() != hipSuccess) {
      tg_set_error("element split: the reorder pass failed to run");
      rc = 1;
    }
    pl->max_k = hmax;
  }
  if (!rc) {
    pl->k_rowptr = cnt;
    cnt = nullptr;
    pl->k_nnz = nnz;
  } else {
    tg_dfree(pl->slot);
    tg_dfree(pl->k_col);
    pl->slot = nullptr, pl->k_col = nullptr;
  }
  cleanup();
  return rc;
}

/* K rows [dof0, dof1) (tg_elemplan_info) of M^T A M summed over the own cells of the plan, global columns; MatZeroRowsColumns
 * (tIGAr/common.py:1196-1204) applied when zero_dofs is given (a caller that adds chunks applies it to the sum instead).
 * a: the rows [a_row0, a_row0 + rows of a) of A, global columns; every entry of the rows [check_row0, check_row1) must couple
 * two nodes of a listed cell -- 100 otherwise (take tg_ptap_*). */
extern "C" int tg_elemplan_ptap(tg_elemplan_t pl, tg_csr_t a, int64_t a_row0, int64_t check_row0, int64_t check_row1,
                                const int32_t *zero_dofs, int64_t nzero, double diag, tg_csr_t *k_out) {
  TG_REQUIRE_INIT();
  TG_REQUIRE(pl && a && k_out, "null argument to tg_elemplan_ptap");
  TG_REQUIRE_CANONICAL(a);
  const int64_t nown = pl->own1 - pl->own0, ndof = pl->dof1 - pl->dof0;
  const int S = pl->S;
  double *blocks = nullptr;
  uint8_t *mask = nullptr;
  tg_csr_s *k = nullptr;
  auto cleanup = [&]() {
    tg_dfree(blocks);
    tg_dfree(mask);
  };
  int rc = tg_dmalloc(&blocks, nown * (int64_t)S * S + TG_CSR_PAD);
  if (rc) return rc;
  unsigned long long *counters = (unsigned long long *)(g_tg.scratch + 8), hc[4];
  hipMemsetAsync(blocks, 0, (size_t)(nown * (int64_t)S * S) * sizeof(double), g_tg.stream);
  hipMemsetAsync(counters, 0, sizeof(hc), g_tg.stream);
  if (a->nrows > 0 && pl->nfix)
    hipLaunchKernelGGL(k_el_scatter8, dim3(el_grid8(std::min<int64_t>(tg_cdiv(a->nrows, 4), (int64_t)g_tg.num_cu * 32))), dim3(256), 0,
                       g_tg.stream, a->rowptr, a->col, a->val, a->nrows, a_row0, pl->node0, pl->nnode, pl->nfix, pl->own0, pl->own1, S, check_row0,
                       check_row1, blocks, counters);
  else if (a->nrows > 0)
    hipLaunchKernelGGL(k_el_scatter, dim3(el_grid8(std::min<int64_t>(tg_cdiv(a->nrows, 4), (int64_t)g_tg.num_cu * 32))), dim3(256), 0,
                       g_tg.stream, a->rowptr, a->col, a->val, a->nrows, a_row0, pl->node0, pl->nnode, pl->nptr, pl->nlist, pl->own0, pl->own1,
                       S, check_row0, check_row1, blocks, counters);
  if (hipMemcpyAsync(hc, counters, sizeof(hc), hipMemcpyDeviceToHost, g_tg.stream) != hipSuccess ||
      hipStreamSynchronize(g_tg.stream) != hipSuccess || hipGetLastError() != hipSuccess) {
    tg_set_error("element split: the splitting pass failed to run");
    cleanup();
    return 1;
  }
  if (hc[2] || hc[3]) {          // an entry whose nodes share no cell (or a node in more than 64 cells): the row-wise kernels
    if (getenv("TIGAR_DEBUG"))
      fprintf(stderr, "[tigar] element split declined: %llu entries in blocks, %llu in other chunks' cells, %llu without a common cell%s\n",
              hc[0], hc[1], hc[2], hc[3] ? ", a node in more than 64 cells" : "");
    cleanup();
    return 100;
  }
  {
    const unsigned grid = el_grid8(std::min<int64_t>(nown, (int64_t)g_tg.num_cu * 64));
#define EL_DENSE(KERNEL)                                                                                                        \
  hipLaunchKernelGGL((KERNEL), dim3(grid), dim3(256), 0, g_tg.stream, blocks, S, nown, pl->cells->nodes, pl->own0, pl->b, pl->m->rowptr, \
                     pl->m->col, pl->m->val, pl->m_row0, pl->fl, pl->nf)
#define EL_DENSE_MFMA(NBV)                                                                                                      \
  hipLaunchKernelGGL((k_el_dense_mfma<NBV>), dim3(grid), dim3(256), 0, g_tg.stream, blocks, S, nown, pl->cells->nodes, pl->own0, pl->b, \
                     pl->m->rowptr, pl->m->col, pl->m->val, pl->m_row0, pl->fl, pl->nf, pl->mpos)
    const bool valu = getenv("TIGAR_EL_VALU") != nullptr;       // (the register-tile kernels: A/B runs)
    if (S <= 16) {
      if (valu) EL_DENSE(k_el_dense<1>); else EL_DENSE_MFMA(1);
    } else if (S <= 32) {
      if (valu) EL_DENSE(k_el_dense<2>); else EL_DENSE_MFMA(2);
    } else if (S <= 48) {
      if (valu) EL_DENSE(k_el_dense<4>); else EL_DENSE_MFMA(3);
    } else if (S <= 64) {
      if (valu) EL_DENSE(k_el_dense<4>); else EL_DENSE_MFMA(4);
    } else if (S <= EL_BIG_MS) {   // (3-D p = 4: 125 nodes; one workgroup per CU)
      const size_t lds = (size_t)(128 * EL_BIG_MS + 16 * 129 + 16 * 128) * sizeof(double);
      hipFuncSetAttribute((const void *)k_el_dense_big, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      hipLaunchKernelGGL(k_el_dense_big, dim3(el_grid8(std::min<int64_t>(nown, (int64_t)g_tg.num_cu * 8))), dim3(256), lds, g_tg.stream, blocks, S,
                         nown, pl->cells->nodes, pl->own0, pl->b, pl->m->rowptr, pl->m->col, pl->m->val, pl->m_row0, pl->fl, pl->nf);
    } else {
      cleanup();
      return 100;                 // (cells of more than 126 nodes / functions)
    }
#undef EL_DENSE
#undef EL_DENSE_MFMA
    if (hipGetLastError() != hipSuccess) {
      tg_set_error("element split: the element kernel failed to launch");
      cleanup();
      return 1;
    }
  }
  if (pl->k_nnz < 0) {
    rc = el_symbolic(pl);
    if (rc) {
      cleanup();
      return rc;
    }
  }
  if (nzero > 0) rc = tg_build_dof_mask(zero_dofs, nzero, pl->m->ncols, &mask);
  if (!rc) rc = tg_csr_alloc(ndof, pl->m->ncols, pl->k_nnz, &k);
  if (rc) {
    cleanup();
    return rc;
  }
  hipMemcpyAsync(k->rowptr, pl->k_rowptr, (size_t)(ndof + 1) * sizeof(int64_t), hipMemcpyDeviceToDevice, g_tg.stream);
  hipMemcpyAsync(k->col, pl->k_col, (size_t)pl->k_nnz * sizeof(int32_t), hipMemcpyDeviceToDevice, g_tg.stream);
  {
    const int capk = (std::max(pl->max_k, 1) + 7) & ~7;
    const unsigned grid = el_grid8(std::min<int64_t>(tg_cdiv(ndof, 4), (int64_t)g_tg.num_cu * 32));
#define EL_MERGE(LGV)                                                                                                                   \
  do {                                                                                                                                  \
    const size_t lds = (size_t)4 * (64 >> LGV) * capk * sizeof(double);                                                                 \
    hipFuncSetAttribute((const void *)k_el_merge<LGV>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);                         \
    hipLaunchKernelGGL((k_el_merge<LGV>), dim3(grid), dim3(256), lds, g_tg.stream, pl->iptr, pl->ient, blocks, pl->slot, pl->nf, S, ndof, capk, \
                       k->rowptr, k->col, mask, pl->dof0, diag, k->val);                                                                \
  } while (0)
    if (pl->nfmax <= 8) EL_MERGE(3);
    else if (pl->nfmax <= 16) EL_MERGE(4);
    else if (pl->nfmax <= 32) EL_MERGE(5);
    else EL_MERGE(6);
#undef EL_MERGE
  }
  if (hipStreamSynchronize(g_tg.stream) != hipSuccess || hipGetLastError() != hipSuccess) {
    tg_set_error("element split: the merge failed to run");
    rc = 1;
  }
  cleanup();
  if (rc) {
    tg_csr_destroy(k);
    return rc;
  }
  *k_out = k;
  return 0;
}
