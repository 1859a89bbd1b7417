// Half-storage product for the K p of a CG solve (tIGAr/common.py:1255-1258 hands K to a PETSc KSP; CG is only
// defined for a symmetric K, and PETSc's own answer to that is the SBAIJ format).
//
// The sliced copy of tg_sell.hip moves 8 B per stored entry and sits at 0.71-0.78 of the HBM peak: the only way to a
// faster product is fewer bytes.  K = M^T A M of ONE scalar field on a 3-D tensor-product patch is a box stencil of
// radius P on an n0 x n1 x n2 grid of control points, and symmetric.  This plan stores the diagonal and the entries
// above it only ((2P+1)^3 + 1) / 2 per row: 172 of 343 for P = 3) and uses each of them twice,
//
//     y[i]       += K[i][i+off] * x[i+off]       (the row, as before)
//     y[i+off]   += K[i][i+off] * x[i]           (the transposed entry),
//
// so a product moves HALF the bytes.  The second line is a scatter; what makes it cheap is the geometry, not the CSR
// arrays: the grid is cut into patches of at most 24 x 16 points in (x, y), a wave owns one patch and walks it plane by
// plane along z.  Everything the wave scatters lands in the window (patch + P points on every side) of the current and
// the next P planes, which it keeps in LDS as a ring of P + 1 planes (21 KB for P = 3, beside a ring of the same planes of
// x) and adds to by read - add - write -- one wave per window, LDS operations of a wave complete in order, so the sums are
// the same bit for bit in every run.  When
// a plane is finished its window goes to a staging array (1 % of the value bytes); a second small kernel adds, for
// every row, the windows that cover it (its own patch, up to 3 x 3 neighbours in the plane, the previous z chunk) in a
// fixed order.  No global atomics.
//
// Nothing about the origin of the matrix is assumed: P, n0, n1, n2 are read off the column offsets of one interior
// row, the conversion checks the length and every column index of every row it copies against the box stencil, and the
// finished plan is compared with the CSR product on a pseudo-random vector (which a matrix that is not symmetric fails
// by O(1)).  A matrix that does not pass keeps the sliced copy / the CSR kernel.
#include "tg_common.h"
#include <algorithm>
#include <utility>
#include <vector>

#ifndef SG_PX
#define SG_PX 24          // widest / highest patch (points), P <= 3
#endif
#ifndef SG_PY
#define SG_PY 16
#endif
// P = 4 (round 6; 3-D quartics: 729 entries per row, 365 stored): patches of 16 x 12 points -- two rings of five window
// planes of 24 x 20 entries are 38 KB, as many waves per CU as at P = 3 (with 24 x 16 patches: 61 KB, two waves)
#define SG_PX4 16
#define SG_PY4 12

typedef double sg_d2 __attribute__((ext_vector_type(2)));
typedef unsigned int sg_u2 __attribute__((ext_vector_type(2)));

template <int P>
struct sg_c {
  static constexpr int S = 2 * P + 1, NF = S * S * S, LC = (NF - 1) / 2;
  static constexpr int NP = LC + 1;          // stored positions per row: the diagonal and what follows it
  static constexpr int NG = (NP + 1) / 2;    // pairs of positions (16 bytes per lane)
  static constexpr int PX = P == 4 ? SG_PX4 : SG_PX, PY = P == 4 ? SG_PY4 : SG_PY, TAB = PX * PY;
  static constexpr int Wx = PX + 2 * P, Wy = PY + 2 * P, W = Wx * Wy;
};
static inline int sg_px(int P) { return P == 4 ? SG_PX4 : SG_PX; }
static inline int sg_py(int P) { return P == 4 ? SG_PY4 : SG_PY; }
// position -> offset in the box, positions in ascending column order starting at the diagonal
__host__ __device__ constexpr int sg_dx(int P, int pos) { return (pos + ((2 * P + 1) * (2 * P + 1) * (2 * P + 1) - 1) / 2) % (2 * P + 1) - P; }
__host__ __device__ constexpr int sg_dy(int P, int pos) {
  return ((pos + ((2 * P + 1) * (2 * P + 1) * (2 * P + 1) - 1) / 2) / (2 * P + 1)) % (2 * P + 1) - P;
}
__host__ __device__ constexpr int sg_dz(int P, int pos) {
  return (pos + ((2 * P + 1) * (2 * P + 1) * (2 * P + 1) - 1) / 2) / ((2 * P + 1) * (2 * P + 1)) - P;
}

// STORAGE order of the positions of a row.  Positions with the same in-plane offset (dx, dy) form a group, one member
// per plane dz = 1 .. P (and dz = 0 where (0, dy, dx) lies above the diagonal): consecutive stored values scatter into
// DIFFERENT planes of the window ring, so the read - add - write sequences of a group are independent (see the product
// kernel).  The S^2 - n0 groups without a dz = 0 member come first (P values each), then the n0 = (S^2 + 1) / 2 groups
// with one (P + 1 values each, the diagonal in the first of them).  Batches of the product kernel are whole groups.
template <int P>
struct sg_lay {
  static constexpr int S = 2 * P + 1, N0 = (S * S + 1) / 2;
  static constexpr int NA = (S * S - N0) * P, NP = NA + N0 * (P + 1);
  __host__ __device__ static constexpr int pos(int k) {       // storage index -> position
    const int i = k < NA ? N0 + k / P : (k - NA) / (P + 1);
    const int dz = k < NA ? k % P + 1 : (((k - NA) % (P + 1)) < P ? (k - NA) % (P + 1) + 1 : 0);
    return dz == 0 ? i : N0 + (dz - 1) * S * S + i;
  }
  __host__ __device__ static constexpr int inv(int pos) {     // position -> storage index
    const int dz = pos < N0 ? 0 : (pos - N0) / (S * S) + 1;
    const int i = pos < N0 ? pos : (pos - N0) % (S * S);
    return i < N0 ? NA + i * (P + 1) + (dz == 0 ? P : dz - 1) : (i - N0) * P + (dz - 1);
  }
  // batches: storage indices [bstart(b), bstart(b + 1)), even boundaries on group boundaries; their number is a multiple
  // of 4 (the product kernel keeps 4 batches of values in registers, buffer = batch mod 4)
  // (P = 4: NA = 160 = 17 batches of two groups (8 values) + 6 of one (4), then 41 groups of 5: 20 batches of two (10) and
  //  the last group with the padding (6): 44 batches)
  static constexpr int NBATCH = P == 4 ? 44 : P == 3 ? 24 : P == 2 ? 12 : 4;
  __host__ __device__ static constexpr int bstart(int b) {
    return P == 4   ? (b <= 17 ? 8 * b : b <= 23 ? 136 + 4 * (b - 17) : b < 44 ? 160 + 10 * (b - 23) : 366)
           : P == 3 ? (b <= 12 ? 6 * b : b < 24 ? 72 + 8 * (b - 12) : 172)
           : P == 2 ? (b <= 6 ? 4 * b : b < 12 ? 24 + 6 * (b - 6) : 64)
                    : (b <= 2 ? 2 * b : b == 3 ? 8 : 14);
  }
  static constexpr int GBMAX = P == 4 ? 5 : P == 3 ? 6 : P == 2 ? 5 : 3;   // pairs in the longest batch
};
static_assert(sg_lay<3>::NP == sg_c<3>::NP && sg_lay<2>::NP == sg_c<2>::NP && sg_lay<1>::NP == sg_c<1>::NP &&
                  sg_lay<4>::NP == sg_c<4>::NP && sg_lay<4>::NA == 160, "positions");
static_assert(sg_lay<3>::bstart(24) == 2 * sg_c<3>::NG && sg_lay<2>::bstart(12) == 2 * sg_c<2>::NG &&
                  sg_lay<1>::bstart(4) == 2 * sg_c<1>::NG && sg_lay<4>::bstart(44) == 2 * sg_c<4>::NG &&
                  sg_lay<4>::bstart(23) == sg_lay<4>::NA && sg_lay<4>::bstart(43) == 360,
              "batches cover the stored pairs");
static_assert(sg_lay<4>::inv(sg_lay<4>::pos(364)) == 364 && sg_lay<4>::inv(sg_lay<4>::pos(161)) == 161 &&
                  sg_lay<4>::pos(sg_lay<4>::NA + 4) == 0, "storage order, P = 4");
static_assert(sg_lay<3>::inv(sg_lay<3>::pos(171)) == 171 && sg_lay<3>::inv(sg_lay<3>::pos(72)) == 72 &&
                  sg_lay<2>::inv(sg_lay<2>::pos(40)) == 40 && sg_lay<3>::pos(sg_lay<3>::NA + 3) == 0, "storage order");

template <int... I, typename F>
__device__ __forceinline__ void sg_each(std::integer_sequence<int, I...>, F &&f) {
  (f(std::integral_constant<int, I>{}), ...);
}

struct tg_symgrid_s {
  int P = 0, n0 = 0, n1 = 0, n2 = 0;     // n2: planes of THIS row block
  int n2g = 0, zoff = 0;                 // planes of the whole grid, first plane of the block (several ranks: z slabs)
  int64_t row0 = 0;
  int npx = 0, npy = 0, nch = 0, m = 0, czmax = 0;
  int32_t *tabs = nullptr;      // device: x0[npx+1] | y0[npy+1] | z0[nch+1] | px_of[n0] | py_of[n1] | pc_of[n2]
  sg_d2 *val = nullptr;         // [patch][plane][sub-step][pair][lane]
  double *stage = nullptr;      // [patch][chunk][plane of the chunk + P][W]
  int64_t val_bytes = 0, stage_bytes = 0;
  int64_t rows_stored = 0;      // lanes that hold a row (the bytes a product reads: rows_stored * NG * 16)
  // several fields on one scalar basis (nf > 1): this object is the container -- gd / gf hold the geometry (tables, staging
  // array) of the diagonal and of the full blocks, vd[f] / vf[f * 4 + g] (f < g) the values
  int nf = 1;
  int64_t ncp = 0;
  // numbering of the fields: entry (field f, plane z, in-plane index ij) of x / y at f * fs + z * zs + ij -- field after field
  // (fs = ncp, zs = n0 n1: tIGAr/common.py:242-252) or plane by plane (fs = n0 n1, zs = nF n0 n1: dist.FieldSlabPath)
  int64_t fs = 0, zs = 0;
  tg_symgrid_s *gd = nullptr, *gf = nullptr;
  sg_d2 *vd[4] = {nullptr, nullptr, nullptr, nullptr};
  sg_d2 *vf[16] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr,
                   nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
};

struct sg_dev {
  int n0, n1, n2, npx, npy, nch, m, czmax, n2g, zoff;
  int zs;                  // entries of x / y between two planes of the grid (n0 n1; several fields plane by plane: nF n0 n1)
  const int32_t *x0, *y0, *z0, *px_of, *py_of, *pc_of;
};

static sg_dev sg_view(const tg_symgrid_s *s) {
  sg_dev d;
  d.n0 = s->n0, d.n1 = s->n1, d.n2 = s->n2, d.npx = s->npx, d.npy = s->npy, d.nch = s->nch, d.m = s->m, d.czmax = s->czmax;
  d.n2g = s->n2g, d.zoff = s->zoff;
  d.zs = s->zs > 0 ? (int)s->zs : s->n0 * s->n1;
  d.x0 = s->tabs;
  d.y0 = d.x0 + s->npx + 1;
  d.z0 = d.y0 + s->npy + 1;
  d.px_of = d.z0 + s->nch + 1;
  d.py_of = d.px_of + s->n0;
  d.pc_of = d.py_of + s->n1;
  return d;
}

void tg_symgrid_free(tg_symgrid_s *s) {
  if (!s) return;
  if (g_tg.ready) {
    tg_dfree(s->tabs);
    tg_dfree(s->val);
    tg_dfree(s->stage);
    for (int i = 0; i < 4; i++) tg_dfree(s->vd[i]);
    for (int i = 0; i < 16; i++) tg_dfree(s->vf[i]);
  }
  tg_symgrid_free(s->gd);
  tg_symgrid_free(s->gf);
  delete s;
}

// ---- conversion.  Row (ix, iy, iz) of a box stencil holds the entries dx in [-min(P, ix), min(P, n0-1-ix)] x (same in
// y, z) in lexicographic (dz, dy, dx) = ascending column order: the stored half is the TAIL of the CSR row from the
// diagonal on, one contiguous piece per row.  A workgroup takes 32 rows of a (patch, plane, sub-step) block: every wave
// streams the tails of 8 rows (coalesced), places the values by position in an LDS tile [row][position] and the tile goes
// out as [pair of positions][lane] pieces of 512 bytes.  fail[0] is set when a row has another length than its box or
// another column index at one of its places.
// (P = 4: 16 rows per workgroup -- the tile of 32 rows x 367 values would not fit the 64 KB of static LDS)
template <int P>
struct sg_cv {
  static constexpr int ROWS = P == 4 ? 16 : 32, RW = ROWS / 4, PARTS = 64 / ROWS;
};
template <int P>
__global__ void __launch_bounds__(256)
    k_symgrid_convert(sg_dev G, const int64_t *__restrict__ rowptr, const int32_t *__restrict__ col,
                      const double *__restrict__ val, sg_d2 *__restrict__ out, int64_t nblk, int *__restrict__ fail) {
  typedef sg_c<P> C;
  constexpr int SG_CV_ROWS = sg_cv<P>::ROWS, RW = sg_cv<P>::RW, PARTS = sg_cv<P>::PARTS;
  constexpr int LD = 2 * C::NG + 1;          // odd row length of the tile
  __shared__ double tile[SG_CV_ROWS * LD];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int64_t blk = (int64_t)blockIdx.x / PARTS;
  const int half = (int)(blockIdx.x % PARTS);
  const int sub = (int)(blk % G.m);
  const int64_t pz = blk / G.m;
  const int z = (int)(pz % G.n2), patch = (int)(pz / G.n2);
  const int a = patch % G.npx, b = patch / G.npx;
  const int xa = G.x0[a], pxv = G.x0[a + 1] - xa, ya = G.y0[b], pyv = G.y0[b + 1] - ya;
  const int cnt = pxv * pyv;
  const int t0 = sub * 64 + half * SG_CV_ROWS;
  if (sub * 64 >= cnt) return;                 // (a block no product touches)
  for (int i = tid; i < SG_CV_ROWS * LD; i += 256) tile[i] = 0.0;
  __syncthreads();
  const int n0 = G.n0, n01 = G.n0 * G.n1;
  const int zg = z + G.zoff;                  // (a z slab of the grid: the rows hold the box of the WHOLE grid)
  const int dzlo = -min(P, zg), dzhi = min(P, G.n2g - 1 - zg), nz = dzhi - dzlo + 1;
  const int grow0 = G.zoff * n01;              // global index of local row 0
  // lane rr < RW of a wave looks up row rr of the wave's eight (four at P = 4)
  int my_row = -1, my_len = 0;
  int64_t my_e0 = 0;
  {
    const int t = t0 + w * RW + lane;
    if (lane < RW && t < cnt) {                 // (lanes of the block beyond the patch get zeros: the product loads them)
      const int ly = t / pxv, lx = t - ly * pxv;
      my_row = (z * G.n1 + ya + ly) * n0 + xa + lx;
      my_e0 = rowptr[my_row];
      my_len = (int)(rowptr[my_row + 1] - my_e0);
    }
  }
  // phase 1: every load of the wave's eight tails is issued (addresses from the row pointers alone) ...
  constexpr int IT = (C::NP + 63) / 64;       // a tail holds at most NP entries
  double vv[RW][IT];
  int cc[RW][IT];
  bool bad = false;
#pragma unroll
  for (int rr = 0; rr < RW; rr++) {
    const int row = __builtin_amdgcn_readlane(my_row, rr);
    const int len = __builtin_amdgcn_readlane(my_len, rr);
    const int64_t e0 = ((int64_t)__builtin_amdgcn_readlane((int)(my_e0 >> 32), rr) << 32) |
                       (unsigned)__builtin_amdgcn_readlane((int)(my_e0 & 0xffffffff), rr);
    const int ix = row % n0, iy = (row / n0) % G.n1;
    const int kd = row < 0 ? 0 : ((0 - dzlo) * (min(P, G.n1 - 1 - iy) + min(P, iy) + 1) + min(P, iy)) * (min(P, n0 - 1 - ix) + min(P, ix) + 1) + min(P, ix);
#pragma unroll
    for (int it = 0; it < IT; it++) {
      const int k = kd + lane + 64 * it;
      const bool on = row >= 0 && k < len;
      vv[rr][it] = on ? __builtin_nontemporal_load(val + e0 + k) : 0.0;
      cc[rr][it] = on ? __builtin_nontemporal_load(col + e0 + k) : 0;
    }
  }
  // ... phase 2: the values go to their places in the tile
#pragma unroll
  for (int rr = 0; rr < RW; rr++) {
    const int row = __builtin_amdgcn_readlane(my_row, rr);
    if (row < 0) break;
    const int len = __builtin_amdgcn_readlane(my_len, rr);
    const int ix = row % n0, iy = (row / n0) % G.n1;
    const int dxlo = -min(P, ix), dxhi = min(P, n0 - 1 - ix), nx = dxhi - dxlo + 1;
    const int dylo = -min(P, iy), dyhi = min(P, G.n1 - 1 - iy), ny = dyhi - dylo + 1;
    if (len != nx * ny * nz) {
      bad = true;
      continue;
    }
    const int nxy = nx * ny;
    const int mxy = (65536 + nxy - 1) / nxy, mx = (65536 + nx - 1) / nx;    // k / d = (k * m) >> 16 for k d < 65536
    const int kd = ((0 - dzlo) * ny + (0 - dylo)) * nx + (0 - dxlo);          // the diagonal
    double *trow = tile + (w * RW + rr) * LD;
#pragma unroll
    for (int it = 0; it < IT; it++) {
      const int k = kd + lane + 64 * it;
      if (k < len) {
        const int qz = (k * mxy) >> 16, rem = k - qz * nxy;
        const int qy = (rem * mx) >> 16, qx = rem - qy * nx;
        const int dz = qz + dzlo, dy = qy + dylo, dx = qx + dxlo;
        bad |= cc[rr][it] != grow0 + row + dx + n0 * dy + n01 * dz;
        trow[sg_lay<P>::inv(((dz + P) * C::S + dy + P) * C::S + dx + P - C::LC)] = vv[rr][it];
      }
    }
  }
  __syncthreads();
  sg_d2 *o = out + blk * (int64_t)(C::NG * 64) + half * SG_CV_ROWS;
  for (int i = tid; i < C::NG * SG_CV_ROWS; i += 256) {
    const int g = i / SG_CV_ROWS, l = i - g * SG_CV_ROWS;
    sg_d2 v;
    v.x = tile[l * LD + 2 * g], v.y = tile[l * LD + 2 * g + 1];
    o[g * 64 + l] = v;
  }
  if (bad) atomicExch(fail, 1);
}

// ---- the product: one wave per (patch, z chunk).  Two rings of P + 1 window planes in LDS: the sums (current plane and
// the P planes the transposed entries reach) and the x the wave multiplies with (loaded once per plane, 43 KB together at
// P = 3 -> 3 waves per CU).  For a stored value a = K[i][i+off] ONE window index serves both uses: sum_i += a * xs[idx],
// acc[idx] += a * x_i.  The vector memory pipe carries the value stream only; its loads are issued three batches (18 KB
// per wave) before they are multiplied.  (With x gathered from global memory -- 171 gathers per row through the L1, whose
// 32 KB do not hold the windows of the waves of a CU -- the product took 0.55 ms longer at cfg3's size, and vmcnt, which
// counts in issue order, tied the depth of the value ring to that of the gathers.)
//
// acc[idx] += v for the 64 lanes of the wave (64 distinct places) is a plain read - add - write, not ds_add_f64: the LDS
// adds fp64 atomics at about one lane per clock, which alone would take as long as the whole product.  It is safe because
// the wave is the only one on its window and its LDS operations complete in order -- but the places of DIFFERENT lanes
// overlap between positions (lane i at dx + 1 is lane i + 1 at dx), so the order of these accesses must stay what the
// program says, whatever the compiler can prove about one lane's own addresses: a compiler barrier after every group
// (`volatile` would do as well but makes the backend wait for ALL outstanding loads at every access).  The members of a
// group (see sg_lay) go to different planes of the ring, their reads are issued together and then their writes.
//
// xw: the window of x this rank may read (its own rows and the halo planes of its z neighbours), local row r at
// xw[xoff + r]; reads beyond it return 0 (buffer range check; such entries of K are stored zeros: rows at the faces of the
// grid).  What is scattered beyond the block's last plane is dropped: those rows belong to the next rank, which holds the
// transposed entries in ITS rows (k_symgrid_lowhalo).
template <int P>
__global__ void __launch_bounds__(64)
    k_symgrid_spmv(sg_dev G, const sg_d2 *__restrict__ val, const double *__restrict__ xw, int xoff, int xlen,
                      double *__restrict__ stage, int64_t nwaves, int c_begin, int c_count, const double *__restrict__ gate,
                      double gate_tol) {
  typedef sg_c<P> C;
  typedef sg_lay<P> Y;
  constexpr int Wx = C::Wx, W = C::W, GB = Y::GBMAX, NB = Y::NBATCH, NXL = (W + 63) / 64;
  __shared__ double acc[(P + 1) * W];
  __shared__ double xs[(P + 1) * W];
  __shared__ unsigned short tab[C::TAB];
  if (gate && !(*gate > gate_tol)) return;
  const int lane = threadIdx.x;
  const int64_t L = tg_xcd_block(blockIdx.x, nwaves);
  if (L >= nwaves) return;
  const int c = c_begin + (int)(L % c_count), patch = (int)(L / c_count);
  const int a = patch % G.npx, b = patch / G.npx;
  const int xa = G.x0[a], pxv = G.x0[a + 1] - xa, ya = G.y0[b], pyv = G.y0[b + 1] - ya;
  const int za = G.z0[c], zb = G.z0[c + 1];
  const int cnt = pxv * pyv, msub = (cnt + 63) >> 6;
  for (int t = lane; t < cnt; t += 64) {
    const int ly = t / pxv;
    tab[t] = (unsigned short)((ly << 8) | (t - ly * pxv));
  }
  for (int e = lane; e < (P + 1) * W; e += 64) acc[e] = 0.0;
  const __amdgpu_buffer_rsrc_t xr =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<double *>(xw), 0, (unsigned)xlen * 8u, 0x00020000);
  const int n0 = G.n0;
  // in-plane offset of this lane's window entries (the same for every plane); out of the grid: no read (0)
  unsigned woff[NXL];
#pragma unroll
  for (int k = 0; k < NXL; k++) {
    const int e = k * 64 + lane, wy = e / Wx, wx = e - wy * Wx;
    const int gy = ya - P + wy, gx = xa - P + wx;
    woff[k] = (e < W && gy >= 0 && gy < G.n1 && gx >= 0 && gx < n0) ? (unsigned)(gy * n0 + gx + xoff) : 0xffffffffu;
  }
  auto load_plane = [&](int zp, double *dst) {       // plane zp of the window (beyond the block: the halo, else 0)
#pragma unroll
    for (int k = 0; k < NXL; k++)
      dst[k] = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(
                                              xr, woff[k] == 0xffffffffu ? 0xffffffffu : (woff[k] + (unsigned)(zp * G.zs)) * 8u, 0, 0));
  };
  auto store_plane = [&](int zp, const double *src) {
    const int s0 = (zp % (P + 1)) * W;
#pragma unroll
    for (int k = 0; k < NXL; k++)
      if (k * 64 + lane < W) xs[s0 + k * 64 + lane] = src[k];
  };
  {
    double tmp[NXL];
#pragma unroll
    for (int d = 0; d <= P; d++) {
      load_plane(za + d, tmp);
      store_plane(za + d, tmp);
    }
  }
  __syncthreads();
  double *st = stage + ((int64_t)patch * G.nch + c) * (int64_t)(G.czmax + P) * W;
  const sg_d2 *vp = val + (int64_t)patch * G.n2 * G.m * (int64_t)(C::NG * 64) + lane;
  struct ctx_t {
    int lb;
    const sg_d2 *v;
  };
  auto ctx_of = [&](int z, int sub) {
    ctx_t k;
    const int t = sub * 64 + lane, zc = min(z, G.n2 - 1);
    const int tl = t < cnt ? tab[t] : 0, ly = tl >> 8, lx = tl & 255;
    k.lb = ly * Wx + lx;
    k.v = vp + ((int64_t)zc * G.m + sub) * (C::NG * 64);
    return k;
  };
  constexpr int VD = 4;      // batches of values held in registers (deeper rings: no gain)
  sg_d2 vv[VD][GB];
  auto issue_v = [&](const ctx_t &k, auto btc) {
    constexpr int BT = decltype(btc)::value, b0 = Y::bstart(BT), b1 = Y::bstart(BT + 1);
#pragma unroll
    for (int j = 0; j < (b1 - b0) / 2; j++) vv[BT % VD][j] = __builtin_nontemporal_load(k.v + (b0 / 2 + j) * 64);
  };
  ctx_t cur = ctx_of(za, 0);
  sg_each(std::make_integer_sequence<int, VD - 1>{}, [&](auto btc) { issue_v(cur, btc); });
  for (int z = za; z < zb; z++) {
    int so[P + 1];
#pragma unroll
    for (int d = 0; d <= P; d++) so[d] = ((z + d) % (P + 1)) * W;
    double xnext[NXL];
    load_plane(z + P + 1, xnext);                      // (arrives while the plane is multiplied)
    for (int sub = 0; sub < msub; sub++) {
      const ctx_t nxt = (sub + 1 < msub) ? ctx_of(z, sub + 1) : ctx_of(z + 1, 0);
      const double xi = xs[so[0] + cur.lb + P * Wx + P];
      double sum = 0.0;
      sg_each(std::make_integer_sequence<int, NB>{}, [&](auto btc) {
        constexpr int BT = decltype(btc)::value;
        if constexpr (BT + VD - 1 < NB)
          issue_v(cur, std::integral_constant<int, BT + VD - 1>{});
        else
          issue_v(nxt, std::integral_constant<int, BT + VD - 1 - NB>{});
        __builtin_amdgcn_sched_barrier(0);
        constexpr int b0 = Y::bstart(BT), b1 = Y::bstart(BT + 1);
        constexpr int gs = b0 < Y::NA ? P : P + 1, ng = (b1 - b0 + gs - 1) / gs;
#pragma unroll
        for (int gq = 0; gq < ng; gq++) {
          double r[gs], cc[gs];
          int idx[gs];
#pragma unroll
          for (int t = 0; t < gs; t++) {
            constexpr int Pc = P;
            const int kk = b0 + gq * gs + t, l = kk - b0;
            const int pos = (kk < b1 && kk < Y::NP) ? Y::pos(kk) : -1;
            const double vq = (l & 1) ? vv[BT % VD][l >> 1].y : vv[BT % VD][l >> 1].x;
            if (pos == 0) {
              sum += vq * xi;
            } else if (pos > 0) {
              idx[t] = so[sg_dz(Pc, pos)] + cur.lb + (sg_dy(Pc, pos) + P) * Wx + sg_dx(Pc, pos) + P;
              sum += vq * xs[idx[t]];
              r[t] = acc[idx[t]];
              cc[t] = vq * xi;
            }
          }
#pragma unroll
          for (int t = 0; t < gs; t++) {
            const int kk = b0 + gq * gs + t;
            const int pos = (kk < b1 && kk < Y::NP) ? Y::pos(kk) : -1;
            if (pos > 0) acc[idx[t]] = r[t] + cc[t];
          }
          asm volatile("" ::: "memory");
        }
        __builtin_amdgcn_sched_barrier(0);
      });
      acc[so[0] + cur.lb + P * Wx + P] += sum;
      asm volatile("" ::: "memory");
      cur = nxt;
    }
    __syncthreads();
    double *sp = st + (int64_t)(z - za) * W;
    for (int e = lane; e < W; e += 64) {
      sp[e] = acc[so[0] + e];
      acc[so[0] + e] = 0.0;
    }
    store_plane(z + P + 1, xnext);                     // (into the slot of plane z, which no row reads any more)
    __syncthreads();
  }
#pragma unroll
  for (int d = 0; d < P; d++) {
    const int z = zb + d;
    if (z >= G.n2) break;
    const int s0 = (z % (P + 1)) * W;
    double *sp = st + (int64_t)(z - za) * W;
    for (int e = lane; e < W; e += 64) sp[e] = acc[s0 + e];
  }
}

// ------------------------------------------------------------------------------------------------------------------------
// Several fields on one scalar basis (round 6): K of nF x nF blocks K_fg, every block the box stencil of the scalar grid,
// K symmetric as a whole (K_gf = K_fg^T) -- the stiffness matrix of linear elasticity, tIGAr/common.py:1891-1914 +
// 1255-1258.  The diagonal blocks are symmetric box stencils themselves: half storage and the kernels above, block by
// block.  An off-diagonal pair is stored ONCE, as the full box of K_fg (f < g, (2P+1)^3 positions per row), and used twice,
//     y_f[i]       += K_fg[i][i+off] * x_g[i+off]      (the row)
//     y_g[i+off]   += K_fg[i][i+off] * x_f[i]          (the row of K_gf, transposed),
// so a product moves nF (S^3 + 1) / 2 + nF (nF - 1) / 2 S^3 values per node instead of nF^2 S^3: a half.  The scatter now
// reaches the P planes BELOW the row's plane as well: rings of 2P + 1 window planes (patches of 16 x 12 points: 44 KB at
// P = 3), a finished plane is the one P below the current, and a chunk hands P planes of windows to the chunk before it as
// well as to the next.  The row sums go straight to y_f (each row belongs to one lane of one wave of the launch: a plain
// read - add - write, the launches of a product in a fixed order), the windows through the staging array to y_g.
template <int P>
struct sg_cf {
  static constexpr int S = 2 * P + 1, NP = S * S * S, NG = (NP + 1) / 2, NR = S;     // NR: planes of a ring
  static constexpr int PX = SG_PX4, PY = SG_PY4, TAB = PX * PY;
  static constexpr int Wx = PX + 2 * P, Wy = PY + 2 * P, W = Wx * Wy;
  // storage order: groups of equal (dx, dy), members dz = -P .. P (consecutive values go to different planes of the ring)
  __host__ __device__ static constexpr int inv(int dx, int dy, int dz) { return ((dy + P) * S + (dx + P)) * S + (dz + P); }
  static constexpr int NBR = (S * S + 1) / 2, NBATCH = (NBR + 3) / 4 * 4;           // batches of two groups; empty ones pad
  __host__ __device__ static constexpr int bstart(int b) { return b < NBR ? b * 2 * S : 2 * NG; }
  static constexpr int GBMAX = S;
};
static_assert(sg_cf<3>::bstart(sg_cf<3>::NBR) - sg_cf<3>::bstart(sg_cf<3>::NBR - 1) == 8 && sg_cf<2>::NBATCH == 16, "full boxes");

// conversion of block (f, g) of the rows of K: one wave per row of the scalar grid, the entries of the block straight to
// their places (the copies are zeroed before: lanes beyond a patch, positions beyond the grid).  FULL: the whole box;
// otherwise the tail from the diagonal on in the order of sg_lay (the diagonal blocks).  Every row's length and every
// column index of the block are checked against the box stencil.
struct sg_blk {
  int nf, f, g;
  int64_t ncp;                 // points of the scalar grid = rows per field
  int64_t fs, zs;              // numbering: (field, plane, in-plane index) at f fs + z zs + ij
};
template <int P, bool FULL>
__global__ void __launch_bounds__(256)
    k_symgrid_convert_blk(sg_dev G, sg_blk B, const int64_t *__restrict__ rowptr, const int32_t *__restrict__ col,
                          const double *__restrict__ val, double *__restrict__ out, int *__restrict__ fail) {
  typedef sg_c<P> C;
  constexpr int NG = FULL ? sg_cf<P>::NG : C::NG;
  const int lane = threadIdx.x & 63;
  const int64_t nw = (int64_t)gridDim.x * 4;
  const int n0 = G.n0, n01 = G.n0 * G.n1;
  bool bad = false;
  for (int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); row < B.ncp; row += nw) {
    const int ix = (int)(row % n0), iy = (int)((row / n0) % G.n1), z = (int)(row / n01);
    const int a = G.px_of[ix], b = G.py_of[iy];
    const int xa = G.x0[a], pxv = G.x0[a + 1] - xa, ya = G.y0[b];
    const int t = (iy - ya) * pxv + (ix - xa), sub = t >> 6, l = t & 63;
    const int64_t blk = ((int64_t)(b * G.npx + a) * G.n2 + z) * G.m + sub;
    double *o = out + (blk * (int64_t)(NG * 64) + l) * 2;
    const int dxlo = -min(P, ix), nx = min(P, n0 - 1 - ix) - dxlo + 1;
    const int dylo = -min(P, iy), ny = min(P, G.n1 - 1 - iy) - dylo + 1;
    const int dzlo = -min(P, z), nz = min(P, G.n2 - 1 - z) - dzlo + 1;
    const int len = nx * ny * nz;
    const int64_t ij = row - (int64_t)z * n01;
    const int64_t r = B.f * B.fs + z * B.zs + ij, e0 = rowptr[r];
    if (rowptr[r + 1] - e0 != (int64_t)len * B.nf) {
      bad = true;
      continue;
    }
    const int nxy = nx * ny;
    const int kd = FULL ? 0 : ((0 - dzlo) * ny + (0 - dylo)) * nx + (0 - dxlo);
    // (the columns of a row ascend: field after field, the nF boxes follow each other; plane by plane, the nF pieces of a
    //  plane dz do -- a piece = the nx ny entries of box g in that plane)
    const bool fm = B.zs == n01;
    const int64_t cb = B.g * B.fs + z * B.zs + ij;
    for (int k = kd + lane; k < len; k += 64) {
      const int qz = k / nxy, rem = k - qz * nxy, qy = rem / nx, qx = rem - qy * nx;
      const int dz = qz + dzlo, dy = qy + dylo, dx = qx + dxlo;
      const int64_t e = fm ? e0 + (int64_t)B.g * len + k : e0 + ((int64_t)qz * B.nf + B.g) * nxy + rem;
      bad |= (int64_t)col[e] != cb + dx + n0 * dy + B.zs * dz;
      const int si = FULL ? sg_cf<P>::inv(dx, dy, dz) : sg_lay<P>::inv(((dz + P) * C::S + dy + P) * C::S + dx + P - C::LC);
      o[(int64_t)(si >> 1) * 128 + (si & 1)] = val[e];
    }
  }
  if (bad) atomicExch(fail, 1);
}

// product with a stored full block: one wave per (patch, z chunk), as k_symgrid_spmv with rings of 2P + 1 planes
template <int P>
__global__ void __launch_bounds__(64)
    k_symgrid_spmv_full(sg_dev G, const sg_d2 *__restrict__ val, const double *__restrict__ xg, const double *__restrict__ xf,
                        double *__restrict__ yf, int xlen, int ylen, double *__restrict__ stage, int64_t nwaves,
                        const double *__restrict__ gate, double gate_tol) {
  typedef sg_cf<P> Y;
  constexpr int Wx = Y::Wx, W = Y::W, GB = Y::GBMAX, NB = Y::NBATCH, NXL = (W + 63) / 64, NR = Y::NR, S = Y::S;
  __shared__ double acc[NR * W];
  __shared__ double xs[NR * W];
  __shared__ unsigned short tab[Y::TAB];
  if (gate && !(*gate > gate_tol)) return;
  const int lane = threadIdx.x;
  const int64_t L = tg_xcd_block(blockIdx.x, nwaves);
  if (L >= nwaves) return;
  const int c = (int)(L % G.nch), patch = (int)(L / G.nch);
  const int a = patch % G.npx, b = patch / G.npx;
  const int xa = G.x0[a], pxv = G.x0[a + 1] - xa, ya = G.y0[b], pyv = G.y0[b + 1] - ya;
  const int za = G.z0[c], zb = G.z0[c + 1];
  const int cnt = pxv * pyv, msub = (cnt + 63) >> 6;
  for (int t = lane; t < cnt; t += 64) {
    const int ly = t / pxv;
    tab[t] = (unsigned short)((ly << 8) | (t - ly * pxv));
  }
  for (int e = lane; e < NR * W; e += 64) acc[e] = 0.0;
  const __amdgpu_buffer_rsrc_t xr =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<double *>(xg), 0, (unsigned)xlen * 8u, 0x00020000);
  const int n0 = G.n0;
  unsigned woff[NXL];
#pragma unroll
  for (int k = 0; k < NXL; k++) {
    const int e = k * 64 + lane, wy = e / Wx, wx = e - wy * Wx;
    const int gy = ya - P + wy, gx = xa - P + wx;
    woff[k] = (e < W && gy >= 0 && gy < G.n1 && gx >= 0 && gx < n0) ? (unsigned)(gy * n0 + gx) : 0xffffffffu;
  }
  auto slot = [&](int zp) { return ((zp + NR) % NR) * W; };      // (zp >= -P)
  auto load_plane = [&](int zp, double *dst) {       // plane zp of the window of x_g (outside the grid: 0)
    const bool in = zp >= 0 && zp < G.n2;
#pragma unroll
    for (int k = 0; k < NXL; k++)
      dst[k] = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(
                                              xr, (woff[k] == 0xffffffffu || !in) ? 0xffffffffu : (woff[k] + (unsigned)(zp * G.zs)) * 8u, 0, 0));
  };
  auto store_plane = [&](int zp, const double *src) {
    const int s0 = slot(zp);
#pragma unroll
    for (int k = 0; k < NXL; k++)
      if (k * 64 + lane < W) xs[s0 + k * 64 + lane] = src[k];
  };
  {
    double tmp[NXL];
#pragma unroll
    for (int d = -P; d <= P; d++) {
      load_plane(za + d, tmp);
      store_plane(za + d, tmp);
    }
  }
  __syncthreads();
  double *st = stage + ((int64_t)patch * G.nch + c) * (int64_t)(G.czmax + 2 * P) * W;      // planes za - P .. zb + P - 1
  const sg_d2 *vp = val + (int64_t)patch * G.n2 * G.m * (int64_t)(Y::NG * 64) + lane;
  // (no branch inside the walk: a lane without a row reads row 0 and stores through the buffer descriptor with an offset
  //  beyond its range, which the hardware drops -- a divergent `if` around the store splits the block the value loads are
  //  scheduled in and the loads of a whole row end up in flight at once: 4 KB of scratch per lane)
  const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc(yf, 0, (unsigned)ylen * 8u, 0x00020000);
  struct ctx_t {
    int lb;
    const sg_d2 *v;
    unsigned yoff;           // byte offset of the row in y_f, 0xffffffff: no row in this lane
    double xi, yo;
  };
  auto ctx_of = [&](int z, int sub) {
    ctx_t k;
    const int t = sub * 64 + lane, zc = min(z, G.n2 - 1);
    const bool on = t < cnt && z < zb;
    const int tl = t < cnt ? tab[t] : 0, ly = tl >> 8, lx = tl & 255;
    k.lb = ly * Wx + lx;
    k.v = vp + ((int64_t)zc * G.m + sub) * (Y::NG * 64);
    const int row = on ? zc * G.zs + (ya + ly) * n0 + xa + lx : 0;
    k.yoff = on ? (unsigned)row * 8u : 0xffffffffu;
    k.xi = xf[row];
    k.yo = yf[row];
    return k;
  };
  constexpr int VD = 4;
  sg_d2 vv[VD][GB];
  auto issue_v = [&](const ctx_t &k, auto btc) {
    constexpr int BT = decltype(btc)::value, b0 = Y::bstart(BT), b1 = Y::bstart(BT + 1);
#pragma unroll
    for (int j = 0; j < (b1 - b0) / 2; j++) vv[BT % VD][j] = __builtin_nontemporal_load(k.v + (b0 / 2 + j) * 64);
  };
  ctx_t cur = ctx_of(za, 0);
  sg_each(std::make_integer_sequence<int, VD - 1>{}, [&](auto btc) { issue_v(cur, btc); });
  for (int z = za; z < zb; z++) {
    int so[NR];
#pragma unroll
    for (int d = 0; d < NR; d++) so[d] = slot(z + d - P);
    double xnext[NXL];
    load_plane(z + P + 1, xnext);
    for (int sub = 0; sub < msub; sub++) {
      const ctx_t nxt = (sub + 1 < msub) ? ctx_of(z, sub + 1) : ctx_of(z + 1, 0);
      const double xi = cur.xi;
      double sum = 0.0;
      sg_each(std::make_integer_sequence<int, NB>{}, [&](auto btc) {
        constexpr int BT = decltype(btc)::value;
        if constexpr (BT + VD - 1 < NB)
          issue_v(cur, std::integral_constant<int, BT + VD - 1>{});
        else
          issue_v(nxt, std::integral_constant<int, BT + VD - 1 - NB>{});
        __builtin_amdgcn_sched_barrier(0);
        constexpr int b0 = Y::bstart(BT), b1 = Y::bstart(BT + 1);
        constexpr int ng = (b1 - b0 + S - 1) / S;
#pragma unroll
        for (int gq = 0; gq < ng; gq++) {
          double r[S], cc[S];
          int idx[S];
#pragma unroll
          for (int t = 0; t < S; t++) {
            const int kk = b0 + gq * S + t, l = kk - b0;
            const bool on = kk < b1 && kk < Y::NP;
            const int gi = kk / S, dz = kk - gi * S - P, dx = gi % S - P, dy = gi / S - P;
            const double vq = (l & 1) ? vv[BT % VD][l >> 1].y : vv[BT % VD][l >> 1].x;
            if (on) {
              idx[t] = so[dz + P] + cur.lb + (dy + P) * Wx + dx + P;
              sum += vq * xs[idx[t]];
              r[t] = acc[idx[t]];
              cc[t] = vq * xi;
            }
          }
#pragma unroll
          for (int t = 0; t < S; t++) {
            const int kk = b0 + gq * S + t;
            if (kk < b1 && kk < Y::NP) acc[idx[t]] = r[t] + cc[t];
          }
          asm volatile("" ::: "memory");
        }
        __builtin_amdgcn_sched_barrier(0);
      });
      __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(sg_u2, cur.yo + sum), yr, cur.yoff, 0, 0);
      cur = nxt;
    }
    __syncthreads();
    // plane z - P is finished: no later row of the chunk reaches it
    double *sp = st + (int64_t)(z - za) * W;
    for (int e = lane; e < W; e += 64) {
      sp[e] = acc[so[0] + e];
      acc[so[0] + e] = 0.0;
    }
    store_plane(z + P + 1, xnext);                     // (into the slot of plane z - P, which no row reads any more)
    __syncthreads();
  }
#pragma unroll
  for (int d = 0; d < 2 * P; d++) {                    // planes zb - P .. zb + P - 1
    const int z = zb - P + d;
    double *sp = st + (int64_t)(z - (za - P)) * W;
    for (int e = lane; e < W; e += 64) sp[e] = acc[slot(z) + e];
  }
}

// y_g[i] += sum of the windows of a full block that cover point i (its own chunk, P planes handed down by the next chunk,
// P planes handed up by the one before), in a fixed order
template <int P>
__global__ void __launch_bounds__(256)
    k_symgrid_combine_full(sg_dev G, const double *__restrict__ stage, double *__restrict__ y, int64_t nlines,
                           const double *__restrict__ gate, double gate_tol) {
  typedef sg_cf<P> Y;
  constexpr int Wx = Y::Wx, W = Y::W;
  if (gate && !(*gate > gate_tol)) return;
  const int64_t cstride = (int64_t)(G.czmax + 2 * P) * W;
  for (int ix = threadIdx.x; ix < G.n0; ix += 256) {
    const int a0 = G.px_of[ix];
    const int alo = (a0 > 0 && ix - G.x0[a0] < P) ? a0 - 1 : a0;
    const int ahi = (a0 + 1 < G.npx && G.x0[a0 + 1] - ix <= P) ? a0 + 1 : a0;
    const int64_t astride = (int64_t)G.nch * cstride;
    const int64_t o0 = (int64_t)a0 * astride + (ix - G.x0[a0]);
    const int64_t olo = alo < a0 ? (int64_t)alo * astride + (ix - G.x0[alo]) : -1;
    const int64_t ohi = ahi > a0 ? (int64_t)ahi * astride + (ix - G.x0[ahi]) : -1;
    for (int64_t line = blockIdx.x; line < nlines; line += gridDim.x) {
      const int iy = (int)(line % G.n1), iz = (int)(line / G.n1);
      const int b0 = G.py_of[iy], c0 = G.pc_of[iz];
      const int blo = (b0 > 0 && iy - G.y0[b0] < P) ? b0 - 1 : b0;
      const int bhi = (b0 + 1 < G.npy && G.y0[b0 + 1] - iy <= P) ? b0 + 1 : b0;
      const int clo = (c0 > 0 && iz - G.z0[c0] < P) ? c0 - 1 : c0;
      const int chi = (c0 + 1 < G.nch && G.z0[c0 + 1] - iz <= P) ? c0 + 1 : c0;
      double s = 0.0;
      for (int c = clo; c <= chi; c++)
        for (int b = blo; b <= bhi; b++) {
          const double *sb = stage + ((int64_t)(b * G.npx) * G.nch + c) * cstride + (int64_t)(iz - (G.z0[c] - P)) * W +
                             (iy - G.y0[b] + P) * Wx + P;
          if (olo >= 0) s += sb[olo];
          s += sb[o0];
          if (ohi >= 0) s += sb[ohi];
        }
      y[(int64_t)iz * G.zs + iy * G.n0 + ix] += s;
    }
  }
}

// y[i] = sum of the windows that cover point i, in a fixed order.  One workgroup per grid line (iy, iz): which patch rows
// and chunks cover the line is wave-uniform, a thread only looks up the patch column of its ix.
template <int P>
__global__ void __launch_bounds__(256)
    k_symgrid_combine(sg_dev G, const double *__restrict__ stage, double *__restrict__ y, int64_t nlines,
                      const double *__restrict__ gate, double gate_tol) {
  typedef sg_c<P> C;
  constexpr int Wx = C::Wx, W = C::W;
  if (gate && !(*gate > gate_tol)) return;
  const int64_t cstride = (int64_t)(G.czmax + P) * W;
  // a thread keeps its ix for every line of the workgroup: which patch columns cover it is looked up once
  for (int ix = threadIdx.x; ix < G.n0; ix += 256) {
    const int a0 = G.px_of[ix];
    const int alo = (a0 > 0 && ix - G.x0[a0] < P) ? a0 - 1 : a0;
    const int ahi = (a0 + 1 < G.npx && G.x0[a0 + 1] - ix <= P) ? a0 + 1 : a0;
    const int64_t astride = (int64_t)G.nch * cstride;
    const int64_t o0 = (int64_t)a0 * astride + (ix - G.x0[a0]);
    const int64_t olo = alo < a0 ? (int64_t)alo * astride + (ix - G.x0[alo]) : -1;
    const int64_t ohi = ahi > a0 ? (int64_t)ahi * astride + (ix - G.x0[ahi]) : -1;
    for (int64_t line = blockIdx.x; line < nlines; line += gridDim.x) {
      const int iy = (int)(line % G.n1), iz = (int)(line / G.n1);
      const int b0 = G.py_of[iy], c0 = G.pc_of[iz];
      const int blo = (b0 > 0 && iy - G.y0[b0] < P) ? b0 - 1 : b0;
      const int bhi = (b0 + 1 < G.npy && G.y0[b0 + 1] - iy <= P) ? b0 + 1 : b0;
      const int clo = (c0 > 0 && iz - G.z0[c0] < P) ? c0 - 1 : c0;
      double s = 0.0;
      for (int c = clo; c <= c0; c++)
        for (int b = blo; b <= bhi; b++) {
          const double *sb = stage + ((int64_t)(b * G.npx) * G.nch + c) * cstride + (int64_t)(iz - G.z0[c]) * W +
                             (iy - G.y0[b] + P) * Wx + P;
          if (olo >= 0) s += sb[olo];        // (ascending patch column: the order of the sum is fixed)
          s += sb[o0];
          if (ohi >= 0) s += sb[ohi];
        }
      y[(int64_t)iz * G.zs + iy * G.n0 + ix] = s;
    }
  }
}

// Rows in the first P planes of a z slab that is not the first: their entries in planes BELOW the slab (the head of the CSR
// row, columns < row0) are the transposed entries of rows the previous rank holds; they are read from the CSR arrays as
// they are (3 planes of rows, <= 147 entries each) and added to y after the windows.  One wave per row, fixed tree.
__global__ void __launch_bounds__(256)
    k_symgrid_lowhalo(const int64_t *__restrict__ rowptr, const int32_t *__restrict__ col, const double *__restrict__ val,
                      int64_t nrows_low, int row0, const double *__restrict__ xs, double *__restrict__ y,
                      const double *__restrict__ gate, double gate_tol) {
  if (gate && !(*gate > gate_tol)) return;
  const int lane = threadIdx.x & 63;
  const int64_t nw = (int64_t)gridDim.x * 4;
  for (int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); r < nrows_low; r += nw) {
    const int64_t e0 = rowptr[r], e1 = rowptr[r + 1];
    double s = 0.0;
    for (int64_t e = e0 + lane; e < e1; e += 64) {
      const int c = col[e];
      if (c < row0) s += val[e] * xs[c];
    }
    s = tg_wave_sum(s);
    if (lane == 0) y[r] += s;
  }
}

// ---- check of a finished plan against the CSR product
__global__ void k_symgrid_random(double *x, int64_t n, int64_t first) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    unsigned long long z = (unsigned long long)(i + first) * 0x9E3779B97F4A7C15ull + 0x7F4A7C15ull;
    z ^= z >> 31;
    z *= 0xD6E8FEB86659FD93ull;
    z ^= z >> 29;
    x[i] = (double)(long long)(z >> 11) * (1.0 / 4503599627370496.0) - 1.0;   // (-1, 1)
  }
}
__global__ void k_symgrid_compare(const double *__restrict__ y1, const double *__restrict__ y2, int64_t n,
                                  unsigned long long *out) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  double d = 0.0, m = 0.0;
  bool nan = false;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const double u = y1[i], v = y2[i];
    nan |= !(fabs(u - v) <= 1.7e308);
    d = fmax(d, fabs(u - v));
    m = fmax(m, fabs(u));
  }
  if (nan) d = 1.7e308;
  // (non-negative doubles compare like their bit patterns)
  atomicMax(out, (unsigned long long)__double_as_longlong(d));
  atomicMax(out + 1, (unsigned long long)__double_as_longlong(m));
}

template <int P>
static void sg_launch_convert(const tg_symgrid_s *s, tg_csr_s *a, int *fail) {
  const int64_t nblk = (int64_t)s->npx * s->npy * s->n2 * s->m;
  hipLaunchKernelGGL(k_symgrid_convert<P>, dim3((unsigned)(nblk * sg_cv<P>::PARTS)), dim3(256), 0, g_tg.stream, sg_view(s),
                     a->rowptr, a->col, a->val, s->val, nblk, fail);
}
// part 0: all of it; 1: the chunks that read no halo plane of x (all but the last one of every patch) -- what may run while
// the halo of x is still travelling; 2: the rest (last chunks, the sum of the windows, the rows next to the previous slab)
template <int P>
static void sg_launch_spmv(const tg_symgrid_s *s, tg_csr_s *a, const double *x_shifted, int64_t cmin, int64_t cmax, double *y,
                           int part, const double *gate, double tol) {
  const double *xw = x_shifted + cmin;
  const int xoff = (int)(s->row0 - cmin), xlen = (int)(cmax - cmin + 1);
  const int64_t npatch = (int64_t)s->npx * s->npy;
  auto chunks = [&](int c0, int cn) {
    if (cn <= 0) return;
    const int64_t nw = npatch * cn;
    hipLaunchKernelGGL(k_symgrid_spmv<P>, dim3((unsigned)(tg_cdiv(nw, 8) * 8)), dim3(64), 0, g_tg.stream, sg_view(s), s->val,
                       xw, xoff, xlen, s->stage, nw, c0, cn, gate, tol);
  };
  if (part == 0) chunks(0, s->nch);
  if (part == 1) chunks(0, s->nch - 1);
  if (part == 2) chunks(s->nch - 1, 1);
  if (part == 1) return;
  const int64_t nlines = (int64_t)s->n1 * s->n2;
  hipLaunchKernelGGL(k_symgrid_combine<P>, dim3((unsigned)std::min<int64_t>(nlines, (int64_t)g_tg.num_cu * 64)), dim3(256), 0,
                     g_tg.stream, sg_view(s), s->stage, y, nlines, gate, tol);
  if (s->zoff > 0) {
    const int64_t low = (int64_t)std::min(s->P, s->n2) * s->n0 * s->n1;
    hipLaunchKernelGGL(k_symgrid_lowhalo, dim3((unsigned)std::min<int64_t>(tg_cdiv(low, 4), (int64_t)g_tg.num_cu * 32)),
                       dim3(256), 0, g_tg.stream, a->rowptr, a->col, a->val, low, (int)s->row0, x_shifted, y, gate, tol);
  }
}

// y = K x for the row block the plan was built for; x addressed by GLOBAL column index (x_shifted[col]), readable in
// [cmin, cmax] (the rank's rows and the halo of its z neighbours; one rank: [0, n - 1]).  part: see sg_launch_spmv
static int sg_multi_spmv(tg_symgrid_s *s, tg_csr_s *a, const double *x, double *y, const double *gate, double gate_tol);
int tg_symgrid_spmv(tg_symgrid_s *s, tg_csr_s *a, const double *x_shifted, int64_t cmin, int64_t cmax, double *y, int part,
                    const double *gate, double gate_tol) {
  if (s->nf > 1) {         // (one rank, all columns: part is 0)
    TG_REQUIRE(part == 0 && cmin == 0 && cmax == a->ncols - 1, "the several-field half-storage product runs on one rank");
    return sg_multi_spmv(s, a, x_shifted, y, gate, gate_tol);
  }
  switch (s->P) {
    case 1: sg_launch_spmv<1>(s, a, x_shifted, cmin, cmax, y, part, gate, gate_tol); break;
    case 2: sg_launch_spmv<2>(s, a, x_shifted, cmin, cmax, y, part, gate, gate_tol); break;
    case 4: sg_launch_spmv<4>(s, a, x_shifted, cmin, cmax, y, part, gate, gate_tol); break;
    default: sg_launch_spmv<3>(s, a, x_shifted, cmin, cmax, y, part, gate, gate_tol); break;
  }
  TG_LAUNCH_CHECK();
  return 0;
}
int tg_symgrid_chunks(const tg_symgrid_s *s) { return s->nf > 1 ? 1 : s->nch; }

// the longest row (the first of them): an interior row of a box stencil, if there is one
__global__ void k_symgrid_longest(const int64_t *__restrict__ rowptr, int64_t n, unsigned long long *out) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  unsigned long long best = 0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const unsigned long long len = (unsigned long long)(rowptr[i + 1] - rowptr[i]);
    const unsigned long long key = (len << 32) | (unsigned long long)(0xffffffffu - (unsigned)i);
    best = best > key ? best : key;
  }
  atomicMax(out, best);
}

// P, n0, n1, n2 from the column offsets of one interior row ((2P+1)^3 entries, the diagonal in the middle)
static int sg_detect(tg_csr_s *a, int64_t row0, int *Pout, int *n0o, int *n1o, int *n2o, bool *found) {
  *found = false;
  const int64_t n = a->nrows;
  unsigned long long *o = (unsigned long long *)(g_tg.scratch + 96);
  unsigned long long key = 0;
  TG_CHECK_HIP(hipMemcpyAsync(o, &key, sizeof(key), hipMemcpyHostToDevice, g_tg.stream));
  hipLaunchKernelGGL(k_symgrid_longest, dim3(tg_grid_1d(n, 256)), dim3(256), 0, g_tg.stream, a->rowptr, n, o);
  TG_LAUNCH_CHECK();
  TG_CHECK_HIP(hipMemcpyAsync(&key, o, sizeof(key), hipMemcpyDeviceToHost, g_tg.stream));
  TG_CHECK_HIP(hipStreamSynchronize(g_tg.stream));
  const int64_t len = (int64_t)(key >> 32), rl = (int64_t)(0xffffffffu - (unsigned)(key & 0xffffffffu));
  const int64_t r = rl + row0;          // (global index of that row)
  int P = 0;
  for (int p = 1; p <= 4; p++)
    if (len == (int64_t)(2 * p + 1) * (2 * p + 1) * (2 * p + 1)) P = p;
  if (!P || rl < 0 || rl >= n) return 0;
  int64_t e0 = 0;
  TG_CHECK_HIP(hipMemcpyAsync(&e0, a->rowptr + rl, sizeof(e0), hipMemcpyDeviceToHost, g_tg.stream));
  TG_CHECK_HIP(hipStreamSynchronize(g_tg.stream));
  std::vector<int32_t> c((size_t)len);
  TG_CHECK_HIP(hipMemcpyAsync(c.data(), a->col + e0, sizeof(int32_t) * (size_t)len, hipMemcpyDeviceToHost, g_tg.stream));
  TG_CHECK_HIP(hipStreamSynchronize(g_tg.stream));
  const int S = 2 * P + 1, LC = (S * S * S - 1) / 2;
  if ((int64_t)c[(size_t)LC] != r) return 0;
  const int64_t n0 = (int64_t)c[(size_t)(LC + P + 1)] - r + P;                 // (dx, dy, dz) = (-P, 1, 0)
  const int64_t o1 = (int64_t)c[(size_t)((P + 1) * S * S)] - r;                // (-P, -P, 1)
  if (n0 < 2 * P + 1) return 0;
  const int64_t n01 = o1 + P * n0 + P;
  if (n01 <= 0 || n01 % n0 || n % n01 || row0 % n01 || a->ncols % n01) return 0;
  for (int l = 0; l < S * S * S; l++) {
    const int dx = l % S - P, dy = (l / S) % S - P, dz = l / (S * S) - P;
    if ((int64_t)c[(size_t)l] - r != dx + n0 * dy + n01 * dz) return 0;
  }
  *Pout = P, *n0o = (int)n0, *n1o = (int)(n01 / n0), *n2o = (int)(n / n01);
  *found = true;
  return 0;
}

// ---- several fields on one scalar grid (see k_symgrid_spmv_full)
// nF, P, n0, n1, n2 from the longest row: nF boxes of (2P+1)^3 columns, box g shifted by g * ncp
static int sg_detect_multi(tg_csr_s *a, int *nfo, int *Pout, int *n0o, int *n1o, int *n2o, int *byplane, bool *found) {
  *found = false;
  const int64_t n = a->nrows;
  unsigned long long *o = (unsigned long long *)(g_tg.scratch + 96);
  unsigned long long key = 0;
  TG_CHECK_HIP(hipMemcpyAsync(o, &key, sizeof(key), hipMemcpyHostToDevice, g_tg.stream));
  hipLaunchKernelGGL(k_symgrid_longest, dim3(tg_grid_1d(n, 256)), dim3(256), 0, g_tg.stream, a->rowptr, n, o);
  TG_LAUNCH_CHECK();
  TG_CHECK_HIP(hipMemcpyAsync(&key, o, sizeof(key), hipMemcpyDeviceToHost, g_tg.stream));
  TG_CHECK_HIP(hipStreamSynchronize(g_tg.stream));
  const int64_t len = (int64_t)(key >> 32), rl = (int64_t)(0xffffffffu - (unsigned)(key & 0xffffffffu));
  int P = 0, nf = 0;
  for (int f = 2; f <= 4 && !P; f++)
    for (int p = 1; p <= 3; p++)
      if (len == (int64_t)f * (2 * p + 1) * (2 * p + 1) * (2 * p + 1) && n % f == 0) P = p, nf = f;
  if (!P || rl < 0 || rl >= n) return 0;
  const int64_t ncp = n / nf;
  int64_t e0 = 0;
  TG_CHECK_HIP(hipMemcpyAsync(&e0, a->rowptr + rl, sizeof(e0), hipMemcpyDeviceToHost, g_tg.stream));
  TG_CHECK_HIP(hipStreamSynchronize(g_tg.stream));
  std::vector<int32_t> c((size_t)len);
  TG_CHECK_HIP(hipMemcpyAsync(c.data(), a->col + e0, sizeof(int32_t) * (size_t)len, hipMemcpyDeviceToHost, g_tg.stream));
  TG_CHECK_HIP(hipStreamSynchronize(g_tg.stream));
  const int S = 2 * P + 1, S3 = S * S * S;
  // n0 from the first two lines of the first box (both numberings start a row with the dx run of (dy, dz) = (-P, -P) of
  // field 0, followed by the run of dy = -P + 1), the plane from the numbering that fits all columns
  const int64_t n0 = (int64_t)c[(size_t)S] - (int64_t)c[0];
  if (n0 < 2 * P + 1) return 0;
  for (int bp = 0; bp < 2 && !*found; bp++) {
    // field after field: box g at [g S3, (g+1) S3), plane stride n01; plane by plane: plane dz holds the nf pieces of S*S
    const int64_t second = bp ? (int64_t)c[(size_t)(nf * S * S)] : (int64_t)c[(size_t)(S * S)];      // first column of plane dz = -P + 1
    const int64_t zs = second - (int64_t)c[0];
    const int64_t n01 = bp ? zs / nf : zs;
    if (zs <= 0 || (bp && zs % nf) || n01 <= 0 || n01 % n0 || ncp % n01) continue;
    const int64_t fs = bp ? n01 : ncp;
    // the row itself: (field f, plane z, in-plane ij)
    int64_t f, z, ij;
    if (bp) {
      z = rl / zs, f = (rl % zs) / n01, ij = rl % n01;
    } else {
      f = rl / ncp, z = (rl % ncp) / n01, ij = rl % n01;
    }
    (void)f;
    bool ok = true;
    for (int g = 0; g < nf && ok; g++)
      for (int l = 0; l < S3 && ok; l++) {
        const int dx = l % S - P, dy = (l / S) % S - P, dz = l / (S * S) - P;
        const size_t at = bp ? (size_t)(((dz + P) * nf + g) * S * S + l % (S * S)) : (size_t)(g * S3 + l);
        ok = (int64_t)c[at] == g * fs + (z + dz) * zs + ij + dx + n0 * dy;
      }
    if (!ok) continue;
    *nfo = nf, *Pout = P, *n0o = (int)n0, *n1o = (int)(n01 / n0), *n2o = (int)(ncp / n01), *byplane = bp;
    *found = true;
  }
  return 0;
}

// patches, chunks and their tables for an n0 x n1 x n2 grid cut into px x py patches (what tg_symgrid_build does inline for
// the scalar plan); staging for `overhang` planes of windows beyond a chunk
static int sg_geometry(tg_symgrid_s *s, int px, int py, int overhang, int W) {
  const int n0 = s->n0, n1 = s->n1, n2 = s->n2, P = s->P;
  s->npx = (n0 + px - 1) / px;
  s->npy = (n1 + py - 1) / py;
  const int64_t npatch = (int64_t)s->npx * s->npy;
  int want = getenv("TIGAR_SYMGRID_CHUNKS") ? atoi(getenv("TIGAR_SYMGRID_CHUNKS")) : 0;
  if (want <= 0) want = (int)std::max<int64_t>(n2 / 12, tg_cdiv((int64_t)g_tg.num_cu * 2, npatch));
  s->nch = std::max(1, std::min(want, n2 / std::max(P, 4)));
  std::vector<int32_t> h;
  auto split = [&](int n, int parts) {
    for (int k = 0; k <= parts; k++) h.push_back((int32_t)((int64_t)k * n / parts));
  };
  const size_t ox = 0;
  split(n0, s->npx);
  const size_t oy = h.size();
  split(n1, s->npy);
  const size_t oz = h.size();
  split(n2, s->nch);
  auto owner = [&](size_t o, int parts, int n) {
    int k = 0;
    for (int i = 0; i < n; i++) {
      while (k + 1 < parts && h[o + k + 1] <= i) k++;
      h.push_back(k);
    }
  };
  owner(ox, s->npx, n0);
  owner(oy, s->npy, n1);
  owner(oz, s->nch, n2);
  int cmax = 0;
  for (int x = 0; x < s->npx; x++)
    for (int y = 0; y < s->npy; y++) cmax = std::max(cmax, (h[ox + x + 1] - h[ox + x]) * (h[oy + y + 1] - h[oy + y]));
  s->m = (cmax + 63) / 64;
  s->czmax = 0;
  for (int c = 0; c < s->nch; c++) s->czmax = std::max(s->czmax, h[oz + c + 1] - h[oz + c]);
  s->stage_bytes = npatch * s->nch * (int64_t)(s->czmax + overhang) * W * 8;
  TG_TRY(tg_dmalloc(&s->tabs, (int64_t)h.size()));
  TG_TRY(tg_h2d_staged(s->tabs, h.data(), h.size() * sizeof(int32_t)));
  void *p = nullptr;
  if (tg_dmalloc_bytes(&p, (size_t)s->stage_bytes)) return 100;      // no room: declined
  s->stage = (double *)p;
  return 0;
}

template <int P>
static int sg_multi_convert(tg_symgrid_s *s, tg_csr_s *a, int *ctl) {
  const int nf = s->nf;
  for (int f = 0; f < nf; f++)
    for (int g = f; g < nf; g++) {
      tg_symgrid_s *geo = f == g ? s->gd : s->gf;
      const int NG = f == g ? sg_c<P>::NG : sg_cf<P>::NG;
      const int64_t bytes = (int64_t)geo->npx * geo->npy * geo->n2 * geo->m * NG * 64 * 16;
      void *p = nullptr;
      if (tg_dmalloc_bytes(&p, (size_t)bytes)) return 100;
      (f == g ? s->vd[f] : s->vf[f * 4 + g]) = (sg_d2 *)p;
      TG_CHECK_HIP(hipMemsetAsync(p, 0, (size_t)bytes, g_tg.stream));
      s->val_bytes += s->ncp * (int64_t)NG * 16;       // (what a product reads of the block: the lanes that hold a row)
      const sg_blk B = {nf, f, g, s->ncp, s->fs, s->zs};
      const unsigned grid = (unsigned)std::min<int64_t>(tg_cdiv(s->ncp, 4), (int64_t)g_tg.num_cu * 64);
      if (f == g)
        hipLaunchKernelGGL((k_symgrid_convert_blk<P, false>), dim3(grid), dim3(256), 0, g_tg.stream, sg_view(geo), B, a->rowptr,
                           a->col, a->val, (double *)p, ctl);
      else
        hipLaunchKernelGGL((k_symgrid_convert_blk<P, true>), dim3(grid), dim3(256), 0, g_tg.stream, sg_view(geo), B, a->rowptr,
                           a->col, a->val, (double *)p, ctl);
      TG_LAUNCH_CHECK();
    }
  return 0;
}

template <int P>
static int sg_multi_spmv_p(tg_symgrid_s *s, tg_csr_s *a, const double *x, double *y, const double *gate, double tol) {
  const int nf = s->nf;
  const int64_t fs = s->fs, n = a->ncols;
  // the diagonal blocks: y_f = K_ff x_f (the scalar kernels on the geometry gd; x / y from the field's first entry on)
  for (int f = 0; f < nf; f++) {
    tg_symgrid_s view = *s->gd;
    view.val = s->vd[f];
    view.gd = view.gf = nullptr;
    sg_launch_spmv<P>(&view, a, x + f * fs, 0, n - f * fs - 1, y + f * fs, 0, gate, tol);
    view.tabs = nullptr, view.val = nullptr, view.stage = nullptr;
  }
  // the pairs f < g: y_f += K_fg x_g in the product kernel, y_g += K_fg^T x_f from its windows
  const tg_symgrid_s *G = s->gf;
  const int64_t nw = (int64_t)G->npx * G->npy * G->nch, nlines = (int64_t)G->n1 * G->n2;
  for (int f = 0; f < nf; f++)
    for (int g = f + 1; g < nf; g++) {
      hipLaunchKernelGGL(k_symgrid_spmv_full<P>, dim3((unsigned)(tg_cdiv(nw, 8) * 8)), dim3(64), 0, g_tg.stream, sg_view(G),
                         s->vf[f * 4 + g], x + g * fs, x + f * fs, y + f * fs, (int)(n - g * fs), (int)(n - f * fs), G->stage, nw,
                         gate, tol);
      hipLaunchKernelGGL(k_symgrid_combine_full<P>, dim3((unsigned)std::min<int64_t>(nlines, (int64_t)g_tg.num_cu * 64)),
                         dim3(256), 0, g_tg.stream, sg_view(G), G->stage, y + g * fs, nlines, gate, tol);
    }
  TG_LAUNCH_CHECK();
  return 0;
}
static int sg_multi_spmv(tg_symgrid_s *s, tg_csr_s *a, const double *x, double *y, const double *gate, double gate_tol) {
  switch (s->P) {
    case 1: return sg_multi_spmv_p<1>(s, a, x, y, gate, gate_tol);
    case 2: return sg_multi_spmv_p<2>(s, a, x, y, gate, gate_tol);
    default: return sg_multi_spmv_p<3>(s, a, x, y, gate, gate_tol);
  }
}

static int sg_build_multi(tg_csr_s *a, int verify, tg_symgrid_s **out) {
  const bool trace = getenv("TIGAR_TRACE") != nullptr;
  int nf = 0, P = 0, n0 = 0, n1 = 0, n2 = 0, byplane = 0;
  bool found = false;
  TG_TRY(sg_detect_multi(a, &nf, &P, &n0, &n1, &n2, &byplane, &found));
  if (!found || n0 < 16 || n1 < 16 || n2 < 2 * P + 2) {
    if (trace) fprintf(stderr, "[trace] symgrid: no 3-D box stencil found (%lld rows)\n", (long long)a->nrows);
    return 0;
  }
  tg_symgrid_s *s = new tg_symgrid_s;
  s->P = P, s->n0 = n0, s->n1 = n1, s->n2 = n2, s->n2g = n2, s->nf = nf, s->ncp = (int64_t)n0 * n1 * n2;
  s->fs = byplane ? (int64_t)n0 * n1 : s->ncp;
  s->zs = byplane ? (int64_t)nf * n0 * n1 : (int64_t)n0 * n1;
  s->rows_stored = a->nrows;
  int rc = 0;
  bool declined = false;
  do {
    for (int k = 0; k < 2; k++) {
      tg_symgrid_s *g = new tg_symgrid_s;
      (k == 0 ? s->gd : s->gf) = g;
      g->P = P, g->n0 = n0, g->n1 = n1, g->n2 = n2, g->n2g = n2, g->zs = s->zs;
      const int W = k == 0 ? (P == 1 ? sg_c<1>::W : P == 2 ? sg_c<2>::W : sg_c<3>::W)
                           : (P == 1 ? sg_cf<1>::W : P == 2 ? sg_cf<2>::W : sg_cf<3>::W);
      rc = sg_geometry(g, k == 0 ? SG_PX : SG_PX4, k == 0 ? SG_PY : SG_PY4, k == 0 ? P : 2 * P, W);
      if (rc) break;
      s->stage_bytes += g->stage_bytes;
    }
    if (rc) break;
    int *ctl = (int *)(g_tg.scratch + 64);
    int hflag[2] = {0, 0};
    if (hipMemcpyAsync(ctl, hflag, sizeof(hflag), hipMemcpyHostToDevice, g_tg.stream) != hipSuccess) {
      rc = 1;
      break;
    }
    rc = P == 1 ? sg_multi_convert<1>(s, a, ctl) : P == 2 ? sg_multi_convert<2>(s, a, ctl) : sg_multi_convert<3>(s, a, ctl);
    if (rc) break;
    if (hipMemcpyAsync(hflag, ctl, sizeof(hflag), hipMemcpyDeviceToHost, g_tg.stream) != hipSuccess ||
        hipStreamSynchronize(g_tg.stream) != hipSuccess) {
      tg_set_error("tg_symgrid_build: conversion failed to run");
      rc = 1;
      break;
    }
    if (hflag[0]) {
      if (trace) fprintf(stderr, "[trace] symgrid: a row is not %d box stencils (P=%d, %d x %d x %d): declined\n", nf, P, n0, n1, n2);
      declined = true;
      break;
    }
    if (verify && a->sym_verified != 1) {          // against the CSR product on a pseudo-random vector, as the scalar plan
      const int64_t nx = a->ncols;
      double *t = nullptr;
      if ((rc = tg_dmalloc(&t, nx + 2 * a->nrows))) break;
      double *x = t, *y1 = t + nx, *y2 = y1 + a->nrows;
      unsigned long long *o = (unsigned long long *)(g_tg.scratch + 80);
      unsigned long long ho[2] = {0, 0};
      hipLaunchKernelGGL(k_symgrid_random, dim3(tg_grid_1d(nx, 256)), dim3(256), 0, g_tg.stream, x, nx, (int64_t)0);
      rc = tg_spmv_plan(a);
      if (!rc) rc = tg_spmv_raw(a, x, 0, nx - 1, y1);
      if (!rc) rc = sg_multi_spmv(s, a, x, y2, nullptr, 0.0);
      if (!rc && hipMemcpyAsync(o, ho, sizeof(ho), hipMemcpyHostToDevice, g_tg.stream) != hipSuccess) rc = 1;
      if (!rc) {
        hipLaunchKernelGGL(k_symgrid_compare, dim3(tg_grid_1d(a->nrows, 256)), dim3(256), 0, g_tg.stream, y1, y2, a->nrows, o);
        if (hipMemcpyAsync(ho, o, sizeof(ho), hipMemcpyDeviceToHost, g_tg.stream) != hipSuccess ||
            hipStreamSynchronize(g_tg.stream) != hipSuccess)
          rc = 1;
      }
      tg_dfree(t);
      if (rc) break;
      double d, mx;
      memcpy(&d, &ho[0], 8);
      memcpy(&mx, &ho[1], 8);
      if (trace) fprintf(stderr, "[trace] symgrid (%d fields): check vs CSR product: max |diff| %.3e, max |y| %.3e\n", nf, d, mx);
      if (!(d <= 1e-10 * mx)) {
        if (trace) fprintf(stderr, "[trace] symgrid: the matrix is not symmetric (or the copy is wrong): declined\n");
        a->sym_verified = -1;
        declined = true;
        break;
      }
      a->sym_verified = 1;
    }
  } while (0);
  if (rc == 100) rc = 0, declined = true;
  if (rc || declined) {
    if (g_tg.ready) hipStreamSynchronize(g_tg.stream);
    tg_symgrid_free(s);
    return rc;
  }
  if (trace)
    fprintf(stderr, "[trace] symgrid: %d fields (numbered %s), P=%d grid %d x %d x %d, values %.2f GB (CSR: %.2f GB of values), staging %.2f GB\n",
            nf, byplane ? "plane by plane" : "field after field", P, n0, n1, n2, s->val_bytes / 1e9, 8.0 * a->nnz / 1e9, s->stage_bytes / 1e9);
  *out = s;
  return 0;
}

// Builds the plan for the rows [row0, row0 + nrows) of a square matrix (the whole matrix, or the z slab of planes one
// rank holds: whole planes of the grid); *out stays nullptr when the matrix is not a symmetric box stencil on a 3-D grid
// (or there is no room for the copy).  verify: compare with the CSR product on a pseudo-random vector.
int tg_symgrid_build(tg_csr_s *a, int64_t row0, int verify, tg_symgrid_s **out) {
  *out = nullptr;
  const bool trace = getenv("TIGAR_TRACE") != nullptr;
  if (a->rowcnt || a->view || a->nrows < 1024 || a->ncols >= (int64_t)1 << 28 || row0 < 0 || row0 + a->nrows > a->ncols)
    return 0;
  if (a->sym_verified < 0) return 0;          // (these values failed the comparison before: not symmetric)
  int P = 0, n0 = 0, n1 = 0, n2 = 0;
  bool found = false;
  TG_TRY(sg_detect(a, row0, &P, &n0, &n1, &n2, &found));
  if (!found && row0 == 0 && a->nrows == a->ncols && !(getenv("TIGAR_SPMV_SYM_FIELDS") && atoi(getenv("TIGAR_SPMV_SYM_FIELDS")) == 0))
    return sg_build_multi(a, verify, out);          // several fields on one scalar grid? (*out stays nullptr if not)
  if (!found || n0 < 16 || n1 < 16 || n2 < 2 * P + 2) {
    if (trace) fprintf(stderr, "[trace] symgrid: no 3-D box stencil found (%lld rows)\n", (long long)a->nrows);
    return 0;
  }
  tg_symgrid_s *s = new tg_symgrid_s;
  s->P = P, s->n0 = n0, s->n1 = n1, s->n2 = n2;
  s->row0 = row0, s->zoff = (int)(row0 / ((int64_t)n0 * n1)), s->n2g = (int)(a->ncols / ((int64_t)n0 * n1));
  s->npx = (n0 + sg_px(P) - 1) / sg_px(P);
  s->npy = (n1 + sg_py(P) - 1) / sg_py(P);
  const int64_t npatch = (int64_t)s->npx * s->npy;
  {
    int want = getenv("TIGAR_SYMGRID_CHUNKS") ? atoi(getenv("TIGAR_SYMGRID_CHUNKS")) : 0;
    // (measured, 64^3 .. 256^3, p = 2, 3: chunks of 11-13 planes are best -- a chunk pays for P extra window planes and
    // the start of its load pipeline -- as long as there are about two waves per CU)
    const bool chosen = want > 0;
    if (want <= 0) want = (int)std::max<int64_t>(n2 / 12, tg_cdiv((int64_t)g_tg.num_cu * 2, npatch));
    const int minplanes = std::max(P, 4);
    s->nch = std::max(1, std::min(want, n2 / minplanes));
    // Few planes (the z slab of one of several ranks, a short patch): the waves of a product are one or two rounds of what
    // the chip holds, and the number of chunks decides how full the last round is -- 67 planes of cfg3's grid: 5 chunks
    // (935 waves on 768 places) 1.38 ms, 4 chunks (748 waves) 1.15 ms.  Cost of a choice: rounds x (planes per chunk + the P + 1
    // planes a chunk pays for its start); with three rounds or more the rounds overlap and the 11-13 planes above stand.
    const int64_t lds_wave = (int64_t)2 * (P + 1) * (sg_px(P) + 2 * P) * (sg_py(P) + 2 * P) * 8 + 1024;
    const int64_t places = (int64_t)g_tg.num_cu * std::max<int64_t>(1, std::min<int64_t>(8, (160 * 1024) / lds_wave));
    if (!chosen && npatch * s->nch < 3 * places) {
      double best = 1e300;
      int pick = s->nch;
      for (int c = 1; c <= std::max(1, n2 / minplanes); c++) {
        const double cost = (double)tg_cdiv(npatch * c, places) * ((double)n2 / c + P + 1);
        if (cost < best - 1e-9) best = cost, pick = c;
      }
      s->nch = pick;
    }
  }
  std::vector<int32_t> h;
  auto split = [&](int n, int parts) {
    for (int k = 0; k <= parts; k++) h.push_back((int32_t)((int64_t)k * n / parts));
  };
  const size_t ox = 0;
  split(n0, s->npx);
  const size_t oy = h.size();
  split(n1, s->npy);
  const size_t oz = h.size();
  split(n2, s->nch);
  auto owner = [&](size_t o, int parts, int n) {
    int k = 0;
    for (int i = 0; i < n; i++) {
      while (k + 1 < parts && h[o + k + 1] <= i) k++;
      h.push_back(k);
    }
  };
  owner(ox, s->npx, n0);
  owner(oy, s->npy, n1);
  owner(oz, s->nch, n2);
  int cmax = 0;
  for (int x = 0; x < s->npx; x++)
    for (int y = 0; y < s->npy; y++) cmax = std::max(cmax, (h[ox + x + 1] - h[ox + x]) * (h[oy + y + 1] - h[oy + y]));
  s->m = (cmax + 63) / 64;
  for (int c = 0; c < s->nch; c++) s->czmax = std::max(s->czmax, h[oz + c + 1] - h[oz + c]);
  const int NG = P == 1 ? sg_c<1>::NG : P == 2 ? sg_c<2>::NG : P == 4 ? sg_c<4>::NG : sg_c<3>::NG;
  const int W = P == 1 ? sg_c<1>::W : P == 2 ? sg_c<2>::W : P == 4 ? sg_c<4>::W : sg_c<3>::W;
  s->val_bytes = npatch * n2 * s->m * (int64_t)NG * 64 * 16;
  s->stage_bytes = npatch * s->nch * (int64_t)(s->czmax + P) * W * 8;
  s->rows_stored = a->nrows;
  int rc = 0;
  bool declined = false;
  do {
    if ((rc = tg_dmalloc(&s->tabs, (int64_t)h.size()))) break;
    if ((rc = tg_h2d_staged(s->tabs, h.data(), h.size() * sizeof(int32_t)))) break;
    void *p = nullptr;
    if (tg_dmalloc_bytes(&p, (size_t)s->val_bytes)) {
      declined = true;      // no room: not an error
      break;
    }
    s->val = (sg_d2 *)p;
    if (tg_dmalloc_bytes(&p, (size_t)s->stage_bytes)) {
      declined = true;
      break;
    }
    s->stage = (double *)p;
    int *ctl = (int *)(g_tg.scratch + 64);
    int hflag[2] = {0, 0};
    if (hipMemcpyAsync(ctl, hflag, sizeof(hflag), hipMemcpyHostToDevice, g_tg.stream) != hipSuccess) {
      rc = 1;
      break;
    }
    switch (P) {
      case 1: sg_launch_convert<1>(s, a, ctl); break;
      case 2: sg_launch_convert<2>(s, a, ctl); break;
      case 4: sg_launch_convert<4>(s, a, ctl); break;
      default: sg_launch_convert<3>(s, a, ctl); break;
    }
    if (hipGetLastError() != hipSuccess ||
        hipMemcpyAsync(hflag, ctl, sizeof(hflag), hipMemcpyDeviceToHost, g_tg.stream) != hipSuccess ||
        hipStreamSynchronize(g_tg.stream) != hipSuccess) {
      tg_set_error("tg_symgrid_build: conversion failed to run");
      rc = 1;
      break;
    }
    if (hflag[0]) {
      if (trace) fprintf(stderr, "[trace] symgrid: a row is not the box stencil (P=%d, %d x %d x %d): declined\n", P, n0, n1, n2);
      declined = true;
      break;
    }
    if (verify && a->sym_verified != 1) {
      // x on the window of columns the block reads: P planes on either side (as far as the grid goes)
      const int64_t n01 = (int64_t)n0 * n1;
      const int64_t cmin = std::max<int64_t>(0, row0 - P * n01), cmax = std::min<int64_t>(a->ncols, row0 + a->nrows + P * n01) - 1;
      const int64_t nx = cmax - cmin + 1;
      double *t = nullptr;
      if ((rc = tg_dmalloc(&t, nx + 2 * a->nrows))) break;
      double *x = t, *y1 = t + nx, *y2 = y1 + a->nrows;
      unsigned long long *o = (unsigned long long *)(g_tg.scratch + 80);
      unsigned long long ho[2] = {0, 0};
      const int vg = tg_grid_1d(a->nrows, 256);
      hipLaunchKernelGGL(k_symgrid_random, dim3(tg_grid_1d(nx, 256)), dim3(256), 0, g_tg.stream, x, nx, cmin);
      rc = tg_spmv_plan(a);
      if (!rc) rc = tg_spmv_raw(a, x - cmin, cmin, cmax, y1);
      if (!rc) rc = tg_symgrid_spmv(s, a, x - cmin, cmin, cmax, y2, 0, nullptr, 0.0);
      if (!rc && hipMemcpyAsync(o, ho, sizeof(ho), hipMemcpyHostToDevice, g_tg.stream) != hipSuccess) rc = 1;
      if (!rc) {
        hipLaunchKernelGGL(k_symgrid_compare, dim3(vg), dim3(256), 0, g_tg.stream, y1, y2, a->nrows, o);
        if (hipMemcpyAsync(ho, o, sizeof(ho), hipMemcpyDeviceToHost, g_tg.stream) != hipSuccess ||
            hipStreamSynchronize(g_tg.stream) != hipSuccess)
          rc = 1;
      }
      tg_dfree(t);
      if (rc) break;
      double d, mx;
      memcpy(&d, &ho[0], 8);
      memcpy(&mx, &ho[1], 8);
      if (trace) fprintf(stderr, "[trace] symgrid: check vs CSR product: max |diff| %.3e, max |y| %.3e\n", d, mx);
      if (!(d <= 1e-10 * mx)) {
        if (trace) fprintf(stderr, "[trace] symgrid: the matrix is not symmetric (or the copy is wrong): declined\n");
        a->sym_verified = -1;
        declined = true;
        break;
      }
      a->sym_verified = 1;
    }
  } while (0);
  if (rc || declined) {
    if (g_tg.ready) hipStreamSynchronize(g_tg.stream);
    tg_symgrid_free(s);
    return rc;
  }
  if (trace)
    fprintf(stderr, "[trace] symgrid: P=%d grid %d x %d x %d (planes %d..%d of %d), %d x %d patches, %d chunks, %d sub-steps, values %.2f GB, staging %.2f GB\n",
            P, n0, n1, n2, s->zoff, s->zoff + n2, s->n2g, s->npx, s->npy, s->nch, s->m, s->val_bytes / 1e9, s->stage_bytes / 1e9);
  *out = s;
  return 0;
}

void tg_symgrid_info(const tg_symgrid_s *s, int64_t *val_bytes, int64_t *stage_bytes) {
  if (s->nf > 1) {
    if (val_bytes) *val_bytes = s->val_bytes;
    if (stage_bytes) *stage_bytes = s->stage_bytes;
    return;
  }
  const int NG = s->P == 1 ? sg_c<1>::NG : s->P == 2 ? sg_c<2>::NG : s->P == 4 ? sg_c<4>::NG : sg_c<3>::NG;
  if (val_bytes) *val_bytes = s->rows_stored * NG * 16;
  if (stage_bytes) *stage_bytes = s->stage_bytes;
}

/* Plans the half-storage product for a matrix -- or for the block of rows [row0, row0 + nrows) of one, whole planes of
 * the grid -- and runs y = K x with it (tests, bench accounting); x covers ALL columns.  *accepted = 0 when the matrix is
 * not a symmetric 3-D box stencil.  value_bytes: what one product reads of K; staging_bytes: window staging. */
extern "C" int tg_spmv_symgrid(tg_csr_t a, int64_t row0, tg_vec_t x, tg_vec_t y, int *accepted, int64_t *value_bytes,
                               int64_t *staging_bytes) {
  TG_REQUIRE_INIT();
  TG_REQUIRE(a && accepted, "null argument to tg_spmv_symgrid");
  TG_REQUIRE_CANONICAL(a);
  *accepted = 0;
  tg_symgrid_s *s = nullptr;
  TG_TRY(tg_symgrid_build(a, row0, 1, &s));
  if (!s) return 0;
  int rc = 0;
  if (x && y) {
    if (x->n != a->ncols || y->n != a->nrows) {
      tg_set_error("tg_spmv_symgrid: vector sizes %lld / %lld for a %lld x %lld matrix", (long long)x->n, (long long)y->n,
                   (long long)a->nrows, (long long)a->ncols);
      rc = 2;
    } else {
      rc = tg_symgrid_spmv(s, a, x->d, 0, a->ncols - 1, y->d, 0, nullptr, 0.0);
    }
  }
  *accepted = 1;
  tg_symgrid_info(s, value_bytes, staging_bytes);
  if (g_tg.ready) hipStreamSynchronize(g_tg.stream);
  tg_symgrid_free(s);
  return rc;
}
