// Host side and kernel wrappers of the tensor-pattern extractMatrix path (see tg_tensor_body.h for the
// algorithm).  K = M^T A M for a tensor-product B-spline patch in three line-walk passes:
//
//   tg_tensor_planes   FE planes [z0,z1) of A (any CSR matrix; its pattern is VERIFIED against the closed form
//                      while it is read -- status 100 = "not this pattern, use the general kernels")
//                      -> x pass -> y pass -> dense B2 planes (plane-local, cached by the caller across sub-slabs)
//   tg_tensor_zstage   B2 planes of one or several pieces -> rows of K for dof planes [ka,kb), written in CSR
//                      order at closed-form positions into a new matrix or straight into the slab-wise builder,
//                      MatZeroRowsColumns(zeroDofs, diag) fused (tIGAr/common.py:1199-1200).
#include "tg_common.h"
#include "tg_tensor_body.h"
#include <algorithm>
#include <vector>

struct tg_tensor_plan_s {
  int P = 0;
  int d = 3;               // 2: patch with two parametric directions and nF fields (tg_tensor2_ptap)
  int nF = 1;
  tt_dir_t dir[3];
  // device tables (owned)
  double *wl[3] = {nullptr, nullptr, nullptr};
  int32_t *rps[3] = {nullptr, nullptr, nullptr}, *kps[3] = {nullptr, nullptr, nullptr};
  // lines of direction 1 by block extent: short (P+1) and vertex (2P+1)
  int32_t *lines1[2] = {nullptr, nullptr};
  int nlines1[2] = {0, 0};
  std::vector<int32_t> h_rps[3], h_kps[3];
  // 1-D value tables of the Kronecker-sum form used last by tg_tensor_planes_kron (device, term-major), and their key
  double *kcv[3] = {nullptr, nullptr, nullptr};
  uint64_t kcv_key = 0;
  int kcv_terms = 0;
  // blocks with different spline bases on the row and the column side (tg_tensor_plan_create_pair): true function counts
  // and degrees per direction; pair == false: square (ncr = ncc = nel + P, pr = pc = P)
  bool pair = false;
  int ncr[3] = {0, 0, 0}, ncc[3] = {0, 0, 0}, pr[3] = {0, 0, 0}, pc[3] = {0, 0, 0};
  double *wlr[3] = {nullptr, nullptr, nullptr};
  uint64_t expect_tag = 0; // tg_pattern_hash of the element-coupling pattern the passes rely on
  std::vector<int32_t> h_ecol[3];   // that pattern's 1-D column indices (rows as in h_rps)
  int *status = nullptr;   // device flag
};

struct tg_tensor_planes_s {
  int z0 = 0, z1 = 0;
  int *status = nullptr;            // device flag of the passes that produced these planes (read by the z stage)
  double *buf = nullptr;            // B2 planes
  std::vector<int64_t> pb;          // offset of plane r2 in buf, indexed r2 - z0
};

template <int P>
__global__ void __launch_bounds__(256) k_tt_check_rows(tt_check_args A, int64_t n, int *status) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n && tt_check_row<P>(A, i)) atomicOr(status, 1);
}
// All (plane class, line class) combinations of a pass in ONE launch: every wave walks a whole line, so a launch ends with
// a partly filled last round of waves (26.9 k waves on 4096 slots: 6.6 rounds); four launches pay that four times
// (x+y over 36 planes at cfg3, both variants alternating inside one process: 12.36 vs 12.49 ms).
// Workgroups [first[c], first[c+1]) belong to class c, x fastest; the classes with the widest windows come first.
struct tt_x_multi {
  tt_x_args c[4];
  unsigned first[5], gx[4];
  int n;
};
struct tt_y_multi {
  tt_y_args c[2];
  unsigned first[3], gx[2];
  int n;
};
// piece length of the y walk (TIGAR_TT_Y_ECH=n for experiments; default 0 = one walk: the pass is HBM-bound, two / three /
// four pieces at cfg3 gave 3.07 / 3.12 / 3.05 ms against 3.07)
static int tt_y_ech(int nel, int P) {
  static const int env = getenv("TIGAR_TT_Y_ECH") ? atoi(getenv("TIGAR_TT_Y_ECH")) : 0;
  const int pick = env;
  return pick > 0 && pick < nel ? std::max(pick, 2 * P) : 0;
}
static unsigned tt_pieces(int nel, int ech) { return ech > 0 ? (unsigned)tg_cdiv(nel, ech) : 1u; }
template <int P, bool V>
__global__ void __launch_bounds__(64) k_tt_x_multi(tt_x_multi M) {
  // (the piece of the walk, if it is cut into pieces, is the slowest-varying index: tt_xg_args::ech)
  const unsigned piece = blockIdx.x / M.first[M.n], b = blockIdx.x - piece * M.first[M.n];
  int c = 0;
  while (c + 1 < M.n && b >= M.first[c + 1]) c++;
  const tt_x_args &A = M.c[c];
  if (*(volatile int *)A.status) return;     // row lengths differ from the pattern: the closed-form addresses do not apply
  const unsigned local = b - M.first[c];
  const int bad = tt_x_lane<P, V>(A, (int)(local % M.gx[c]), (int)(local / M.gx[c]), threadIdx.x, (int)piece);
  if (bad) atomicOr(A.status, 1);
}
template <int P>
__global__ void __launch_bounds__(64) k_tt_y_multi(tt_y_multi M) {
  const unsigned piece = blockIdx.x / M.first[M.n], b = blockIdx.x - piece * M.first[M.n];
  int c = 0;
  while (c + 1 < M.n && b >= M.first[c + 1]) c++;
  const unsigned local = b - M.first[c];
  tt_y_lane<P>(M.c[c], (int)(local % M.gx[c]), (int)(local / M.gx[c]), threadIdx.x, (int)piece);
}
template <int P>
__global__ void __launch_bounds__(64) k_tt_z(tt_z_args A) {
  tt_z_lane<P>(A, blockIdx.x, threadIdx.x);
}
__global__ void __launch_bounds__(256) k_tt_rowptr(tt_rowptr_args A, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) tt_rowptr_one(A, i);
}

template <typename T>
static int tt_upload(T **dst, const std::vector<T> &h) {
  TG_TRY(tg_dmalloc(dst, (int64_t)std::max<size_t>(h.size(), 1)));
  if (!h.empty()) TG_TRY(tg_h2d_staged(*dst, h.data(), h.size() * sizeof(T)));   // (h may be a temporary)
  return 0;
}

static int tt_rn_host(int P, int a, int nfe) { return (a % P == 0 && a > 0 && a < nfe - 1) ? 2 * P + 1 : P + 1; }
static int tt_rlo_host(int P, int a, int nfe) {
  (void)nfe;
  return (a % P == 0 && a > 0) ? a - P : (a / P) * P;
}

extern "C" int tg_tensor_plan_destroy(tg_tensor_plan_t p) {
  if (!p) return 0;
  if (g_tg.ready) hipStreamSynchronize(g_tg.stream);
  for (int k = 0; k < 3; k++) {
    tg_dfree(p->wl[k]);
    tg_dfree(p->wlr[k]);
    tg_dfree(p->rps[k]);
    tg_dfree(p->kps[k]);
  }
  tg_dfree(p->lines1[0]);
  tg_dfree(p->lines1[1]);
  tg_dfree(p->status);
  for (int k = 0; k < 3; k++) tg_dfree(p->kcv[k]);
  delete p;
  return 0;
}

static int tt_plan_create3(const tg_tensor_pair_dir_t *dirs, bool pair, tg_tensor_plan_t *out) {
  const int P = dirs[0].p;
  TG_REQUIRE(P >= 1 && P <= 3 && dirs[1].p == P && dirs[2].p == P,
             "tg_tensor_plan_create: equal degrees 1..3 in all directions ((2p+1)^2 lanes must fit a wave)");
  tg_tensor_plan_s *pl = new tg_tensor_plan_s();
  pl->P = P;
  pl->pair = pair;
  int rc = 0;
  for (int k = 0; k < 3 && !rc; k++) {
    const int nel = dirs[k].nel, nfe = P * nel + 1, ncp = nel + P;
    const int pr = pair ? dirs[k].pr : P, pc = pair ? dirs[k].pc : P;
    if (nel < 1 || !dirs[k].wlc || (pair && !dirs[k].wlr) || pr < 1 || pr > P || pc < 1 || pc > P) {
      tg_set_error("tg_tensor_plan_create: bad direction %d (spline degrees 1..%d on both sides)", k, P);
      rc = 2;
      break;
    }
    const int ncr = nel + pr, ncc = nel + pc;
    pl->pr[k] = pr;
    pl->pc[k] = pc;
    pl->ncr[k] = ncr;
    pl->ncc[k] = ncc;
    std::vector<double> w(dirs[k].wlc, dirs[k].wlc + (size_t)nel * (P + 1) * (P + 1));
    pl->h_rps[k].assign(nfe + 1, 0);
    for (int a = 0; a < nfe; a++) pl->h_rps[k][a + 1] = pl->h_rps[k][a] + tt_rn_host(P, a, nfe);
    // 1-D pattern of the product: row function i couples to the column functions [i - pr, i + pc], clipped
    pl->h_kps[k].assign(ncr + 1, 0);
    for (int i = 0; i < ncr; i++)
      pl->h_kps[k][i + 1] = pl->h_kps[k][i] + (std::min(ncc - 1, i + pc) - std::max(0, i - pr) + 1);
    rc = tt_upload(&pl->wl[k], w);
    if (!rc && pair) {
      std::vector<double> wr(dirs[k].wlr, dirs[k].wlr + (size_t)nel * (P + 1) * (P + 1));
      rc = tt_upload(&pl->wlr[k], wr);
    }
    if (!rc) rc = tt_upload(&pl->rps[k], pl->h_rps[k]);
    if (!rc) rc = tt_upload(&pl->kps[k], pl->h_kps[k]);
    pl->dir[k].nel = nel;
    pl->dir[k].nfe = nfe;
    pl->dir[k].ncp = ncp;                 // (padded: the layout of the intermediates)
    pl->dir[k].wl = pl->wl[k];
    pl->dir[k].wlr = pair ? pl->wlr[k] : nullptr;
    pl->dir[k].rps = pl->rps[k];
    pl->dir[k].kps = pl->kps[k];
  }
  if (!rc) {
    // the pattern the FE matrix must have, as 1-D CSR patterns: row a couples to the columns [lo(a), lo(a) + n(a))
    std::vector<int32_t> ecol[3];
    const int32_t *rps[3], *cls[3];
    int64_t nr[3], nc[3];
    for (int k = 0; k < 3; k++) {
      const int nfe = pl->dir[k].nfe;
      for (int a = 0; a < nfe; a++) {
        const int lo = tt_rlo_host(P, a, nfe), n = tt_rn_host(P, a, nfe);
        for (int j = 0; j < n; j++) ecol[k].push_back(lo + j);
      }
      rps[k] = pl->h_rps[k].data();
      cls[k] = ecol[k].data();
      nr[k] = nc[k] = nfe;
    }
    pl->expect_tag = tg_pattern_hash(3, nr, nc, rps, cls, 0);
    for (int k = 0; k < 3; k++) pl->h_ecol[k] = ecol[k];
  }
  if (!rc) {
    std::vector<int32_t> ls, lv;
    const int nfe1 = pl->dir[1].nfe;
    for (int a = 0; a < nfe1; a++) (tt_rn_host(P, a, nfe1) == P + 1 ? ls : lv).push_back(a);
    pl->nlines1[0] = (int)ls.size();
    pl->nlines1[1] = (int)lv.size();
    rc = tt_upload(&pl->lines1[0], ls);
    if (!rc) rc = tt_upload(&pl->lines1[1], lv);
    if (!rc) rc = tg_dmalloc(&pl->status, 4);
  }
  if (rc) {
    tg_tensor_plan_destroy(pl);
    return rc;
  }
  *out = pl;
  return 0;
}

extern "C" int tg_tensor_plan_create(int d, const tg_tensor_dir_t *dirs, tg_tensor_plan_t *out) {
  TG_REQUIRE_INIT();
  TG_REQUIRE(d == 3 && dirs && out, "tg_tensor_plan_create: three parametric directions expected");
  tg_tensor_pair_dir_t pd[3];
  for (int k = 0; k < 3; k++) {
    pd[k].p = dirs[k].p;
    pd[k].nel = dirs[k].nel;
    pd[k].pr = pd[k].pc = dirs[k].p;
    pd[k].wlr = nullptr;
    pd[k].wlc = dirs[k].wl;
  }
  return tt_plan_create3(pd, false, out);
}

extern "C" int tg_tensor_plan_create_pair(int d, const tg_tensor_pair_dir_t *dirs, tg_tensor_plan_t *out) {
  TG_REQUIRE_INIT();
  TG_REQUIRE(d == 3 && dirs && out, "tg_tensor_plan_create_pair: three parametric directions expected");
  return tt_plan_create3(dirs, true, out);
}

extern "C" int tg_tensor_planes_destroy(tg_tensor_planes_t p) {
  if (!p) return 0;
  if (g_tg.ready && !g_tg.multi) hipStreamSynchronize(g_tg.stream);
  tg_dfree(p->buf);
  tg_dfree(p->status);
  delete p;
  return 0;
}

#define TT_DISPATCH_P(P, CALL) \
  do {                         \
    if ((P) == 1) {            \
      CALL(1);                 \
    } else if ((P) == 2) {     \
      CALL(2);                 \
    } else {                   \
      CALL(3);                 \
    }                          \
  } while (0)

extern "C" int tg_tensor_planes(tg_tensor_plan_t pl, tg_csr_t a, int64_t a_row0, int z0, int z1,
                                tg_tensor_planes_t *out) {
  TG_REQUIRE_INIT();
  TG_REQUIRE(pl && a && out, "null argument to tg_tensor_planes");
  TG_REQUIRE(pl->d == 3, "tg_tensor_planes: the plan belongs to a 2-D patch (tg_tensor2_ptap)");
  TG_REQUIRE_CANONICAL(a);
  const int P = pl->P, W = 2 * P + 1;
  const tt_dir_t &D0 = pl->dir[0], &D1 = pl->dir[1], &D2 = pl->dir[2];
  const int64_t plane_fe = (int64_t)D0.nfe * D1.nfe;
  TG_REQUIRE(z0 >= 0 && z1 > z0 && z1 <= D2.nfe, "tg_tensor_planes: plane range out of bounds");
  TG_REQUIRE(a_row0 % plane_fe == 0 && a_row0 <= z0 * plane_fe && a_row0 + a->nrows >= z1 * plane_fe,
             "tg_tensor_planes: the FE rows given do not cover whole planes [%d,%d)", z0, z1);
  if (a->ncols != plane_fe * D2.nfe) return 100;     // not a matrix on this node grid
  const int aplane0 = (int)(a_row0 / plane_fe);
  const int np = z1 - z0;
  // plane bases of the two intermediates (B1: [c2][m0][c1] blocks, B2: [m1][m0][c2] blocks)
  std::vector<int64_t> pb1(np + 1, 0), pb2(np + 1, 0);
  std::vector<int32_t> pls[2];
  for (int q = 0; q < np; q++) {
    const int n2 = tt_rn_host(P, z0 + q, D2.nfe);
    pb1[q + 1] = pb1[q] + (int64_t)W * n2 * D0.ncp * pl->h_rps[1][D1.nfe];
    pb2[q + 1] = pb2[q] + (int64_t)W * W * n2 * D0.ncp * D1.ncp;
    pls[n2 == P + 1 ? 0 : 1].push_back(z0 + q);
  }
  double *b1 = nullptr;
  int64_t *d_pb1 = nullptr, *d_pb2 = nullptr;
  int32_t *d_pl[2] = {nullptr, nullptr};
  tg_tensor_planes_s *res = new tg_tensor_planes_s();
  res->z0 = z0;
  res->z1 = z1;
  res->pb = pb2;
  int rc = tg_dmalloc(&b1, pb1[np]);
  if (!rc) rc = tg_dmalloc(&res->buf, pb2[np]);
  if (!rc) rc = tg_dmalloc(&res->status, 4);
  if (!rc) rc = tt_upload(&d_pb1, pb1);
  if (!rc) rc = tt_upload(&d_pb2, pb2);
  if (!rc) rc = tt_upload(&d_pl[0], pls[0]);
  if (!rc) rc = tt_upload(&d_pl[1], pls[1]);
  // (one flag per set of planes, not per plan: with a certified matrix nobody waits here, and the flag of an earlier
  //  piece of the ring must still be there when the z stage reads it)
  if (!rc && hipMemsetAsync(res->status, 0, sizeof(int), g_tg.stream) != hipSuccess) rc = 1;
  int bad = 0;
  bool certified_pass = false;
  if (!rc) {
    // row lengths of the planes against the pattern (8 B per row; the x pass then needs no row pointers)
    {
      tt_check_args Cq;
      Cq.rowptr = a->rowptr;
      Cq.nfe0 = D0.nfe;
      Cq.nfe1 = D1.nfe;
      Cq.nfe2 = D2.nfe;
      Cq.aplane0 = aplane0;
      Cq.z0 = z0;
      Cq.dense2 = 0;
      const int64_t nr = (int64_t)np * plane_fe;
#define TT_C(PP) hipLaunchKernelGGL((k_tt_check_rows<PP>), dim3((unsigned)tg_cdiv(nr, 256)), dim3(256), 0, g_tg.stream, Cq, nr, res->status)
      TT_DISPATCH_P(P, TT_C);
#undef TT_C
    }
    // x pass: the (plane class, line class) combinations in one launch, widest windows first
    // The walk in TWO pieces (each lane walks half of the direction; the second piece re-reads the rows of A of p elements:
    // +1-2 % of A): 6.71 -> 6.27 ms per sub-slab at cfg3 with the FE matrix materialised, 3.76 -> 3.50 ms at cfg2; three and
    // more pieces (96, 64, 48 elements at cfg3) gave nothing (6.76-6.86 ms) -- unlike the pass that forms A's entries itself
    // this one is not waiting for its scalar tables.  TIGAR_TT_X_ECH=n: pieces of n elements, 0: one walk.
    static const int x_ech_env = getenv("TIGAR_TT_X_ECH") ? atoi(getenv("TIGAR_TT_X_ECH")) : -1;
    const int x_ech_pick = x_ech_env >= 0 ? x_ech_env : (D0.nel >= 64 ? (D0.nel + 1) / 2 : 0);
    const int x_ech = x_ech_pick > 0 && x_ech_pick < D0.nel ? std::max(x_ech_pick, 2 * P) : 0;
    const unsigned x_pieces = x_ech > 0 ? (unsigned)tg_cdiv(D0.nel, x_ech) : 1u;
    {
      tt_x_multi XM;
      memset(&XM, 0, sizeof(XM));
      for (int pc = 1; pc >= 0; pc--) {
        if (pls[pc].empty()) continue;
        const int n2 = pc == 0 ? P + 1 : W;
        for (int lc = 1; lc >= 0; lc--) {
          if (!pl->nlines1[lc]) continue;
          const int n1 = lc == 0 ? P + 1 : W;
          tt_x_args &X = XM.c[XM.n];
          X.rowptr = a->rowptr;
          X.col = a->col;
          X.val = a->val;
          X.rps2 = D2.rps;
          X.aplane0 = aplane0;
          X.d0 = D0;
          X.nfe1 = D1.nfe;
          X.nfe2 = D2.nfe;
          X.rps1 = D1.rps;
          X.lines = pl->lines1[lc];
          X.nlines = pl->nlines1[lc];
          X.n1 = n1;
          X.L = std::max(1, 64 / (n1 * n2));
          X.planes = d_pl[pc];
          X.n2 = n2;
          X.b1 = b1;
          X.pb1 = d_pb1;
          X.z0 = z0;
          X.status = res->status;
          X.ech = x_ech;
          XM.gx[XM.n] = (unsigned)tg_cdiv(X.nlines, X.L);
          XM.first[XM.n + 1] = XM.first[XM.n] + XM.gx[XM.n] * (unsigned)pls[pc].size();
          XM.n++;
        }
      }
      // A matrix written by this library with exactly this pattern says so (tg_csr_s::pattern_tag): its column indices
      // need not be read again (227 GB per pass at cfg3).  Any other matrix is verified entry by entry while it is read;
      // TIGAR_PTAP_VERIFY=1 verifies always.
      const bool certified = certified_pass = a->pattern_tag != 0 && a->pattern_tag == pl->expect_tag && a->pattern_row0 == a_row0 &&
                             !(getenv("TIGAR_PTAP_VERIFY") && atoi(getenv("TIGAR_PTAP_VERIFY")));
      if (XM.n > 0 && XM.first[XM.n] > 0) {
#define TT_X(PP) hipLaunchKernelGGL((k_tt_x_multi<PP, true>), dim3(XM.first[XM.n] * x_pieces), dim3(64), 0, g_tg.stream, XM)
#define TT_XC(PP) hipLaunchKernelGGL((k_tt_x_multi<PP, false>), dim3(XM.first[XM.n] * x_pieces), dim3(64), 0, g_tg.stream, XM)
        if (certified) TT_DISPATCH_P(P, TT_XC);
        else TT_DISPATCH_P(P, TT_X);
#undef TT_X
#undef TT_XC
        g_tg.prof_n[TG_PROF_PTAP_CERTIFIED] += certified ? 1 : 0;
      }
    }
    // y pass: both plane classes in one launch
    {
      tt_y_multi YM;
      memset(&YM, 0, sizeof(YM));
      for (int pc = 1; pc >= 0; pc--) {
        if (pls[pc].empty()) continue;
        const int n2 = pc == 0 ? P + 1 : W;
        tt_y_args &Y = YM.c[YM.n];
        Y.b1 = b1;
        Y.pb1 = d_pb1;
        Y.b2 = res->buf;
        Y.pb2 = d_pb2;
        Y.z0 = z0;
        Y.d1 = D1;
        Y.ncp0 = D0.ncp;
        Y.planes = d_pl[pc];
        Y.n2 = n2;
        Y.L = std::max(1, 64 / (W * n2));
        Y.ech = tt_y_ech(D1.nel, P);
        YM.gx[YM.n] = (unsigned)tg_cdiv(D0.ncp, Y.L);
        YM.first[YM.n + 1] = YM.first[YM.n] + YM.gx[YM.n] * (unsigned)pls[pc].size();
        YM.n++;
      }
      if (YM.n > 0 && YM.first[YM.n] > 0) {
#define TT_Y(PP) \
  hipLaunchKernelGGL((k_tt_y_multi<PP>), dim3(YM.first[YM.n] * tt_pieces(D1.nel, tt_y_ech(D1.nel, P))), dim3(64), 0, g_tg.stream, YM)
        TT_DISPATCH_P(P, TT_Y);
#undef TT_Y
      }
    }
    if (hipGetLastError() != hipSuccess) {
      tg_set_error("tg_tensor_planes: kernel launch failed");
      rc = 1;
    }
    // a matrix whose pattern was verified entry by entry may have failed: the caller is told now (status 100).  A
    // certified matrix cannot, and the host goes on enqueueing (the z stage reads the flag when it waits anyway).
    if (!certified_pass) {
      if (!rc && hipMemcpyAsync(&bad, res->status, sizeof(int), hipMemcpyDeviceToHost, g_tg.stream) != hipSuccess) rc = 1;
      if (!rc && hipStreamSynchronize(g_tg.stream) != hipSuccess) {
        tg_set_error("tg_tensor_planes: %s", hipGetErrorString(hipGetLastError()));
        rc = 1;
      }
    }
  }
  tg_dfree(b1);
  tg_dfree(d_pb1);
  tg_dfree(d_pb2);
  tg_dfree(d_pl[0]);
  tg_dfree(d_pl[1]);
  if (rc || bad) {
    tg_tensor_planes_destroy(res);
    return rc ? rc : 100;          // 100: A does not have the element-coupling pattern -> general path
  }
  *out = res;
  return 0;
}

// ------------------------------------------------------------------------------------------------------
// x and y passes for an FE matrix given as a Kronecker sum of 1-D matrices on the element-coupling pattern: the matrix
// is never materialised (tt_xg_lane); everything downstream (y pass, z stage) is unchanged and the planes returned are
// bit for bit those tg_tensor_planes computes from the matrix tg_kron_sum_csr would have written.
template <int NT>
struct tt_xg_multi {
  tt_xg_args<NT> c[4];
  unsigned first[5], gx[4];
  int n;
};
template <int P, int NT>
__global__ void __launch_bounds__(64) k_tt_xg_multi(tt_xg_multi<NT> M) {
  // (the piece of the walk is the slowest-varying index: see tt_xg_args::ech)
  const unsigned piece = blockIdx.x / M.first[M.n], b = blockIdx.x - piece * M.first[M.n];
  int c = 0;
  while (c + 1 < M.n && b >= M.first[c + 1]) c++;
  const unsigned local = b - M.first[c];
  tt_xg_lane<P, NT>(M.c[c], (int)(local % M.gx[c]), (int)(local / M.gx[c]), threadIdx.x, (int)piece);
}

template <int NT>
static int tt_launch_xg(tg_tensor_plan_s *pl, int z0, const std::vector<int32_t> *pls, int32_t *const *d_pl, double *b1,
                        const int64_t *d_pb1, const int *nnz1d) {
  const int P = pl->P, W = 2 * P + 1;
  const tt_dir_t &D0 = pl->dir[0], &D1 = pl->dir[1], &D2 = pl->dir[2];
  tt_xg_multi<NT> XM;
  memset(&XM, 0, sizeof(XM));
  // The walk in pieces, the piece as the slowest-varying block index: what a wave reads through the scalar unit per element --
  // the rows of the x factor of every term (P nodes x NT terms x ~1.5 (P + 1) doubles) and the local weights ((P + 1)^2
  // doubles) -- is 0.56 KB at P = 3 with three terms, 140 KB for the whole direction at cfg3: the waves of a CU, spread over
  // the direction, kept missing the 16 KB scalar cache (six batches of scalar loads per element, each waited for: 36 % of the
  // vector cycles busy).  With all resident workgroups inside one window of ~12 KB the loads hit: 3.41 -> 2.34 ms per
  // sub-slab at cfg3 (pieces of 8 / 12 / 16 / 20 / 24 / 32 / 40 / 48 / 64 elements: 2.60 / 2.46 / 2.38 / 2.34 / 2.35 / 2.38 /
  // 2.63 / 2.97 / 3.34 ms; a piece re-walks P elements, its rows are bit for bit those of the whole walk).
  // TIGAR_TT_XG_ECH=n sets the piece length, 0 walks the direction at once.
  static const int ech_env = getenv("TIGAR_TT_XG_ECH") ? atoi(getenv("TIGAR_TT_XG_ECH")) : -1;
  const double per_element = 8.0 * (P * NT * 1.5 * (P + 1) + (P + 1) * (P + 1));
  int ech = ech_env >= 0 ? ech_env : (int)(12288.0 / per_element);
  ech = ech > 0 && ech < D0.nel ? std::max(ech, 2 * P) : 0;
  for (int pc = 1; pc >= 0; pc--) {
    if (pls[pc].empty()) continue;
    const int n2 = pc == 0 ? P + 1 : W;
    for (int lc = 1; lc >= 0; lc--) {
      if (!pl->nlines1[lc]) continue;
      const int n1 = lc == 0 ? P + 1 : W;
      tt_xg_args<NT> &X = XM.c[XM.n];
      X.d0 = D0;
      X.cv0 = pl->kcv[0];
      X.cv1 = pl->kcv[1];
      X.cv2 = pl->kcv[2];
      X.nnz0 = nnz1d[0];
      X.nnz1 = nnz1d[1];
      X.nnz2 = nnz1d[2];
      X.rps1 = D1.rps;
      X.rps2 = D2.rps;
      X.nfe1 = D1.nfe;
      X.nfe2 = D2.nfe;
      X.lines = pl->lines1[lc];
      X.nlines = pl->nlines1[lc];
      X.n1 = n1;
      X.L = std::max(1, 64 / (n1 * n2));
      X.planes = d_pl[pc];
      X.n2 = n2;
      X.b1 = b1;
      X.pb1 = d_pb1;
      X.z0 = z0;
      X.ech = ech;
      XM.gx[XM.n] = (unsigned)tg_cdiv(X.nlines, X.L);
      XM.first[XM.n + 1] = XM.first[XM.n] + XM.gx[XM.n] * (unsigned)pls[pc].size();
      XM.n++;
    }
  }
  if (XM.n == 0 || XM.first[XM.n] == 0) return 0;
  const unsigned npieces = ech > 0 ? (unsigned)tg_cdiv(D0.nel, ech) : 1u;
#define TT_XG(PP) hipLaunchKernelGGL((k_tt_xg_multi<PP, NT>), dim3(XM.first[XM.n] * npieces), dim3(64), 0, g_tg.stream, XM)
  TT_DISPATCH_P(P, TT_XG);
#undef TT_XG
  return 0;
}

extern "C" int tg_tensor_planes_kron(tg_tensor_plan_t pl, int nterms, const tg_kron_dir_t *dirs, int z0, int z1,
                                     tg_tensor_planes_t *out) {
  TG_REQUIRE_INIT();
  TG_REQUIRE(pl && dirs && out, "null argument to tg_tensor_planes_kron");
  TG_REQUIRE(pl->d == 3, "tg_tensor_planes_kron: 3-D plans only");
  if (nterms < 1 || nterms > 3) return 100;
  const int P = pl->P, W = 2 * P + 1;
  const tt_dir_t &D0 = pl->dir[0], &D1 = pl->dir[1], &D2 = pl->dir[2];
  TG_REQUIRE(z0 >= 0 && z1 > z0 && z1 <= D2.nfe, "tg_tensor_planes_kron: plane range out of bounds");
  // the 1-D factors must sit on exactly the 1-D element-coupling patterns of the plan
  int nnz1d[3];
  uint64_t key = 1469598103934665603ull ^ (uint64_t)nterms;
  for (int k = 0; k < 3; k++) {
    const int nfe = pl->dir[k].nfe;
    if (dirs[k].n != nfe || !dirs[k].rowptr || !dirs[k].col || !dirs[k].val) return 100;
    if (memcmp(dirs[k].rowptr, pl->h_rps[k].data(), (size_t)(nfe + 1) * sizeof(int32_t)) != 0) return 100;
    nnz1d[k] = pl->h_rps[k][nfe];
    if (memcmp(dirs[k].col, pl->h_ecol[k].data(), (size_t)nnz1d[k] * sizeof(int32_t)) != 0) return 100;
    const uint64_t *w = (const uint64_t *)dirs[k].val;
    for (int64_t i = 0; i < (int64_t)nterms * nnz1d[k]; i++) key = (key ^ w[i]) * 1099511628211ull + (uint64_t)k;
  }
  if (key != pl->kcv_key || nterms != pl->kcv_terms || !pl->kcv[0]) {
    // (the previous tables may still be read by kernels in flight: released to the pool in stream order)
    for (int k = 0; k < 3; k++) {
      tg_dfree(pl->kcv[k]);
      pl->kcv[k] = nullptr;
      TG_TRY(tg_dmalloc(&pl->kcv[k], (int64_t)nterms * nnz1d[k]));
      TG_TRY(tg_h2d_staged(pl->kcv[k], dirs[k].val, (size_t)nterms * nnz1d[k] * sizeof(double)));
    }
    pl->kcv_key = key;
    pl->kcv_terms = nterms;
  }
  const int np = z1 - z0;
  std::vector<int64_t> pb1(np + 1, 0), pb2(np + 1, 0);
  std::vector<int32_t> pls[2];
  for (int q = 0; q < np; q++) {
    const int n2 = tt_rn_host(P, z0 + q, D2.nfe);
    pb1[q + 1] = pb1[q] + (int64_t)W * n2 * D0.ncp * pl->h_rps[1][D1.nfe];
    pb2[q + 1] = pb2[q] + (int64_t)W * W * n2 * D0.ncp * D1.ncp;
    pls[n2 == P + 1 ? 0 : 1].push_back(z0 + q);
  }
  double *b1 = nullptr;
  int64_t *d_pb1 = nullptr, *d_pb2 = nullptr;
  int32_t *d_pl[2] = {nullptr, nullptr};
  tg_tensor_planes_s *res = new tg_tensor_planes_s();
  res->z0 = z0;
  res->z1 = z1;
  res->pb = pb2;
  int rc = tg_dmalloc(&b1, pb1[np]);
  if (!rc) rc = tg_dmalloc(&res->buf, pb2[np]);
  if (!rc) rc = tg_dmalloc(&res->status, 4);
  if (!rc) rc = tt_upload(&d_pb1, pb1);
  if (!rc) rc = tt_upload(&d_pb2, pb2);
  if (!rc) rc = tt_upload(&d_pl[0], pls[0]);
  if (!rc) rc = tt_upload(&d_pl[1], pls[1]);
  if (!rc && hipMemsetAsync(res->status, 0, sizeof(int), g_tg.stream) != hipSuccess) rc = 1;
  if (!rc) {
    if (nterms == 1) rc = tt_launch_xg<1>(pl, z0, pls, d_pl, b1, d_pb1, nnz1d);
    else if (nterms == 2) rc = tt_launch_xg<2>(pl, z0, pls, d_pl, b1, d_pb1, nnz1d);
    else rc = tt_launch_xg<3>(pl, z0, pls, d_pl, b1, d_pb1, nnz1d);
    tt_y_multi YM;
    memset(&YM, 0, sizeof(YM));
    for (int pc = 1; pc >= 0; pc--) {
      if (pls[pc].empty()) continue;
      const int n2 = pc == 0 ? P + 1 : W;
      tt_y_args &Y = YM.c[YM.n];
      Y.b1 = b1;
      Y.pb1 = d_pb1;
      Y.b2 = res->buf;
      Y.pb2 = d_pb2;
      Y.z0 = z0;
      Y.d1 = D1;
      Y.ncp0 = D0.ncp;
      Y.planes = d_pl[pc];
      Y.n2 = n2;
      Y.L = std::max(1, 64 / (W * n2));
      Y.ech = tt_y_ech(D1.nel, P);
      YM.gx[YM.n] = (unsigned)tg_cdiv(D0.ncp, Y.L);
      YM.first[YM.n + 1] = YM.first[YM.n] + YM.gx[YM.n] * (unsigned)pls[pc].size();
      YM.n++;
    }
    if (!rc && YM.n > 0 && YM.first[YM.n] > 0) {
#define TT_Y(PP) \
  hipLaunchKernelGGL((k_tt_y_multi<PP>), dim3(YM.first[YM.n] * tt_pieces(D1.nel, tt_y_ech(D1.nel, P))), dim3(64), 0, g_tg.stream, YM)
      TT_DISPATCH_P(P, TT_Y);
#undef TT_Y
    }
    if (hipGetLastError() != hipSuccess) {
      tg_set_error("tg_tensor_planes_kron: kernel launch failed");
      rc = 1;
    }
  }
  tg_dfree(b1);
  tg_dfree(d_pb1);
  tg_dfree(d_pb2);
  tg_dfree(d_pl[0]);
  tg_dfree(d_pl[1]);
  if (rc) {
    tg_tensor_planes_destroy(res);
    return rc;
  }
  *out = res;
  return 0;
}

// ------------------------------------------------------------------------------------------------------
// An FE matrix on the node grid whose pattern is NOT the element-coupling pattern (couplings added by hand: contact
// terms, constraints; or entries missing): split into the part that lies on the pattern -- copied to the closed-form
// positions of a matrix that has exactly the pattern (zeros where A has no entry) -- and the remainder, a CSR matrix of
// the entries outside.  M^T A M = M^T A_on M (the line-walk passes) + M^T A_off M (general kernels, few entries).
struct tt_split_args {
  const int64_t *rowptr;     // A
  const int32_t *col;
  const double *val;
  int nfe0, nfe1, nfe2;
  const int64_t *crowptr;    // the conforming matrix (rows in closed form)
  double *cval;
  int64_t nrows;
};

template <int P>
__device__ __forceinline__ bool tt_split_slot(const tt_split_args &A, int a0, int a1, int a2, int32_t c, int &slot) {
  const uint32_t cu = (uint32_t)c, t = cu / (uint32_t)A.nfe0;
  const int b0 = (int)(cu - t * (uint32_t)A.nfe0);
  const int b2 = (int)(t / (uint32_t)A.nfe1), b1 = (int)(t - (uint32_t)b2 * (uint32_t)A.nfe1);
  const int l0 = tt_rlo<P>(a0, A.nfe0), n0 = tt_rn<P>(a0, A.nfe0);
  const int l1 = tt_rlo<P>(a1, A.nfe1), n1 = tt_rn<P>(a1, A.nfe1);
  const int l2 = tt_rlo<P>(a2, A.nfe2), n2 = tt_rn<P>(a2, A.nfe2);
  const int d0 = b0 - l0, d1 = b1 - l1, d2 = b2 - l2;
  const bool in = d0 >= 0 && d0 < n0 && d1 >= 0 && d1 < n1 && d2 >= 0 && d2 < n2;
  slot = (d2 * n1 + d1) * n0 + d0;
  return in;
}

// one wave per row.  FILL = false: entries on the pattern go to their slots, the others are counted (rem_len);
// FILL = true: the others are copied in order to the remainder (rem_rowptr scanned in between)
template <int P, bool FILL>
__global__ void __launch_bounds__(256)
    k_tt_split(tt_split_args A, int64_t *__restrict__ rem_rowptr, int32_t *__restrict__ rem_col,
               double *__restrict__ rem_val) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t r = wave; r < A.nrows; r += nwaves) {
    const uint32_t ru = (uint32_t)r, t = ru / (uint32_t)A.nfe0;
    const int a0 = (int)(ru - t * (uint32_t)A.nfe0);
    const int a2 = (int)(t / (uint32_t)A.nfe1), a1 = (int)(t - (uint32_t)a2 * (uint32_t)A.nfe1);
    const int64_t b = A.rowptr[r], e = A.rowptr[r + 1], cb = A.crowptr[r];
    int64_t out = FILL ? rem_rowptr[r] : 0;
    int cnt = 0;
    for (int64_t q0 = b; q0 < e; q0 += 64) {
      const int64_t q = q0 + lane;
      bool off = false;
      int32_t c = 0;
      double v = 0.0;
      if (q < e) {
        c = A.col[q];
        v = A.val[q];
        int slot;
        const bool in = tt_split_slot<P>(A, a0, a1, a2, c, slot);
        if (in) {
          if (!FILL) A.cval[cb + slot] = v;
        } else
          off = true;
      }
      const unsigned long long m = __ballot(off);
      if (FILL) {
        if (off) {
          const int64_t pos = out + __popcll(m & ((1ull << lane) - 1ull));
          rem_col[pos] = c;
          rem_val[pos] = v;
        }
        out += __popcll(m);
      } else
        cnt += __popcll(m);
    }
    if (!FILL && lane == 0) rem_rowptr[r] = cnt;
  }
}

extern "C" int tg_tensor_split(tg_tensor_plan_t pl, tg_csr_t a, tg_csr_t *on_pattern, tg_csr_t *remainder) {
  TG_REQUIRE_INIT();
  TG_REQUIRE(pl && a && on_pattern && remainder, "null argument to tg_tensor_split");
  TG_REQUIRE(pl->d == 3, "tg_tensor_split: 3-D plans only");
  TG_REQUIRE_CANONICAL(a);
  const int P = pl->P;
  const tt_dir_t &D0 = pl->dir[0], &D1 = pl->dir[1], &D2 = pl->dir[2];
  const int64_t n = (int64_t)D0.nfe * D1.nfe * D2.nfe;
  if (a->nrows != n || a->ncols != n || n >= 0xffffffffll) return 100;     // not a (whole) matrix on this node grid
  // the matrix with exactly the pattern, values zero: the Kronecker product of the 1-D patterns (carries the certificate)
  std::vector<double> zeros[3];
  tg_kron_dir_t dirs[3];
  for (int k = 0; k < 3; k++) {
    zeros[k].assign(pl->h_ecol[k].size(), 0.0);
    dirs[k].n = pl->dir[k].nfe;
    dirs[k].rowptr = pl->h_rps[k].data();
    dirs[k].col = pl->h_ecol[k].data();
    dirs[k].val = zeros[k].data();
  }
  tg_csr_s *conf = nullptr, *rem = nullptr;
  TG_TRY(tg_kron_sum_csr(3, 1, dirs, 0, n, &conf));
  int rc = 0;
  if (conf->pattern_tag != pl->expect_tag) {
    tg_set_error("tg_tensor_split: the pattern matrix does not carry the plan's certificate");
    rc = 1;
  }
  int64_t *rlen = nullptr;
  if (!rc) rc = tg_dmalloc(&rlen, n + 1);
  tt_split_args S;
  S.rowptr = a->rowptr;
  S.col = a->col;
  S.val = a->val;
  S.nfe0 = D0.nfe;
  S.nfe1 = D1.nfe;
  S.nfe2 = D2.nfe;
  S.crowptr = conf->rowptr;
  S.cval = conf->val;
  S.nrows = n;
  const unsigned grid = (unsigned)std::min<int64_t>(tg_cdiv(n, 4), (int64_t)g_tg.num_cu * 32);
  if (!rc) {
    // (the values written by the Kronecker kernel are 0.0 * ... = 0.0: every slot A does not fill stays zero)
#define TT_S0(PP) hipLaunchKernelGGL((k_tt_split<PP, false>), dim3(grid), dim3(256), 0, g_tg.stream, S, rlen, (int32_t *)nullptr, (double *)nullptr)
    TT_DISPATCH_P(P, TT_S0);
#undef TT_S0
    if (hipGetLastError() != hipSuccess) rc = 1;
  }
  int64_t total = 0;
  if (!rc) rc = tg_exclusive_scan_i64(rlen, n, &total);
  if (!rc) rc = tg_csr_alloc(n, n, total, &rem);
  if (!rc) {
    if (hipMemcpyAsync(rem->rowptr, rlen, (size_t)(n + 1) * sizeof(int64_t), hipMemcpyDeviceToDevice, g_tg.stream) != hipSuccess)
      rc = 1;
    if (!rc && total > 0) {
#define TT_S1(PP) hipLaunchKernelGGL((k_tt_split<PP, true>), dim3(grid), dim3(256), 0, g_tg.stream, S, rem->rowptr, rem->col, rem->val)
      TT_DISPATCH_P(P, TT_S1);
#undef TT_S1
      if (hipGetLastError() != hipSuccess) rc = 1;
    }
  }
  if (hipStreamSynchronize(g_tg.stream) != hipSuccess) rc = 1;
  tg_dfree(rlen);
  if (rc) {
    tg_csr_destroy(conf);
    if (rem) tg_csr_destroy(rem);
    tg_set_error("tg_tensor_split failed");
    return 1;
  }
  *on_pattern = conf;
  *remainder = rem;
  return 0;
}

extern "C" int tg_tensor_zstage(tg_tensor_plan_t pl, int npieces, const tg_tensor_planes_t *pieces, int ka, int kb,
                                const int32_t *zero_dofs, int64_t nzero, double diag, tg_csr_builder_t dest,
                                tg_csr_t *out) {
  TG_REQUIRE_INIT();
  TG_REQUIRE(pl && pieces && npieces >= 1 && (dest || out), "null argument to tg_tensor_zstage");
  TG_REQUIRE(pl->d == 3, "tg_tensor_zstage: 3-D plans only");
  const int P = pl->P, W = 2 * P + 1;
  const tt_dir_t &D0 = pl->dir[0], &D1 = pl->dir[1], &D2 = pl->dir[2];
  const int ncr0 = pl->ncr[0], ncr1 = pl->ncr[1], ncr2 = pl->ncr[2];
  TG_REQUIRE(ka >= 0 && kb > ka && kb <= ncr2, "tg_tensor_zstage: dof planes out of range");
  TG_REQUIRE(!pl->pair || !(zero_dofs && nzero > 0), "tg_tensor_zstage: blocks with different bases on the two sides take no zero dofs "
                                                     "(MatZeroRowsColumns belongs to the assembled matrix)");
  const int e_begin = std::max(0, ka - pl->pr[2]), e_end = std::min(D2.nel, kb);   // (row function i lives on the elements [i - pr, i])
  const int plo = e_begin == 0 ? 0 : P * e_begin + 1, phi = P * e_end;      // FE planes read: [plo, phi]
  std::vector<const double *> ptr(phi - plo + 1, nullptr);
  for (int q = 0; q < npieces; q++) {
    const tg_tensor_planes_s *pc = pieces[q];
    TG_REQUIRE(pc, "tg_tensor_zstage: null piece");
    for (int r = std::max(plo, pc->z0); r <= std::min(phi, pc->z1 - 1); r++) ptr[r - plo] = pc->buf + pc->pb[r - pc->z0];
  }
  for (size_t i = 0; i < ptr.size(); i++)
    TG_REQUIRE(ptr[i], "tg_tensor_zstage: FE plane %d is in none of the pieces", plo + (int)i);
  const int64_t pd = (int64_t)ncr0 * ncr1;                   // rows of K per dof plane (true functions of the row side)
  const int64_t pd_pad = (int64_t)D0.ncp * D1.ncp;           // lines the walk covers (padded functions included)
  const int64_t nrows = (int64_t)(kb - ka) * pd;
  const int64_t w01 = (int64_t)pl->h_kps[0][ncr0] * pl->h_kps[1][ncr1];
  const int64_t nnz = w01 * (pl->h_kps[2][kb] - pl->h_kps[2][ka]);
  const int64_t ncols = (int64_t)pl->ncc[0] * pl->ncc[1] * pl->ncc[2];
  // destination
  tg_csr_s *m = nullptr;
  int64_t row_at = 0, nnz_at = 0;
  if (dest) {
    TG_REQUIRE(dest->m && dest->m->ncols == ncols, "tg_tensor_zstage: builder has other dimensions");
    TG_TRY(tg_csr_builder_reserve(dest, nrows, nnz));
    m = dest->m;
    row_at = dest->rows_done;
    nnz_at = dest->nnz_done;
  } else {
    TG_TRY(tg_csr_alloc(nrows, ncols, nnz, &m));
  }
  uint8_t *mask = nullptr;
  const double **d_ptr = nullptr;
  int rc = 0;
  if (zero_dofs && nzero > 0) rc = tg_build_dof_mask(zero_dofs, nzero, ncols, &mask);
  if (!rc) rc = tt_upload(&d_ptr, ptr);
  if (!rc) {
    tt_rowptr_args R;
    R.kps0 = D0.kps;
    R.kps1 = D1.kps;
    R.kps2 = D2.kps;
    R.ncp0 = ncr0;
    R.ncp1 = ncr1;
    R.ka = ka;
    R.kb = kb;
    R.base = nnz_at;
    R.rowptr_out = m->rowptr + row_at;
    hipLaunchKernelGGL(k_tt_rowptr, dim3((unsigned)tg_cdiv(nrows, 256)), dim3(256), 0, g_tg.stream, R, nrows);
    const int64_t end = nnz_at + nnz;
    if (hipMemcpyAsync(m->rowptr + row_at + nrows, &end, sizeof(int64_t), hipMemcpyHostToDevice, g_tg.stream) != hipSuccess)
      rc = 1;
    tt_z_args Z;
    Z.planes = d_ptr;
    Z.plane_lo = plo;
    Z.d2 = D2;
    Z.ncp0 = D0.ncp;
    Z.ncp1 = D1.ncp;
    Z.kps0 = D0.kps;
    Z.kps1 = D1.kps;
    Z.ka = ka;
    Z.kb = kb;
    Z.ncr0 = ncr0;
    Z.ncr1 = ncr1;
    Z.ncc0 = pl->ncc[0];
    Z.ncc1 = pl->ncc[1];
    Z.pr0 = pl->pr[0];
    Z.pr1 = pl->pr[1];
    Z.pr2 = pl->pr[2];
    Z.L = std::max(1, 64 / (W * W));
    // the diagonal is recorded while the rows are written (only while every row so far came from here)
    if (row_at == 0 && !m->diag_cache && !pl->pair) {
      if (tg_dmalloc(&m->diag_cache, m->nrows)) m->diag_cache = nullptr;
      m->diag_rows = 0;
    }
    const bool keep_diag = m->diag_cache && m->diag_rows == row_at;
    Z.kdiag = keep_diag ? m->diag_cache + row_at : nullptr;
    Z.kcol = m->col + nnz_at;
    Z.kval = m->val + nnz_at;
    Z.mask = mask;
    Z.diag = diag;
    const unsigned grid = (unsigned)tg_cdiv(pd_pad, Z.L);
#define TT_Z(PP) hipLaunchKernelGGL((k_tt_z<PP>), dim3(grid), dim3(64), 0, g_tg.stream, Z)
    TT_DISPATCH_P(P, TT_Z);
#undef TT_Z
    if (hipGetLastError() != hipSuccess) {
      tg_set_error("tg_tensor_zstage: kernel launch failed");
      rc = 1;
    }
    // the flag of the x / y passes that fed this stage (read here when they ran on a certified matrix without waiting)
    std::vector<int> piece_bad((size_t)npieces, 0);
    for (int q = 0; q < npieces && !rc; q++)
      if (pieces[q]->status &&
          hipMemcpyAsync(&piece_bad[(size_t)q], pieces[q]->status, sizeof(int), hipMemcpyDeviceToHost, g_tg.stream) != hipSuccess)
        rc = 1;
    if (!rc && hipStreamSynchronize(g_tg.stream) != hipSuccess) {   // (`end`, the mask and the pointer table are released below)
      tg_set_error("tg_tensor_zstage: %s", hipGetErrorString(hipGetLastError()));
      rc = 1;
    }
    int passes_bad = 0;
    for (int q = 0; q < npieces; q++) passes_bad |= piece_bad[(size_t)q];
    if (!rc && passes_bad) {
      tg_set_error("tg_tensor_zstage: the planes of this stage come from a matrix whose rows do not have the lengths of "
                   "the pattern it was certified for");
      rc = 1;
    }
  }
  tg_dfree(mask);
  tg_dfree(d_ptr);
  if (rc) {
    if (!dest) tg_csr_destroy(m);
    return rc;
  }
  if (m->diag_cache && m->diag_rows == row_at) m->diag_rows = row_at + nrows;
  g_tg.prof_n[TG_PROF_PTAP_TENSOR_WALKS] += 1;
  if (dest) {
    dest->rows_done += nrows;
    dest->nnz_done += nnz;
  } else {
    m->nnz = nnz;
    *out = m;
  }
  return 0;
}

// ------------------------------------------------------------------------------------------------------
// 2-D patches, nF fields on one scalar basis (cfg4: biharmonic p = 4; cfg5: three fields p = 3):
// K = P_y^T (P_x^T A P_x) P_y with M = I_nF (x) M_y (x) M_x.  A = [A_fg] must hold nF x nF blocks that ALL carry the
// element-coupling pattern of the Q_p grid (row (a, r1, f): nF * n1 * n0 entries [g][c1][c0]); verified on the device
// like the 3-D path (status 100 = another pattern: general kernels).  Two passes, both cut into pieces of `ech`
// elements along the walked direction -- a 2-D patch has only ~nfe lines, far too few waves otherwise.
#define TT_DISPATCH_P4(P, CALL) \
  do {                          \
    if ((P) == 4) {             \
      CALL(4);                  \
    } else                      \
      TT_DISPATCH_P(P, CALL);   \
  } while (0)

struct tt_x2_multi {
  tt_x_args c[2];
  unsigned first[3], gx[2];
  int n, nF;
};
template <int P, bool V>
__global__ void __launch_bounds__(64) k_tt_x2(tt_x2_multi M) {
  int c = 0;
  while (c + 1 < M.n && blockIdx.x >= M.first[c + 1]) c++;
  const tt_x_args &A = M.c[c];
  if (*(volatile int *)A.status) return;
  const unsigned local = blockIdx.x - M.first[c];
  const unsigned bx = local % M.gx[c], rest = local / M.gx[c];
  const int bad = tt_x_lane<P, V>(A, (int)bx, (int)(rest % (unsigned)M.nF), threadIdx.x, (int)(rest / (unsigned)M.nF));
  if (bad) atomicOr(A.status, 1);
}
template <int P>
__global__ void __launch_bounds__(64) k_tt_y2(tt_y2_args A, unsigned gx) {
  const unsigned bx = blockIdx.x % gx, rest = blockIdx.x / gx;
  tt_y2_lane<P>(A, (int)bx, (int)(rest % (unsigned)A.nF), (int)(rest / (unsigned)A.nF), threadIdx.x);
}
__global__ void __launch_bounds__(256) k_tt_rowptr2(tt_rowptr2_args A, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) tt_rowptr2_one(A, i);
}

static int tt_plan_create2(int nfields, const tg_tensor_pair_dir_t *dirs, bool pair, tg_tensor_plan_t *out) {
  const int P = dirs[0].p;
  TG_REQUIRE(P >= 1 && P <= 4 && dirs[1].p == P, "tg_tensor2_plan_create: equal degrees 1..4 in both directions");
  TG_REQUIRE((2 * P + 1) * nfields <= 64, "tg_tensor2_plan_create: (2p+1) * nfields lanes must fit a wave");
  TG_REQUIRE(!pair || nfields == 1, "tg_tensor2_plan_create_pair: one field on either side");
  tg_tensor_plan_s *pl = new tg_tensor_plan_s();
  pl->P = P;
  pl->d = 2;
  pl->nF = nfields;
  pl->pair = pair;
  int rc = 0;
  for (int k = 0; k < 2 && !rc; k++) {
    const int nel = dirs[k].nel, nfe = P * nel + 1, ncp = nel + P;
    const int pr = pair ? dirs[k].pr : P, pc = pair ? dirs[k].pc : P;
    if (nel < 1 || !dirs[k].wlc || (pair && !dirs[k].wlr) || pr < 1 || pr > P || pc < 1 || pc > P) {
      tg_set_error("tg_tensor2_plan_create: bad direction %d (spline degrees 1..%d on both sides)", k, P);
      rc = 2;
      break;
    }
    const int ncr = nel + pr, ncc = nel + pc;
    pl->pr[k] = pr, pl->pc[k] = pc, pl->ncr[k] = ncr, pl->ncc[k] = ncc;
    std::vector<double> w(dirs[k].wlc, dirs[k].wlc + (size_t)nel * (P + 1) * (P + 1));
    pl->h_rps[k].assign(nfe + 1, 0);
    for (int a = 0; a < nfe; a++) pl->h_rps[k][a + 1] = pl->h_rps[k][a] + tt_rn_host(P, a, nfe);
    // 1-D pattern of the product: row function i couples to the column functions [i - pr, i + pc], clipped
    pl->h_kps[k].assign(ncr + 1, 0);
    for (int i = 0; i < ncr; i++)
      pl->h_kps[k][i + 1] = pl->h_kps[k][i] + (std::min(ncc - 1, i + pc) - std::max(0, i - pr) + 1);
    rc = tt_upload(&pl->wl[k], w);
    if (!rc && pair) {
      std::vector<double> wr(dirs[k].wlr, dirs[k].wlr + (size_t)nel * (P + 1) * (P + 1));
      rc = tt_upload(&pl->wlr[k], wr);
    }
    if (!rc) rc = tt_upload(&pl->rps[k], pl->h_rps[k]);
    if (!rc) rc = tt_upload(&pl->kps[k], pl->h_kps[k]);
    pl->dir[k].nel = nel;
    pl->dir[k].nfe = nfe;
    pl->dir[k].ncp = ncp;                 // (padded: the layout of the intermediate)
    pl->dir[k].wl = pl->wl[k];
    pl->dir[k].wlr = pair ? pl->wlr[k] : nullptr;
    pl->dir[k].rps = pl->rps[k];
    pl->dir[k].kps = pl->kps[k];
  }
  if (!rc) {
    // direction 2 = the field index: row f couples to all nF fields
    pl->h_rps[2].assign(nfields + 1, 0);
    for (int f = 0; f <= nfields; f++) pl->h_rps[2][f] = f * nfields;
    rc = tt_upload(&pl->rps[2], pl->h_rps[2]);
    pl->dir[2].nel = 0;
    pl->dir[2].nfe = pl->dir[2].ncp = nfields;
    pl->dir[2].rps = pl->rps[2];
  }
  if (!rc && nfields == 1) {
    // certificate of a scalar matrix written by tg_kron_sum_csr on this grid (two directions)
    std::vector<int32_t> ecol[2];
    const int32_t *rps[2], *cls[2];
    int64_t nr[2], nc[2];
    for (int k = 0; k < 2; k++) {
      const int nfe = pl->dir[k].nfe;
      for (int a = 0; a < nfe; a++) {
        const int lo = tt_rlo_host(P, a, nfe), n = tt_rn_host(P, a, nfe);
        for (int j = 0; j < n; j++) ecol[k].push_back(lo + j);
      }
      rps[k] = pl->h_rps[k].data();
      cls[k] = ecol[k].data();
      nr[k] = nc[k] = nfe;
    }
    pl->expect_tag = tg_pattern_hash(2, nr, nc, rps, cls, 0);
  }
  if (!rc) {
    std::vector<int32_t> ls, lv;
    const int nfe1 = pl->dir[1].nfe;
    for (int a = 0; a < nfe1; a++) (tt_rn_host(P, a, nfe1) == P + 1 ? ls : lv).push_back(a);
    pl->nlines1[0] = (int)ls.size();
    pl->nlines1[1] = (int)lv.size();
    rc = tt_upload(&pl->lines1[0], ls);
    if (!rc) rc = tt_upload(&pl->lines1[1], lv);
    if (!rc) rc = tg_dmalloc(&pl->status, 4);
  }
  if (rc) {
    tg_tensor_plan_destroy(pl);
    return rc;
  }
  *out = pl;
  return 0;
}

extern "C" int tg_tensor2_plan_create(int nfields, const tg_tensor_dir_t *dirs, tg_tensor_plan_t *out) {
  TG_REQUIRE_INIT();
  TG_REQUIRE(dirs && out && nfields >= 1, "bad arguments to tg_tensor2_plan_create");
  tg_tensor_pair_dir_t pd[2];
  for (int k = 0; k < 2; k++) {
    pd[k].p = dirs[k].p;
    pd[k].nel = dirs[k].nel;
    pd[k].pr = pd[k].pc = dirs[k].p;
    pd[k].wlr = nullptr;
    pd[k].wlc = dirs[k].wl;
  }
  return tt_plan_create2(nfields, pd, false, out);
}

/* Block (f, g) of a space whose fields sit on DIFFERENT tensor bases over one 2-D Q_P node grid (the components of a 2-D
 * compatible B-spline, tIGAr/compatibleSplines.py:21-66; demos/taylor-green/taylor-green-2d.py): K_fg = M_f^T A_fg M_g in
 * the same two passes with separate row- and column-side weights (padded to P + 1 functions per element), the last pass
 * writing the true pattern. */
extern "C" int tg_tensor2_plan_create_pair(const tg_tensor_pair_dir_t *dirs, tg_tensor_plan_t *out) {
  TG_REQUIRE_INIT();
  TG_REQUIRE(dirs && out, "bad arguments to tg_tensor2_plan_create_pair");
  return tt_plan_create2(1, dirs, true, out);
}

extern "C" int tg_tensor2_ptap(tg_tensor_plan_t pl, tg_csr_t a, const int32_t *zero_dofs, int64_t nzero, double diag,
                               tg_csr_t *out) {
  TG_REQUIRE_INIT();
  TG_REQUIRE(pl && a && out, "null argument to tg_tensor2_ptap");
  TG_REQUIRE(pl->d == 2, "tg_tensor2_ptap: the plan belongs to a 3-D patch");
  TG_REQUIRE_CANONICAL(a);
  const int P = pl->P, W = 2 * P + 1, nF = pl->nF;
  const tt_dir_t &D0 = pl->dir[0], &D1 = pl->dir[1];
  const int64_t nfe = (int64_t)D0.nfe * D1.nfe, ncp = (int64_t)D0.ncp * D1.ncp;
  if (a->nrows != nF * nfe || a->ncols != nF * nfe) return 100;        // not a matrix on this mixed space
  const int64_t t0 = pl->h_rps[0][D0.nfe], t1 = pl->h_rps[1][D1.nfe];
  if (a->nnz != (int64_t)nF * nF * t0 * t1) return 100;                 // (cheap: the pattern has exactly this many entries)
  const int ncr0 = pl->pair ? pl->ncr[0] : D0.ncp, ncr1 = pl->pair ? pl->ncr[1] : D1.ncp;
  const int ncc0 = pl->pair ? pl->ncc[0] : D0.ncp, ncc1 = pl->pair ? pl->ncc[1] : D1.ncp;
  const int64_t w0tot = pl->h_kps[0][ncr0], w1tot = pl->h_kps[1][ncr1];
  const int64_t knnz = (int64_t)nF * nF * w0tot * w1tot, krows = (int64_t)nF * ncr0 * ncr1, kcols = (int64_t)nF * ncc0 * ncc1;
  (void)ncp;
  TG_REQUIRE(nF * nfe < 0x7fffffffll && krows < 0x7fffffffll && kcols < 0x7fffffffll, "tg_tensor2_ptap: index range");
  TG_REQUIRE(!pl->pair || !(zero_dofs && nzero > 0), "tg_tensor2_ptap: boundary conditions belong to the assembled matrix, not to a block");
  // elements per piece of the walks: enough pieces to give the chip a few thousand waves, not so short that the P
  // re-read elements of a piece dominate
  auto pick_ech = [&](int nel, int64_t waves_per_piece, const char *env, int floor_ech) {
    if (getenv(env)) {                       // (experiments)
      const int e = atoi(getenv(env));
      return e >= nel ? 0 : std::max(1, e);
    }
    int ech = nel;
    while (ech > floor_ech && (int64_t)tg_cdiv(nel, ech) * waves_per_piece < 2048) ech = (ech + 1) / 2;
    return ech >= nel ? 0 : ech;
  };
  const int64_t plane_b1 = (int64_t)W * nF * D0.ncp * t1;
  double *b1 = nullptr;
  tg_csr_s *m = nullptr;
  uint8_t *mask = nullptr;
  int64_t *d_pb1 = nullptr;
  int32_t *d_planes = nullptr;
  int rc = tg_dmalloc(&b1, plane_b1 * nF);
  if (!rc) rc = tg_csr_alloc(krows, kcols, knnz, &m);
  if (!rc && zero_dofs && nzero > 0) rc = tg_build_dof_mask(zero_dofs, nzero, krows, &mask);
  {
    std::vector<int64_t> pb1(nF + 1, 0);
    std::vector<int32_t> planes(nF, 0);
    for (int f = 0; f < nF; f++) {
      pb1[f + 1] = pb1[f] + plane_b1;
      planes[f] = f;
    }
    if (!rc) rc = tt_upload(&d_pb1, pb1);
    if (!rc) rc = tt_upload(&d_planes, planes);
  }
  if (!rc && hipMemsetAsync(pl->status, 0, sizeof(int), g_tg.stream) != hipSuccess) rc = 1;
  int bad = 0;
  if (!rc) {
    tt_check_args Cq;
    Cq.rowptr = a->rowptr;
    Cq.nfe0 = D0.nfe;
    Cq.nfe1 = D1.nfe;
    Cq.nfe2 = nF;
    Cq.aplane0 = 0;
    Cq.z0 = 0;
    Cq.dense2 = 1;
    const int64_t nr = nF * nfe;
#define TT_C(PP) hipLaunchKernelGGL((k_tt_check_rows<PP>), dim3((unsigned)tg_cdiv(nr, 256)), dim3(256), 0, g_tg.stream, Cq, nr, pl->status)
    TT_DISPATCH_P4(P, TT_C);
#undef TT_C
    // x pass: both line classes in one launch
    tt_x2_multi XM;
    memset(&XM, 0, sizeof(XM));
    XM.nF = nF;
    int64_t waves = 0;
    for (int lc = 1; lc >= 0; lc--)
      if (pl->nlines1[lc]) waves += tg_cdiv(pl->nlines1[lc], std::max(1, 64 / ((lc == 0 ? P + 1 : W) * nF))) * nF;
    const int ech0 = pick_ech(D0.nel, waves, "TIGAR_TT2_ECH_X", 4 * P);
    const unsigned np0 = ech0 ? (unsigned)tg_cdiv(D0.nel, ech0) : 1u;
    for (int lc = 1; lc >= 0; lc--) {
      if (!pl->nlines1[lc]) continue;
      const int n1 = lc == 0 ? P + 1 : W;
      tt_x_args &X = XM.c[XM.n];
      X.rowptr = a->rowptr;
      X.col = a->col;
      X.val = a->val;
      X.rps2 = pl->rps[2];
      X.aplane0 = 0;
      X.d0 = D0;
      X.nfe1 = D1.nfe;
      X.nfe2 = nF;
      X.rps1 = D1.rps;
      X.lines = pl->lines1[lc];
      X.nlines = pl->nlines1[lc];
      X.n1 = n1;
      X.n2 = nF;
      X.L = std::max(1, 64 / (n1 * nF));
      X.planes = d_planes;
      X.b1 = b1;
      X.pb1 = d_pb1;
      X.z0 = 0;
      X.status = pl->status;
      X.dense2 = 1;
      X.ech = ech0;
      XM.gx[XM.n] = (unsigned)tg_cdiv(X.nlines, X.L);
      XM.first[XM.n + 1] = XM.first[XM.n] + XM.gx[XM.n] * (unsigned)nF * np0;
      XM.n++;
    }
    const bool certified = a->pattern_tag != 0 && a->pattern_tag == pl->expect_tag && a->pattern_row0 == 0 &&
                           !(getenv("TIGAR_PTAP_VERIFY") && atoi(getenv("TIGAR_PTAP_VERIFY")));
    if (XM.n > 0 && XM.first[XM.n] > 0) {
#define TT_X(PP) hipLaunchKernelGGL((k_tt_x2<PP, true>), dim3(XM.first[XM.n]), dim3(64), 0, g_tg.stream, XM)
#define TT_XC(PP) hipLaunchKernelGGL((k_tt_x2<PP, false>), dim3(XM.first[XM.n]), dim3(64), 0, g_tg.stream, XM)
      if (certified) TT_DISPATCH_P4(P, TT_XC);
      else TT_DISPATCH_P4(P, TT_X);
#undef TT_X
#undef TT_XC
      g_tg.prof_n[TG_PROF_PTAP_CERTIFIED] += certified ? 1 : 0;
    }
    // final pass along direction 1: rows of K
    tt_rowptr2_args R;
    R.kps0 = D0.kps;
    R.kps1 = D1.kps;
    R.ncp0 = ncr0;
    R.ncp1 = ncr1;
    R.nF = nF;
    R.rowptr_out = m->rowptr;
    hipLaunchKernelGGL(k_tt_rowptr2, dim3((unsigned)tg_cdiv(krows, 256)), dim3(256), 0, g_tg.stream, R, krows);
    if (hipMemcpyAsync(m->rowptr + krows, &knnz, sizeof(int64_t), hipMemcpyHostToDevice, g_tg.stream) != hipSuccess) rc = 1;
    tt_y2_args Y;
    Y.b1 = b1;
    Y.plane_b1 = plane_b1;
    Y.d1 = D1;
    Y.ncp0 = D0.ncp;
    Y.nF = nF;
    Y.kps0 = D0.kps;
    Y.ncr0 = Y.ncr1 = Y.ncc0 = Y.ncc1 = Y.pr0 = Y.pr1 = 0;
    if (pl->pair) Y.ncr0 = ncr0, Y.ncr1 = ncr1, Y.ncc0 = ncc0, Y.ncc1 = ncc1, Y.pr0 = pl->pr[0], Y.pr1 = pl->pr[1];
    Y.L = std::max(1, 64 / (W * nF));
    const unsigned gx = (unsigned)tg_cdiv(D0.ncp, Y.L);
    Y.ech = pick_ech(D1.nel, (int64_t)gx * nF, "TIGAR_TT2_ECH_Y", 2 * P);
    const unsigned np1 = Y.ech ? (unsigned)tg_cdiv(D1.nel, Y.ech) : 1u;
    if (!pl->pair && !m->diag_cache && tg_dmalloc(&m->diag_cache, krows)) m->diag_cache = nullptr;
    Y.kdiag = m->diag_cache;
    Y.kcol = m->col;
    Y.kval = m->val;
    Y.mask = mask;
    Y.diag = diag;
#define TT_Y2(PP) hipLaunchKernelGGL((k_tt_y2<PP>), dim3(gx * (unsigned)nF * np1), dim3(64), 0, g_tg.stream, Y, gx)
    TT_DISPATCH_P4(P, TT_Y2);
#undef TT_Y2
    if (hipGetLastError() != hipSuccess) {
      tg_set_error("tg_tensor2_ptap: kernel launch failed");
      rc = 1;
    }
    if (!rc && hipMemcpyAsync(&bad, pl->status, sizeof(int), hipMemcpyDeviceToHost, g_tg.stream) != hipSuccess) rc = 1;
    if (!rc && hipStreamSynchronize(g_tg.stream) != hipSuccess) {
      tg_set_error("tg_tensor2_ptap: %s", hipGetErrorString(hipGetLastError()));
      rc = 1;
    }
  }
  tg_dfree(b1);
  tg_dfree(mask);
  tg_dfree(d_pb1);
  tg_dfree(d_planes);
  if (rc || bad) {
    if (m) tg_csr_destroy(m);
    return rc ? rc : 100;
  }
  m->nnz = knnz;
  m->diag_rows = m->diag_cache ? krows : 0;
  g_tg.prof_n[TG_PROF_PTAP_TENSOR_WALKS] += 1;
  *out = m;
  return 0;
}
