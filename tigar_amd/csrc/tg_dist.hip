// Multi-GPU plumbing: RCCL over xGMI.  The only data-path collectives of the hot path are
// (1) the z-slab halo exchange of the Krylov direction vector before each SpMV (p planes,
//     point-to-point with the two z-neighbours), and
// (2) all-reduces of 1-31 doubles for dot products / norms.
// Reference counterparts: PETSc VecScatter inside MatMult and MPI_Allreduce inside VecDot /
// VecNorm of the KSP called at tIGAr/common.py:1255-1258 [ext].
#include "tg_dist.h"
#include <algorithm>

extern "C" int tg_comm_unique_id(char *id128) {
  TG_REQUIRE(id128, "null id buffer");
  static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId size");
  ncclUniqueId id;
  TG_CHECK_NCCL(ncclGetUniqueId(&id));
  memcpy(id128, &id, 128);
  return 0;
}

extern "C" int tg_comm_create(const char *id128, int rank, int world, tg_comm_t *out) {
  TG_REQUIRE_INIT();
  TG_REQUIRE(id128 && out && world >= 1 && rank >= 0 && rank < world, "bad arguments to tg_comm_create");
  tg_comm_s *c = new tg_comm_s();
  c->rank = rank;
  c->world = world;
  ncclUniqueId id;
  memcpy(&id, id128, 128);
  ncclResult_t r = ncclCommInitRank(&c->comm, world, id, rank);
  if (r != ncclSuccess) {
    tg_set_error("ncclCommInitRank failed: %s", ncclGetErrorString(r));
    delete c;
    return 1;
  }
  *out = c;
  return 0;
}

extern "C" int tg_comm_create_host(int rank, int world, tg_host_allreduce_fn allreduce, tg_host_sendrecv_fn sendrecv,
                                   void *ctx, tg_comm_t *out) {
  TG_REQUIRE_INIT();
  TG_REQUIRE(out && world >= 1 && rank >= 0 && rank < world && allreduce && sendrecv, "bad arguments to tg_comm_create_host");
  tg_comm_s *c = new tg_comm_s();
  c->rank = rank;
  c->world = world;
  c->kind = 1;
  c->h_allreduce = allreduce;
  c->h_sendrecv = sendrecv;
  c->h_ctx = ctx;
  *out = c;
  return 0;
}

extern "C" int tg_comm_info(tg_comm_t c, int *rank, int *world, int *kind) {
  TG_REQUIRE(c && rank && world && kind, "null argument to tg_comm_info");
  *rank = c->rank;
  *world = c->world;
  *kind = c->kind;
  if (c->kind == 0 && c->comm) {   // what RCCL itself reports, not what the caller claimed
    int n = 0, r = -1;
    TG_CHECK_NCCL(ncclCommCount(c->comm, &n));
    TG_CHECK_NCCL(ncclCommUserRank(c->comm, &r));
    *world = n;
    *rank = r;
  }
  return 0;
}

extern "C" int tg_device_count(int *n) {
  TG_REQUIRE(n, "null argument to tg_device_count");
  TG_CHECK_HIP(hipGetDeviceCount(n));
  return 0;
}

extern "C" int tg_comm_destroy(tg_comm_t c) {
  if (!c) return 0;
  if (g_tg.ready) hipStreamSynchronize(g_tg.stream);
  if (c->xstream) {
    hipStreamSynchronize(c->xstream);
    hipEventDestroy(c->x_ready);
    hipEventDestroy(c->x_done);
    hipStreamDestroy(c->xstream);
  }
  if (c->comm) ncclCommDestroy(c->comm);
  if (c->stage) hipHostFree(c->stage);
  delete c;
  return 0;
}

static int tg_comm_stage_reserve(tg_comm_s *c, int64_t doubles) {
  if (doubles <= c->stage_cap) return 0;
  if (c->stage) hipHostFree(c->stage);
  c->stage = nullptr;
  c->stage_cap = 0;
  TG_CHECK_HIP(hipHostMalloc((void **)&c->stage, (size_t)doubles * sizeof(double), hipHostMallocDefault));
  c->stage_cap = doubles;
  return 0;
}

int tg_comm_allreduce_dev(tg_comm_s *c, double *dev, int n) {
  if (!c || c->world == 1) return 0;
  TG_REQUIRE(!c->x_open, "all-reduce between tg_comm_halo_begin and tg_comm_halo_end");
  if (c->kind == 1) {
    TG_TRY(tg_comm_stage_reserve(c, std::max<int64_t>(n, 64)));
    TG_CHECK_HIP(hipMemcpyAsync(c->stage, dev, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, g_tg.stream));
    TG_CHECK_HIP(hipStreamSynchronize(g_tg.stream));
    TG_REQUIRE(c->h_allreduce(c->h_ctx, c->stage, n) == 0, "host transport: allreduce failed (rank %d)", c->rank);
    TG_CHECK_HIP(hipMemcpyAsync(dev, c->stage, (size_t)n * sizeof(double), hipMemcpyHostToDevice, g_tg.stream));
    // the staging buffer is reused by the next exchange: the copy must have left it
    TG_CHECK_HIP(hipStreamSynchronize(g_tg.stream));
    return 0;
  }
  TG_CHECK_NCCL(ncclAllReduce(dev, dev, (size_t)n, ncclDouble, ncclSum, c->comm, g_tg.stream));
  return 0;
}

extern "C" int tg_comm_allreduce_sum(tg_comm_t c, double *host_inout, int n) {
  TG_REQUIRE_INIT();
  TG_REQUIRE(c && host_inout && n >= 1 && n <= 1024, "bad arguments to tg_comm_allreduce_sum");
  double *d = g_tg.scratch + TG_SCRATCH_DOUBLES - 1024;
  TG_CHECK_HIP(hipMemcpyAsync(d, host_inout, n * sizeof(double), hipMemcpyHostToDevice, g_tg.stream));
  TG_TRY(tg_comm_allreduce_dev(c, d, n));
  TG_CHECK_HIP(hipMemcpyAsync(host_inout, d, n * sizeof(double), hipMemcpyDeviceToHost, g_tg.stream));
  TG_CHECK_HIP(hipStreamSynchronize(g_tg.stream));
  return 0;
}

extern "C" int tg_comm_set_slab(tg_comm_t c, int64_t g0, int64_t g1, int64_t halo_lo, int64_t halo_hi,
                                int64_t nglobal) {
  TG_REQUIRE_INIT();
  TG_REQUIRE(c && g0 >= 0 && g1 >= g0 && g1 <= nglobal && halo_lo >= 0 && halo_hi >= 0, "bad slab descriptor");
  TG_REQUIRE(halo_lo <= g0 && g1 + halo_hi <= nglobal, "halo reaches outside the global range");
  c->g0 = g0;
  c->g1 = g1;
  c->halo_lo = halo_lo;
  c->halo_hi = halo_hi;
  c->nglobal = nglobal;
  // learn what the neighbours need from this rank: all-gather {g0,g1,halo_lo,halo_hi}
  const int W = c->world;
  std::vector<double> all((size_t)4 * W, 0.0);
  all[4 * c->rank + 0] = (double)g0;
  all[4 * c->rank + 1] = (double)g1;
  all[4 * c->rank + 2] = (double)halo_lo;
  all[4 * c->rank + 3] = (double)halo_hi;
  if (W > 1) {
    TG_REQUIRE(4 * W <= 1024, "world too large");
    TG_TRY(tg_comm_allreduce_sum(c, all.data(), 4 * W));
  }
  c->send_lo = c->send_hi = 0;
  if (c->rank > 0) {
    TG_REQUIRE((int64_t)all[4 * (c->rank - 1) + 1] == g0, "slabs are not contiguous (rank %d)", c->rank);
    c->send_lo = (int64_t)all[4 * (c->rank - 1) + 3];  // lower neighbour's halo_hi
    TG_REQUIRE(halo_lo <= g0 - (int64_t)all[4 * (c->rank - 1) + 0], "halo_lo spans more than one neighbour slab");
  } else
    TG_REQUIRE(halo_lo == 0, "rank 0 cannot have a lower halo");
  if (c->rank < W - 1) {
    c->send_hi = (int64_t)all[4 * (c->rank + 1) + 2];  // upper neighbour's halo_lo
    TG_REQUIRE(halo_hi <= (int64_t)all[4 * (c->rank + 1) + 1] - g1, "halo_hi spans more than one neighbour slab");
  } else
    TG_REQUIRE(halo_hi == 0, "last rank cannot have an upper halo");
  TG_REQUIRE(c->send_lo <= g1 - g0 && c->send_hi <= g1 - g0, "neighbour halo larger than this slab");
  c->slab_set = true;
  return 0;
}

static int tg_comm_xstream(tg_comm_s *c) {
  if (c->xstream) return 0;
  int prio_lo = 0, prio_hi = 0;
  TG_CHECK_HIP(hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi));
  TG_CHECK_HIP(hipStreamCreateWithPriority(&c->xstream, hipStreamNonBlocking, prio_hi));
  TG_CHECK_HIP(hipEventCreateWithFlags(&c->x_ready, hipEventDisableTiming));
  TG_CHECK_HIP(hipEventCreateWithFlags(&c->x_done, hipEventDisableTiming));
  return 0;
}

// staging layout of the host variant: [send_lo | send_hi | recv_lo (halo_lo) | recv_hi (halo_hi)]
int tg_comm_halo_begin(tg_comm_s *c, double *xext) {
  if (!c || c->world == 1) return 0;
  TG_REQUIRE(c->slab_set, "tg_comm_set_slab() has not been called");
  TG_REQUIRE(!c->x_open, "tg_comm_halo_begin: the previous exchange has not been ended");
  TG_TRY(tg_comm_xstream(c));
  double *own = xext + c->halo_lo;
  const int64_t nloc = c->g1 - c->g0;
  // the exchange stream starts from the state of the current stream
  TG_CHECK_HIP(hipEventRecord(c->x_ready, g_tg.stream));
  TG_CHECK_HIP(hipStreamWaitEvent(c->xstream, c->x_ready, 0));
  if (c->kind == 1) {
    const int64_t total = c->send_lo + c->send_hi + c->halo_lo + c->halo_hi;
    TG_TRY(tg_comm_stage_reserve(c, std::max<int64_t>(total, 64)));
    double *s_lo = c->stage, *s_hi = s_lo + c->send_lo;
    if (c->send_lo > 0)
      TG_CHECK_HIP(hipMemcpyAsync(s_lo, own, (size_t)c->send_lo * sizeof(double), hipMemcpyDeviceToHost, c->xstream));
    if (c->send_hi > 0)
      TG_CHECK_HIP(hipMemcpyAsync(s_hi, own + nloc - c->send_hi, (size_t)c->send_hi * sizeof(double),
                                  hipMemcpyDeviceToHost, c->xstream));
    c->x_open = true;
    return 0;
  }
  TG_CHECK_NCCL(ncclGroupStart());
  // between GroupStart and GroupEnd the first error is remembered and the group is still closed
  ncclResult_t first = ncclSuccess;
  auto note = [&](ncclResult_t r) {
    if (r != ncclSuccess && first == ncclSuccess) first = r;
  };
  if (c->rank > 0) {
    if (c->send_lo > 0) note(ncclSend(own, (size_t)c->send_lo, ncclDouble, c->rank - 1, c->comm, c->xstream));
    if (c->halo_lo > 0) note(ncclRecv(xext, (size_t)c->halo_lo, ncclDouble, c->rank - 1, c->comm, c->xstream));
  }
  if (c->rank < c->world - 1) {
    if (c->send_hi > 0)
      note(ncclSend(own + nloc - c->send_hi, (size_t)c->send_hi, ncclDouble, c->rank + 1, c->comm, c->xstream));
    if (c->halo_hi > 0) note(ncclRecv(own + nloc, (size_t)c->halo_hi, ncclDouble, c->rank + 1, c->comm, c->xstream));
  }
  note(ncclGroupEnd());
  if (first != ncclSuccess) {
    tg_set_error("halo exchange (rank %d): %s", c->rank, ncclGetErrorString(first));
    return 1;
  }
  c->x_open = true;
  return 0;
}

int tg_comm_halo_end(tg_comm_s *c, double *xext) {
  if (!c || c->world == 1) return 0;
  TG_REQUIRE(c->x_open, "tg_comm_halo_end without tg_comm_halo_begin");
  c->x_open = false;
  if (c->kind == 1) {
    double *own = xext + c->halo_lo;
    const int64_t nloc = c->g1 - c->g0;
    double *s_lo = c->stage, *s_hi = s_lo + c->send_lo, *r_lo = s_hi + c->send_hi, *r_hi = r_lo + c->halo_lo;
    TG_CHECK_HIP(hipStreamSynchronize(c->xstream));
    // lower neighbour first, then the upper one: the chain rank 0 <-> 1, 1 <-> 2, ... cannot dead-lock
    // because every exchange sends and receives at once
    if (c->rank > 0 && (c->send_lo > 0 || c->halo_lo > 0))
      TG_REQUIRE(c->h_sendrecv(c->h_ctx, c->rank - 1, s_lo, c->send_lo, r_lo, c->halo_lo) == 0,
                 "host transport: exchange with rank %d failed", c->rank - 1);
    if (c->rank < c->world - 1 && (c->send_hi > 0 || c->halo_hi > 0))
      TG_REQUIRE(c->h_sendrecv(c->h_ctx, c->rank + 1, s_hi, c->send_hi, r_hi, c->halo_hi) == 0,
                 "host transport: exchange with rank %d failed", c->rank + 1);
    if (c->halo_lo > 0)
      TG_CHECK_HIP(hipMemcpyAsync(xext, r_lo, (size_t)c->halo_lo * sizeof(double), hipMemcpyHostToDevice, c->xstream));
    if (c->halo_hi > 0)
      TG_CHECK_HIP(hipMemcpyAsync(own + nloc, r_hi, (size_t)c->halo_hi * sizeof(double), hipMemcpyHostToDevice,
                                  c->xstream));
  }
  TG_CHECK_HIP(hipEventRecord(c->x_done, c->xstream));
  TG_CHECK_HIP(hipStreamWaitEvent(g_tg.stream, c->x_done, 0));
  return 0;
}

int tg_comm_halo_exchange(tg_comm_s *c, double *xext) {
  TG_TRY(tg_comm_halo_begin(c, xext));
  return tg_comm_halo_end(c, xext);
}

extern "C" int tg_comm_halo_extend(tg_comm_t c, tg_vec_t x_local, tg_vec_t xext) {
  TG_REQUIRE_INIT();
  TG_REQUIRE(c && x_local && xext && c->slab_set, "bad arguments to tg_comm_halo_extend");
  const int64_t nloc = c->g1 - c->g0;
  TG_REQUIRE(x_local->n == nloc && xext->n == c->halo_lo + nloc + c->halo_hi, "tg_comm_halo_extend: size mismatch");
  TG_CHECK_HIP(hipMemcpyAsync(xext->d + c->halo_lo, x_local->d, (size_t)nloc * sizeof(double), hipMemcpyDeviceToDevice,
                              g_tg.stream));
  return tg_comm_halo_exchange(c, xext->d);
}
