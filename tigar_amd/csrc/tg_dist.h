// RCCL communicator + z-slab descriptor (one process per GPU).
#pragma once
#include "tg_common.h"
#include <rccl/rccl.h>

struct tg_comm_s {
  ncclComm_t comm = nullptr;
  int rank = 0, world = 1;
  // host-staged variant (tg_comm_create_host): exchanges go through pinned host memory and the caller's transport
  int kind = 0;                       // 0 = RCCL, 1 = host-staged
  tg_host_allreduce_fn h_allreduce = nullptr;
  tg_host_sendrecv_fn h_sendrecv = nullptr;
  void *h_ctx = nullptr;
  double *stage = nullptr;            // pinned
  int64_t stage_cap = 0;
  // this rank owns global dofs [g0,g1); its SpMV needs halo_lo dofs below g0 and halo_hi
  // above g1; send_lo / send_hi = what the neighbours need from this rank's ends
  int64_t g0 = 0, g1 = 0, halo_lo = 0, halo_hi = 0, nglobal = 0;
  int64_t send_lo = 0, send_hi = 0;
  bool slab_set = false;
  // the halo exchange runs on a stream of its own so that the rows of a product that need no halo can be
  // computed meanwhile (tg_comm_halo_begin / tg_comm_halo_end); created on first use
  hipStream_t xstream = nullptr;
  hipEvent_t x_ready = nullptr, x_done = nullptr;
  bool x_open = false;
};

#define TG_CHECK_NCCL(expr)                                                           \
  do {                                                                                \
    ncclResult_t _r = (expr);                                                         \
    if (_r != ncclSuccess) {                                                          \
      tg_set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, ncclGetErrorString(_r)); \
      return 1;                                                                       \
    }                                                                                 \
  } while (0)

// exchanges the halo of the extended vector xext = [halo_lo | owned | halo_hi] (device)
int tg_comm_halo_exchange(tg_comm_s *c, double *xext);
// the same in two halves: `begin` starts the exchange of the owned ends as they are on the current stream at
// this point; work enqueued on the current stream between the two calls must not read the halo entries; after
// `end` the current stream sees the received halo
int tg_comm_halo_begin(tg_comm_s *c, double *xext);
int tg_comm_halo_end(tg_comm_s *c, double *xext);
// in-place sum over ranks of n device doubles
int tg_comm_allreduce_dev(tg_comm_s *c, double *dev, int n);
