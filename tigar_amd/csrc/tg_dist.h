// RCCL communicator + z-slab descriptor (one process per GPU).
#pragma once
#include "tg_common.h"
#include <rccl/rccl.h>

// ---- IPC communicator (kind 2): no host in the loop, no RCCL --------------------------------------------------
// Flags and the slots of the small all-reduce live in a POSIX shared-memory file that every rank maps and registers
// with hipHostRegister (fine-grained: visible to every GPU of the node without cache maintenance); halo planes are
// PUSHED by the sender's kernel into the receiver's device mailbox, opened through hipIpcOpenMemHandle (the same
// GPU when ranks share one, an xGMI peer otherwise).  Waits are done by the kernels themselves (system-scope
// acquire loads with a wall-clock timeout), so an exchange is enqueue-only like the RCCL one.
#define TG_IPC_MAXW 16
#define TG_IPC_AR_MAX 1024
struct tg_ipc_shm {
  // host-side rendezvous of the mailbox handles (CPU atomics)
  volatile int mail_gen[TG_IPC_MAXW];
  int64_t mail_cap[TG_IPC_MAXW];
  hipIpcMemHandle_t mail_h[TG_IPC_MAXW];
  volatile int device_of[TG_IPC_MAXW];
  // device-visible part
  alignas(128) unsigned long long ar_flag[2][TG_IPC_MAXW];
  alignas(128) double ar_slot[2][TG_IPC_MAXW][TG_IPC_AR_MAX];
  alignas(128) unsigned long long halo_flag[TG_IPC_MAXW][2];   // [receiver][0 = from below, 1 = from above]: landed
  alignas(128) unsigned long long halo_ack[TG_IPC_MAXW][2];    // [receiver][side]: consumed
  alignas(128) unsigned long long abort_word;                  // != 0: some rank gave up waiting
  unsigned long long status[TG_IPC_MAXW];                      // != 0: what this rank gave up on
};

struct tg_comm_s {
  ncclComm_t comm = nullptr;          // all-reduces (solver stream)
  ncclComm_t comm_x = nullptr;        // halo send/recv (exchange stream); == comm when only one could be made
  int rank = 0, world = 1;
  // host-staged variant (tg_comm_create_host): exchanges go through pinned host memory and the caller's transport
  int kind = 0;                       // 0 = RCCL, 1 = host-staged, 2 = IPC (device mailboxes + shared flags)
  // kind 2
  tg_ipc_shm *shm = nullptr, *shm_dev = nullptr;
  bool shm_registered = false;
  double *mail = nullptr;             // own mailbox: 2 slots x [from below | from above]
  int64_t mail_cap = 0;               // doubles (both slots)
  double *peer_mail[2] = {nullptr, nullptr};   // lower / upper neighbour's mailbox (IPC mapping)
  int64_t peer_halo[2][2] = {{0, 0}, {0, 0}};  // [neighbour][halo_lo, halo_hi] of that neighbour
  unsigned long long ar_seq = 0, tx[2] = {0, 0}, rx[2] = {0, 0};
  unsigned *done_ctr = nullptr;       // device: block counters of the push / pull kernels
  long long tmo_ticks = 0;
  int khz = 0;                       // wall-clock rate of the device (ticks per millisecond)
  int mail_generation = 0;
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
  // > 0: host waits of tg_comm_allreduce_sum give up after this many seconds (tg_comm_selftest)
  double host_wait_limit = 0.0;
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
// after a host wait: 0 when every exchange enqueued so far went through (IPC: no rank gave up waiting)
int tg_comm_check(tg_comm_s *c);
