// Multi-GPU plumbing: RCCL over xGMI.  The only data-path collectives of the hot path are
// (1) the z-slab halo exchange of the Krylov direction vector before each SpMV (p planes,
//     point-to-point with the two z-neighbours), and
// (2) all-reduces of 1-31 doubles for dot products / norms.
// Reference counterparts: PETSc VecScatter inside MatMult and MPI_Allreduce inside VecDot /
// VecNorm of the KSP called at tIGAr/common.py:1255-1258 [ext].
//
// Three communicators carry the same solver code:
//   kind 0  RCCL: ncclSend/ncclRecv for the halo (own communicator, exchange stream), ncclAllReduce (second
//           communicator, solver stream);
//   kind 1  host-staged: pinned staging + the caller's transport (a host wait per exchange);
//   kind 2  IPC: device mailboxes opened through HIP IPC + flags in shared host memory, waits inside the kernels
//           -- enqueue-only like RCCL, and it also works for ranks that SHARE a GPU (SURVEY 8e's "direct xGMI
//           peer copies").
#include "tg_dist.h"
#include <algorithm>
#include <chrono>
#include <future>
#include <thread>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

// ------------------------------------------------------------------------------------------ IPC kernels
// Everything another process reads or writes goes through SYSTEM-scope relaxed atomics (sc0 sc1 on gfx950: stores
// write through to memory, loads miss every non-coherent cache level), so no exchange needs a cache-wide
// write-back or invalidate -- a release / acquire FENCE at system scope costs ~0.1 ms while the products of two
// ranks stream through the L2s (measured) -- and the order "payload, then flag" is kept by waiting for the stores
// of the workgroup to be ACKNOWLEDGED before the flag goes out: an explicit `s_waitcnt vmcnt(0)` (gfx9 family: vmcnt
// counts stores as well; a system-scope store is acknowledged once it is visible at system scope), then the barrier.
// The workgroup-scope release fence that stood here alone in round 3 happens to lower to the same wait with this
// toolchain, but LLVM's memory model for gfx90a+ allows it to omit the wait outside tgsplit mode: the order
// must not depend on that.  gfx950 only (gfx10+ counts stores on vscnt).
__device__ __forceinline__ unsigned long long tg_ld_sys(const unsigned long long *p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ void tg_st_sys(unsigned long long *p, unsigned long long v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ double tg_ld_sys(const double *p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ void tg_st_sys(double *p, double v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
// all stores this workgroup has issued are acknowledged by memory when every thread has passed this point
__device__ __forceinline__ void tg_stores_done_block() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");          // compiler: nothing moves below
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");               // hardware: every store of this wave acknowledged
  __syncthreads();
}
// spins until *p >= v; false after `tmo` wall-clock ticks or when another rank has given up
__device__ __forceinline__ bool tg_spin_ge(const unsigned long long *p, unsigned long long v, tg_ipc_shm *s,
                                           long long tmo) {
  if (tg_ld_sys(p) >= v) return true;
  const long long t0 = wall_clock64();
  unsigned n = 0;
  // exponential back-off: the first polls come 128 cycles apart (a peer on its own GPU answers within microseconds), the
  // interval doubles every 64 polls up to ~8 k cycles -- ranks that SHARE a GPU (verification runs: eight ranks of cfg2
  // spent 1.16 s per solve spinning at the short interval, their polls and the peers' kernels on the same CUs) leave the
  // issue slots and the memory pipeline to whoever they are waiting for
  while (tg_ld_sys(p) < v) {
    const unsigned lvl = min(n >> 6, 6u);
    for (unsigned k = 0; k < (1u << lvl); k++) __builtin_amdgcn_s_sleep(2);
    if ((++n & 15u) == 0 || lvl >= 4u) {
      if (tg_ld_sys(&s->abort_word) != 0ull) return false;
      if (wall_clock64() - t0 > tmo) return false;
    }
  }
  return true;
}
__device__ __forceinline__ void tg_ipc_give_up(tg_ipc_shm *s, int rank, unsigned long long what) {
  tg_st_sys(&s->status[rank], what);
  tg_st_sys(&s->abort_word, 1ull + (unsigned long long)rank);
}

// in-place sum over ranks of n <= TG_IPC_AR_MAX doubles: every rank publishes its contribution in its slot of the
// shared table, waits for the others' and adds the slots up in rank order -- every rank gets the same bits, the bits
// of the host-staged reduction (which also adds in rank order)
__global__ void __launch_bounds__(256) k_ipc_allreduce(tg_ipc_shm *s, int rank, int world, double *dev, int n,
                                                       unsigned long long seq, long long tmo) {
  __shared__ int ok_lds;
  const int par = (int)(seq & 1ull), tid = threadIdx.x;
  if (tid == 0) ok_lds = 1;
  for (int t = tid; t < n; t += 256) tg_st_sys(&s->ar_slot[par][rank][t], dev[t]);
  tg_stores_done_block();
  if (tid == 0) tg_st_sys(&s->ar_flag[par][rank], seq);
  if (tid < world && tid != rank)
    if (!tg_spin_ge(&s->ar_flag[par][tid], seq, s, tmo)) ok_lds = 0;
  __syncthreads();
  if (!ok_lds) {
    if (tid == 0) tg_ipc_give_up(s, rank, 1ull);
    return;
  }
  for (int t = tid; t < n; t += 256) {
    double acc = 0.0;
    for (int r = 0; r < world; r++) acc += tg_ld_sys(&s->ar_slot[par][r][t]);
    dev[t] = acc;
  }
}

struct tg_ipc_leg {          // one direction of a halo exchange
  double *dst;
  const double *src;
  long long n;               // doubles; 0 = this leg does not exist
  unsigned long long *wait;  // flag to wait for before copying (ack of the receiver / arrival flag)
  unsigned long long wait_val;
  unsigned long long *post;  // flag to set when every block has copied
  unsigned long long post_val;
};
struct tg_ipc_legs {
  tg_ipc_leg leg[2];
};

// copies both legs; REMOTE = 1: the destination is another rank's mailbox (system-scope stores), REMOTE = 0: the
// source is the own mailbox another rank filled (system-scope loads).  Every block waits for a leg's condition
// itself; the last block to finish posts the flags.
template <int REMOTE>
__global__ void __launch_bounds__(256) k_ipc_move(tg_ipc_legs L, tg_ipc_shm *s, int rank, unsigned *done, long long tmo,
                                                  unsigned long long what) {
  __shared__ int ok_lds;
  const int tid = threadIdx.x;
  if (tid == 0) ok_lds = 1;
  __syncthreads();
  for (int k = 0; k < 2; k++) {
    const tg_ipc_leg &g = L.leg[k];
    if (g.n <= 0) continue;
    if (tid == 0 && g.wait)
      if (!tg_spin_ge(g.wait, g.wait_val, s, tmo)) ok_lds = 0;
    __syncthreads();
    if (!ok_lds) break;
    const long long stride = (long long)gridDim.x * 256;
    for (long long i = (long long)blockIdx.x * 256 + tid; i < g.n; i += stride) {
      if (REMOTE)
        tg_st_sys(g.dst + i, g.src[i]);
      else
        g.dst[i] = tg_ld_sys(g.src + i);
    }
  }
  tg_stores_done_block();
  if (tid == 0) {
    if (!ok_lds) tg_ipc_give_up(s, rank, what);
    const unsigned prev = __hip_atomic_fetch_add(done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (prev == gridDim.x - 1) {
      __hip_atomic_store(done, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (ok_lds)
        for (int k = 0; k < 2; k++)
          if (L.leg[k].n > 0 && L.leg[k].post) tg_st_sys(L.leg[k].post, L.leg[k].post_val);
    }
  }
}

static inline void tg_comm_host_wait() { g_tg.prof_n[TG_PROF_COMM_HOST_WAITS] += 1; }

extern "C" int tg_comm_unique_id(char *id128) {
  TG_REQUIRE(id128, "null id buffer");
  static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId size");
  ncclUniqueId id;
  TG_CHECK_NCCL(ncclGetUniqueId(&id));
  memcpy(id128, &id, 128);
  return 0;
}

// ncclCommInitRank blocks for ever when a rank never arrives: run it on a helper thread and give up after
// TIGAR_RCCL_TIMEOUT_S seconds (default 180; the thread is abandoned then -- the caller falls back to another
// communicator or fails the run)
static int tg_nccl_init_with_timeout(ncclComm_t *comm, int world, const ncclUniqueId &id, int rank, const char *what) {
  const char *e = getenv("TIGAR_RCCL_TIMEOUT_S");
  const double limit = e ? atof(e) : 180.0;
  struct shared_t {
    std::promise<ncclResult_t> done;
    ncclComm_t comm = nullptr;
  };
  auto sh = std::make_shared<shared_t>();
  std::future<ncclResult_t> fut = sh->done.get_future();
  const int device = g_tg.device;
  std::thread th([sh, world, id, rank, device]() {
    ncclResult_t r = ncclSystemError;
    if (hipSetDevice(device) == hipSuccess) r = ncclCommInitRank(&sh->comm, world, id, rank);
    sh->done.set_value(r);
  });
  if (fut.wait_for(std::chrono::duration<double>(limit)) != std::future_status::ready) {
    th.detach();
    tg_set_error("ncclCommInitRank (%s, rank %d of %d) did not return within %.0f s", what, rank, world, limit);
    return 1;
  }
  th.join();
  const ncclResult_t r = fut.get();
  if (r != ncclSuccess) {
    tg_set_error("ncclCommInitRank (%s) failed: %s", what, ncclGetErrorString(r));
    return 1;
  }
  *comm = sh->comm;
  return 0;
}

extern "C" int tg_comm_create(const char *id128, int rank, int world, tg_comm_t *out) {
  return tg_comm_create2(id128, nullptr, rank, world, out);
}

// two RCCL communicators over the same ranks: `id_reduce` carries the all-reduces on the solver's stream, `id_halo`
// (may be null: one communicator for both) the send/recv pairs on the exchange stream -- RCCL orders the operations of
// ONE communicator across streams with extra event dependencies, two communicators keep the streams independent
extern "C" int tg_comm_create2(const char *id_reduce, const char *id_halo, int rank, int world, tg_comm_t *out) {
  TG_REQUIRE_INIT();
  TG_REQUIRE(id_reduce && out && world >= 1 && rank >= 0 && rank < world, "bad arguments to tg_comm_create");
  tg_comm_s *c = new tg_comm_s();
  c->rank = rank;
  c->world = world;
  ncclUniqueId id;
  memcpy(&id, id_reduce, 128);
  if (tg_nccl_init_with_timeout(&c->comm, world, id, rank, "all-reduce") != 0) {
    delete c;
    return 1;
  }
  c->comm_x = c->comm;
  if (id_halo && world > 1) {
    memcpy(&id, id_halo, 128);
    if (tg_nccl_init_with_timeout(&c->comm_x, world, id, rank, "halo") != 0) {
      ncclCommDestroy(c->comm);
      delete c;
      return 1;
    }
  }
  *out = c;
  return 0;
}

// ------------------------------------------------------------------------------------------ IPC communicator
extern "C" int tg_comm_ipc_shm_bytes(int64_t *bytes) {
  TG_REQUIRE(bytes, "null argument");
  *bytes = (int64_t)((sizeof(tg_ipc_shm) + 4095) / 4096 * 4096);
  return 0;
}

static int tg_ipc_allreduce_host(tg_comm_s *c, double *host_inout, int n);

static int tg_ipc_close_peers(tg_comm_s *c) {
  for (int k = 0; k < 2; k++)
    if (c->peer_mail[k]) {
      TG_CHECK_HIP(hipIpcCloseMemHandle(c->peer_mail[k]));
      c->peer_mail[k] = nullptr;
    }
  return 0;
}

// (re)allocates the own mailbox with room for `cap` doubles, publishes its handle and opens the neighbours'.
// Collective: every rank calls it at the same point.
static int tg_ipc_mailboxes(tg_comm_s *c, int64_t cap, bool first) {
  TG_CHECK_HIP(hipStreamSynchronize(g_tg.stream));
  if (c->xstream) TG_CHECK_HIP(hipStreamSynchronize(c->xstream));
  TG_TRY(tg_ipc_close_peers(c));
  if (!first) {   // nobody may still map the block that is about to be freed
    double one = 1.0;
    TG_TRY(tg_ipc_allreduce_host(c, &one, 1));
  }
  if (cap > c->mail_cap || !c->mail) {
    if (c->mail) TG_CHECK_HIP(hipFree(c->mail));
    c->mail = nullptr;
    cap = std::max<int64_t>(cap, 8192);
    TG_CHECK_HIP(hipMalloc((void **)&c->mail, (size_t)cap * sizeof(double)));
    TG_CHECK_HIP(hipMemset(c->mail, 0, (size_t)cap * sizeof(double)));
    c->mail_cap = cap;
  }
  tg_ipc_shm *s = c->shm;
  TG_CHECK_HIP(hipIpcGetMemHandle(&s->mail_h[c->rank], c->mail));
  s->mail_cap[c->rank] = c->mail_cap;
  s->device_of[c->rank] = g_tg.device;
  c->mail_generation += 1;
  __atomic_store_n(&s->mail_gen[c->rank], c->mail_generation, __ATOMIC_RELEASE);
  const char *e = getenv("TIGAR_IPC_TIMEOUT_S");
  const double limit = e ? atof(e) : 1800.0;
  const auto t0 = std::chrono::steady_clock::now();
  for (int k = 0; k < 2; k++) {
    const int nb = k == 0 ? c->rank - 1 : c->rank + 1;
    if (nb < 0 || nb >= c->world) continue;
    while (__atomic_load_n(&s->mail_gen[nb], __ATOMIC_ACQUIRE) < c->mail_generation) {
      if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > limit) {
        tg_set_error("IPC communicator: rank %d did not publish its mailbox within %.0f s", nb, limit);
        return 1;
      }
      usleep(100);
    }
    hipIpcMemHandle_t h = s->mail_h[nb];
    TG_CHECK_HIP(hipIpcOpenMemHandle((void **)&c->peer_mail[k], h, hipIpcMemLazyEnablePeerAccess));
  }
  return 0;
}

extern "C" int tg_comm_create_ipc(const char *shm_path, int rank, int world, tg_comm_t *out) {
  TG_REQUIRE_INIT();
  TG_REQUIRE(shm_path && out && world >= 1 && world <= TG_IPC_MAXW && rank >= 0 && rank < world,
             "bad arguments to tg_comm_create_ipc (at most %d ranks)", TG_IPC_MAXW);
  int64_t bytes = 0;
  tg_comm_ipc_shm_bytes(&bytes);
  const int fd = open(shm_path, O_RDWR);
  TG_REQUIRE(fd >= 0, "IPC communicator: cannot open %s", shm_path);
  struct stat st;
  if (fstat(fd, &st) != 0 || st.st_size < bytes) {
    close(fd);
    tg_set_error("IPC communicator: %s is smaller than %lld bytes", shm_path, (long long)bytes);
    return 2;
  }
  void *m = mmap(nullptr, (size_t)bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  close(fd);
  TG_REQUIRE(m != MAP_FAILED, "IPC communicator: mmap of %s failed", shm_path);
  tg_comm_s *c = new tg_comm_s();
  c->rank = rank;
  c->world = world;
  c->kind = 2;
  c->shm = (tg_ipc_shm *)m;
  auto fail = [&](int rc) {
    tg_comm_destroy(c);
    return rc;
  };
  if (hipHostRegister(m, (size_t)bytes, hipHostRegisterMapped | hipHostRegisterPortable) != hipSuccess) {
    tg_set_error("IPC communicator: hipHostRegister of the shared flags failed");
    return fail(1);
  }
  c->shm_registered = true;
  if (hipHostGetDevicePointer((void **)&c->shm_dev, m, 0) != hipSuccess) {
    tg_set_error("IPC communicator: no device pointer for the shared flags");
    return fail(1);
  }
  int khz = 0;
  if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, g_tg.device) != hipSuccess || khz <= 0) khz = 100000;
  // Ranks reach a matching exchange far apart in ordinary runs (rank-0-only I/O, uneven assembly, ranks sharing a GPU):
  // the device-side waits give up after 30 minutes unless TIGAR_IPC_TIMEOUT_S says otherwise; the self-test, whose
  // exchanges are back to back on every rank, runs with its own short limit (tg_comm_selftest).
  const char *e = getenv("TIGAR_IPC_TIMEOUT_S");
  c->khz = khz;
  c->tmo_ticks = (long long)((e ? atof(e) : 1800.0) * 1000.0 * khz);
  if (hipMalloc((void **)&c->done_ctr, 4 * sizeof(unsigned)) != hipSuccess ||
      hipMemset(c->done_ctr, 0, 4 * sizeof(unsigned)) != hipSuccess) {
    tg_set_error("IPC communicator: allocation failed");
    return fail(1);
  }
  const char *mb = getenv("TIGAR_IPC_MAILBOX_MB");
  const int64_t cap = (int64_t)((mb ? atof(mb) : 8.0) * 1048576.0 / 8.0);
  if (world > 1) {
    int rc = tg_ipc_mailboxes(c, cap, true);
    if (rc) return fail(rc);
  }
  *out = c;
  return 0;
}

// status of the exchanges enqueued so far (call after a host wait): 0 = fine
int tg_comm_check(tg_comm_s *c) {
  if (!c || c->kind != 2 || !c->shm) return 0;
  const unsigned long long ab = __atomic_load_n(&c->shm->abort_word, __ATOMIC_ACQUIRE);
  if (ab == 0ull) return 0;
  const unsigned long long mine = __atomic_load_n(&c->shm->status[c->rank], __ATOMIC_ACQUIRE);
  static const char *what[] = {"", "an all-reduce", "the receiver's acknowledgement of a halo push", "the arrival of a halo"};
  if (mine)
    tg_set_error("IPC communicator (rank %d): gave up waiting for %s (a peer rank is gone or stuck)", c->rank,
                 what[mine < 4 ? mine : 0]);
  else
    tg_set_error("IPC communicator (rank %d): rank %llu gave up waiting; the exchange is void", c->rank, ab - 1ull);
  return 3;
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

// devices the ranks of an IPC communicator sit on (as each rank published it); other kinds: -1
extern "C" int tg_comm_rank_device(tg_comm_t c, int rank, int *device) {
  TG_REQUIRE(c && device && rank >= 0 && rank < c->world, "bad arguments to tg_comm_rank_device");
  *device = (c->kind == 2 && c->shm && c->world > 1) ? c->shm->device_of[rank] : (rank == c->rank ? g_tg.device : -1);
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
  if (c->comm_x && c->comm_x != c->comm) ncclCommDestroy(c->comm_x);
  if (c->comm) ncclCommDestroy(c->comm);
  if (c->stage) hipHostFree(c->stage);
  if (c->kind == 2) {
    for (int k = 0; k < 2; k++)
      if (c->peer_mail[k]) hipIpcCloseMemHandle(c->peer_mail[k]);
    if (c->mail) hipFree(c->mail);
    if (c->done_ctr) hipFree(c->done_ctr);
    if (c->shm) {
      int64_t bytes = 0;
      tg_comm_ipc_shm_bytes(&bytes);
      if (c->shm_registered) hipHostUnregister(c->shm);
      munmap(c->shm, (size_t)bytes);
    }
  }
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
    tg_comm_host_wait();
    return 0;
  }
  if (c->kind == 2) {
    TG_REQUIRE(n >= 1 && n <= TG_IPC_AR_MAX, "IPC all-reduce of %d doubles (at most %d)", n, TG_IPC_AR_MAX);
    hipLaunchKernelGGL(k_ipc_allreduce, dim3(1), dim3(256), 0, g_tg.stream, c->shm_dev, c->rank, c->world, dev, n,
                       ++c->ar_seq, c->tmo_ticks);
    TG_LAUNCH_CHECK();
    return 0;
  }
  TG_CHECK_NCCL(ncclAllReduce(dev, dev, (size_t)n, ncclDouble, ncclSum, c->comm, g_tg.stream));
  return 0;
}

// hipStreamSynchronize, or -- with a limit -- polling so that a collective whose peer never arrives comes back
static int tg_comm_wait_stream(tg_comm_s *c, hipStream_t st) {
  if (!c || c->host_wait_limit <= 0.0) {
    TG_CHECK_HIP(hipStreamSynchronize(st));
    return 0;
  }
  const auto t0 = std::chrono::steady_clock::now();
  for (;;) {
    const hipError_t q = hipStreamQuery(st);
    if (q == hipSuccess) return 0;
    if (q != hipErrorNotReady) {
      tg_set_error("communicator: stream failed while waiting: %s", hipGetErrorString(q));
      return 1;
    }
    if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > c->host_wait_limit) {
      tg_set_error("communicator (rank %d): an exchange did not complete within %.0f s", c->rank, c->host_wait_limit);
      return 4;
    }
    usleep(200);
  }
}

extern "C" int tg_comm_allreduce_sum(tg_comm_t c, double *host_inout, int n) {
  TG_REQUIRE_INIT();
  TG_REQUIRE(c && host_inout && n >= 1 && n <= 1024, "bad arguments to tg_comm_allreduce_sum");
  double *d = g_tg.scratch + TG_SCRATCH_DOUBLES - 1024;
  TG_CHECK_HIP(hipMemcpyAsync(d, host_inout, n * sizeof(double), hipMemcpyHostToDevice, g_tg.stream));
  TG_TRY(tg_comm_allreduce_dev(c, d, n));
  TG_CHECK_HIP(hipMemcpyAsync(host_inout, d, n * sizeof(double), hipMemcpyDeviceToHost, g_tg.stream));
  TG_TRY(tg_comm_wait_stream(c, g_tg.stream));
  return tg_comm_check(c);
}

static int tg_ipc_allreduce_host(tg_comm_s *c, double *host_inout, int n) { return tg_comm_allreduce_sum(c, host_inout, n); }

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
  if (c->kind == 2 && W > 1) {
    // layout of the neighbours' mailboxes (2 slots of [from below: their halo_lo | from above: their halo_hi]) and
    // room in the own one; the mailboxes grow together when any rank needs more
    for (int k = 0; k < 2; k++) {
      const int nb = k == 0 ? c->rank - 1 : c->rank + 1;
      const bool there = nb >= 0 && nb < W;
      c->peer_halo[k][0] = there ? (int64_t)all[4 * nb + 2] : 0;
      c->peer_halo[k][1] = there ? (int64_t)all[4 * nb + 3] : 0;
    }
    double grow = 2 * (halo_lo + halo_hi) > c->mail_cap ? 1.0 : 0.0;
    TG_TRY(tg_comm_allreduce_sum(c, &grow, 1));
    if (grow > 0.0) TG_TRY(tg_ipc_mailboxes(c, 2 * (halo_lo + halo_hi), false));
  }
  c->slab_set = true;
  return 0;
}

static int tg_comm_xstream(tg_comm_s *c) {
  if (c->xstream) return 0;
  int prio_lo = 0, prio_hi = 0;
  TG_CHECK_HIP(hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi));
  if (getenv("TIGAR_XSTREAM_PRIO") && atoi(getenv("TIGAR_XSTREAM_PRIO")) == 0) prio_hi = prio_lo;   // (experiments)
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
  if (c->kind == 2) {
    // push the own ends into the neighbours' mailboxes (slot = message number & 1) once they have consumed the
    // message that used the slot before; the last block posts the arrival flags
    tg_ipc_legs L;
    memset(&L, 0, sizeof(L));
    tg_ipc_shm *sd = c->shm_dev;
    if (c->rank > 0 && c->send_lo > 0) {          // to the lower neighbour's "from above" area
      const unsigned long long t = ++c->tx[0];
      const int64_t slot = (c->peer_halo[0][0] + c->peer_halo[0][1]) * (int64_t)(t & 1ull);
      L.leg[0] = {c->peer_mail[0] + slot + c->peer_halo[0][0], own, (long long)c->send_lo,
                  &sd->halo_ack[c->rank - 1][1], t >= 2 ? t - 2 : 0ull, &sd->halo_flag[c->rank - 1][1], t};
    }
    if (c->rank < c->world - 1 && c->send_hi > 0) {   // to the upper neighbour's "from below" area
      const unsigned long long t = ++c->tx[1];
      const int64_t slot = (c->peer_halo[1][0] + c->peer_halo[1][1]) * (int64_t)(t & 1ull);
      L.leg[1] = {c->peer_mail[1] + slot, own + nloc - c->send_hi, (long long)c->send_hi,
                  &sd->halo_ack[c->rank + 1][0], t >= 2 ? t - 2 : 0ull, &sd->halo_flag[c->rank + 1][0], t};
    }
    const int64_t total = c->send_lo + c->send_hi;
    if (total > 0) {
      const unsigned grid = (unsigned)std::min<int64_t>(64, std::max<int64_t>(1, tg_cdiv(total, 4096)));
      hipLaunchKernelGGL(k_ipc_move<1>, dim3(grid), dim3(256), 0, c->xstream, L, sd, c->rank, c->done_ctr,
                         c->tmo_ticks, 2ull);
      TG_LAUNCH_CHECK();
    }
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
    if (c->send_lo > 0) note(ncclSend(own, (size_t)c->send_lo, ncclDouble, c->rank - 1, c->comm_x, c->xstream));
    if (c->halo_lo > 0) note(ncclRecv(xext, (size_t)c->halo_lo, ncclDouble, c->rank - 1, c->comm_x, c->xstream));
  }
  if (c->rank < c->world - 1) {
    if (c->send_hi > 0)
      note(ncclSend(own + nloc - c->send_hi, (size_t)c->send_hi, ncclDouble, c->rank + 1, c->comm_x, c->xstream));
    if (c->halo_hi > 0) note(ncclRecv(own + nloc, (size_t)c->halo_hi, ncclDouble, c->rank + 1, c->comm_x, c->xstream));
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
    tg_comm_host_wait();
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
  if (c->kind == 2) {
    // pull the arrived planes out of the own mailbox into the halo of xext, acknowledge
    double *own = xext + c->halo_lo;
    const int64_t nloc = c->g1 - c->g0;
    tg_ipc_legs L;
    memset(&L, 0, sizeof(L));
    tg_ipc_shm *sd = c->shm_dev;
    if (c->rank > 0 && c->halo_lo > 0) {
      const unsigned long long t = ++c->rx[0];
      const int64_t slot = (c->halo_lo + c->halo_hi) * (int64_t)(t & 1ull);
      L.leg[0] = {xext, c->mail + slot, (long long)c->halo_lo, &sd->halo_flag[c->rank][0], t, &sd->halo_ack[c->rank][0], t};
    }
    if (c->rank < c->world - 1 && c->halo_hi > 0) {
      const unsigned long long t = ++c->rx[1];
      const int64_t slot = (c->halo_lo + c->halo_hi) * (int64_t)(t & 1ull);
      L.leg[1] = {own + nloc, c->mail + slot + c->halo_lo, (long long)c->halo_hi, &sd->halo_flag[c->rank][1], t,
                  &sd->halo_ack[c->rank][1], t};
    }
    const int64_t total = c->halo_lo + c->halo_hi;
    if (total > 0) {
      const unsigned grid = (unsigned)std::min<int64_t>(64, std::max<int64_t>(1, tg_cdiv(total, 4096)));
      hipLaunchKernelGGL(k_ipc_move<0>, dim3(grid), dim3(256), 0, c->xstream, L, sd, c->rank, c->done_ctr + 1,
                         c->tmo_ticks, 3ull);
      TG_LAUNCH_CHECK();
    }
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

// One small all-reduce and one halo exchange with known values on a stream of their own, every host wait bounded by
// `timeout_s`: 0 = the communicator works on every rank's side of it; 4 = an exchange did not complete (the stream is
// abandoned: the caller should drop the communicator WITHOUT destroying it and fall back to another kind); other
// codes = wrong values / errors.  Collective.  Leaves no slab set.
extern "C" int tg_comm_selftest(tg_comm_t c, double timeout_s) {
  TG_REQUIRE_INIT();
  TG_REQUIRE(c, "null communicator");
  if (c->world == 1) return 0;
  hipStream_t saved = g_tg.stream, tmp = nullptr;
  TG_CHECK_HIP(hipStreamCreateWithFlags(&tmp, hipStreamNonBlocking));
  g_tg.stream = tmp;
  c->host_wait_limit = timeout_s > 0.0 ? timeout_s : 60.0;
  const long long saved_tmo = c->tmo_ticks;
  if (c->kind == 2 && c->khz > 0) c->tmo_ticks = std::min(saved_tmo, (long long)(c->host_wait_limit * 1000.0 * c->khz));
  const int W = c->world, R = c->rank;
  const int64_t nloc = 4096, h = 512;
  double *xext = nullptr;
  int rc = 0;
  auto body = [&]() -> int {
    double v[2] = {(double)(R + 1), 1.0};
    TG_TRY(tg_comm_allreduce_sum(c, v, 2));
    TG_REQUIRE(v[0] == 0.5 * W * (W + 1) && v[1] == (double)W, "self-test: all-reduce returned %g, %g on rank %d", v[0], v[1], R);
    const int64_t hlo = R > 0 ? h : 0, hhi = R < W - 1 ? h : 0;
    TG_TRY(tg_comm_set_slab(c, R * nloc, (R + 1) * nloc, hlo, hhi, (int64_t)W * nloc));
    const int64_t next = hlo + nloc + hhi;
    TG_CHECK_HIP(hipMalloc((void **)&xext, (size_t)next * sizeof(double)));
    std::vector<double> host((size_t)next, -1.0);
    for (int64_t i = 0; i < nloc; i++) host[(size_t)(hlo + i)] = (double)(R * nloc + i) + 0.25;
    TG_CHECK_HIP(hipMemcpyAsync(xext, host.data(), (size_t)next * sizeof(double), hipMemcpyHostToDevice, g_tg.stream));
    for (int rep = 0; rep < 3; rep++) TG_TRY(tg_comm_halo_exchange(c, xext));   // (three: both mailbox slots and a reuse)
    TG_TRY(tg_comm_wait_stream(c, g_tg.stream));
    TG_TRY(tg_comm_check(c));
    TG_CHECK_HIP(hipMemcpy(host.data(), xext, (size_t)next * sizeof(double), hipMemcpyDeviceToHost));
    for (int64_t i = 0; i < next; i++)
      TG_REQUIRE(host[(size_t)i] == (double)(R * nloc - hlo + i) + 0.25, "self-test: halo entry %lld of rank %d is %g",
                 (long long)i, R, host[(size_t)i]);
    double ok = 1.0;
    TG_TRY(tg_comm_allreduce_sum(c, &ok, 1));
    TG_REQUIRE(ok == (double)W, "self-test: closing all-reduce returned %g", ok);
    return 0;
  };
  rc = body();
  g_tg.stream = saved;
  c->host_wait_limit = 0.0;
  c->tmo_ticks = saved_tmo;
  c->slab_set = false;
  if (rc != 4) {   // (a stream with an exchange stuck on it cannot be waited for)
    if (xext) hipFree(xext);
    hipStreamDestroy(tmp);
  }
  return rc;
}
