// developer probe: can W processes that share one GPU (or sit on xGMI peers) exchange data with no host in the loop?
//   * flags / small reduction slots in a POSIX shared-memory file, registered with hipHostRegister in every process
//     (fine-grained, visible to every GPU of the node)
//   * bulk halo data pushed into the neighbour's device mailbox through hipIpcOpenMemHandle
//   * waits are single-wave kernels spinning on the flags with a wall-clock timeout
// hipcc --offload-arch=gfx950 -O3 ipc_probe.hip -o ipc_probe.exe ;  ./ipc_probe.exe 2 [ndev]
#include <hip/hip_runtime.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/wait.h>
#include <unistd.h>
#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#define CK(e)                                                                                         \
  do {                                                                                                \
    hipError_t _r = (e);                                                                              \
    if (_r != hipSuccess) {                                                                           \
      fprintf(stderr, "[rank %d] %s:%d %s -> %s\n", g_rank, __FILE__, __LINE__, #e, hipGetErrorString(_r)); \
      _exit(3);                                                                                       \
    }                                                                                                 \
  } while (0)
static int g_rank = -1;

#define MAXW 8
#define NSLOT 64
struct shm_t {
  std::atomic<int> ready[MAXW];
  std::atomic<int> phase[MAXW];
  hipIpcMemHandle_t h[MAXW];
  // device-visible part
  alignas(64) unsigned long long ar_flag[2][MAXW];
  alignas(64) double ar_slot[2][MAXW][NSLOT];
  alignas(64) unsigned long long halo_flag[MAXW][2];   // [receiver][0 = from below, 1 = from above]
  alignas(64) unsigned long long status[MAXW];
};

__device__ __forceinline__ unsigned long long ld_sys(const unsigned long long *p) {
  return __hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ void st_sys(unsigned long long *p, unsigned long long v) {
  __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// one wave: publish own contribution, wait for everybody's, sum in rank order
__global__ void __launch_bounds__(64) k_allreduce(shm_t *s, int rank, int world, double *dev, int n, unsigned long long seq,
                                                  long long timeout_ticks) {
  const int par = (int)(seq & 1), lane = threadIdx.x;
  if (lane < n) __hip_atomic_store(&s->ar_slot[par][rank][lane], dev[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  __atomic_thread_fence(__ATOMIC_RELEASE);
  if (lane == 0) st_sys(&s->ar_flag[par][rank], seq);
  const long long t0 = wall_clock64();
  bool ok = true;
  if (lane < world) {
    while (ld_sys(&s->ar_flag[par][lane]) < seq) {
      __builtin_amdgcn_s_sleep(8);
      if (wall_clock64() - t0 > timeout_ticks) {
        ok = false;
        break;
      }
    }
  }
  ok = __all(ok);
  if (!ok) {
    if (lane == 0) st_sys(&s->status[rank], 1ull);
    return;
  }
  __atomic_thread_fence(__ATOMIC_ACQUIRE);
  if (lane < n) {
    double acc = 0.0;
    for (int r = 0; r < world; r++)
      acc += __hip_atomic_load(&s->ar_slot[par][r][lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    dev[lane] = acc;
  }
}

__global__ void __launch_bounds__(256) k_copy(double *__restrict__ dst, const double *__restrict__ src, int64_t n) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) dst[i] = src[i];
}
__global__ void k_set(unsigned long long *p, unsigned long long v) { st_sys(p, v); }
__global__ void k_wait(unsigned long long *p, unsigned long long v, unsigned long long *status, long long timeout_ticks) {
  const long long t0 = wall_clock64();
  while (ld_sys(p) < v) {
    __builtin_amdgcn_s_sleep(8);
    if (wall_clock64() - t0 > timeout_ticks) {
      st_sys(status, 2ull);
      return;
    }
  }
}
__global__ void k_fill(double *p, int64_t n, double v) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) p[i] = v + (double)i;
}

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

static void cpu_barrier(shm_t *s, int rank, int world, int phase) {
  s->phase[rank].store(phase);
  for (int r = 0; r < world; r++)
    while (s->phase[r].load() < phase) usleep(50);
}

static int child(const char *path, int rank, int world, int ndev) {
  g_rank = rank;
  int fd = open(path, O_RDWR);
  shm_t *s = (shm_t *)mmap(nullptr, sizeof(shm_t), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  if (s == MAP_FAILED) return 4;
  int have = 0;
  CK(hipGetDeviceCount(&have));
  const int dev = rank % (ndev < have ? ndev : have);
  CK(hipSetDevice(dev));
  int wc = 0;
  CK(hipDeviceGetAttribute(&wc, hipDeviceAttributeWallClockRate, dev));   // kHz
  const long long tmo = (long long)wc * 1000ll * 10ll;                    // 10 s
  const int64_t H = 1 << 18;   // 2 MB of halo per side
  double *mail = nullptr, *own = nullptr, *halo = nullptr;
  CK(hipMalloc(&mail, 2 * H * sizeof(double)));
  CK(hipMalloc(&own, 4 * H * sizeof(double)));
  CK(hipMalloc(&halo, 2 * H * sizeof(double)));
  CK(hipMemset(mail, 0, 2 * H * sizeof(double)));
  CK(hipIpcGetMemHandle(&s->h[rank], mail));
  s->ready[rank].store(1);
  for (int r = 0; r < world; r++)
    while (!s->ready[r].load()) usleep(50);
  double *peer_lo = nullptr, *peer_hi = nullptr;
  if (rank > 0) CK(hipIpcOpenMemHandle((void **)&peer_lo, s->h[rank - 1], hipIpcMemLazyEnablePeerAccess));
  if (rank < world - 1) CK(hipIpcOpenMemHandle((void **)&peer_hi, s->h[rank + 1], hipIpcMemLazyEnablePeerAccess));
  CK(hipHostRegister(s, sizeof(shm_t), hipHostRegisterMapped | hipHostRegisterPortable));
  shm_t *ds = nullptr;
  CK(hipHostGetDevicePointer((void **)&ds, s, 0));
  hipStream_t st;
  CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  double *dv = nullptr;
  CK(hipMalloc(&dv, NSLOT * sizeof(double)));
  cpu_barrier(s, rank, world, 1);
  fprintf(stderr, "[rank %d] dev %d of %d, wall clock %d kHz, peers %p %p, shm dev ptr %p\n", rank, dev, have, wc,
          (void *)peer_lo, (void *)peer_hi, (void *)ds);

  // ---- test 1: all-reduce kernel, enqueued back to back with no host wait
  const int iters = 2000;
  double h3[3] = {1.0 + rank, 0.5 * (rank + 1), 1e-3 * rank};
  unsigned long long seq = 0;
  int bad = 0;
  double t0 = 0;
  for (int rep = 0; rep < 2; rep++) {
    cpu_barrier(s, rank, world, 10 + rep);
    t0 = now();
    for (int it = 0; it < iters; it++) {
      CK(hipMemcpyAsync(dv, h3, sizeof(h3), hipMemcpyHostToDevice, st));
      hipLaunchKernelGGL(k_allreduce, dim3(1), dim3(64), 0, st, ds, rank, world, dv, 3, ++seq, tmo);
    }
    CK(hipStreamSynchronize(st));
  }
  const double t_ar = (now() - t0) / iters;
  double out[3], exp3[3] = {0, 0, 0};
  CK(hipMemcpy(out, dv, sizeof(out), hipMemcpyDeviceToHost));
  for (int r = 0; r < world; r++) {
    exp3[0] += 1.0 + r;
    exp3[1] += 0.5 * (r + 1);
    exp3[2] += 1e-3 * r;
  }
  for (int k = 0; k < 3; k++)
    if (out[k] != exp3[k]) bad++;
  fprintf(stderr, "[rank %d] all-reduce: %.2f us per (copy + kernel), result %s, status %llu\n", rank, 1e6 * t_ar,
          bad ? "WRONG" : "ok", (unsigned long long)s->status[rank]);

  // ---- test 2: halo push through the IPC mailbox + flags in shm
  hipLaunchKernelGGL(k_fill, dim3(256), dim3(256), 0, st, own, 4 * H, 1e6 * (rank + 1));
  CK(hipStreamSynchronize(st));
  cpu_barrier(s, rank, world, 20);
  unsigned long long hs = 0;
  const int hiters = 200;
  t0 = now();
  for (int it = 0; it < hiters; it++) {
    ++hs;
    // (flow control: the all-reduce between two exchanges orders them, as in the CG loop)
    if (peer_lo) {   // my lower end goes into the lower neighbour's "from above" half
      hipLaunchKernelGGL(k_copy, dim3(128), dim3(256), 0, st, peer_lo + H, own, H);
      hipLaunchKernelGGL(k_set, dim3(1), dim3(1), 0, st, &ds->halo_flag[rank - 1][1], hs);
    }
    if (peer_hi) {
      hipLaunchKernelGGL(k_copy, dim3(128), dim3(256), 0, st, peer_hi, own + 3 * H, H);
      hipLaunchKernelGGL(k_set, dim3(1), dim3(1), 0, st, &ds->halo_flag[rank + 1][0], hs);
    }
    if (peer_lo) {
      hipLaunchKernelGGL(k_wait, dim3(1), dim3(1), 0, st, &ds->halo_flag[rank][0], hs, &ds->status[rank], tmo);
      hipLaunchKernelGGL(k_copy, dim3(128), dim3(256), 0, st, halo, mail, H);
    }
    if (peer_hi) {
      hipLaunchKernelGGL(k_wait, dim3(1), dim3(1), 0, st, &ds->halo_flag[rank][1], hs, &ds->status[rank], tmo);
      hipLaunchKernelGGL(k_copy, dim3(128), dim3(256), 0, st, halo + H, mail + H, H);
    }
    hipLaunchKernelGGL(k_allreduce, dim3(1), dim3(64), 0, st, ds, rank, world, dv, 3, ++seq, tmo);
    // change the data so that a stale read is seen
    hipLaunchKernelGGL(k_fill, dim3(256), dim3(256), 0, st, own, 4 * H, 1e6 * (rank + 1) + it + 1);
  }
  CK(hipStreamSynchronize(st));
  const double t_h = (now() - t0) / hiters;
  // check the LAST exchange: values of the fill before the last one (it = hiters-2 + 1) -> base + hiters - 1
  double *hh = (double *)malloc(2 * H * sizeof(double));
  CK(hipMemcpy(hh, halo, 2 * H * sizeof(double), hipMemcpyDeviceToHost));
  int64_t wrong = 0;
  const double shift = (double)(hiters - 1);
  if (peer_lo)
    for (int64_t i = 0; i < H; i++)
      if (hh[i] != 1e6 * rank + shift + (double)(3 * H + i)) wrong++;
  if (peer_hi)
    for (int64_t i = 0; i < H; i++)
      if (hh[H + i] != 1e6 * (rank + 2) + shift + (double)i) wrong++;
  fprintf(stderr, "[rank %d] halo: %.1f us per exchange (2 x %.1f MB) + all-reduce + fill, wrong entries %lld, status %llu\n",
          rank, 1e6 * t_h, H * 8 / 1e6, (long long)wrong, (unsigned long long)s->status[rank]);

  // ---- test 3: hipStreamWaitValue64 / hipStreamWriteValue64 on the shm flags
  cpu_barrier(s, rank, world, 30);
  hipError_t e1 = hipStreamWriteValue64(st, &ds->halo_flag[(rank + 1) % world][0], 777777ull, 0);
  hipError_t e2 = hipStreamWaitValue64(st, &ds->halo_flag[rank][0], 777777ull, hipStreamWaitValueGte, ~0ull);
  hipError_t e3 = hipStreamSynchronize(st);
  fprintf(stderr, "[rank %d] stream write/wait value: %s / %s / %s\n", rank, hipGetErrorString(e1), hipGetErrorString(e2),
          hipGetErrorString(e3));
  cpu_barrier(s, rank, world, 40);
  if (peer_lo) hipIpcCloseMemHandle(peer_lo);
  if (peer_hi) hipIpcCloseMemHandle(peer_hi);
  cpu_barrier(s, rank, world, 50);
  return (bad || wrong || s->status[rank]) ? 5 : 0;
}

int main(int argc, char **argv) {
  const int world = argc > 1 ? atoi(argv[1]) : 2;
  const int ndev = argc > 2 ? atoi(argv[2]) : 1;
  char path[256];
  snprintf(path, sizeof(path), "/dev/shm/tigar_ipc_probe_%d", (int)getpid());
  int fd = open(path, O_RDWR | O_CREAT | O_TRUNC, 0600);
  if (fd < 0 || ftruncate(fd, sizeof(shm_t)) != 0) {
    perror("shm");
    return 1;
  }
  shm_t *s = (shm_t *)mmap(nullptr, sizeof(shm_t), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  memset((void *)s, 0, sizeof(shm_t));
  pid_t pids[MAXW];
  for (int r = 0; r < world; r++) {
    pids[r] = fork();
    if (pids[r] == 0) _exit(child(path, r, world, ndev));
  }
  int rc = 0;
  for (int r = 0; r < world; r++) {
    int stw = 0;
    waitpid(pids[r], &stw, 0);
    if (!WIFEXITED(stw) || WEXITSTATUS(stw) != 0) rc = 1;
  }
  unlink(path);
  printf("ipc_probe: %s\n", rc ? "FAILED" : "ok");
  return rc;
}
