// Cost of a device-wide barrier in a persistent kernel on MI355X (developer micro-benchmark; see csrc/tg_krylov_small.hip).
//   hipcc -O3 --offload-arch=gfx950 tools/mb/grid_barrier.hip -o /tmp/grid_barrier && /tmp/grid_barrier
// Variants: levels (1 = one counter, 2 = groups of 16 + top), fences (0 = none, 1 = release before / acquire after by thread 0,
// 2 = the same in every wave, 3 = no cache maintenance at all: the payload goes through agent-scope relaxed atomic stores and
// loads, every wave waits for its stores before it arrives), payload (each workgroup writes 2 KB before the barrier and reads 2 KB of a neighbour after it).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
struct ctrl_t {
  unsigned gen, pad0[31];
  unsigned top, pad1[31];
  unsigned grp[32 * 32];
};
template <int LEVELS, int FENCES>
__device__ __forceinline__ void barrier(ctrl_t *c, unsigned G, unsigned &gen) {
  if (FENCES == 2) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
  if (FENCES == 3) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's write-through stores have arrived
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned target = gen + 1;
    if (FENCES == 1) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    bool last = false;
    if (LEVELS == 1) {
      if (__hip_atomic_fetch_add(&c->top, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == G - 1) last = true;
    } else {
      const unsigned grp = blockIdx.x / 16, ngrp = (G + 15) / 16, gsz = grp + 1 < ngrp ? 16 : G - grp * 16;
      if (__hip_atomic_fetch_add(&c->grp[32 * grp], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gsz - 1) {
        __hip_atomic_store(&c->grp[32 * grp], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (__hip_atomic_fetch_add(&c->top, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == ngrp - 1) last = true;
      }
    }
    if (last) {
      __hip_atomic_store(&c->top, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (FENCES == 1 || FENCES == 2) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      __hip_atomic_store(&c->gen, target, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      while (__hip_atomic_load(&c->gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != target) __builtin_amdgcn_s_sleep(1);
    }
    if (FENCES == 1) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  }
  gen++;
  __syncthreads();
  if (FENCES == 2) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
}
template <int LEVELS, int FENCES, int PAYLOAD>
__global__ void __launch_bounds__(512) k(ctrl_t *c, double *buf, int reps, double *out) {
  unsigned gen = 0;
  const unsigned G = gridDim.x;
  double acc = 0.0;
  for (int r = 0; r < reps; r++) {
    // two buffers used alternately: a fast workgroup that is one barrier ahead writes the OTHER buffer
    double *b = buf + (size_t)(r & 1) * G * 256;
    if (PAYLOAD && threadIdx.x < 256) {
      const double val = (double)(r + blockIdx.x);
      if (FENCES == 3)
        __hip_atomic_store(&b[(size_t)blockIdx.x * 256 + threadIdx.x], val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      else
        b[(size_t)blockIdx.x * 256 + threadIdx.x] = val;
    }
    barrier<LEVELS, FENCES>(c, G, gen);
    if (PAYLOAD && threadIdx.x < 256) {
      const size_t at = (size_t)((blockIdx.x + 37) % G) * 256 + threadIdx.x;
      acc += FENCES == 3 ? __hip_atomic_load(&b[at], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : b[at];
    }
  }
  if (PAYLOAD && threadIdx.x < 256) out[(size_t)blockIdx.x * 256 + threadIdx.x] = acc;
}
template <int LEVELS, int FENCES, int PAYLOAD>
static void run(const char *name, int G, int reps, ctrl_t *c, double *buf, double *out) {
  hipMemset(c, 0, sizeof(ctrl_t));
  void *params[] = {&c, &buf, &reps, &out};
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  for (int w = 0; w < 2; w++) {
    hipMemset(c, 0, sizeof(ctrl_t));
    hipEventRecord(a, 0);
    hipError_t e = hipLaunchCooperativeKernel((const void *)k<LEVELS, FENCES, PAYLOAD>, dim3(G), dim3(512), params, 0, 0);
    hipEventRecord(b, 0);
    hipDeviceSynchronize();
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    if (w == 1) {
      bool ok = true;
      if (PAYLOAD) {
        std::vector<double> h((size_t)G * 256);
        hipMemcpy(h.data(), out, h.size() * 8, hipMemcpyDeviceToHost);
        for (int blk = 0; blk < G && ok; blk++) {
          double want = 0.0;
          for (int r = 0; r < reps; r++) want += (double)(r + (blk + 37) % G);
          ok = h[(size_t)blk * 256 + 5] == want;
        }
      }
      printf("%-52s %7.2f us per barrier   %s %s\n", name, 1e3 * ms / reps, e == hipSuccess ? "" : hipGetErrorString(e),
             PAYLOAD ? (ok ? "payload seen" : "PAYLOAD STALE") : "");
    }
  }
}
int main() {
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, 0);
  const int G = p.multiProcessorCount, reps = 20000;
  ctrl_t *c;
  double *buf, *out;
  hipMalloc(&c, sizeof(ctrl_t));
  hipMalloc(&buf, (size_t)2 * G * 256 * 8);
  hipMalloc(&out, (size_t)G * 256 * 8);
  printf("%d workgroups of 512 threads, %d barriers\n", G, reps);
  run<1, 0, 0>("one counter, no fences", G, reps, c, buf, out);
  run<2, 0, 0>("two levels, no fences", G, reps, c, buf, out);
  run<2, 1, 0>("two levels, fences by thread 0", G, reps, c, buf, out);
  run<2, 2, 0>("two levels, fences in every wave", G, reps, c, buf, out);
  run<2, 0, 1>("two levels, no fences, 2 KB written / read", G, reps, c, buf, out);
  run<2, 1, 1>("two levels, fences by thread 0, 2 KB written / read", G, reps, c, buf, out);
  run<1, 1, 1>("one counter, fences by thread 0, 2 KB written / read", G, reps, c, buf, out);
  run<2, 3, 1>("two levels, NO cache maintenance: agent-scope stores / loads + s_waitcnt", G, reps, c, buf, out);
  // fine-grained (coherent) allocation for the payload and the control block: no cache maintenance needed
  ctrl_t *c2;
  double *buf2;
  if (hipExtMallocWithFlags((void **)&c2, sizeof(ctrl_t), hipDeviceMallocFinegrained) == hipSuccess &&
      hipExtMallocWithFlags((void **)&buf2, (size_t)2 * G * 256 * 8, hipDeviceMallocFinegrained) == hipSuccess) {
    run<2, 0, 1>("fine-grained memory: two levels, no fences, payload", G, reps, c2, buf2, out);
    run<2, 1, 1>("fine-grained memory: two levels, fences, payload", G, reps, c2, buf2, out);
  } else
    printf("fine-grained allocation not available\n");
  return 0;
}
