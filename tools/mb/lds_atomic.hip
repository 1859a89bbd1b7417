#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
template <int MODE>
__global__ void __launch_bounds__(256) k(double *out, int iters, int ts) {
  extern __shared__ double tab[];
  for (int s = threadIdx.x; s < ts; s += 256) tab[s] = 0.0;
  __syncthreads();
  unsigned x = threadIdx.x * 2654435761u + blockIdx.x;
  double acc = 0;
  for (int i = 0; i < iters; i++) {
    x = x * 1664525u + 1013904223u;
    unsigned slot = (x >> 10) & (ts - 1);
    if (MODE == 0) unsafeAtomicAdd(&tab[slot], 1.0);
    if (MODE == 1) atomicAdd((int *)&tab[slot], 1);
    if (MODE == 2) acc += tab[slot];
    if (MODE == 3) tab[slot] = acc + i;
    if (MODE == 4) atomicAdd((unsigned long long *)&tab[slot], 1ull);
    if (MODE == 5) { float *f = (float *)tab; unsafeAtomicAdd(&f[slot], 1.0f); }
  }
  __syncthreads();
  if (threadIdx.x == 0) out[blockIdx.x] = tab[1] + acc;
}
template <int MODE> void run(const char *name, int ts) {
  double *out; hipMalloc(&out, 8 * 4096);
  int iters = 4096, blocks = 256 * 8;
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  k<MODE><<<blocks, 256, ts * 8>>>(out, 16, ts);
  hipEventRecord(a);
  k<MODE><<<blocks, 256, ts * 8>>>(out, iters, ts);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  double ops = (double)blocks * 256 * iters;
  printf("%-28s ts=%5d  %8.3f ms  %7.1f Gop/s  %.2f ops/clk/CU (2.4GHz,256CU)\n", name, ts, ms, ops / ms / 1e6, ops / (ms * 1e-3) / 2.4e9 / 256);
}
int main() {
  for (int ts : {256, 1024, 4096}) {
    run<0>("ds_add_f64", ts); run<1>("ds_add_u32", ts); run<4>("ds_add_u64", ts); run<5>("ds_add_f32", ts); run<2>("ds_read_b64", ts); run<3>("ds_write_b64", ts);
  }
  return 0;
}
