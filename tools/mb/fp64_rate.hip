// fp64 issue rates on MI355X (developer micro-benchmark for csrc/tg_assemble.hip: would the element matrices gain from the
// matrix cores?).   hipcc -O3 --offload-arch=gfx950 tools/mb/fp64_rate.hip -o /tmp/fp64_rate && /tmp/fp64_rate
//   mode 0: v_fma_f64 only (16 independent chains per lane)
//   mode 1: v_mfma_f64_16x16x4_f64 only (4 independent accumulator tiles per wave)
//   mode 2: both interleaved (one MFMA per 4 FMAs)
// waves per SIMD: 1, 2, 4.  Reported: TFLOP/s counting 2 flops per multiply-add (FMA: 64 per instruction, MFMA: 1024).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef double v4d __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ void __launch_bounds__(256) k(double *out, int iters, double seed) {
  double a[16];
  v4d c[4];
  for (int i = 0; i < 16; i++) a[i] = seed + threadIdx.x * 1e-3 + i;
  for (int i = 0; i < 4; i++) c[i] = v4d{seed, seed + 1, seed + 2, seed + 3};
  const double x = 1.0 + seed * 1e-9, y = seed * 1e-12;
  for (int it = 0; it < iters; it++) {
    if (MODE == 0 || MODE == 2) {
#pragma unroll
      for (int i = 0; i < 16; i++) a[i] = fma(a[i], x, y);
    }
    if (MODE == 1 || MODE == 2) {
#pragma unroll
      for (int i = 0; i < 4; i++) c[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y + i, c[i], 0, 0, 0);
    }
  }
  double s = 0.0;
  for (int i = 0; i < 16; i++) s += a[i];
  for (int i = 0; i < 4; i++) s += c[i].x + c[i].y + c[i].z + c[i].w;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int MODE>
static void run(int wps, int iters, double *d) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int blocks = 256 * wps;               // 256 CUs x wps blocks of 4 waves
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, 10, 1.0);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, iters, 1.0);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double waves = blocks * 4.0;
  const double fma = (MODE == 1 ? 0.0 : 16.0 * 64) * iters * waves, mfma = (MODE == 0 ? 0.0 : 4.0 * 1024) * iters * waves;
  printf("mode %d  waves/SIMD %d: %8.3f ms   VALU %6.1f TFLOP/s   MFMA %6.1f TFLOP/s   sum %6.1f\n", MODE, wps, ms, 2 * fma / ms / 1e9,
         2 * mfma / ms / 1e9, 2 * (fma + mfma) / ms / 1e9);
}
int main() {
  double *d;
  hipMalloc(&d, 256 * 8 * 256 * sizeof(double));
  for (int wps = 1; wps <= 4; wps *= 2) {
    run<0>(wps, 20000, d);
    run<1>(wps, 20000, d);
    run<2>(wps, 20000, d);
  }
  return 0;
}
