// Does a producer -> consumer hand-over through a buffer of S bytes stay on the die (256 MiB Infinity Cache) on MI355X?
// (developer micro-benchmark for the tensor-pattern PtAP: the x pass writes an intermediate the y pass reads once;
// DESIGN.md section 4a.)   hipcc -O3 --offload-arch=gfx950 tools/mb/mall_handoff.hip -o /tmp/mall && /tmp/mall
// For every S: kernel W writes the buffer (8 B per lane), kernel R reads it (sum), 20 pairs back to back on ONE buffer that
// is reused; reported: time per pair and the rate 2 S / t.  A pair that runs well above the HBM streaming rates (write
// 5.9-6.6, read 6.2-6.4 TB/s -> 3.1 TB/s for the pair) is served by the cache.  Second column: the same with the writes
// of pair k going to buffer k mod 8 of eight buffers (no reuse of addresses; total footprint 8 S).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
__global__ void __launch_bounds__(256) kw(double *p, size_t n, double v) {
  const size_t stride = (size_t)gridDim.x * 256;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) p[i] = v + (double)i;
}
__global__ void __launch_bounds__(256) kr(const double *p, size_t n, double *out) {
  const size_t stride = (size_t)gridDim.x * 256;
  double s = 0.0;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) s += p[i];
  if (s == 1.2345e300) out[0] = s;
}
int main() {
  const size_t sizes_mb[] = {16, 32, 64, 96, 128, 192, 256, 384, 512, 1024, 4096};
  double *buf[8], *out;
  const size_t maxb = 4096ull << 20;
  for (int k = 0; k < 8; k++) hipMalloc(&buf[k], maxb);
  hipMalloc(&out, 64);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  printf("%8s %14s %12s %14s %12s\n", "S (MB)", "reuse: us/pair", "TB/s", "8 bufs: us/pair", "TB/s");
  for (size_t mb : sizes_mb) {
    const size_t n = (mb << 20) / 8;
    const unsigned grid = (unsigned)((n / 256 < 256 * 32) ? n / 256 : 256 * 32);
    float ms[2];
    for (int mode = 0; mode < 2; mode++) {
      for (int w = 0; w < 3; w++) {
        kw<<<grid, 256>>>(buf[0], n, 1.0);
        kr<<<grid, 256>>>(buf[0], n, out);
      }
      hipDeviceSynchronize();
      hipEventRecord(e0);
      for (int k = 0; k < 20; k++) {
        double *b = buf[mode ? k % 8 : 0];
        kw<<<grid, 256>>>(b, n, (double)k);
        kr<<<grid, 256>>>(b, n, out);
      }
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      hipEventElapsedTime(&ms[mode], e0, e1);
    }
    const double bytes = 2.0 * (double)(mb << 20);
    printf("%8zu %14.1f %12.2f %14.1f %12.2f\n", mb, ms[0] / 20 * 1e3, bytes / (ms[0] / 20 * 1e-3) / 1e12, ms[1] / 20 * 1e3,
           bytes / (ms[1] / 20 * 1e-3) / 1e12);
  }
  return 0;
}
