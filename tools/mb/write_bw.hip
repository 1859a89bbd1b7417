// developer micro-benchmark: ceiling of large pure-write streams (what bounds the FE input generator and the
// output halves of the PtAP passes).  hipcc --offload-arch=gfx950 -O3 write_bw.hip -o write_bw.exe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
typedef double d2 __attribute__((ext_vector_type(2)));
typedef double d4 __attribute__((ext_vector_type(4)));

template <int W, bool NT>
__global__ void __launch_bounds__(256) k_write(double *__restrict__ o, int64_t n_vec, double v) {
  // grid-stride over vectors of W doubles
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < n_vec; i += stride) {
    if (W == 1) {
      if (NT) __builtin_nontemporal_store(v, o + i); else o[i] = v;
    } else if (W == 2) {
      d2 t = {v, v + 1};
      if (NT) __builtin_nontemporal_store(t, (d2 *)o + i); else ((d2 *)o)[i] = t;
    } else {
      d4 t = {v, v + 1, v + 2, v + 3};
      if (NT) __builtin_nontemporal_store(t, (d4 *)o + i); else ((d4 *)o)[i] = t;
    }
  }
}

// one workgroup writes a contiguous chunk (like the pencil stream), thread <-> 16 B
template <bool NT>
__global__ void __launch_bounds__(256) k_write_chunk(double *__restrict__ o, int64_t chunk_vec, double v) {
  d2 *p = (d2 *)o + (int64_t)blockIdx.x * chunk_vec;
  for (int64_t i = threadIdx.x; i < chunk_vec; i += 256) {
    d2 t = {v, v + 1};
    if (NT) __builtin_nontemporal_store(t, p + i); else p[i] = t;
  }
}

// CSR-like: 8 bytes to one stream and 4 bytes to another per lane
__global__ void __launch_bounds__(256) k_write_csr(double *__restrict__ v, int *__restrict__ c, int64_t n, double x) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    v[i] = x;
    c[i] = (int)i;
  }
}
// the same with a contiguous chunk per workgroup and one row of `rowlen` entries per wave at a time
__global__ void __launch_bounds__(256) k_write_csr_rows(double *__restrict__ v, int *__restrict__ c, int64_t chunk,
                                                        int rowlen, double x) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t b0 = (int64_t)blockIdx.x * chunk;
  for (int64_t r = wave; r * rowlen < chunk; r += 4) {
    double *vr = v + b0 + r * rowlen;
    int *cr = c + b0 + r * rowlen;
    for (int t = lane; t < rowlen; t += 64) {
      vr[t] = x;
      cr[t] = t;
    }
  }
}

// mixed streams: R reads of 16 B per lane for every write of 16 B (R = 0: pure write handled above; W = 0: pure read)
template <int R, int W>
__global__ void __launch_bounds__(256) k_mix(const d2 *__restrict__ in, d2 *__restrict__ out, int64_t n_vec, double *sink) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  d2 acc = {0.0, 0.0};
  for (; i < n_vec; i += stride) {
    d2 t = {1.0, 2.0};
#pragma unroll
    for (int r = 0; r < R; r++) t += in[i + (int64_t)r * n_vec];
    if (W) out[i] = t;
    else acc += t;
  }
  if (!W && acc.x == 12345.678) *sink = acc.y;
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

int main(int argc, char **argv) {
  const double gb = argc > 1 ? atof(argv[1]) : 32.0;
  if (argc > 2) {
    // placement probe: several buffers of `gb` allocated one after the other, the same write stream on each, twice
    const int nb = atoi(argv[2]);
    const int64_t n = (int64_t)(gb * 1e9 / 8) / 1024 * 1024;
    double *buf[16];
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int b = 0; b < nb && b < 16; b++) CK(hipMalloc(&buf[b], n * 8));
    for (int rep = 0; rep < 2; rep++)
      for (int b = 0; b < nb && b < 16; b++) {
        hipLaunchKernelGGL((k_write<1, false>), dim3(8192), dim3(256), 0, 0, buf[b], n, 1.0);
        hipDeviceSynchronize();
        hipEventRecord(e0, 0);
        for (int r = 0; r < 5; r++) hipLaunchKernelGGL((k_write<1, false>), dim3(8192), dim3(256), 0, 0, buf[b], n, 1.0);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        printf("buffer %d at %p: write %.2f TB/s", b, (void *)buf[b], 5 * n * 8.0 / ms / 1e9);
        // the same buffer as a read stream
        hipLaunchKernelGGL((k_mix<1, 0>), dim3(8192), dim3(256), 0, 0, (const d2 *)buf[b], (d2 *)nullptr, n / 2, buf[b]);
        hipDeviceSynchronize();
        hipEventRecord(e0, 0);
        for (int r = 0; r < 5; r++)
          hipLaunchKernelGGL((k_mix<1, 0>), dim3(8192), dim3(256), 0, 0, (const d2 *)buf[b], (d2 *)nullptr, n / 2, buf[b]);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        printf("   read %.2f TB/s\n", 5 * n * 8.0 / ms / 1e9);
      }
    return 0;
  }
  const int64_t n = (int64_t)(gb * 1e9 / 8) / 1024 * 1024;
  double *o = nullptr;
  CK(hipMalloc(&o, n * 8));
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipStream_t st;
  hipStreamCreate(&st);
  double launch_scale = 1.0;
  auto timeit = [&](const char *name, auto launch) {
    launch();
    hipStreamSynchronize(st);
    hipEventRecord(e0, st);
    for (int r = 0; r < 5; r++) launch();
    hipEventRecord(e1, st);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    ms /= 5;
    printf("%-52s %8.3f ms  %6.2f TB/s\n", name, ms, launch_scale * n * 8.0 / ms / 1e9);
  };
  printf("pure writes of %.1f GB\n", n * 8.0 / 1e9);
  timeit("hipMemsetAsync", [&] { hipMemsetAsync(o, 0, n * 8, st); });
  for (int g : {256 * 8, 256 * 32, 256 * 128}) {
    char nm[96];
    snprintf(nm, 96, "8 B/lane grid %d", g);
    timeit(nm, [&] { hipLaunchKernelGGL((k_write<1, false>), dim3(g), dim3(256), 0, st, o, n, 1.0); });
    snprintf(nm, 96, "16 B/lane grid %d", g);
    timeit(nm, [&] { hipLaunchKernelGGL((k_write<2, false>), dim3(g), dim3(256), 0, st, o, n / 2, 1.0); });
    snprintf(nm, 96, "16 B/lane nt grid %d", g);
    timeit(nm, [&] { hipLaunchKernelGGL((k_write<2, true>), dim3(g), dim3(256), 0, st, o, n / 2, 1.0); });
    snprintf(nm, 96, "32 B/lane grid %d", g);
    timeit(nm, [&] { hipLaunchKernelGGL((k_write<4, false>), dim3(g), dim3(256), 0, st, o, n / 4, 1.0); });
    snprintf(nm, 96, "32 B/lane nt grid %d", g);
    timeit(nm, [&] { hipLaunchKernelGGL((k_write<4, true>), dim3(g), dim3(256), 0, st, o, n / 4, 1.0); });
  }
  for (int64_t chunk : {4096, 65536, 1048576}) {   // 16-B vectors per workgroup: 64 KB, 1 MB, 16 MB
    char nm[96];
    const int64_t nwg = (n / 2) / chunk;
    snprintf(nm, 96, "chunk per workgroup %lld KB", (long long)(chunk * 16 / 1024));
    timeit(nm, [&] { hipLaunchKernelGGL((k_write_chunk<false>), dim3((unsigned)nwg), dim3(256), 0, st, o, chunk, 1.0); });
    snprintf(nm, 96, "chunk per workgroup %lld KB nt", (long long)(chunk * 16 / 1024));
    timeit(nm, [&] { hipLaunchKernelGGL((k_write_chunk<true>), dim3((unsigned)nwg), dim3(256), 0, st, o, chunk, 1.0); });
  }
  {
    // 2/3 of the bytes as doubles, 1/3 as ints: n12 entries of 12 bytes
    const int64_t ne = n * 8 / 12 / 4096 * 4096;
    double *v = o;
    int *c = (int *)(o + ne);
    for (int g : {256 * 8, 256 * 32, 256 * 128}) {
      char nm[96];
      snprintf(nm, 96, "csr-like 8+4 B/lane grid %d (x 12/8 bytes)", g);
      launch_scale = 12.0 * ne / (8.0 * n);
      timeit(nm, [&] { hipLaunchKernelGGL(k_write_csr, dim3(g), dim3(256), 0, st, v, c, ne, 1.0); });
    }
    for (int rowlen : {2401, 4096}) {
      const int64_t chunk = (int64_t)rowlen * 768;
      char nm[96];
      snprintf(nm, 96, "csr-like rows of %d per wave, chunk %lld", rowlen, (long long)chunk);
      timeit(nm, [&] { hipLaunchKernelGGL(k_write_csr_rows, dim3((unsigned)(ne / chunk)), dim3(256), 0, st, v, c, chunk, rowlen, 1.0); });
    }
    launch_scale = 1.0;
  }
  {
    // mixed traffic on the same buffer: the first R quarters are read, the last quarter written
    const int64_t nq = n / 2 / 4;          // 16-byte vectors per quarter
    d2 *in = (d2 *)o, *out = (d2 *)o + 3 * nq;
    double *sink = o;
    const int g = 256 * 64;
    auto run = [&](const char *nm, double bytes, auto launch) {
      launch();
      hipStreamSynchronize(st);
      hipEventRecord(e0, st);
      for (int r = 0; r < 5; r++) launch();
      hipEventRecord(e1, st);
      hipEventSynchronize(e1);
      float ms = 0;
      hipEventElapsedTime(&ms, e0, e1);
      printf("%-52s %8.3f ms  %6.2f TB/s\n", nm, ms / 5, bytes / (ms / 5) / 1e9);
    };
    run("pure read (3 streams)", 3.0 * nq * 16, [&] { hipLaunchKernelGGL((k_mix<3, 0>), dim3(g), dim3(256), 0, st, in, out, nq, sink); });
    run("copy (1 read : 1 write)", 2.0 * nq * 16, [&] { hipLaunchKernelGGL((k_mix<1, 1>), dim3(g), dim3(256), 0, st, in, out, nq, sink); });
    run("2 reads : 1 write", 3.0 * nq * 16, [&] { hipLaunchKernelGGL((k_mix<2, 1>), dim3(g), dim3(256), 0, st, in, out, nq, sink); });
    run("3 reads : 1 write", 4.0 * nq * 16, [&] { hipLaunchKernelGGL((k_mix<3, 1>), dim3(g), dim3(256), 0, st, in, out, nq, sink); });
  }
  hipFree(o);
  return 0;
}
