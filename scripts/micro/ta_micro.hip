// Microbenchmark: cost of one wave64 global-load instruction per CU by access pattern (gfx950).
// Build: hipcc -O3 --offload-arch=gfx950 -o ta_micro ta_micro.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

template <int PAT>
__global__ __launch_bounds__(256) void k(const double2* __restrict__ buf, size_t nelem16, int iters, double* out) {
  const int lane = threadIdx.x & 63;
  const size_t wave = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  double acc = 0.0;
  unsigned h = (unsigned)(wave * 64 + lane) * 2654435761u + 12345u;
  for (int it = 0; it < iters; ++it) {
    size_t idx;
    if (PAT == 0 || PAT == 1 || PAT == 4 || PAT == 5) {        // coalesced, a fresh KiB per wave and iteration
      idx = ((wave * 977 + (size_t)it * 131) * 64 + lane) % nelem16;
    } else if (PAT == 6) {                                        // broadcast
      idx = ((wave * 977 + (size_t)it * 131) * 64) % nelem16;
    } else if (PAT == 7) {                                        // half coalesced, half scattered
      h = h * 1664525u + 1013904223u;
      idx = (lane < 32) ? ((wave * 977 + (size_t)it * 131) * 64 + lane) % nelem16 : (size_t)(h >> 4) % nelem16;
    } else if (PAT == 8) {                                        // scattered but local: within +-4096 elements (64 KiB) of a wave base
      h = h * 1664525u + 1013904223u;
      idx = (((wave * 977 + (size_t)it * 131) * 64) + (h >> 20)) % nelem16;
    } else {                                                      // scattered
      h = h * 1664525u + 1013904223u;
      idx = (size_t)(h >> 4) % nelem16;
    }
    bool active = true;
    if (PAT == 4) active = (lane & 15) == 0;
    if (PAT == 5) active = lane == 0;
    if (active) {
      if (PAT == 1 || PAT == 3) {
        const double v = reinterpret_cast<const double*>(buf)[2 * idx];
        acc += v;
      } else {
        const double2 v = buf[idx];
        acc += v.x + v.y;
      }
    }
  }
  if (acc == 123.456) out[0] = acc;
}

template <int PAT>
double run(const double2* buf, size_t nelem16, int iters, double* out, int grid) {
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  hipLaunchKernelGGL(k<PAT>, dim3(grid), dim3(256), 0, 0, buf, nelem16, iters, out);
  hipEventRecord(a, 0);
  hipLaunchKernelGGL(k<PAT>, dim3(grid), dim3(256), 0, 0, buf, nelem16, iters, out);
  hipEventRecord(b, 0);
  hipDeviceSynchronize();
  float ms;
  hipEventElapsedTime(&ms, a, b);
  return ms;
}

int main(int argc, char** argv) {
  const size_t mb = argc > 1 ? atoi(argv[1]) : 16;
  const int iters = argc > 2 ? atoi(argv[2]) : 64;
  const size_t nelem16 = mb * 1024 * 1024 / 16;
  double2* buf;
  double* out;
  hipMalloc(&buf, nelem16 * 16);
  hipMalloc(&out, 8);
  hipMemset(buf, 0, nelem16 * 16);
  const int grid = 2048;
  const char* names[] = {"coalesced x4", "coalesced x2", "scattered x4", "scattered x2", "x4, 4 active lanes", "x4, 1 active lane",
                         "broadcast x4", "half coalesced half scattered x4", "scattered-local x4"};
  double ms[9];
  ms[0] = run<0>(buf, nelem16, iters, out, grid);
  ms[1] = run<1>(buf, nelem16, iters, out, grid);
  ms[2] = run<2>(buf, nelem16, iters, out, grid);
  ms[3] = run<3>(buf, nelem16, iters, out, grid);
  ms[4] = run<4>(buf, nelem16, iters, out, grid);
  ms[5] = run<5>(buf, nelem16, iters, out, grid);
  ms[6] = run<6>(buf, nelem16, iters, out, grid);
  ms[7] = run<7>(buf, nelem16, iters, out, grid);
  ms[8] = run<8>(buf, nelem16, iters, out, grid);
  const double ninstr_per_cu = (double)grid * 4 * iters / 256.0;
  printf("buffer %zu MiB, %d loads per wave, %d waves\n", mb, iters, grid * 4);
  for (int p = 0; p < 9; ++p)
    printf("%-36s %8.1f us  -> %7.1f ns per wave-instruction per CU (%.0f cycles at 2.4 GHz)\n", names[p], 1e3 * ms[p],
           1e6 * ms[p] / ninstr_per_cu, 2.4e6 * ms[p] / ninstr_per_cu);
  return 0;
}
