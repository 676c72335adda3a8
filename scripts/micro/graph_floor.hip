// graph_floor.hip -- what a kernel node of a replayed hipGraph costs on this GPU: chains of K launches captured into a graph,
// kernels with D dependent loads (pointer chase through a small buffer), grids of 8 / 256 / 2048 workgroups of 256 threads.
// hipcc --offload-arch=gfx950 -O3 graph_floor.hip -o graph_floor && ./graph_floor
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int D>
__global__ __launch_bounds__(256) void k_chase(const int* __restrict__ next, int* __restrict__ out, int n) {
  int i = (blockIdx.x * 256 + threadIdx.x) % n;
#pragma unroll
  for (int d = 0; d < D; ++d) i = next[i];
  if (D == 0 || i == -1) out[blockIdx.x * 256 + threadIdx.x] = i;   // D > 0: never true, the loads stay
  if (D > 0 && threadIdx.x == 0 && blockIdx.x == 0) out[0] = i;
}

// producer -> consumer: every launch reads what the previous launch wrote (D dependent reads through it), then writes its own
// buffer.  shift = 0: a workgroup reads what the workgroup with the same index wrote (same XCD under round-robin dispatch);
// shift = 1: what its neighbour wrote (the next XCD).
template <int D>
__global__ __launch_bounds__(256) void k_fresh(const int* __restrict__ in, int* __restrict__ outb, int nblk, int shift) {
  const int t = threadIdx.x;
  int i = ((blockIdx.x + shift) % nblk) * 256 + t;
#pragma unroll
  for (int d = 0; d < D; ++d) i = in[i];
  outb[blockIdx.x * 256 + t] = ((blockIdx.x + shift) % nblk) * 256 + ((t * 7 + 1) & 255) + (i < 0 ? 1 : 0);
}
template <int D>
double run_fresh(hipStream_t s, int* a, int* b, int grid, int shift, int K, int reps) {
  hipGraph_t g;
  hipGraphExec_t ex;
  hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
  for (int k = 0; k < K; ++k) hipLaunchKernelGGL(k_fresh<D>, dim3(grid), dim3(256), 0, s, (k & 1) ? b : a, (k & 1) ? a : b, grid, shift);
  hipStreamEndCapture(s, &g);
  hipGraphInstantiate(&ex, g, nullptr, nullptr, 0);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int r = 0; r < 3; ++r) hipGraphLaunch(ex, s);
  hipStreamSynchronize(s);
  hipEventRecord(e0, s);
  for (int r = 0; r < reps; ++r) hipGraphLaunch(ex, s);
  hipEventRecord(e1, s);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  hipGraphExecDestroy(ex);
  hipGraphDestroy(g);
  return 1e3 * ms / (reps * K);
}

// what a kernel of the solver's kind adds to the floor: a large by-value argument block; a fixed-order block reduction of
// partial sums (two barriers) in front of a vector update
struct BigArgs {
  const int* p[24];
  int v[16];
};
__global__ __launch_bounds__(256) void k_bigargs(BigArgs a, int* __restrict__ out) {
  int i = (blockIdx.x * 256 + threadIdx.x) & 65535;
  i = a.p[3][i];
  i = a.p[17][i];
  if (i == -1) out[threadIdx.x] = i + a.v[5];
}
__global__ __launch_bounds__(256) void k_reduce_update(const double* __restrict__ parts, int nparts, const double* __restrict__ z,
                                                       double* __restrict__ p, int n) {
  __shared__ double sm[4];
  __shared__ double res;
  const int i0 = blockIdx.x * 256 + threadIdx.x;
  double z0 = 0, p0 = 0;
  if (i0 < n) {
    z0 = z[i0];
    p0 = p[i0];
  }
  double s = 0.0;
  for (int i = threadIdx.x; i < nparts; i += 256) s += parts[i];
  for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) res = sm[0] + sm[1] + sm[2] + sm[3];
  __syncthreads();
  const double beta = res * 1e-300;
  if (i0 < n) p[i0] = z0 + beta * p0;
}
template <class F>
double run_any(hipStream_t s, int K, int reps, F launch) {
  hipGraph_t g;
  hipGraphExec_t ex;
  hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
  for (int k = 0; k < K; ++k) launch(k);
  hipStreamEndCapture(s, &g);
  hipGraphInstantiate(&ex, g, nullptr, nullptr, 0);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int r = 0; r < 3; ++r) hipGraphLaunch(ex, s);
  hipStreamSynchronize(s);
  hipEventRecord(e0, s);
  for (int r = 0; r < reps; ++r) hipGraphLaunch(ex, s);
  hipEventRecord(e1, s);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  hipGraphExecDestroy(ex);
  hipGraphDestroy(g);
  return 1e3 * ms / (reps * K);
}

// a grid barrier inside one launch (all workgroups resident): arrive on a counter, spin until the generation's count is
// reached; agent-scope release / acquire around it, as a producer -> consumer phase change needs.  NB barriers per launch,
// between them one dependent read of what another workgroup wrote in the previous phase.
__global__ __launch_bounds__(256) void k_barriers(unsigned* __restrict__ counter, int* __restrict__ buf, int nb, int* __restrict__ fail) {
  const unsigned G = gridDim.x;
  int v = threadIdx.x;
  for (int b = 0; b < nb; ++b) {
    buf[blockIdx.x * 256 + threadIdx.x] = v + b;
    __threadfence();
    __syncthreads();
    if (threadIdx.x == 0) {
      const unsigned target = (unsigned)(b + 1) * G;
      __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      int spins = 0;
      while (__hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) {
        if (++spins > 2000000) {
          *fail = 1;
          break;
        }
      }
    }
    __syncthreads();
    v = __hip_atomic_load(&buf[((blockIdx.x + 1) % G) * 256 + threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  if (v == -12345) fail[1] = v;
}

template <int D>
double run(hipStream_t s, const int* next, int* out, int n, int grid, int K, int reps) {
  hipGraph_t g;
  hipGraphExec_t ex;
  hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
  for (int k = 0; k < K; ++k) hipLaunchKernelGGL(k_chase<D>, dim3(grid), dim3(256), 0, s, next, out, n);
  hipStreamEndCapture(s, &g);
  hipGraphInstantiate(&ex, g, nullptr, nullptr, 0);
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  for (int r = 0; r < 3; ++r) hipGraphLaunch(ex, s);
  hipStreamSynchronize(s);
  hipEventRecord(a, s);
  for (int r = 0; r < reps; ++r) hipGraphLaunch(ex, s);
  hipEventRecord(b, s);
  hipEventSynchronize(b);
  float ms = 0;
  hipEventElapsedTime(&ms, a, b);
  hipGraphExecDestroy(ex);
  hipGraphDestroy(g);
  return 1e3 * ms / (reps * K);
}

int main() {
  const int n = 1 << 16;
  std::vector<int> h(n);
  for (int i = 0; i < n; ++i) h[i] = (int)(((long long)i * 40503 + 12345) % n);
  int *next, *out;
  hipMalloc(&next, n * sizeof(int));
  hipMalloc(&out, 2048 * 256 * sizeof(int));
  hipMemcpy(next, h.data(), n * sizeof(int), hipMemcpyHostToDevice);
  hipStream_t s;
  hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  const int K = 64, reps = 50;
  for (int grid : {8, 256, 2048}) {
    std::printf("grid %4d: D=0 %.2f us  D=1 %.2f  D=2 %.2f  D=3 %.2f  D=4 %.2f  D=6 %.2f per node\n", grid, run<0>(s, next, out, n, grid, K, reps),
                run<1>(s, next, out, n, grid, K, reps), run<2>(s, next, out, n, grid, K, reps), run<3>(s, next, out, n, grid, K, reps),
                run<4>(s, next, out, n, grid, K, reps), run<6>(s, next, out, n, grid, K, reps));
  }
  {
    BigArgs ba;
    for (int i = 0; i < 24; ++i) ba.p[i] = next;
    for (int i = 0; i < 16; ++i) ba.v[i] = i;
    double *parts, *z, *p;
    hipMalloc(&parts, 4096 * sizeof(double));
    hipMalloc(&z, 30000 * sizeof(double));
    hipMalloc(&p, 30000 * sizeof(double));
    hipMemset(parts, 0, 4096 * sizeof(double));
    hipMemset(z, 0, 30000 * sizeof(double));
    hipMemset(p, 0, 30000 * sizeof(double));
    std::printf("256-byte argument block, two dependent loads, grid 256: %.2f us per node\n",
                run_any(s, K, reps, [&](int) { hipLaunchKernelGGL(k_bigargs, dim3(256), dim3(256), 0, s, ba, out); }));
    {   // the same kernel as plain stream launches, back to back
      hipEvent_t e0, e1;
      hipEventCreate(&e0);
      hipEventCreate(&e1);
      for (int r = 0; r < 64; ++r) hipLaunchKernelGGL(k_reduce_update, dim3(118), dim3(256), 0, s, (const double*)parts, 352, (const double*)z, p, 30000);
      hipStreamSynchronize(s);
      hipEventRecord(e0, s);
      for (int r = 0; r < 2048; ++r) hipLaunchKernelGGL(k_reduce_update, dim3(118), dim3(256), 0, s, (const double*)parts, 352, (const double*)z, p, 30000);
      hipEventRecord(e1, s);
      hipEventSynchronize(e1);
      float ms = 0;
      hipEventElapsedTime(&ms, e0, e1);
      std::printf("the same kernel as 2048 plain stream launches: %.2f us per launch\n", 1e3 * ms / 2048);
    }
    for (int kk : {4, 8, 16, 32, 64, 128})
      std::printf("graph of %3d such nodes (352 partial sums), replayed back to back: %.2f us per node\n", kk,
                  run_any(s, kk, reps * 64 / kk, [&](int) { hipLaunchKernelGGL(k_reduce_update, dim3(118), dim3(256), 0, s, (const double*)parts, 352, (const double*)z, p, 30000); }));
    for (int np : {64, 352, 2048})
      std::printf("reduction of %4d partial sums + update of 30000 values, grid 118: %.2f us per node\n", np,
                  run_any(s, K, reps, [&](int) { hipLaunchKernelGGL(k_reduce_update, dim3(118), dim3(256), 0, s, (const double*)parts, np, (const double*)z, p, 30000); }));
  }
  {
    unsigned* counter;
    int *bbuf, *fail;
    hipMalloc(&counter, 4);
    hipMalloc(&bbuf, 256 * 256 * sizeof(int));
    hipMalloc(&fail, 8);
    hipMemset(fail, 0, 8);
    for (int G : {8, 32, 64, 128, 256}) {
      for (int nb : {1, 5, 9}) {
        hipEvent_t e0, e1;
        hipEventCreate(&e0);
        hipEventCreate(&e1);
        float best = 1e9f;
        for (int rep = 0; rep < 20; ++rep) {
          hipMemsetAsync(counter, 0, 4, s);
          hipEventRecord(e0, s);
          hipLaunchKernelGGL(k_barriers, dim3(G), dim3(256), 0, s, counter, bbuf, nb, fail);
          hipEventRecord(e1, s);
          hipEventSynchronize(e1);
          float ms = 0;
          hipEventElapsedTime(&ms, e0, e1);
          best = ms < best ? ms : best;
        }
        std::printf("one launch of %3d workgroups with %d grid barriers: %.2f us%s", G, nb, 1e3 * best, nb == 9 ? "\n" : ";  ");
      }
    }
    int hf[2] = {0, 0};
    hipMemcpy(hf, fail, 8, hipMemcpyDeviceToHost);
    if (hf[0]) std::printf("  (a barrier gave up spinning)\n");
  }
  int *fa, *fb;
  hipMalloc(&fa, 2048 * 256 * sizeof(int));
  hipMalloc(&fb, 2048 * 256 * sizeof(int));
  {
    std::vector<int> init(2048 * 256);
    for (int i = 0; i < 2048 * 256; ++i) init[i] = (i / 256) * 256 + ((i * 7 + 1) & 255);
    hipMemcpy(fa, init.data(), init.size() * sizeof(int), hipMemcpyHostToDevice);
    hipMemcpy(fb, init.data(), init.size() * sizeof(int), hipMemcpyHostToDevice);
  }
  for (int grid : {8, 256}) {
    for (int shift : {0, 1})
      std::printf("fresh data, grid %4d, reads the output of workgroup + %d: D=1 %.2f us  D=2 %.2f  D=3 %.2f per node\n", grid, shift,
                  run_fresh<1>(s, fa, fb, grid, shift, K, reps), run_fresh<2>(s, fa, fb, grid, shift, K, reps),
                  run_fresh<3>(s, fa, fb, grid, shift, K, reps));
  }
  return 0;
}
