// Microbenchmark: cost of one dependent kernel in a same-stream chain, as a function of what the kernel does.
//   hipcc --offload-arch=gfx950 -O3 scripts/launch_floor_bench.hip -o /tmp/launch_floor && /tmp/launch_floor
// Variants: empty kernel; one dependent load->store; a 160-byte by-value argument struct (the size of ConvArgs);
// a kernel with 32 KB of static LDS; 512-thread blocks; and grids of 1 / 75 / 256 / 1200 workgroups.
#include <hip/hip_runtime.h>
#include <cstdio>

struct Big { const float* p[3]; int v[34]; };

__global__ void k_empty() {}
__global__ void k_touch(const float* in, float* out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  out[i] = in[i] + 1.0f;
}
__global__ void k_big(const Big a, float* out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  out[i] = a.p[0][i] + (float)a.v[33];
}
__global__ __launch_bounds__(256) void k_lds(const float* in, float* out) {
  __shared__ float s[8192];
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  s[threadIdx.x] = in[i];
  __syncthreads();
  out[i] = s[threadIdx.x ^ 1];
}
// dependent chain of 3 loads (pointer chase through indices), like a prologue that reads params, then offsets, then data
__global__ void k_chase(const int* idx, const float* in, float* out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int j = idx[i];
  const int k = idx[j];
  out[i] = in[k];
}

template <typename F>
float chain(int n, F launch) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int i = 0; i < 20; ++i) launch();
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int i = 0; i < n; ++i) launch();
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  return ms * 1e3f / n;
}

int main() {
  float *a, *b;
  int* idx;
  const size_t N = 1200 * 512;
  hipMalloc(&a, N * 4);
  hipMalloc(&b, N * 4);
  hipMalloc(&idx, N * 4);
  hipMemset(a, 0, N * 4);
  hipMemset(idx, 0, N * 4);
  Big big{};
  big.p[0] = a;
  const int grids[] = {1, 75, 256, 1200};
  for (int g : grids) {
    printf("grid %4d x 256: empty %.2f us | load->store %.2f | 160B args %.2f | 32KB LDS %.2f | 3-deep chase %.2f | 512-thread blocks %.2f\n", g,
           chain(400, [&] { hipLaunchKernelGGL(k_empty, dim3(g), dim3(256), 0, 0); }),
           chain(400, [&] { hipLaunchKernelGGL(k_touch, dim3(g), dim3(256), 0, 0, a, b); }),
           chain(400, [&] { hipLaunchKernelGGL(k_big, dim3(g), dim3(256), 0, 0, big, b); }),
           chain(400, [&] { hipLaunchKernelGGL(k_lds, dim3(g), dim3(256), 0, 0, a, b); }),
           chain(400, [&] { hipLaunchKernelGGL(k_chase, dim3(g), dim3(256), 0, 0, idx, a, b); }),
           chain(400, [&] { hipLaunchKernelGGL(k_touch, dim3(g), dim3(512), 0, 0, a, b); }));
  }
  // alternating two different kernels (instruction cache refill between dependent launches)
  printf("alternating touch/lds, grid 256: %.2f us per kernel\n",
         chain(200, [&] { hipLaunchKernelGGL(k_touch, dim3(256), dim3(256), 0, 0, a, b);
                          hipLaunchKernelGGL(k_lds, dim3(256), dim3(256), 0, 0, a, b); }) / 2);
  // graph replay of the same chain
  hipStream_t s;
  hipStreamCreate(&s);
  hipGraph_t graph;
  hipGraphExec_t exec;
  hipStreamBeginCapture(s, hipStreamCaptureModeGlobal);
  for (int i = 0; i < 100; ++i) hipLaunchKernelGGL(k_touch, dim3(256), dim3(256), 0, s, a, b);
  hipStreamEndCapture(s, &graph);
  hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
  hipGraphLaunch(exec, s);
  hipStreamSynchronize(s);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipEventRecord(e0, s);
  for (int i = 0; i < 5; ++i) hipGraphLaunch(exec, s);
  hipEventRecord(e1, s);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  printf("graph of 100 load->store kernels, grid 256: %.2f us per kernel\n", ms * 1e3f / 500);
  return 0;
}
