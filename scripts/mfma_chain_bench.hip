// Microbenchmark: issue rate of v_mfma_f32_32x32x2_f32 as a function of the number of independent
// accumulator chains per wave and of waves per SIMD.  Build + run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 scripts/mfma_chain_bench.hip -o /tmp/mfma_chain && /tmp/mfma_chain
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ __launch_bounds__(256) void chain_kernel(float* out, int iters, float a, float b) {
  f32x16 acc[NACC];
  for (int q = 0; q < NACC; ++q)
    for (int r = 0; r < 16; ++r) acc[q][r] = (float)threadIdx.x;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int q = 0; q < NACC; ++q) acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[q], 0, 0, 0);
  }
  float s = 0.f;
  for (int q = 0; q < NACC; ++q)
    for (int r = 0; r < 16; ++r) s += acc[q][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NACC>
void run(int blocks_per_cu, float* d_out) {
  const int iters = 8192 / NACC * 4;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int blocks = 256 * blocks_per_cu;
  hipLaunchKernelGGL(chain_kernel<NACC>, dim3(blocks), dim3(256), 0, 0, d_out, 16, 1.0f, 1e-9f);
  hipEventRecord(e0);
  hipLaunchKernelGGL(chain_kernel<NACC>, dim3(blocks), dim3(256), 0, 0, d_out, iters, 1.0f, 1e-9f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double mfmas = (double)blocks * 4 * iters * NACC;
  const double tflops = mfmas * 4096.0 / (ms * 1e-3) / 1e12;
  printf("chains/wave=%d waves/SIMD=%d : %8.3f ms  %7.1f TFLOP/s  (%.1f%% of 157.3)\n", NACC, blocks_per_cu, ms, tflops,
         100.0 * tflops / 157.3);
}

int main() {
  float* d_out;
  hipMalloc(&d_out, sizeof(float) * 256 * 8 * 256);
  for (int w = 1; w <= 4; w *= 2) {
    run<1>(w, d_out);
    run<2>(w, d_out);
    run<4>(w, d_out);
    run<8>(w, d_out);
  }
  return 0;
}
