// Layout + rate check of v_mfma_f32_32x32x16_f16 on gfx950 (used by the split-precision MLP volume kernel):
//   A operand of lane l: A[l & 31][8 * (l >> 5) + e], e = 0..7;  B operand: B[8 * (l >> 5) + e][l & 31];
//   D register r of lane l: D[(r & 3) + 8 * (r >> 2) + 4 * (l >> 5)][l & 31].
//   hipcc --offload-arch=gfx950 -O3 scripts/mfma_f16_layout_check.hip -o /tmp/mfma16 && /tmp/mfma16
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ void one(const _Float16* A, const _Float16* B, float* D) {
  const int l = threadIdx.x;
  half8 a, b;
  for (int e = 0; e < 8; ++e) {
    a[e] = A[(l & 31) * 16 + 8 * (l >> 5) + e];
    b[e] = B[(8 * (l >> 5) + e) * 32 + (l & 31)];
  }
  f32x16 c;
  for (int r = 0; r < 16; ++r) c[r] = 0.f;
  c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
  for (int r = 0; r < 16; ++r) D[((r & 3) + 8 * (r >> 2) + 4 * (l >> 5)) * 32 + (l & 31)] = c[r];
}

template <int NACC>
__global__ __launch_bounds__(256) void rate(float* out, int iters) {
  half8 a, b;
  for (int e = 0; e < 8; ++e) { a[e] = (_Float16)(threadIdx.x * 0.001f + e); b[e] = (_Float16)(1.0f / (e + 1)); }
  f32x16 acc[NACC];
  for (int q = 0; q < NACC; ++q) for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
  for (int i = 0; i < iters; ++i)
#pragma unroll
    for (int q = 0; q < NACC; ++q) acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[q], 0, 0, 0);
  float s = 0.f;
  for (int q = 0; q < NACC; ++q) for (int r = 0; r < 16; ++r) s += acc[q][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main() {
  std::vector<_Float16> A(32 * 16), B(16 * 32);
  for (int i = 0; i < 512; ++i) { A[i] = (_Float16)(std::sin(0.37 * i) ); B[i] = (_Float16)(std::cos(0.11 * i + 1)); }
  _Float16 *dA, *dB; float* dD;
  hipMalloc(&dA, 1024); hipMalloc(&dB, 1024); hipMalloc(&dD, 4096);
  hipMemcpy(dA, A.data(), 1024, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), 1024, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(one, dim3(1), dim3(64), 0, 0, dA, dB, dD);
  std::vector<float> D(1024);
  hipMemcpy(D.data(), dD, 4096, hipMemcpyDeviceToHost);
  double worst = 0;
  for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) {
    double ref = 0; for (int k = 0; k < 16; ++k) ref += (double)(float)A[i * 16 + k] * (double)(float)B[k * 32 + j];
    worst = std::fmax(worst, std::fabs(ref - D[i * 32 + j]));
  }
  printf("layout check: max |D - A@B| = %.3e (%s)\n", worst, worst < 1e-4 ? "OK" : "MISMATCH");
  float* out; hipMalloc(&out, 4 * 256 * 256 * 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 20000;
  auto run = [&](auto kern, int nacc, int wpc) {
    hipLaunchKernelGGL(kern, dim3(256 * wpc), dim3(256), 0, 0, out, 100);
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(256 * wpc), dim3(256), 0, 0, out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double mf = (double)256 * wpc * 4 * iters * nacc;
    printf("chains=%d waves/SIMD=%d: %.3f ms, %.1f TFLOP/s, %.1f cycles per MFMA per SIMD (at 2.4 GHz)\n", nacc, wpc, ms,
           mf * 2 * 32 * 32 * 16 / (ms * 1e-3) / 1e12, ms * 1e-3 * 2.4e9 / ((double)iters * nacc * wpc));
  };
  run(rate<1>, 1, 1); run(rate<2>, 2, 1); run(rate<4>, 4, 1); run(rate<1>, 1, 2); run(rate<4>, 4, 2);
  return 0;
}
