// Microbenchmark (round 4): what keeps a stream of v_mfma_f32_32x32x2_f32 below the pipe rate once its operands stop being
// constants?  One 512-thread workgroup per CU (two waves per SIMD, like cv_mlp_mfma_kernel), 4 accumulator chains per wave.
//   MODE 0: constant A and B                      (scripts/mfma_chain_bench.hip: ~98 %)
//   MODE 1: A fragments from LDS, one ds_read_b128 per four MFMAs, read one group ahead
//   MODE 2: MODE 1 + B operand rotates through 8 registers
//   MODE 3: MODE 2 + 4 independent v_fma per MFMA (vector fillers)
//   MODE 4: MODE 2 with a SINGLE accumulator chain (layer-2 form)
//   MODE 5: MODE 1 but the LDS address walks through 84 KB (bank / row behaviour of the real fragment table)
// Build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 scripts/mfma_stream_bench.hip -o /tmp/msb && /tmp/msb
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define FENCE() __builtin_amdgcn_sched_barrier(0)

template <int MODE, int NW>
__global__ __launch_bounds__(NW * 64, NW / 4) void stream_kernel(float* out, int groups, float a, float b) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  for (int i = threadIdx.x; i < 160 * 256 - 256; i += NW * 64) lds[i] = 1e-9f * (float)(i & 1023);
  __syncthreads();
  const int lane = threadIdx.x & 63;
  f32x16 acc[4];
  for (int q = 0; q < 4; ++q)
    for (int r = 0; r < 16; ++r) acc[q][r] = (float)threadIdx.x;
  float bv[8], fill[4] = {a, b, a, b};
  for (int j = 0; j < 8; ++j) bv[j] = b * (float)(j + 1);
  const float4* wl = reinterpret_cast<const float4*>(lds + lane * 4);
  float4 a_nxt = wl[0];
  for (int g = 0; g < groups; g += 8) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      float4 a4 = make_float4(a, a, a, a);
      if (MODE >= 1) {
        a4 = a_nxt;
        const int step = (MODE == 5) ? ((g + u + 1) % 84) : ((u + 1) & 7);
        a_nxt = wl[step * 64];
      }
      const float bb = (MODE >= 2) ? bv[u] : b;
      if (MODE == 4) {
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.x, bb, acc[0], 0, 0, 0);
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.y, bb, acc[0], 0, 0, 0);
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.z, bb, acc[0], 0, 0, 0);
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.w, bb, acc[0], 0, 0, 0);
      } else {
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.x, bb, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.y, bb, acc[1], 0, 0, 0);
        acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.z, bb, acc[2], 0, 0, 0);
        acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.w, bb, acc[3], 0, 0, 0);
      }
      if (MODE == 3) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          fill[0] = fill[0] * a + b; fill[1] = fill[1] * a + b; fill[2] = fill[2] * a + b; fill[3] = fill[3] * a + b;
        }
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      }
      FENCE();
    }
  }
  float s = fill[0] + fill[1] + fill[2] + fill[3];
  for (int q = 0; q < 4; ++q)
    for (int r = 0; r < 16; ++r) s += acc[q][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE, int NW>
void run(float* d_out, const char* what) {
  const int groups = 4096;  // x4 MFMAs per wave
  const size_t lds = 160 * 1024 - 1024;
  hipFuncSetAttribute(reinterpret_cast<const void*>(&stream_kernel<MODE, NW>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL((stream_kernel<MODE, NW>), dim3(256), dim3(NW * 64), lds, 0, d_out, 64, 1.0f, 1e-9f);
  float best = 1e9f;
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((stream_kernel<MODE, NW>), dim3(256), dim3(NW * 64), lds, 0, d_out, groups, 1.0f, 1e-9f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  const double mfmas = 256.0 * NW * groups * 4;
  const double tf = mfmas * 4096.0 / (best * 1e-3) / 1e12;
  printf("mode %d waves/SIMD=%d : %8.3f ms  %7.1f TFLOP/s (%.1f%% of 157.3)  %s\n", MODE, NW / 4, best, tf, 100.0 * tf / 157.3, what);
}

int main() {
  float* d_out;
  hipMalloc(&d_out, sizeof(float) * 256 * 512);
  run<0, 8>(d_out, "constant operands");
  run<1, 8>(d_out, "A from LDS (8 fragments)");
  run<5, 8>(d_out, "A from LDS (84 KB walk)");
  run<2, 8>(d_out, "+ rotating B");
  run<3, 8>(d_out, "+ 4 v_fma per MFMA");
  run<4, 8>(d_out, "single chain, A from LDS, rotating B");
  run<0, 4>(d_out, "constant operands");
  run<1, 4>(d_out, "A from LDS (8 fragments)");
  run<2, 4>(d_out, "+ rotating B");
  run<3, 4>(d_out, "+ 4 v_fma per MFMA");
  run<4, 4>(d_out, "single chain, A from LDS, rotating B");
  return 0;
}
