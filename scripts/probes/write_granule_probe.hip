// Probe (round 6): what does WRITE_SIZE count for partial-line stores into an NHWC volume on gfx950?
// Four kernels write the same 4.9 MB buffer (19200 pixels x 64 planes x 4 B, pixel stride 256 B):
//   full  : every lane one float4, consecutive lanes consecutive addresses (whole 128-B lines per instruction)
//   c64   : per store instruction a lane pair covers 64 contiguous bytes of a pixel (16 planes), pixels 256 B apart
//   c32   : a lane pair covers 32 contiguous bytes (8 planes) -- the shipped volume kernel's staged store
//   c4    : a lane writes 4 bytes per pixel (1 plane) -- the unstaged NHWC store
// Each chunk kernel writes the chunks of a pixel in separate instructions, like the plane loop does.
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/wgp scripts/probes/write_granule_probe.hip
//   rocprofv3 --kernel-trace --pmc WRITE_SIZE -d out -o pmc --output-format csv -- /tmp/wgp
#include <hip/hip_runtime.h>
#include <cstdio>

constexpr int kPix = 19200, kD = 64;

__global__ void full(float* v) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < (size_t)kPix * kD / 4) reinterpret_cast<float4*>(v)[i] = make_float4(1.f, 2.f, 3.f, 4.f);
}
__global__ void c64(float* v) {
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, ls = threadIdx.x & 63;
  const int px = wave * 32 + (ls >> 1), q = (ls & 1) * 8;
  if (px >= kPix) return;
  for (int c = 0; c < 4; ++c) {
    float* dst = v + (size_t)px * kD + c * 16 + q;
    reinterpret_cast<float4*>(dst)[0] = make_float4(1.f, 2.f, 3.f, 4.f);
    reinterpret_cast<float4*>(dst)[1] = make_float4(5.f, 6.f, 7.f, 8.f);
    __builtin_amdgcn_s_sleep(64);
  }
}
__global__ void c32(float* v) {
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, ls = threadIdx.x & 63;
  const int px = wave * 32 + (ls >> 1), q = (ls & 1) * 4;
  if (px >= kPix) return;
  for (int c = 0; c < 8; ++c) {
    float* dst = v + (size_t)px * kD + c * 8 + q;
    reinterpret_cast<float4*>(dst)[0] = make_float4(1.f, 2.f, 3.f, 4.f);
    __builtin_amdgcn_s_sleep(64);
  }
}
__global__ void c4(float* v) {
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, ls = threadIdx.x & 63;
  const int px = wave * 32 + (ls & 31);
  if (px >= kPix || ls >= 32) return;
  for (int c = 0; c < 64; ++c) {
    v[(size_t)px * kD + c] = 1.f;
    __builtin_amdgcn_s_sleep(8);
  }
}

int main() {
  float* v;
  const size_t bytes = (size_t)kPix * kD * 4;
  if (hipMalloc(&v, bytes) != hipSuccess) return 1;
  const int waves = (kPix + 31) / 32, blocks = (waves * 64 + 255) / 256;
  for (int rep = 0; rep < 10; ++rep) {
    hipLaunchKernelGGL(full, dim3((kPix * kD / 4 + 255) / 256), dim3(256), 0, 0, v);
    hipLaunchKernelGGL(c64, dim3(blocks), dim3(256), 0, 0, v);
    hipLaunchKernelGGL(c32, dim3(blocks), dim3(256), 0, 0, v);
    hipLaunchKernelGGL(c4, dim3(blocks), dim3(256), 0, 0, v);
  }
  if (hipDeviceSynchronize() != hipSuccess) return 2;
  printf("wrote %zu bytes per kernel\n", bytes);
  return 0;
}
