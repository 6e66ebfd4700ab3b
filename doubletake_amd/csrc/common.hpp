// Shared helpers for the gfx950 kernels (error reporting, launch checks, small device math).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <cstdarg>
#include <cstdio>

#include "../../include/doubletake_hip.h"

namespace dt {

// thread-local error text returned by dt_last_error()
char* err_buf();
int fail(const char* fmt, ...);

inline int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail("%s: %s", what, hipGetErrorString(e));
  return 0;
}

inline hipStream_t to_stream(dt_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

// compute units of HIP's CURRENT device (cached per device id: one process may drive several GPUs); 256 if unknown
int device_cu_count();

// Every kernel launch of the library goes through DT_LAUNCH so that dt_kernel_launch_count() can report how many
// kernels a step needed (bench.py: launches of the conv stack; cheap: one relaxed atomic add per launch).
void note_launch();
#define DT_LAUNCH(...)                \
  do {                                \
    ::dt::note_launch();              \
    hipLaunchKernelGGL(__VA_ARGS__);  \
  } while (0)

#define DT_REQUIRE(cond, ...)            \
  do {                                   \
    if (!(cond)) return dt::fail(__VA_ARGS__); \
  } while (0)

constexpr int kWave = 64;

__device__ __forceinline__ float lrelu(float v, float slope) { return v >= 0.f ? v : v * slope; }

// layout of the per-batch-element cost-volume parameter block (see doubletake_hip.h)
constexpr int kCvInvK = 0;
constexpr int kCvPlanes = 12;
constexpr int kCvViewFloats = 20;
__host__ __device__ inline int cv_params_floats(int D, int K) { return kCvPlanes + D + kCvViewFloats * K; }
__host__ __device__ inline int cv_view_off(int D, int k) { return kCvPlanes + D + kCvViewFloats * k; }

}  // namespace dt
