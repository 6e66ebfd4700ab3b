// Shared helpers for the gfx950 kernels (error reporting, launch checks, small device math).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <cstdarg>
#include <cstdio>
#include <cstddef>
#include <tuple>
#include <type_traits>
#include <utility>

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

// Every kernel launch of the library goes through DT_LAUNCH (dt::launch) so that
//   * dt_kernel_launch_count() can report how many kernels a step needed (bench.py: launches of the conv stack; one relaxed
//     atomic add per launch), and
//   * a launch program can be recorded (program.hip: dt_program_begin .. dt_program_end): while the calling thread records
//     on the launch's stream, the kernel's address, launch geometry and a copy of its arguments -- converted to the kernel's
//     own parameter types -- are appended to the program, and the launch still executes.  dt_program_launch then re-issues the
//     recorded launches with one host call (hipLaunchKernel per node, none of the planning above it).
void note_launch();
// process-wide settings that change which kernels / grids the launchers pick -- the values in force (conv.hip: the plan
// objective mask; cv_mlp_mfma.hip: the volume kernel's CU budget, 0 = whole device).  dt_settings_token() combines them: replay
// mechanisms (hipGraphs, launch programs) bake those choices in at capture time and key their caches on the token
int conv_plan_objective_value();
int mlp_cu_budget_value();
bool recording_on(hipStream_t s);  // this thread records launches of stream s
void record_node(const void* func, dim3 grid, dim3 block, size_t shmem, int nargs, const void* const* arg_ptrs,
                 const size_t* arg_sizes, const size_t* arg_aligns, const int* arg_nptrs, const size_t* ptr_offsets);

// Where the DEVICE POINTERS of a kernel argument sit (byte offsets inside the argument): a launch program patches the pointers
// that refer to the step's input tensors, and it must know them by position -- a scan of the argument bytes for "words that look
// like an input address" also hits padding bytes of by-value structs (uninitialised: often the upper half of a pointer that
// lived in the same stack slot), which made one recording in ~40 patch a non-pointer word (round 6: intermittent GPU memory
// faults on a program's first replay).  Scalars have none, a pointer parameter has one at offset 0, and every struct that is
// passed to a kernel by value declares its own with DT_ARG_POINTERS / DT_ARG_NO_POINTERS next to its definition (the build
// fails for a struct that does not).
constexpr int kMaxArgPointers = 64;
template <typename T, typename Enable = void>
struct arg_pointers {
  static_assert(!std::is_class_v<T> && !std::is_union_v<T>,
                "a struct passed by value to a kernel must declare its pointer members: DT_ARG_POINTERS(T, offsets...) or "
                "DT_ARG_NO_POINTERS(T)");
  static int collect(size_t*) { return 0; }
};
template <typename T>
struct arg_pointers<T*, void> {
  static int collect(size_t* out) {
    out[0] = 0;
    return 1;
  }
};
#define DT_ARG_POINTERS(T, ...)                                      \
  template <>                                                        \
  struct arg_pointers<T, void> {                                     \
    static int collect(size_t* out) {                                \
      const size_t offs[] = {__VA_ARGS__};                           \
      int n = 0;                                                     \
      for (size_t o : offs) out[n++] = o;                            \
      static_assert(sizeof(offs) / sizeof(offs[0]) <= (size_t)kMaxArgPointers, "too many pointer members"); \
      return n;                                                      \
    }                                                                \
  }
#define DT_ARG_NO_POINTERS(T)                    \
  template <>                                    \
  struct arg_pointers<T, void> {                 \
    static int collect(size_t*) { return 0; }    \
  }

template <typename... P, typename... A>
inline void launch(void (*kernel)(P...), dim3 grid, dim3 block, size_t shmem, hipStream_t s, A&&... a) {
  static_assert(sizeof...(P) == sizeof...(A), "kernel launched with the wrong number of arguments");
  note_launch();
  if (recording_on(s)) {
    std::tuple<P...> vals{static_cast<P>(a)...};
    constexpr int n = (int)sizeof...(P);
    const void* ptrs[n > 0 ? n : 1];
    const size_t sizes[n > 0 ? n : 1] = {sizeof(P)...};
    const size_t aligns[n > 0 ? n : 1] = {alignof(P)...};
    int nptrs[n > 0 ? n : 1];
    size_t poffs[(n > 0 ? n : 1) * kMaxArgPointers];
    int i = 0;
    std::apply([&](const auto&... v) { ((ptrs[i++] = static_cast<const void*>(&v)), ...); }, vals);
    int j = 0;
    ((nptrs[j] = arg_pointers<std::remove_cv_t<P>>::collect(poffs + j * kMaxArgPointers), ++j), ...);
    record_node(reinterpret_cast<const void*>(kernel), grid, block, shmem, n, ptrs, sizes, aligns, nptrs, poffs);
  }
  hipLaunchKernelGGL(kernel, grid, block, shmem, s, std::forward<A>(a)...);
}
#define DT_LAUNCH(...) ::dt::launch(__VA_ARGS__)

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
