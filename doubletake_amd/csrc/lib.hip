// Library-level entry points: version, error text, device count, layout helpers, exp.
#include <atomic>
#include <map>
#include <mutex>

#include "common.hpp"

namespace dt {

char* err_buf() {
  static thread_local char buf[512] = {0};
  return buf;
}

int fail(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(err_buf(), 512, fmt, ap);
  va_end(ap);
  return 1;
}

static std::atomic<long long> g_launches{0};
void note_launch() { g_launches.fetch_add(1, std::memory_order_relaxed); }


int device_cu_count() {
  static std::mutex mtx;
  static std::map<int, int> cus;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) {
    (void)hipGetLastError();
    return 256;
  }
  std::lock_guard<std::mutex> lock(mtx);
  auto it = cus.find(dev);
  if (it == cus.end()) {
    int n = 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) {
      (void)hipGetLastError();
      n = 256;
    }
    it = cus.emplace(dev, n).first;
  }
  return it->second;
}

// [n, c, h*w] -> [n, h*w, c] through a 32x32 LDS tile (+1 pad: conflict-free column reads)
__global__ __launch_bounds__(256) void transpose_last2_kernel(const float* __restrict__ src, float* __restrict__ dst,
                                                             int rows, int cols) {
  __shared__ float tile[32][33];
  const int n = blockIdx.z;
  const float* s = src + (size_t)n * rows * cols;
  float* d = dst + (size_t)n * rows * cols;
  const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
#pragma unroll
  for (int i = 0; i < 32; i += 8) {
    const int r = r0 + ty + i, c = c0 + tx;
    if (r < rows && c < cols) tile[ty + i][tx] = s[(size_t)r * cols + c];
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 32; i += 8) {
    const int c = c0 + ty + i, r = r0 + tx;
    if (r < rows && c < cols) d[(size_t)c * rows + r] = tile[tx][ty + i];
  }
}

__global__ void exp_kernel(const float* __restrict__ in, float* __restrict__ out, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    out[i] = expf(in[i]);
}

static int transpose_last2(const float* src, float* dst, int n, int rows, int cols, hipStream_t s) {
  dim3 grid((cols + 31) / 32, (rows + 31) / 32, n);
  DT_LAUNCH(transpose_last2_kernel, grid, dim3(256), 0, s, src, dst, rows, cols);
  return check_launch("transpose");
}

}  // namespace dt

using namespace dt;

extern "C" {

int dt_version(void) { return DT_ABI_VERSION; }

const char* dt_last_error(void) { return err_buf(); }

int64_t dt_kernel_launch_count(void) { return (int64_t)g_launches.load(std::memory_order_relaxed); }

int64_t dt_settings_token(void) { return (int64_t)conv_plan_objective_value() | ((int64_t)mlp_cu_budget_value() << 24); }

int dt_device_count(void) {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    fail("hipGetDeviceCount: %s", hipGetErrorString(e));
    return -1;
  }
  return n;
}

int dt_nchw_to_nhwc_f32(const float* src, float* dst, int n, int c, int h, int w, dt_stream_t s) {
  DT_REQUIRE(src && dst && n > 0 && c > 0 && h > 0 && w > 0, "dt_nchw_to_nhwc_f32: bad arguments");
  return transpose_last2(src, dst, n, c, h * w, to_stream(s));
}

int dt_nhwc_to_nchw_f32(const float* src, float* dst, int n, int c, int h, int w, dt_stream_t s) {
  DT_REQUIRE(src && dst && n > 0 && c > 0 && h > 0 && w > 0, "dt_nhwc_to_nchw_f32: bad arguments");
  return transpose_last2(src, dst, n, h * w, c, to_stream(s));
}

int dt_exp_f32(const float* in, float* out, int64_t count, dt_stream_t s) {
  DT_REQUIRE(in && out && count >= 0, "dt_exp_f32: bad arguments");
  if (count == 0) return 0;
  const int blocks = (int)((count + 255) / 256 < 2048 ? (count + 255) / 256 : 2048);
  DT_LAUNCH(exp_kernel, dim3(blocks), dim3(256), 0, to_stream(s), in, out, count);
  return check_launch("dt_exp_f32");
}

}  // extern "C"
