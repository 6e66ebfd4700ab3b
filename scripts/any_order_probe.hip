// Does hipExtAnyOrderLaunch let two consecutive kernels of ONE stream overlap on this runtime / gfx950?
// Two single-workgroup spin kernels of ~100 us each: back to back they take ~200 us, overlapped ~100 us.
//   hipcc --offload-arch=gfx950 -O2 scripts/any_order_probe.hip -o /tmp/any_order_probe && /tmp/any_order_probe
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>

__global__ void spin(unsigned long long ticks, int* out) {
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) {}
  if (out) out[0] = 1;
}

static float run(hipStream_t s, int flags, int n) {
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  hipEventRecord(a, s);
  for (int i = 0; i < n; ++i) {
    if (flags < 0)
      hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s, 10000ull, (int*)nullptr);
    else
      hipExtLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s, nullptr, nullptr, (i == 0 || i == n - 1) ? 0 : flags, 10000ull, (int*)nullptr);
  }
  hipEventRecord(b, s);
  hipEventSynchronize(b);
  float ms = 0;
  hipEventElapsedTime(&ms, a, b);
  return ms * 1e3f;
}

int main() {
  hipStream_t s;
  hipStreamCreate(&s);
  run(s, -1, 2);
  printf("6 spin kernels of 100 us, plain launches:            %.1f us\n", run(s, -1, 6));
  printf("6 spin kernels, hipExtLaunchKernelGGL flags=0:        %.1f us\n", run(s, 0, 6));
  printf("6 spin kernels, middle four hipExtAnyOrderLaunch:     %.1f us\n", run(s, hipExtAnyOrderLaunch, 6));
  return 0;
}
