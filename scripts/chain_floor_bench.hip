// Microbenchmark behind the decision NOT to chain conv layers inside one persistent kernel:
// what does a cross-workgroup dependency cost inside a kernel, compared with the kernel boundary it would replace?
//   hipcc --offload-arch=gfx950 -O3 scripts/chain_floor_bench.hip -o /tmp/chain_floor && /tmp/chain_floor
// A "layer" = every workgroup reads a 1 KB slice written by ANOTHER workgroup in the previous layer, does a fixed amount of
// dependent ALU work, and writes its own slice.  (a) one kernel launch per layer, same stream; (b) ONE persistent launch:
// the slice travels with write-through stores (sc0 sc1), the producer drains vmcnt and publishes a per-workgroup flag with a
// write-through store, the consumer polls the flag with coherent loads and then reads the slice with coherent loads -- the
// hand-off csrc/conv.hip uses for its cross-workgroup K reduction (no cache-wide fences); (c) as (b) with one device-scope
// atomic counter per layer (grid barrier) instead of per-producer flags.  All variants are checked against each other.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void st_wt(float4* p, float4 v) {
  const f32x4 x = {v.x, v.y, v.z, v.w};
  asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(x) : "memory");
}
__device__ __forceinline__ float4 ld_co(const float4* p) {
  f32x4 x;
  asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(x) : "v"(p) : "memory");
  return make_float4(x[0], x[1], x[2], x[3]);
}
__device__ __forceinline__ unsigned ld_flag(const unsigned* p) {
  unsigned v;
  asm volatile("global_load_dword %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  return v;
}
__device__ __forceinline__ float4 work(float4 v, int iters) {
  for (int i = 0; i < iters; ++i) {  // dependent chain: iters x 4 FMAs
    v.x = v.x * 1.0001f + 0.5f; v.y = v.y * 0.9999f + v.x; v.z = v.z * 1.0002f - v.y * 1e-3f; v.w = v.w * 0.9998f + v.z * 1e-3f;
  }
  return v;
}

// (a) one launch per layer: 64 threads x float4 = 1 KB per workgroup
__global__ __launch_bounds__(64) void k_layer(const float4* in, float4* out, int iters) {
  const int src = (blockIdx.x + 37) % gridDim.x;
  out[blockIdx.x * 64 + threadIdx.x] = work(in[src * 64 + threadIdx.x], iters);
}

// (b) persistent, per-producer flags
__global__ __launch_bounds__(64) void k_chain_flags(float4* buf0, float4* buf1, unsigned* flags, int layers, int iters, unsigned epoch,
                                                   unsigned* timeout) {
  const int nb = gridDim.x, src = (blockIdx.x + 37) % nb;
  for (int l = 0; l < layers; ++l) {
    const float4* in = (l & 1) ? buf1 : buf0;
    float4* out = (l & 1) ? buf0 : buf1;
    if (l > 0) {  // wait for the producer of my slice in layer l - 1
      const unsigned want = epoch + l - 1;
      int spins = 0;
      while ((int)(ld_flag(flags + src) - want) < 0) {
        if (++spins > (1 << 22)) { *timeout = 1; return; }
        __builtin_amdgcn_s_sleep(1);
      }
    }
    const float4 v = (l > 0) ? ld_co(in + src * 64 + threadIdx.x) : in[src * 64 + threadIdx.x];
    st_wt(out + blockIdx.x * 64 + threadIdx.x, work(v, iters));
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (threadIdx.x == 0) asm volatile("global_store_dword %0, %1, off sc0 sc1" ::"v"(flags + blockIdx.x), "v"(epoch + l) : "memory");
    // (the consumer of MY previous slice may still be reading buf[l-1]: double buffering + the layer-l flag of that
    //  consumer is only needed two layers later; with a ring of two buffers we must know it finished reading)
    if (l + 1 < layers) {
      const int dst = (blockIdx.x + nb - 37) % nb;  // who reads my slice
      // wait until the reader of my layer-(l-1) slice has published layer l, before layer l+1 overwrites that buffer
      const unsigned want = epoch + l;
      int spins = 0;
      while ((int)(ld_flag(flags + dst) - want) < 0) {
        if (++spins > (1 << 22)) { *timeout = 1; return; }
        __builtin_amdgcn_s_sleep(1);
      }
    }
  }
}

// (c) persistent, one arrival counter per layer (grid barrier)
__global__ __launch_bounds__(64) void k_chain_barrier(float4* buf0, float4* buf1, unsigned* counters, int layers, int iters, unsigned* timeout) {
  const int nb = gridDim.x, src = (blockIdx.x + 37) % nb;
  for (int l = 0; l < layers; ++l) {
    const float4* in = (l & 1) ? buf1 : buf0;
    float4* out = (l & 1) ? buf0 : buf1;
    const float4 v = (l > 0) ? ld_co(in + src * 64 + threadIdx.x) : in[src * 64 + threadIdx.x];
    st_wt(out + blockIdx.x * 64 + threadIdx.x, work(v, iters));
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (threadIdx.x == 0) __hip_atomic_fetch_add(counters + l, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    int spins = 0;
    while (__hip_atomic_load(counters + l, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != (unsigned)nb) {
      if (++spins > (1 << 22)) { *timeout = 1; return; }
      __builtin_amdgcn_s_sleep(1);
    }
  }
}

int main() {
  const int layers = 64;
  for (int nb : {128, 256}) {
    for (int iters : {0, 400, 2000}) {
      float4 *b0, *b1, *r0, *r1;
      unsigned *flags, *counters, *timeout;
      const size_t n = (size_t)nb * 64;
      hipMalloc(&b0, n * 16); hipMalloc(&b1, n * 16); hipMalloc(&r0, n * 16); hipMalloc(&r1, n * 16);
      hipMalloc(&flags, nb * 4); hipMalloc(&counters, layers * 4); hipMalloc(&timeout, 4);
      std::vector<float> h(n * 4);
      for (size_t i = 0; i < n * 4; ++i) h[i] = (float)(i % 97) * 0.01f;
      hipEvent_t e0, e1;
      hipEventCreate(&e0); hipEventCreate(&e1);
      float ms_a = 0, ms_b = 0, ms_c = 0;
      std::vector<float> out_a(n * 4), out_b(n * 4), out_c(n * 4);
      unsigned to = 0;
      for (int rep = 0; rep < 3; ++rep) {
        // (a)
        hipMemcpy(r0, h.data(), n * 16, hipMemcpyHostToDevice);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        for (int l = 0; l < layers; ++l) hipLaunchKernelGGL(k_layer, dim3(nb), dim3(64), 0, 0, (l & 1) ? r1 : r0, (l & 1) ? r0 : r1, iters);
        hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms_a, e0, e1);
        hipMemcpy(out_a.data(), (layers & 1) ? r1 : r0, n * 16, hipMemcpyDeviceToHost);
        // (b)
        hipMemcpy(b0, h.data(), n * 16, hipMemcpyHostToDevice);
        hipMemset(flags, 0, nb * 4); hipMemset(timeout, 0, 4);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_chain_flags, dim3(nb), dim3(64), 0, 0, b0, b1, flags, layers, iters, 1000u * (rep + 1), timeout);
        hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms_b, e0, e1);
        hipMemcpy(out_b.data(), (layers & 1) ? b1 : b0, n * 16, hipMemcpyDeviceToHost);
        hipMemcpy(&to, timeout, 4, hipMemcpyDeviceToHost);
        // (c)
        hipMemcpy(b0, h.data(), n * 16, hipMemcpyHostToDevice);
        hipMemset(counters, 0, layers * 4);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_chain_barrier, dim3(nb), dim3(64), 0, 0, b0, b1, counters, layers, iters, timeout);
        hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms_c, e0, e1);
        hipMemcpy(out_c.data(), (layers & 1) ? b1 : b0, n * 16, hipMemcpyDeviceToHost);
      }
      size_t bad_b = 0, bad_c = 0;
      for (size_t i = 0; i < n * 4; ++i) { bad_b += out_a[i] != out_b[i]; bad_c += out_a[i] != out_c[i]; }
      printf("%3d workgroups, %4d-step ALU chain per layer: launch per layer %.2f us | persistent, per-producer flags %.2f us | persistent, "
             "atomic grid barrier %.2f us   (per layer; mismatches vs launches: %zu / %zu; timeout flag %u)\n",
             nb, iters, ms_a * 1e3f / layers, ms_b * 1e3f / layers, ms_c * 1e3f / layers, bad_b, bad_c, to);
      hipFree(b0); hipFree(b1); hipFree(r0); hipFree(r1); hipFree(flags); hipFree(counters); hipFree(timeout);
    }
  }
  return 0;
}
