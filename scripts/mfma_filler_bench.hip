// Microbenchmark (round 4): the price of one vector instruction beside v_mfma_f32_32x32x2_f32.
// The fp32 MFMA runs at the fp32 VECTOR rate on gfx950 (157.3 TFLOP/s both): do the two share execution cycles?
// Stream per wave: [1 MFMA (4 chains round-robin), N fillers of one kind] repeated; one 256- or 512-thread workgroup per
// CU.  Prints cycles per MFMA slot (64 = free fillers) and the implied cost per filler at the measured clock.
//   hipcc --offload-arch=gfx950 -O3 scripts/mfma_filler_bench.hip -o /tmp/mfb && /tmp/mfb
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define FILL_fma(r) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(r) : "v"(a), "v"(b))
#define FILL_mul(r) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(r) : "v"(a))
#define FILL_addf(r) asm volatile("v_add_f32 %0, %0, %1" : "+v"(r) : "v"(b))
#define FILL_addu(r) asm volatile("v_add_u32 %0, %0, %1" : "+v"(r) : "v"(b))
#define FILL_mov(r) asm volatile("v_mov_b32 %0, %1" : "+v"(r) : "v"(b))
#define FILL_cnd(r) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(r) : "v"(b))
#define FILL_maxi(r) asm volatile("v_max_i32 %0, %0, %1" : "+v"(r) : "v"(b))
#define FILL_med3(r) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(r) : "v"(a), "v"(b))
#define FILL_rcp(r) asm volatile("v_rcp_f32 %0, %0" : "+v"(r))
#define FILL_cvt(r) asm volatile("v_cvt_i32_f32 %0, %0" : "+v"(r))
#define FILL_pkfma(r) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(r2) : "v"(a2))
#define FILL_salu(r) asm volatile("s_add_u32 %0, %0, 1" : "+s"(sc))
#define FILL_lshl64(r) asm volatile("v_lshl_add_u64 %0, %0, 1, %1" : "+v"(r2u) : "v"(r2u))
#define FILL_mad64(r) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(r2u) : "v"(b), "v"(b) : "vcc")


template <int KIND, int N, int NW>
__global__ __launch_bounds__(NW * 64, NW / 4) void k_fill(float* out, int iters, float a, float b) {
  f32x16 acc[4];
  for (int q = 0; q < 4; ++q)
    for (int r = 0; r < 16; ++r) acc[q][r] = (float)threadIdx.x;
  float r0 = a, r1 = b, r2s = a + b, r3 = a - b;
  typedef float f2 __attribute__((ext_vector_type(2)));
  f2 r2 = {a, b}, a2 = {a, a};
  unsigned long long r2u = threadIdx.x;
  unsigned sc = 0;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[q], 0, 0, 0);
#pragma unroll
      for (int f = 0; f < N; ++f) {
#define DISPATCH(R) \
        if (KIND == 0) { FILL_fma(R); } \
        if (KIND == 1) { FILL_mul(R); } \
        if (KIND == 2) { FILL_addf(R); } \
        if (KIND == 3) { FILL_addu(R); } \
        if (KIND == 4) { FILL_mov(R); } \
        if (KIND == 5) { FILL_cnd(R); } \
        if (KIND == 6) { FILL_maxi(R); } \
        if (KIND == 7) { FILL_med3(R); } \
        if (KIND == 8) { FILL_rcp(R); } \
        if (KIND == 9) { FILL_cvt(R); } \
        if (KIND == 10) { FILL_pkfma(R); } \
        if (KIND == 11) { FILL_salu(R); } \
        if (KIND == 12) { FILL_lshl64(R); } \
        if (KIND == 13) { FILL_mad64(R); } \
        ;
        if ((f & 3) == 0) { DISPATCH(r0) } else if ((f & 3) == 1) { DISPATCH(r1) } else if ((f & 3) == 2) { DISPATCH(r2s) } else { DISPATCH(r3) }
#undef DISPATCH
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  float s = r0 + r1 + r2s + r3 + r2.x + r2.y + (float)r2u + (float)sc;
  for (int q = 0; q < 4; ++q)
    for (int r = 0; r < 16; ++r) s += acc[q][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

static const char* kKinds[] = {"fma","mul","addf","addu","mov","cnd","maxi","med3","rcp","cvt","pkfma","salu","lshl64","mad64"};

template <int KIND, int N, int NW>
double run(float* d_out) {
  const int iters = 2048;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  hipLaunchKernelGGL((k_fill<KIND, N, NW>), dim3(256), dim3(NW * 64), 0, 0, d_out, 16, 1.0f, 1e-9f);
  float best = 1e9f;
  for (int rep = 0; rep < 3; ++rep) {
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k_fill<KIND, N, NW>), dim3(256), dim3(NW * 64), 0, 0, d_out, iters, 1.0f, 1e-9f);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  // ns per MFMA slot on one SIMD: (NW / 4) waves share it
  return best * 1e6 / ((double)iters * 4 * (NW / 4));
}

template <int KIND, int NW>
void row(float* d_out, double base_ns) {
  const double t2 = run<KIND, 2, NW>(d_out), t4 = run<KIND, 4, NW>(d_out), t8 = run<KIND, 8, NW>(d_out), t12 = run<KIND, 12, NW>(d_out);
  printf("%-7s waves/SIMD=%d  ns per MFMA slot: N=2 %6.2f  N=4 %6.2f  N=8 %6.2f  N=12 %6.2f   -> per filler: %5.2f %5.2f %5.2f %5.2f ns  (MFMA alone %.2f ns)\n",
         kKinds[KIND], NW / 4, t2, t4, t8, t12, (t2 - base_ns) / 2, (t4 - base_ns) / 4, (t8 - base_ns) / 8, (t12 - base_ns) / 12, base_ns);
}

template <int NW>
void table(float* d_out) {
  const double base = run<0, 0, NW>(d_out);
  printf("waves/SIMD=%d: MFMA alone %.2f ns per slot (64 cycles -> %.3f GHz)\n", NW / 4, base, 64.0 / base);
  row<0, NW>(d_out, base);
  row<1, NW>(d_out, base);
  row<2, NW>(d_out, base);
  row<3, NW>(d_out, base);
  row<4, NW>(d_out, base);
  row<5, NW>(d_out, base);
  row<6, NW>(d_out, base);
  row<7, NW>(d_out, base);
  row<8, NW>(d_out, base);
  row<9, NW>(d_out, base);
  row<10, NW>(d_out, base);
  row<11, NW>(d_out, base);
  row<12, NW>(d_out, base);
  row<13, NW>(d_out, base);
}

int main() {
  float* d_out;
  (void)hipMalloc(&d_out, sizeof(float) * 256 * 512);
  table<4>(d_out);
  table<8>(d_out);
  return 0;
}
