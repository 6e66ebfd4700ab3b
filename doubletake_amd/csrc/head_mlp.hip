// Fused regression head of the DoubleTake-small decoder: per pixel Cin -> 128 -> 128 -> 1 with ELU,
// i.e. the three 1x1 convolutions of SkipDecoderRegression.out{1..4}
// (reference modules/networks_fast.py:102-132,134-141) in ONE kernel, gfx950 fp32 MFMA.
//
// Unfused, the 240x320 head costs 37 + 67 + 7 us and moves 157 MB (two 39 MB intermediates written and
// re-read); fused it reads 19.7 MB and writes 0.3 MB.  Same chained-MFMA layout as cv_mlp_mfma.hip:
// i = output feature (A = weights in LDS), j = pixel (B = registers); layer A's accumulators are,
// after bias + ELU, the B operands of layer B for the same lane.
#include "common.hpp"
#include "head_device.hpp"

namespace dt {

#define DT_HMFMA4(ACC, A4, BVAL)                                                    \
  do {                                                                              \
    ACC[0] = __builtin_amdgcn_mfma_f32_32x32x2f32((A4).x, (BVAL), ACC[0], 0, 0, 0); \
    ACC[1] = __builtin_amdgcn_mfma_f32_32x32x2f32((A4).y, (BVAL), ACC[1], 0, 0, 0); \
    ACC[2] = __builtin_amdgcn_mfma_f32_32x32x2f32((A4).z, (BVAL), ACC[2], 0, 0, 0); \
    ACC[3] = __builtin_amdgcn_mfma_f32_32x32x2f32((A4).w, (BVAL), ACC[3], 0, 0, 0); \
  } while (0)

// Waves per workgroup of the persistent kernel (one workgroup per CU: the weights take 99-132 KB of LDS).  Two waves per SIMD:
// while one waits for its next tile's inputs, its LDS operands or runs the ELUs, the other keeps the matrix pipe busy
// (round 4, 240x320 head of cfg2: 51.4 us with four waves -> 44.2 with eight; same tiles, bit-identical output).
#ifndef DT_HEAD_WAVES
#define DT_HEAD_WAVES 8
#endif
static_assert(DT_HEAD_WAVES % 4 == 0 && DT_HEAD_WAVES >= 4 && DT_HEAD_WAVES <= 16, "whole waves per SIMD");
constexpr int kHeadThreads = 64 * DT_HEAD_WAVES;

template <int NG>  // NG = cin / 8 input groups (8 or 16)
__global__ __launch_bounds__(kHeadThreads, 1) void head_mlp_kernel(const HeadArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* lds_wa = lds;
  float* lds_wb = lds + NG * 4 * kHeadStep;
  float* lds_tail = lds_wb + 64 * kHeadStep;
  {
    const float4* g1 = reinterpret_cast<const float4*>(a.wa);
    float4* l1 = reinterpret_cast<float4*>(lds_wa);
    for (int i = threadIdx.x; i < NG * 4 * kHeadStep / 4; i += kHeadThreads) l1[i] = g1[i];
    const float4* g2 = reinterpret_cast<const float4*>(a.wb);
    float4* l2 = reinterpret_cast<float4*>(lds_wb);
    for (int i = threadIdx.x; i < 64 * kHeadStep / 4; i += kHeadThreads) l2[i] = g2[i];
    for (int i = threadIdx.x; i < kHeadTail; i += kHeadThreads) lds_tail[i] = a.tail[i];
  }
  __syncthreads();

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int half = lane >> 5, p = lane & 31;
  const int lane_off = (half * 32 + p) * 4;
  const long tiles = (a.pixels + 31) / 32;
  // balanced contiguous spans of pixel tiles, dealt to SIMDs first (wave w of a workgroup runs on SIMD w & 3) and then split
  // between the waves of a SIMD: with 3072 tiles on 1024 SIMDs every SIMD gets 3 (its waves 2 + 1), where spans balanced per
  // wave would give the wave pairs 2 + 2 and 1 + 1
  constexpr int kWavesPerSimd = DT_HEAD_WAVES / 4;
  const long simds_total = (long)gridDim.x * 4;
  const long sid = (long)blockIdx.x * 4 + (wave & 3);
  const long s_begin = sid * tiles / simds_total, s_end = (sid + 1) * tiles / simds_total;
  const int sub = wave >> 2;
  long t = s_begin + (s_end - s_begin) * sub / kWavesPerSimd;
  const long t_end = s_begin + (s_end - s_begin) * (sub + 1) / kWavesPerSimd;
  const float bc = lds_tail[384];

  float4 xq[NG], xn[NG];
  auto load_x = [&](float4 (&dst)[NG], long tile) {
    const long pix = tile * 32 + p;
    const long pc = pix < a.pixels ? pix : a.pixels - 1;
    const float4* src = reinterpret_cast<const float4*>(a.in + pc * a.cin + half * 4);
#pragma unroll
    for (int g = 0; g < NG; ++g) dst[g] = src[g * 2];
  };
  if (t < t_end) load_x(xn, t);
  for (; t < t_end; ++t) {
#pragma unroll
    for (int g = 0; g < NG; ++g) xq[g] = xn[g];
    if (t + 1 < t_end) load_x(xn, t + 1);
    // ---- layer A: cin -> 128 ------------------------------------------------------------------
    f32x16 acc1[4];
    {
      const float* bl = lds_tail + half * 64;
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc1[i][r] = bl[i * 16 + r];
    }
    {
      const float4* wl = reinterpret_cast<const float4*>(lds_wa + lane_off);
#pragma unroll
      for (int g = 0; g < NG; ++g) {
        DT_HMFMA4(acc1, wl[(g * 4 + 0) * (kHeadStep / 4)], xq[g].x);
        DT_HMFMA4(acc1, wl[(g * 4 + 1) * (kHeadStep / 4)], xq[g].y);
        DT_HMFMA4(acc1, wl[(g * 4 + 2) * (kHeadStep / 4)], xq[g].z);
        DT_HMFMA4(acc1, wl[(g * 4 + 3) * (kHeadStep / 4)], xq[g].w);
      }
    }
    // ---- layer B: 128 -> 128 (ELU applied just in time to each B operand) -----------------------
    f32x16 acc2[4];
    {
      const float* bl = lds_tail + 128 + half * 64;
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc2[i][r] = bl[i * 16 + r];
    }
    {
      const float4* wl = reinterpret_cast<const float4*>(lds_wb + lane_off);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float4 a4 = wl[(i * 16 + r) * (kHeadStep / 4)];
          const float hv = elu1(acc1[i][r]);
          DT_HMFMA4(acc2, a4, hv);
          if ((r & 3) == 3) __builtin_amdgcn_sched_barrier(0);
        }
    }
    // ---- layer C: 128 -> 1 --------------------------------------------------------------------------
    float s = 0.f;
    {
      const float* wl = lds_tail + 256 + half * 64;
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += wl[i * 16 + r] * elu1(acc2[i][r]);
    }
    s += __shfl_xor(s, 32, 64);
    const long pix = t * 32 + p;
    if (half == 0 && pix < a.pixels) {
      const float v = s + bc;
      a.out[pix] = v;
      if (a.out_exp) a.out_exp[pix] = expf(v);
    }
  }
}

template <int NG>
__global__ __launch_bounds__(256) void head_mlp_split_kernel(const HeadArgs a) {
  __shared__ float hbuf[4 * 16 * 64];
  __shared__ float red[4 * 32];
  head_split_body<NG>(a, blockIdx.x, hbuf, red);
}

__global__ __launch_bounds__(256) void head_mlp_multi_kernel(const HeadMultiArgs m) {
  __shared__ float hbuf[4 * 16 * 64];
  __shared__ float red[4 * 32];
  head_multi_block(m, blockIdx.x, hbuf, red);
}


}  // namespace dt

using namespace dt;

extern "C" {

int dt_head_mlp_pack_floats(int cin, int* wa, int* wb, int* tail) {
  DT_REQUIRE(cin == 64 || cin == 128 || cin == 256, "dt_head_mlp_pack_floats: cin=%d (64, 128 or 256 supported)", cin);
  if (wa) *wa = (cin / 2) * kHeadStep;
  if (wb) *wb = 64 * kHeadStep;
  if (tail) *tail = kHeadTail;
  return 0;
}

int dt_head_mlp_f32(const float* in_nhwc, const float* wa, const float* wb, const float* tail, float* out, float* out_exp,
                    int64_t pixels, int cin, dt_stream_t s) {
  DT_REQUIRE(in_nhwc && wa && wb && tail && out, "dt_head_mlp_f32: null pointer");
  DT_REQUIRE(pixels > 0, "dt_head_mlp_f32: pixels=%ld", (long)pixels);
  DT_REQUIRE(cin == 64 || cin == 128 || cin == 256, "dt_head_mlp_f32: cin=%d (64, 128 or 256 supported)", cin);
  const int g_head_cus = device_cu_count();
  HeadArgs a;
  a.in = in_nhwc; a.wa = wa; a.wb = wb; a.tail = tail; a.out = out; a.out_exp = out_exp; a.pixels = pixels; a.cin = cin;
  const long tiles = (pixels + 31) / 32;
  static const long split_max_tiles = [] { const char* e = getenv("DT_HEAD_SPLIT_MAX_TILES"); return e ? atol(e) : kHeadSplitMaxTiles; }();
  if (tiles <= split_max_tiles) {  // small image: one tile per workgroup, hidden features split over the waves
    if (cin == 64)
      DT_LAUNCH(head_mlp_split_kernel<8>, dim3((unsigned)tiles), dim3(256), 0, to_stream(s), a);
    else if (cin == 128)
      DT_LAUNCH(head_mlp_split_kernel<16>, dim3((unsigned)tiles), dim3(256), 0, to_stream(s), a);
    else
      DT_LAUNCH(head_mlp_split_kernel<32>, dim3((unsigned)tiles), dim3(256), 0, to_stream(s), a);
    return check_launch("dt_head_mlp_f32");
  }
  DT_REQUIRE(cin != 256, "dt_head_mlp_f32: cin=256 is only supported up to %ld pixel tiles (got %ld)", split_max_tiles, tiles);
  const long want = (tiles + DT_HEAD_WAVES - 1) / DT_HEAD_WAVES;
  const int blocks = (int)(want < g_head_cus ? want : g_head_cus);
  const size_t lds_bytes = (size_t)((cin / 2) * kHeadStep + 64 * kHeadStep + kHeadTail) * sizeof(float);
#define DT_LAUNCH_HEAD(NG_)                                                                                          \
  do {                                                                                                               \
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&head_mlp_kernel<NG_>),                          \
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);                  \
    if (e != hipSuccess) {                                                                                           \
      (void)hipGetLastError();                                                                                       \
      return fail("dt_head_mlp_f32: cannot reserve %zu B of LDS: %s", lds_bytes, hipGetErrorString(e));               \
    }                                                                                                                \
    DT_LAUNCH(head_mlp_kernel<NG_>, dim3(blocks), dim3(kHeadThreads), lds_bytes, to_stream(s), a);                   \
  } while (0)
  if (cin == 64) DT_LAUNCH_HEAD(8); else DT_LAUNCH_HEAD(16);
#undef DT_LAUNCH_HEAD
  return check_launch("dt_head_mlp_f32");
}

int dt_head_mlp_multi_f32(int n_heads, const float* const* in_nhwc, const float* const* wa, const float* const* wb,
                          const float* const* tail, float* const* out, float* const* out_exp, const int64_t* pixels,
                          const int* cin, dt_stream_t s) {
  HeadMultiArgs m;
  unsigned total = 0;
  if (int rc = head_multi_fill(n_heads, in_nhwc, wa, wb, tail, out, out_exp, pixels, cin, m, total, "dt_head_mlp_multi_f32")) return rc;
  DT_LAUNCH(head_mlp_multi_kernel, dim3(total), dim3(256), 0, to_stream(s), m);
  return check_launch("dt_head_mlp_multi_f32");
}

}  // extern "C"
