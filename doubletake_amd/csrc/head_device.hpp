// Device code of the fused regression heads that more than one translation unit needs: the one-tile-per-workgroup head
// (head_split_body) runs in head_mlp.hip's own kernels and -- round 5 -- as extra workgroups inside the grid of a decoder
// convolution (conv.hip: conv_wino_heads_kernel).
#pragma once
#include "common.hpp"

namespace dt {

typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int kHeadHidden = 128;
constexpr int kHeadStep = 256;        // floats per K step: [2 halves][32 lanes][4 feature blocks]
constexpr int kHeadTail = 388;        // ba_r[128], bb_r[128], wc_r[128], bc, pad[3]
constexpr int kHeadMaxCin = 128;      // LDS: Cin*512 B + 64 KB + tail <= 160 KB

struct HeadArgs {
  const float* in;    // [pixels][cin] NHWC
  const float* wa;    // packed [cin/2 steps][256]
  const float* wb;    // packed [64 steps][256]
  const float* tail;  // kHeadTail
  float* out;         // [pixels]
  float* out_exp;     // optional [pixels]: expf(out) (the depth of a log-depth head)
  long pixels;
  int cin;
};
#define DT_HEAD_ARG_PTRS(BASE)                                                                                              \
  (BASE) + offsetof(HeadArgs, in), (BASE) + offsetof(HeadArgs, wa), (BASE) + offsetof(HeadArgs, wb), (BASE) + offsetof(HeadArgs, tail), \
      (BASE) + offsetof(HeadArgs, out), (BASE) + offsetof(HeadArgs, out_exp)
DT_ARG_POINTERS(HeadArgs, DT_HEAD_ARG_PTRS(0));

__device__ __forceinline__ float elu1(float v) { return v > 0.f ? v : __expf(v) - 1.0f; }  // ATen: exp(x) - 1

// Small images (a few hundred pixel tiles): the persistent kernel above would give each wave a single tile,
// i.e. 384-512 dependent MFMAs behind a 100+ KB weight staging, on a fraction of the CUs.  Here a workgroup
// owns ONE tile and its four waves split the 128 hidden features (wave w = feature block w): 4x shorter
// chains, 4x more workgroups, weights read straight from L2 (each wave only needs its quarter).
template <int NG>
__device__ __forceinline__ void head_split_body(const HeadArgs& a, const unsigned tile, float* hbuf, float* red) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int half = lane >> 5, p = lane & 31;
  const int lane_off = (half * 32 + p) * 4 + wave;  // this wave's component of the packed float4
  const long pix = (long)tile * 32 + p;
  const long pc = pix < a.pixels ? pix : a.pixels - 1;

  float4 xq[NG];
  {
    const float4* src = reinterpret_cast<const float4*>(a.in + pc * (NG * 8) + half * 4);
#pragma unroll
    for (int g = 0; g < NG; ++g) xq[g] = src[g * 2];
  }
  // ---- layer A: cin -> this wave's 32 of 128 hidden features -----------------------------------------
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = a.tail[half * 64 + wave * 16 + r];
  {
    // (round 4: issuing all 96-192 A-operand loads of a wave up front -- 108-128 registers instead of 40 -- measured slower,
    //  36 -> 39 us for the three coarse heads; the launch is bound by its dependent MFMA chains, not by these loads)
    const float* wl = a.wa + lane_off;
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wl[(g * 4 + 0) * kHeadStep], xq[g].x, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wl[(g * 4 + 1) * kHeadStep], xq[g].y, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wl[(g * 4 + 2) * kHeadStep], xq[g].z, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wl[(g * 4 + 3) * kHeadStep], xq[g].w, acc, 0, 0, 0);
    }
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) hbuf[(wave * 16 + r) * 64 + lane] = elu1(acc[r]);
  __syncthreads();
  // ---- layer B: 128 -> this wave's 32 features; B operands of step (i, r) = block i's row r -----------
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = a.tail[128 + half * 64 + wave * 16 + r];
  {
    const float* wl = a.wb + lane_off;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r)
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wl[(i * 16 + r) * kHeadStep], hbuf[(i * 16 + r) * 64 + lane], acc, 0, 0, 0);
  }
  // ---- layer C: partial dot over this wave's features, summed across halves and waves -------------------
  float s = 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) s += a.tail[256 + half * 64 + wave * 16 + r] * elu1(acc[r]);
  s += __shfl_xor(s, 32, 64);
  if (half == 0) red[wave * 32 + p] = s;
  __syncthreads();
  if (wave == 0 && half == 0 && pix < a.pixels) {
    const float v = red[p] + red[32 + p] + red[64 + p] + red[96 + p] + a.tail[384];
    a.out[pix] = v;
    if (a.out_exp) a.out_exp[pix] = expf(v);
  }
}

// The three coarse heads of one decoder pass (s3: 256 ch, s2: 128 ch, s1: 64 ch at 640x480) are independent and
// each far too small to fill the chip (38 / 150 / 600 tiles): ONE grid runs all of them, every workgroup picking
// its head from the tile prefix.  Same body as head_mlp_split_kernel, so the results are bit-identical to separate
// launches; what goes away is two launch boundaries and two latency-bound tails.
constexpr int kHeadMultiMax = 4;
struct HeadMultiArgs {
  HeadArgs h[kHeadMultiMax];
  unsigned first[kHeadMultiMax + 1];  // tile prefix: head i owns virtual blocks [first[i], first[i+1])
};
static_assert(kHeadMultiMax == 4, "DT_ARG_POINTERS(HeadMultiArgs) lists four heads");
DT_ARG_POINTERS(HeadMultiArgs, DT_HEAD_ARG_PTRS(offsetof(HeadMultiArgs, h) + 0 * sizeof(HeadArgs)),
                DT_HEAD_ARG_PTRS(offsetof(HeadMultiArgs, h) + 1 * sizeof(HeadArgs)),
                DT_HEAD_ARG_PTRS(offsetof(HeadMultiArgs, h) + 2 * sizeof(HeadArgs)),
                DT_HEAD_ARG_PTRS(offsetof(HeadMultiArgs, h) + 3 * sizeof(HeadArgs)));

constexpr long kHeadSplitMaxTiles = 1024;

// host side: validate the pointer tables of n_heads heads and fill the kernel argument block; total = number of tiles (blocks)
inline int head_multi_fill(int n_heads, const float* const* in_nhwc, const float* const* wa, const float* const* wb,
                           const float* const* tail, float* const* out, float* const* out_exp, const int64_t* pixels,
                           const int* cin, HeadMultiArgs& m, unsigned& total, const char* who) {
  DT_REQUIRE(n_heads >= 1 && n_heads <= kHeadMultiMax, "%s: n_heads=%d (1..%d)", who, n_heads, kHeadMultiMax);
  DT_REQUIRE(in_nhwc && wa && wb && tail && out && pixels && cin, "%s: null pointer table", who);
  total = 0;
  for (int i = 0; i < kHeadMultiMax; ++i) {
    const int j = i < n_heads ? i : n_heads - 1;  // unused slots repeat the last head and own no blocks
    DT_REQUIRE(in_nhwc[j] && wa[j] && wb[j] && tail[j] && out[j], "%s: null pointer in head %d", who, j);
    DT_REQUIRE(cin[j] == 64 || cin[j] == 128 || cin[j] == 256, "%s: head %d cin=%d (64, 128 or 256 supported)", who, j, cin[j]);
    const long tiles = ((long)pixels[j] + 31) / 32;
    DT_REQUIRE(pixels[j] > 0 && tiles <= kHeadSplitMaxTiles, "%s: head %d has %ld pixels (1..%ld supported; use dt_head_mlp_f32)", who,
               j, (long)pixels[j], kHeadSplitMaxTiles * 32);
    HeadArgs& a = m.h[i];
    a.in = in_nhwc[j]; a.wa = wa[j]; a.wb = wb[j]; a.tail = tail[j]; a.out = out[j]; a.out_exp = out_exp ? out_exp[j] : nullptr;
    a.pixels = pixels[j]; a.cin = cin[j];
    m.first[i] = total;
    if (i < n_heads) total += (unsigned)tiles;
  }
  m.first[kHeadMultiMax] = total;
  for (int i = n_heads; i < kHeadMultiMax; ++i) m.first[i] = 0xffffffffu;  // never selected
  return 0;
}

// head and tile of virtual block b of a multi-head grid, then the head body (hbuf: 4096 floats, red: 128 floats of LDS)
__device__ __forceinline__ void head_multi_block(const HeadMultiArgs& m, const unsigned b, float* hbuf, float* red) {
  int i = 0;
#pragma unroll
  for (int k = 1; k < kHeadMultiMax; ++k) i += (b >= m.first[k]) ? 1 : 0;
  const HeadArgs& a = m.h[i];
  const unsigned tile = b - m.first[i];
  const int cin = a.cin;
  if (cin == 256)
    head_split_body<32>(a, tile, hbuf, red);
  else if (cin == 128)
    head_split_body<16>(a, tile, hbuf, red);
  else
    head_split_body<8>(a, tile, hbuf, red);
}

}  // namespace dt
