// Winograd F(4x4, 3x3) convolution for gfx950, exact-fp32 MFMA: the chip-filling 3x3 stride-1 layers of the conv stacks
// (reference modules/layers.py:77-94 BasicBlock convs, modules/networks_fast.py:17-40 ConvBlock convs, modules/networks.py:20-85
// UNet++ nodes) with 36 multiplies per 4x4 output tile -- 2.25 per output pixel instead of 4 (F(2x2), conv.hip) or 9 (direct).
//
//   Y = A^T [ (G g G^T) .* (B^T d B) ] A      interpolation points 0, +-1, +-2, inf (the textbook matrices below)
//
// Same fused function as dt_conv2d_wino_f32: up to three virtually concatenated NHWC sources (each optionally nearest-x2
// upsampled), zero or replicate padding, bias + residual + activation in the epilogue.
//
// Mapping.  A workgroup (6 waves) owns 4x4 Winograd tiles = 16x16 output pixels of one 32-channel block.  The 6x6 transform
// domain is dealt by ROWS: wave i owns positions (i, 0..5), one accumulator pair (two 16-channel halves) each.  The matrix
// instruction is v_mfma_f32_16x16x4_f32: N = the 16 tiles, M = 16 output channels, K = 4 input channels; lane l = (tile l & 15,
// k-slot l >> 4).  Per 8-channel group a wave reads, for every window column, the (up to) four patch rows its transform row
// combines, forms the six B operands of its row (input transform B^T d B restricted to row i: 72 multiply-adds on two
// channels per lane) and issues 24 MFMAs against the pre-transformed weights.  Epilogue: the column inverse transform runs in
// registers, the row inverse transform across the six waves through LDS, then all threads write the 16x16x32 block.
//
// Accuracy.  F(4x4) amplifies fp32 rounding more than F(2x2): 4-7e-6 mean / 3e-5 worst case per layer on O(1) activations
// against 4e-7 / 2e-6 (numpy model of exactly this arithmetic) -- inside the path's tolerances (log depth 2e-4,
// depth 1e-3), which the whole-tensor parity tests check at full size with this kernel on the path.
#include "common.hpp"

#ifndef DT_W4ABL
#define DT_W4ABL 0  // ablation switches (timing experiments only): 1 no weight loads, 2 no patch loads, 4 no LDS window reads,
                    // 8 no MFMAs, 16 no output loop, 32 no staging writes
#endif

namespace dt {
namespace w4 {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int kT = 4;                               // Winograd tiles per workgroup side (4 x 4 tiles = MFMA N)
constexpr int kOut = 4 * kT;                        // 16 output rows / columns per workgroup
constexpr int kPH = 4 * kT + 2, kPW = 4 * kT + 2;   // staged input patch 18 x 18 pixels
constexpr int kNPix = kPH * kPW;                    // 324
constexpr int kThreads = 384;
constexpr int kNLoad = (kNPix * 2 + kThreads - 1) / kThreads;  // float4 staging loads per thread and group: 2
// LDS layout of a staged 8-channel patch, in 8-byte slots (one channel PAIR 2kq, 2kq+1 of one pixel):
//   slot(kq, y, x) = kq * kPlane + (x & 3) * kSub + y * kPitch + (x >> 2)
// A lane (tile ty, tx; k-slot kq) reads pixel (4 ty + r, 4 tx + c) for fixed (r, c): slot = const + kq * kPlane + 36 ty + tx.
// With kPitch = 9 (4 * 9 = 4 mod 32) and kPlane = 16 mod 32 the 32 lanes of one ds_read_b64 cycle (16 tiles x 2 k-slots)
// land on 32 different slots of the 256-byte LDS line: conflict free.
constexpr int kPitch = 9;
constexpr int kSub = kPH * kPitch;                  // 162
constexpr int kPlane = 656;                         // 4 * 162 = 648, padded to 16 mod 32
constexpr int kPatchSlots = 4 * kPlane;             // 2624 slots = 20992 bytes per buffer
constexpr int kPatchFloats = 2 * kPatchSlots;
constexpr int kZFloat4 = 6 * 4 * 2 * 64;            // Z[i][q][mb][tile][mq] float4: 3072 = 48 KB
constexpr int kLdsFloats = (2 * kPatchFloats > 4 * kZFloat4) ? 2 * kPatchFloats : 4 * kZFloat4;  // 12288 floats = 48 KB

struct Args {
  const float* src[3];
  unsigned src_bytes[3];
  int c[3];
  int up[3];
  int nsrc;
  const float* wp;  // [co/32][groups][pos 36][64 lanes][mb*2 + kk]
  const float* bias;
  const float* res;
  float* out;
  int n, h_out, w_out, c_out, h_in, w_in, act;
  int pad_replicate;
  int xcd_remap;
  int groups;  // 8-channel input groups over all sources
  int co_blocks;
  int tiles_x, tiles_y;  // 16x16-pixel output blocks
};
}  // namespace w4
DT_ARG_POINTERS(w4::Args, offsetof(w4::Args, src) + 0 * sizeof(const float*), offsetof(w4::Args, src) + 1 * sizeof(const float*),
                offsetof(w4::Args, src) + 2 * sizeof(const float*), offsetof(w4::Args, wp), offsetof(w4::Args, bias),
                offsetof(w4::Args, res), offsetof(w4::Args, out));
namespace w4 {

__device__ __forceinline__ float apply_act(float v, int act) {
  if (act == DT_ACT_LRELU02) return v >= 0.f ? v : 0.2f * v;
  if (act == DT_ACT_ELU) return v > 0.f ? v : __expf(v) - 1.0f;
  if (act == DT_ACT_RELU) return fmaxf(v, 0.f);
  return v;
}

__device__ __forceinline__ long xcd_contiguous_block(int xcd_remap, unsigned b, unsigned nb) {
  if (!xcd_remap || (nb & 7u) != 0u) return (long)b;
  return (long)(b & 7u) * (nb >> 3) + (b >> 3);
}

// byte offset of this thread's 4-channel half of input pixel (iy, ix) of image n inside one NHWC source (nearest-upsampled
// when `up`); 0xFFFFFFFC (out of every buffer range: the raw buffer load returns zeros) for padding
__device__ __forceinline__ int pixel_byte_offset(bool inside, int n, int iy, int ix, int h_in, int w_in, int up, int cs, int hq) {
  const int hs = up ? (h_in >> 1) : h_in, ws = up ? (w_in >> 1) : w_in;
  const int sy = up ? (iy >> 1) : iy, sx = up ? (ix >> 1) : ix;
  const int pix = (n * hs + sy) * ws + sx;
  return inside ? (pix * cs + hq * 4) * 4 : -4;
}

__global__ __launch_bounds__(kThreads, 3) void conv_wino4_kernel(const Args a) {
  __shared__ __attribute__((aligned(16))) float lds[kLdsFloats];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // transform row of this wave
  const int n16 = lane & 15, kq = lane >> 4;
  const int ty = n16 >> 2, tx = n16 & 3;

  long bid = xcd_contiguous_block(a.xcd_remap, blockIdx.x, gridDim.x);
  const int cb = (int)(bid % a.co_blocks);
  bid /= a.co_blocks;
  const int bx = (int)(bid % a.tiles_x);
  bid /= a.tiles_x;
  const int by = (int)(bid % a.tiles_y);
  const int n = (int)(bid / a.tiles_y);
  const int iy0 = by * kOut - 1, ix0 = bx * kOut - 1;

  // ---- staging addresses (group invariant) --------------------------------------------------------------------------
  int poff0[kNLoad], poff1[kNLoad], poff2[kNLoad], lslot[kNLoad];
  const int hq = tid & 1;
#pragma unroll
  for (int it = 0; it < kNLoad; ++it) {
    const int idx = (tid >> 1) + it * (kThreads / 2);
    const int ly = idx / kPW, lx = idx - ly * kPW;
    int iy = iy0 + ly, ix = ix0 + lx;
    bool inside = idx < kNPix && iy >= 0 && iy < a.h_in && ix >= 0 && ix < a.w_in;
    if (a.pad_replicate) {
      iy = min(max(iy, 0), a.h_in - 1);
      ix = min(max(ix, 0), a.w_in - 1);
      inside = idx < kNPix;
    }
    poff0[it] = pixel_byte_offset(inside, n, iy, ix, a.h_in, a.w_in, a.up[0], a.c[0], hq);
    poff1[it] = pixel_byte_offset(inside && a.nsrc > 1, n, iy, ix, a.h_in, a.w_in, a.up[1], a.c[1], hq);
    poff2[it] = pixel_byte_offset(inside && a.nsrc > 2, n, iy, ix, a.h_in, a.w_in, a.up[2], a.c[2], hq);
    // channels 4 hq .. 4 hq + 3 = the pairs of k-slots 2 hq and 2 hq + 1
    // (threads past the patch write their zeros to the 8 padding slots at the end of their planes: no branch in the K loop)
    lslot[it] = (2 * hq) * kPlane + ((idx < kNPix) ? (lx & 3) * kSub + ly * kPitch + (lx >> 2) : 4 * kSub + (tid & 7));
  }
  const int ng0 = a.c[0] >> 3, ng1 = a.c[1] >> 3;
  const __amdgpu_buffer_rsrc_t rs0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.src[0]), 0, a.src_bytes[0], 0x00020000);
  const __amdgpu_buffer_rsrc_t rs1 =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.src[1] ? a.src[1] : a.src[0]), 0, a.src[1] ? a.src_bytes[1] : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs2 =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.src[2] ? a.src[2] : a.src[0]), 0, a.src[2] ? a.src_bytes[2] : 0, 0x00020000);

  // packed weights: this wave reads positions 6 * wave .. 6 * wave + 5 of every group
  const float4* wbase = reinterpret_cast<const float4*>(a.wp) + ((size_t)cb * a.groups * 36 + wave * 6) * 64 + lane;

  // ---- the transform row of this wave: B^T row i = coefficients on (up to) four patch rows ---------------------------
  //   0: 4 d0 - 5 d2 + d4        1: -4 d1 - 4 d2 + d3 + d4     2: 4 d1 - 4 d2 - d3 + d4
  //   3: -2 d1 - d2 + 2 d3 + d4  4: 2 d1 - d2 - 2 d3 + d4      5: 4 d1 - 5 d3 + d5
  const bool four = wave >= 1 && wave <= 4;
  const int rw0 = (wave == 0) ? 0 : 1;
  const int rw1 = (wave == 0) ? 2 : ((wave == 5) ? 3 : 2);
  const int rw2 = (wave == 0) ? 4 : ((wave == 5) ? 5 : 3);
  const int rw3 = 4;
  const float cf0 = (wave == 0 || wave == 2 || wave == 5) ? 4.f : ((wave == 1) ? -4.f : ((wave == 3) ? -2.f : 2.f));
  const float cf1 = (wave == 0 || wave == 5) ? -5.f : ((wave == 1 || wave == 2) ? -4.f : -1.f);
  const float cf2 = (wave == 0 || wave == 1 || wave == 5) ? 1.f : ((wave == 2) ? -1.f : ((wave == 3) ? 2.f : -2.f));
  const float cf3 = four ? 1.f : 0.f;  // (rows 0 and 5 have three terms: they read patch row 4 against a zero -- no branch)
  const int lbase = kq * kPlane + (4 * ty) * kPitch + tx;  // slot of window element (0, 0) of this lane
  const int ro0 = lbase + rw0 * kPitch, ro1 = lbase + rw1 * kPitch, ro2 = lbase + rw2 * kPitch, ro3 = lbase + rw3 * kPitch;
  // slot distance of window column c from column 0: (c & 3) * kSub + (c >> 2)
  constexpr int kCol[6] = {0, kSub, 2 * kSub, 3 * kSub, 1, kSub + 1};

  f32x4 acc[6][2];
#pragma unroll
  for (int j = 0; j < 6; ++j)
#pragma unroll
    for (int mb = 0; mb < 2; ++mb) acc[j][mb] = f32x4{0.f, 0.f, 0.f, 0.f};

  float4 patch[kNLoad];
  float4 wa[6], wb[6];
#pragma unroll
  for (int j = 0; j < 6; ++j) wa[j] = wb[j] = make_float4(0.f, 0.f, 0.f, 0.f);

#define DTW4_PREFETCH(G, W)                                                                                         \
  do {                                                                                                              \
    const int g_ = ((G) < a.groups) ? (G) : a.groups - 1;                                                           \
    const int sidx = (g_ < ng0) ? 0 : ((g_ < ng0 + ng1) ? 1 : 2);                                                   \
    const int gl = (sidx == 0) ? g_ : ((sidx == 1) ? g_ - ng0 : g_ - ng0 - ng1);                                    \
    const __amdgpu_buffer_rsrc_t rs = (sidx == 0) ? rs0 : ((sidx == 1) ? rs1 : rs2);                                \
    _Pragma("unroll") for (int it = 0; it < kNLoad; ++it) {                                                         \
      const int off = (sidx == 0) ? poff0[it] : ((sidx == 1) ? poff1[it] : poff2[it]);                              \
      typedef unsigned u32x4 __attribute__((ext_vector_type(4)));                                                   \
      if (!(DT_W4ABL & 2)) {                                                                                        \
        const u32x4 raw = __builtin_amdgcn_raw_buffer_load_b128(rs, off, gl * 32, 0);                               \
        patch[it] = make_float4(__uint_as_float(raw.x), __uint_as_float(raw.y), __uint_as_float(raw.z), __uint_as_float(raw.w)); \
      }                                                                                                             \
    }                                                                                                               \
    const float4* wg = wbase + (size_t)g_ * (36 * 64);                                                              \
    _Pragma("unroll") for (int j = 0; j < 6; ++j)                                                                   \
      W[j] = (DT_W4ABL & 1) ? make_float4((float)g_, (float)j, 1.f, 2.f) : wg[j * 64];                              \
  } while (0)

#if DT_W4ABL & 8
#define DTW4_MFMA(ACC, A_, B_) ACC[0] += (A_) * (B_)
#else
#define DTW4_MFMA(ACC, A_, B_) ACC = __builtin_amdgcn_mfma_f32_16x16x4f32((A_), (B_), ACC, 0, 0, 0)
#endif

#define DTW4_STEP(I, WC, WN)                                                                                        \
  do {                                                                                                              \
    f32x2* buf = reinterpret_cast<f32x2*>(lds) + ((I) & 1) * kPatchSlots;                                           \
    _Pragma("unroll") for (int it = 0; it < kNLoad; ++it) {                                                         \
      if (!(DT_W4ABL & 32)) {                                                                                       \
        buf[lslot[it]] = f32x2{patch[it].x, patch[it].y};                                                           \
        buf[lslot[it] + kPlane] = f32x2{patch[it].z, patch[it].w};                                                  \
      }                                                                                                             \
    }                                                                                                               \
    __syncthreads();                                                                                                \
    DTW4_PREFETCH((I) + 1, WN);                                                                                     \
    __builtin_amdgcn_sched_barrier(0); /* the loads are issued HERE: a full K step ahead of their use */            \
    f32x2 t[6];                                                                                                     \
    _Pragma("unroll") for (int c = 0; c < 6; ++c) {                                                                 \
      const f32x2 d0 = (DT_W4ABL & 4) ? f32x2{acc[c][0][0], 1.f} : buf[ro0 + kCol[c]];                              \
      const f32x2 d1 = (DT_W4ABL & 4) ? f32x2{acc[c][1][0], 2.f} : buf[ro1 + kCol[c]];                              \
      const f32x2 d2 = (DT_W4ABL & 4) ? f32x2{acc[c][0][1], 3.f} : buf[ro2 + kCol[c]];                              \
      const f32x2 d3 = (DT_W4ABL & 4) ? f32x2{acc[c][1][1], (float)(I)} : buf[ro3 + kCol[c]];                       \
      t[c] = cf0 * d0 + cf1 * d1 + cf2 * d2 + cf3 * d3;                                                             \
    }                                                                                                               \
    f32x2 V[6];                                                                                                     \
    {                                                                                                               \
      const f32x2 e = t[4] - 4.f * t[2], f = t[3] - 4.f * t[1], g2 = t[4] - t[2], h2 = t[3] - t[1];                 \
      V[0] = 4.f * t[0] - 5.f * t[2] + t[4];                                                                        \
      V[1] = e + f;                                                                                                 \
      V[2] = e - f;                                                                                                 \
      V[3] = g2 + 2.f * h2;                                                                                         \
      V[4] = g2 - 2.f * h2;                                                                                         \
      V[5] = 4.f * t[1] - 5.f * t[3] + t[5];                                                                        \
    }                                                                                                               \
    _Pragma("unroll") for (int j = 0; j < 6; ++j) {                                                                 \
      DTW4_MFMA(acc[j][0], WC[j].x, V[j].x);                                                                        \
      DTW4_MFMA(acc[j][1], WC[j].z, V[j].x);                                                                        \
      DTW4_MFMA(acc[j][0], WC[j].y, V[j].y);                                                                        \
      DTW4_MFMA(acc[j][1], WC[j].w, V[j].y);                                                                        \
    }                                                                                                               \
  } while (0)

  DTW4_PREFETCH(0, wa);
  for (int i = 0; i < a.groups; i += 2) {
    DTW4_STEP(i, wa, wb);
    if (i + 1 < a.groups) DTW4_STEP(i + 1, wb, wa);
  }
#undef DTW4_STEP
#undef DTW4_MFMA
#undef DTW4_PREFETCH

  // ---- inverse transform: columns (A^T M A restricted to this wave's row) in registers --------------------------------
  //   A^T = [1 1 1 1 1 0; 0 1 -1 2 -2 0; 0 1 1 4 4 0; 0 1 -1 8 -8 1]
  __syncthreads();  // every wave is done with the patch buffers
  float4* zl = reinterpret_cast<float4*>(lds);
#pragma unroll
  for (int mb = 0; mb < 2; ++mb) {
    f32x4 z[4];
    {
      const f32x4 m0 = acc[0][mb], m1 = acc[1][mb], m2 = acc[2][mb], m3 = acc[3][mb], m4 = acc[4][mb], m5 = acc[5][mb];
      const f32x4 s1 = m1 + m2, d1 = m1 - m2, s2 = m3 + m4, d2 = m3 - m4;
      z[0] = m0 + s1 + s2;
      z[1] = d1 + 2.f * d2;
      z[2] = s1 + 4.f * s2;
      z[3] = d1 + 8.f * d2 + m5;
    }
#pragma unroll
    for (int q = 0; q < 4; ++q)
      zl[(((wave * 4 + q) * 2 + mb) * 16 + n16) * 4 + kq] = make_float4(z[q][0], z[q][1], z[q][2], z[q][3]);
  }
  __syncthreads();

  // ---- rows across the six waves + epilogue: 16 tiles x 16 pixels x 8 channel quads = 2048 float4 outputs ---------------
  // Two passes of three outputs per thread (2 x 3 x 384 = 2304 >= 2048): the 18 Z reads of a pass are in flight together.
  constexpr int kOutTotal = (DT_W4ABL & 16) ? 64 : 2048;
#pragma unroll 1
  for (int pass = 0; pass < 2; ++pass) {
    float4 zz[3][6];
    int idxs[3];
#pragma unroll
    for (int u = 0; u < 3; ++u) {
      const int idx = (pass * 3 + u) * kThreads + tid;
      idxs[u] = idx;
      const int ic = idx < 2048 ? idx : 2047;
      const int mq = ic & 3, txo = (ic >> 2) & 3, mb = (ic >> 4) & 1, rest = ic >> 5;
      const int q = rest & 3, tyo = rest >> 4;
      const float4* zp = zl + ((q * 2 + mb) * 16 + tyo * 4 + txo) * 4 + mq;
#pragma unroll
      for (int i = 0; i < 6; ++i) zz[u][i] = zp[i * 512];
    }
#pragma unroll
    for (int u = 0; u < 3; ++u) {
      const int idx = idxs[u];
      const int mq = idx & 3, txo = (idx >> 2) & 3, mb = (idx >> 4) & 1, rest = idx >> 5;
      const int q = rest & 3, p = (rest >> 2) & 3, tyo = rest >> 4;
      // row p of A^T: [e0, 1, sg, k, sg k, e5] with sg = -1 for odd p, k = 2^p
      const float e0 = (p == 0) ? 1.f : 0.f, e5 = (p == 3) ? 1.f : 0.f, sg = (p & 1) ? -1.f : 1.f, kk = (float)(1 << p);
      const float c4 = sg * kk;
      const float4 z0 = zz[u][0], z1 = zz[u][1], z2 = zz[u][2], z3 = zz[u][3], z4 = zz[u][4], z5 = zz[u][5];
      float4 o;
      o.x = e0 * z0.x + z1.x + sg * z2.x + kk * z3.x + c4 * z4.x + e5 * z5.x;
      o.y = e0 * z0.y + z1.y + sg * z2.y + kk * z3.y + c4 * z4.y + e5 * z5.y;
      o.z = e0 * z0.z + z1.z + sg * z2.z + kk * z3.z + c4 * z4.z + e5 * z5.z;
      o.w = e0 * z0.w + z1.w + sg * z2.w + kk * z3.w + c4 * z4.w + e5 * z5.w;
      const int oy = by * kOut + 4 * tyo + p, ox = bx * kOut + 4 * txo + q;
      if (idx < kOutTotal && oy < a.h_out && ox < a.w_out) {
        const int co = cb * 32 + mb * 16 + mq * 4;
        const size_t off = (((size_t)n * a.h_out + oy) * a.w_out + ox) * a.c_out + co;
        if (a.bias) {
          const float4 bv = *reinterpret_cast<const float4*>(a.bias + co);
          o.x += bv.x; o.y += bv.y; o.z += bv.z; o.w += bv.w;
        }
        if (a.res) {
          const float4 rv = *reinterpret_cast<const float4*>(a.res + off);
          o.x += rv.x; o.y += rv.y; o.z += rv.z; o.w += rv.w;
        }
        o.x = apply_act(o.x, a.act); o.y = apply_act(o.y, a.act); o.z = apply_act(o.z, a.act); o.w = apply_act(o.w, a.act);
        *reinterpret_cast<float4*>(a.out + off) = o;
      }
    }
  }
}

// OIHW 3x3 weights -> U = G g G^T per (co, ci) (in double, rounded once), packed [co_block][group][pos 36][64 lanes][mb*2 + kk]:
// lane = (m = lane & 15, kq = lane >> 4), output channel co_block*32 + mb*16 + m, input channel group*8 + 2*kq + kk
//   G = [1/4 0 0; -1/6 -1/6 -1/6; -1/6 1/6 -1/6; 1/24 1/12 1/6; 1/24 -1/12 1/6; 0 0 1]
__global__ void conv_wino4_pack_kernel(const float* __restrict__ W, float* __restrict__ packed, int c_out, int c_in) {
  const int groups = c_in >> 3;
  const size_t total = (size_t)c_out * c_in * 36;
  const double G[6][3] = {{0.25, 0.0, 0.0},          {-1.0 / 6, -1.0 / 6, -1.0 / 6}, {-1.0 / 6, 1.0 / 6, -1.0 / 6},
                          {1.0 / 24, 1.0 / 12, 1.0 / 6}, {1.0 / 24, -1.0 / 12, 1.0 / 6}, {0.0, 0.0, 1.0}};
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
    size_t r = idx;
    const int e = r & 3;
    r >>= 2;
    const int l = r & 63;
    r >>= 6;
    const int pos = (int)(r % 36);
    r /= 36;
    const int g = (int)(r % groups);
    const int cb = (int)(r / groups);
    const int mb = e >> 1, kk = e & 1, m = l & 15, kq = l >> 4;
    const int co = cb * 32 + mb * 16 + m, ci = g * 8 + 2 * kq + kk;
    const float* k = W + ((size_t)co * c_in + ci) * 9;
    const int ra = pos / 6, rb = pos % 6;
    double u = 0.0;
#pragma unroll
    for (int y = 0; y < 3; ++y)
#pragma unroll
      for (int x = 0; x < 3; ++x) u += G[ra][y] * (double)k[y * 3 + x] * G[rb][x];
    packed[idx] = (float)u;
  }
}

}  // namespace w4
}  // namespace dt

using namespace dt;

extern "C" {

int64_t dt_conv_wino4_pack_floats(int c_out, int c_in) {
  if (c_out <= 0 || c_in <= 0 || c_out % 32 != 0 || c_in % 8 != 0) return 0;
  return (int64_t)c_out * c_in * 36;
}

int dt_conv_wino4_pack_f32(const float* W, float* packed, int c_out, int c_in, dt_stream_t s) {
  DT_REQUIRE(W && packed, "dt_conv_wino4_pack_f32: null pointer");
  DT_REQUIRE(c_out > 0 && c_out % 32 == 0 && c_in > 0 && c_in % 8 == 0, "dt_conv_wino4_pack_f32: c_out=%d (multiple of 32), c_in=%d "
             "(multiple of 8)", c_out, c_in);
  const size_t total = (size_t)c_out * c_in * 36;
  const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
  DT_LAUNCH(w4::conv_wino4_pack_kernel, dim3(blocks), dim3(256), 0, to_stream(s), W, packed, c_out, c_in);
  return check_launch("dt_conv_wino4_pack_f32");
}

int64_t dt_conv2d_wino4_blocks(const dt_conv_desc* d) {
  if (!d || d->ksize != 3 || d->stride != 1 || d->nsrc < 1 || d->nsrc > 3 || d->c_out <= 0 || d->c_out % 32 != 0 || d->n <= 0 ||
      d->h_out <= 0 || d->w_out <= 0)
    return 0;
  for (int s = 0; s < d->nsrc; ++s)
    if (d->c[s] <= 0 || d->c[s] % 8 != 0) return 0;
  return (int64_t)d->n * ((d->h_out + w4::kOut - 1) / w4::kOut) * ((d->w_out + w4::kOut - 1) / w4::kOut) * (d->c_out / 32);
}

int dt_conv2d_wino4_f32(const dt_conv_desc* d, const float* in0, const float* in1, const float* in2, const float* packed_w,
                        const float* bias, const float* residual, float* out, dt_stream_t s) {
  DT_REQUIRE(d != nullptr, "dt_conv2d_wino4_f32: null descriptor");
  const int64_t blocks = dt_conv2d_wino4_blocks(d);
  DT_REQUIRE(blocks > 0, "dt_conv2d_wino4_f32: needs a 3x3 stride-1 convolution with c_out %% 32 == 0 and every source a multiple "
                         "of 8 channels");
  DT_REQUIRE(blocks < 2147483647LL, "dt_conv2d_wino4_f32: grid too large");
  DT_REQUIRE(d->h_in == d->h_out && d->w_in == d->w_out, "dt_conv2d_wino4_f32: bad extents");
  DT_REQUIRE(d->act >= 0 && d->act <= 3 && (d->pad_mode == 0 || d->pad_mode == 1) && d->transposed == 0,
             "dt_conv2d_wino4_f32: bad act / pad_mode / transposed");
  DT_REQUIRE(packed_w && out, "dt_conv2d_wino4_f32: null pointer");
  const float* ins[3] = {in0, in1, in2};
  w4::Args a;
  a.groups = 0;
  for (int i = 0; i < 3; ++i) {
    a.src[i] = nullptr;
    a.src_bytes[i] = 0;
    a.c[i] = 0;
    a.up[i] = 0;
  }
  for (int i = 0; i < d->nsrc; ++i) {
    DT_REQUIRE(ins[i] != nullptr, "dt_conv2d_wino4_f32: source %d is null", i);
    DT_REQUIRE(!d->up[i] || (d->h_in % 2 == 0 && d->w_in % 2 == 0), "dt_conv2d_wino4_f32: upsampled source needs even extents");
    a.src[i] = ins[i];
    a.c[i] = d->c[i];
    a.up[i] = d->up[i] ? 1 : 0;
    a.groups += d->c[i] >> 3;
    const size_t px = (size_t)d->n * (d->up[i] ? d->h_in / 2 : d->h_in) * (d->up[i] ? d->w_in / 2 : d->w_in);
    const size_t bytes = px * (size_t)d->c[i] * sizeof(float);
    DT_REQUIRE(bytes < 0xFFFFF000ull, "dt_conv2d_wino4_f32: source %d is larger than a 32-bit buffer range", i);
    a.src_bytes[i] = (unsigned)bytes;
  }
  a.nsrc = d->nsrc;
  a.wp = packed_w;
  a.bias = bias;
  a.res = residual;
  a.out = out;
  a.n = d->n; a.h_out = d->h_out; a.w_out = d->w_out; a.c_out = d->c_out; a.h_in = d->h_in; a.w_in = d->w_in; a.act = d->act;
  a.pad_replicate = d->pad_mode;
  a.xcd_remap = 1;
  a.co_blocks = d->c_out / 32;
  a.tiles_x = (d->w_out + w4::kOut - 1) / w4::kOut;
  a.tiles_y = (d->h_out + w4::kOut - 1) / w4::kOut;
  DT_LAUNCH(w4::conv_wino4_kernel, dim3((unsigned)blocks), dim3(w4::kThreads), 0, to_stream(s), a);
  return check_launch("dt_conv2d_wino4_f32");
}

}  // extern "C"
