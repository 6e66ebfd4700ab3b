// OPT-IN split-precision Winograd F(2x2,3x3) convolution for gfx950: the conv-stack counterpart of csrc/cv_mlp_split.hip.
//
// Same function as conv_wino_kernel<1> (csrc/conv.hip; reference modules/layers.py:77-94 BasicBlock convs,
// modules/networks_fast.py:17-40 ConvBlock convs: 3x3, stride 1, bias + LeakyReLU/ELU + residual + concat + nearest x2 fused),
// same tiling, same LDS patch geometry, same fp32 input / inverse transforms and epilogue -- but the products of the
// Winograd domain run on the fp16 matrix pipe (16x the fp32 MFMA rate on CDNA4) with every operand split in two halves
//     x = x_hi + x_lo,   x_hi = fp16(x),   x_lo = fp16(x - x_hi)
//     U * V  ~=  U_hi*V_hi + U_lo*V_hi + U_hi*V_lo          (fp32 accumulation inside v_mfma_f32_32x32x16_f16)
// where U = G g G^T (weights, transformed and split once by dt_conv_wino_split_pack_f16) and V = B^T d B (input tile,
// transformed in fp32 and split on the fly).  The dropped U_lo*V_lo term is 2^-22 relative: fp32-class results.  NOT the
// default; the headline conv stack stays exact fp32 (conv_ops.CONV_PRECISION = "split16" opts in, bench.py
// --conv-precision split16).  Range assumption: |U|, |V| < 65504 (V is a sum of four activations).
//
// Mapping onto v_mfma_f32_32x32x16_f16 (D[i][j] += A[i][k] B[k][j], k = 0..15; lane l = (row/col l & 31, k-block l >> 5),
// a lane's A / B operand = the 8 consecutive k of its k-block): i = output channel (A = U from global memory / L2),
// j = Winograd tile of the workgroup's 4 x 8 tiles (B = V in registers), k = 16 input channels -- lane half kb supplies
// channels 8 kb .. 8 kb + 7 of the group, i.e. it reads two 4-channel quads per window position where the fp32 kernel reads
// one.  One K step = 16 channels = 12 MFMAs per wave (4 transform positions x 3 partial products) instead of 32 fp32 MFMAs.
#include "common.hpp"

namespace dt {
namespace ws {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));

constexpr int kTH = 4, kTW = 8;                      // Winograd tiles per workgroup (rows, cols) = MFMA N
constexpr int kPH = 2 * kTH + 2, kPW = 2 * kTW + 2;  // staged input patch 10 x 18 pixels
constexpr int kRowPitch = 10;                        // columns per parity (9 used): two rows shift the bank group by 4
// LDS layout of a staged 16-channel patch: [quad (4 channels)][column parity][row][column / 2, pitch 10][4 floats] --
// the fp32 kernel's conflict-free layout (conv.hip: wino_lds_off) with four quads instead of two halves
constexpr int kPatchFloats = 4 * 2 * kPH * kRowPitch * 4;  // 3200
constexpr int kNPix = kPH * kPW;                           // 180
constexpr int kNLoad = (kNPix * 4 + 255) / 256;            // float4 staging loads per thread and group: 3
constexpr int kLdsFloats = 8192;                           // 2 patch buffers (6400), later 4 waves x 2 x 1024 Z values
constexpr int kFragHalves = 64 * 8;                        // one A fragment: 64 lanes x 8 halves = 1 KB

__device__ __forceinline__ int lds_off(int quad, int y, int x) {
  return ((((quad * 2 + (x & 1)) * kPH + y) * kRowPitch) + (x >> 1)) * 4;
}

struct Args {
  const float* src[3];
  int c[3];
  int up[3];
  int nsrc;
  const uint16_t* wp;  // [co/32][groups16][xi 16][part hi|lo][64 lanes][8 halves]
  const float* bias;
  const float* res;
  float* out;
  int n, h_out, w_out, c_out, h_in, w_in, act;
  int pad_replicate;
  int xcd_remap;
  int groups;  // 16-channel input groups over all sources
  int co_blocks;
};
}  // namespace ws
DT_ARG_POINTERS(ws::Args, offsetof(ws::Args, src) + 0 * sizeof(const float*), offsetof(ws::Args, src) + 1 * sizeof(const float*),
                offsetof(ws::Args, src) + 2 * sizeof(const float*), offsetof(ws::Args, wp), offsetof(ws::Args, bias),
                offsetof(ws::Args, res), offsetof(ws::Args, out));
namespace ws {

__device__ __forceinline__ float apply_act(float v, int act) {
  if (act == DT_ACT_LRELU02) return v >= 0.f ? v : 0.2f * v;
  if (act == DT_ACT_ELU) return v > 0.f ? v : __expf(v) - 1.0f;
  if (act == DT_ACT_RELU) return fmaxf(v, 0.f);
  return v;
}

// x[0..7] (fp32) -> hi / lo fp16 operands by truncation: hi = top 11 significant bits (v_cvt_pkrtz per pair), its fp32
// value is x with the low 13 mantissa bits cleared, so the remainder x - hi is exact (csrc/cv_mlp_split.hip: split8)
struct SplitB {
  half8 hi, lo;
};
__device__ __forceinline__ SplitB split8(const float (&x)[8]) {
  union {
    half8 v;
    decltype(__builtin_amdgcn_cvt_pkrtz(0.f, 0.f)) p[4];
  } H, L;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float a = x[2 * i], b = x[2 * i + 1];
    H.p[i] = __builtin_amdgcn_cvt_pkrtz(a, b);
    const float ar = a - __uint_as_float(__float_as_uint(a) & 0xFFFFE000u);
    const float br = b - __uint_as_float(__float_as_uint(b) & 0xFFFFE000u);
    L.p[i] = __builtin_amdgcn_cvt_pkrtz(ar, br);
  }
  SplitB s;
  s.hi = H.v;
  s.lo = L.v;
  return s;
}

__global__ __launch_bounds__(256, 2) void conv_wino_split_kernel(const Args a) {
  __shared__ __attribute__((aligned(16))) float lds[kLdsFloats];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // transform row of this wave
  const int kb = lane >> 5, t = lane & 31;
  // lane -> Winograd tile: ds_read_b128 services lanes {0-3,12-15,20-27} and {4-11,16-19,28-31} (of each half-wave) together;
  // the first group takes tile rows 0 and 2, the second rows 1 and 3 (conflict free with lds_off, as in conv.hip)
  const bool grp_a = (t < 4) || (t >= 12 && t < 16) || (t >= 20 && t < 28);
  const int gi = grp_a ? ((t < 4) ? t : ((t < 16) ? t - 8 : t - 12)) : ((t < 12) ? t - 4 : ((t < 20) ? t - 8 : t - 16));
  const int ty = 2 * (gi >> 3) + (grp_a ? 0 : 1), tx = gi & 7;

  const int wt_x = (a.w_out + 2 * kTW - 1) / (2 * kTW), wt_y = (a.h_out + 2 * kTH - 1) / (2 * kTH);
  // XCD-contiguous block order (conv.hip: xcd_contiguous_block)
  long bid = blockIdx.x;
  if (a.xcd_remap && (gridDim.x & 7u) == 0u) bid = (long)(blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3);
  const int cb = (int)(bid % a.co_blocks);
  bid /= a.co_blocks;
  const int bx = (int)(bid % wt_x);
  bid /= wt_x;
  const int by = (int)(bid % wt_y);
  const int n = (int)(bid / wt_y);
  const int iy0 = by * 2 * kTH - 1, ix0 = bx * 2 * kTW - 1;

  // staging: thread -> (pixel, quad) of the patch; element offset of the pixel inside each source (-1: zero padding)
  int poff0[kNLoad], poff1[kNLoad], poff2[kNLoad];
#pragma unroll
  for (int it = 0; it < kNLoad; ++it) {
    const int idx = (tid + it * 256) >> 2;
    const int ly = idx / kPW, lx = idx - ly * kPW;
    int iy = iy0 + ly, ix = ix0 + lx;
    bool inside = idx < kNPix && iy >= 0 && iy < a.h_in && ix >= 0 && ix < a.w_in;
    if (a.pad_replicate) {
      iy = min(max(iy, 0), a.h_in - 1);
      ix = min(max(ix, 0), a.w_in - 1);
      inside = idx < kNPix;
    }
    auto off = [&](bool ok, int s) {
      const int up = a.up[s];
      const int hs = up ? (a.h_in >> 1) : a.h_in, ws_ = up ? (a.w_in >> 1) : a.w_in;
      const int sy = up ? (iy >> 1) : iy, sx = up ? (ix >> 1) : ix;
      return ok ? ((n * hs + sy) * ws_ + sx) * a.c[s] + (tid & 3) * 4 : -1;
    };
    poff0[it] = off(inside, 0);
    poff1[it] = off(inside && a.nsrc > 1, 1);
    poff2[it] = off(inside && a.nsrc > 2, 2);
  }
  const int ng0 = a.c[0] >> 4, ng1 = a.c[1] >> 4;
  const float* src0 = a.src[0];
  const float* src1 = a.src[1];
  const float* src2 = a.src[2];
  // this wave reads xi = 4 * wave .. 4 * wave + 3, both parts
  const uint16_t* wbase = a.wp + ((size_t)cb * a.groups * 16 + wave * 4) * (2 * kFragHalves) + lane * 8;

  f32x16 acc[4];
#pragma unroll
  for (int b = 0; b < 4; ++b)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[b][r] = 0.f;

  // the two patch rows transform row `wave` combines:  0: d0 - d2   1: d1 + d2   2: d2 - d1   3: d1 - d3
  const int r1 = (wave == 0) ? 0 : ((wave == 2) ? 2 : 1);
  const int r2 = (wave == 0) ? 2 : ((wave == 1) ? 2 : ((wave == 2) ? 1 : 3));
  const float sgn = (wave == 1) ? 1.0f : -1.0f;
  // LDS offsets (quad 2 kb) of the eight window positions this lane reads per group; quad 2 kb + 1 is one quad stride on
  int wo1[4], wo2[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    wo1[c] = lds_off(2 * kb, 2 * ty + r1, 2 * tx + c);
    wo2[c] = lds_off(2 * kb, 2 * ty + r2, 2 * tx + c);
  }
  constexpr int kQuadStride = 2 * kPH * kRowPitch * 4;  // floats between consecutive quads

  float4 patch[kNLoad];
  half8 wh[4], wl[4];
#pragma unroll
  for (int it = 0; it < kNLoad; ++it) patch[it] = make_float4(0.f, 0.f, 0.f, 0.f);
#define DTS_PREFETCH(G)                                                                              \
  do {                                                                                               \
    const int g_ = (G);                                                                              \
    const int sidx = (g_ < ng0) ? 0 : ((g_ < ng0 + ng1) ? 1 : 2);                                    \
    const int gl = (sidx == 0) ? g_ : ((sidx == 1) ? g_ - ng0 : g_ - ng0 - ng1);                     \
    const float* sp = ((sidx == 0) ? src0 : ((sidx == 1) ? src1 : src2)) + gl * 16;                  \
    _Pragma("unroll") for (int it = 0; it < kNLoad; ++it) {                                          \
      const int off = (sidx == 0) ? poff0[it] : ((sidx == 1) ? poff1[it] : poff2[it]);               \
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);                                                    \
      if (off >= 0) v = *reinterpret_cast<const float4*>(sp + off);                                  \
      patch[it] = v;                                                                                 \
    }                                                                                                \
    const uint16_t* wg = wbase + (size_t)g_ * (16 * 2 * kFragHalves);                                \
    _Pragma("unroll") for (int b = 0; b < 4; ++b) {                                                  \
      wh[b] = *reinterpret_cast<const half8*>(wg + (b * 2 + 0) * kFragHalves);                       \
      wl[b] = *reinterpret_cast<const half8*>(wg + (b * 2 + 1) * kFragHalves);                       \
    }                                                                                                \
  } while (0)

  DTS_PREFETCH(0);
  for (int g = 0; g < a.groups; ++g) {
    float* buf = lds + (g & 1) * kPatchFloats;
#pragma unroll
    for (int it = 0; it < kNLoad; ++it) {
      const int idx = (tid + it * 256) >> 2;
      if (idx < kNPix) *reinterpret_cast<float4*>(buf + lds_off(tid & 3, idx / kPW, idx % kPW)) = patch[it];
    }
    half8 ah[4], al[4];
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      ah[b] = wh[b];
      al[b] = wl[b];
    }
    __syncthreads();  // patch visible; everyone is done with this buffer from two iterations ago
    if (g + 1 < a.groups) DTS_PREFETCH(g + 1);
    // B^T d B restricted to transform row `wave`, for this lane's 8 channels (quads 2 kb and 2 kb + 1)
    float tcol[4][8];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const float4 u0 = *reinterpret_cast<const float4*>(buf + wo1[c]);
      const float4 u1 = *reinterpret_cast<const float4*>(buf + wo1[c] + kQuadStride);
      const float4 v0 = *reinterpret_cast<const float4*>(buf + wo2[c]);
      const float4 v1 = *reinterpret_cast<const float4*>(buf + wo2[c] + kQuadStride);
      tcol[c][0] = u0.x + sgn * v0.x; tcol[c][1] = u0.y + sgn * v0.y; tcol[c][2] = u0.z + sgn * v0.z; tcol[c][3] = u0.w + sgn * v0.w;
      tcol[c][4] = u1.x + sgn * v1.x; tcol[c][5] = u1.y + sgn * v1.y; tcol[c][6] = u1.z + sgn * v1.z; tcol[c][7] = u1.w + sgn * v1.w;
    }
    SplitB V[4];
    {
      float v0[8], v1[8], v2[8], v3[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        v0[e] = tcol[0][e] - tcol[2][e];
        v1[e] = tcol[1][e] + tcol[2][e];
        v2[e] = tcol[2][e] - tcol[1][e];
        v3[e] = tcol[1][e] - tcol[3][e];
      }
      V[0] = split8(v0);
      V[1] = split8(v1);
      V[2] = split8(v2);
      V[3] = split8(v3);
    }
    // part-major issue order: the three partial products of one position are dependent (same accumulator), the four
    // positions are independent chains
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[b], V[b].hi, acc[b], 0, 0, 0);
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[b], V[b].lo, acc[b], 0, 0, 0);
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[b], V[b].hi, acc[b], 0, 0, 0);
  }
#undef DTS_PREFETCH

  // ---- inverse transform: columns in registers, rows across the four waves through LDS (as conv_wino_body) ----------
  // C/D layout of the 32x32 MFMA: lane (j, kb) register r = output channel (r & 3) + 8 (r >> 2) + 4 kb of the block
  __syncthreads();  // all waves are done with the patch buffers
  float* zb = lds + wave * 2048;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const float z0 = acc[0][r] + acc[1][r] + acc[2][r];
    const float z1 = acc[1][r] - acc[2][r] - acc[3][r];
    zb[(r >> 2) * 256 + lane * 4 + (r & 3)] = z0;
    zb[1024 + (r >> 2) * 256 + lane * 4 + (r & 3)] = z1;
  }
  __syncthreads();
  const int p = wave >> 1, q = wave & 1;  // output sub-pixel of every tile this wave finishes
  const int oy = by * 2 * kTH + 2 * ty + p, ox = bx * 2 * kTW + 2 * tx + q;
  const bool in_image = oy < a.h_out && ox < a.w_out;
  const size_t pix_off = (((size_t)n * a.h_out + oy) * a.w_out + ox) * a.c_out;
#pragma unroll
  for (int qd = 0; qd < 4; ++qd) {
    const float* z = lds + q * 1024 + qd * 256 + lane * 4;
    const float4 za = *reinterpret_cast<const float4*>(z + (p ? 1 : 0) * 2048);
    const float4 zbv = *reinterpret_cast<const float4*>(z + (p ? 2 : 1) * 2048);
    const float4 zc = *reinterpret_cast<const float4*>(z + (p ? 3 : 2) * 2048);
    float4 o;
    if (p == 0) {
      o = make_float4(za.x + zbv.x + zc.x, za.y + zbv.y + zc.y, za.z + zbv.z + zc.z, za.w + zbv.w + zc.w);
    } else {
      o = make_float4(za.x - zbv.x - zc.x, za.y - zbv.y - zc.y, za.z - zbv.z - zc.z, za.w - zbv.w - zc.w);
    }
    if (in_image) {
      const int co = cb * 32 + qd * 8 + kb * 4;
      const size_t off = pix_off + co;
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

// OIHW 3x3 weights -> U = G g G^T per (co, ci), split into fp16 hi / lo, packed [cb][g16][xi][part][lane][8 halves]:
// lane (i, kb) of fragment (cb, g, xi) holds U[co = 32 cb + i][ci = 16 g + 8 kb + e][xi], e = 0..7
__global__ void conv_wino_split_pack_kernel(const float* __restrict__ W, uint16_t* __restrict__ packed, int c_out, int c_in) {
  const int groups = c_in >> 4;
  const size_t total = (size_t)c_out * c_in * 16;  // (co, ci, xi) triples
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
    size_t r = idx;
    const int e = r & 7;
    r >>= 3;
    const int lane = r & 63;
    r >>= 6;
    const int xi = r & 15;
    r >>= 4;
    const int g = r % groups;
    const int cb = (int)(r / groups);
    const int co = cb * 32 + (lane & 31), ci = g * 16 + (lane >> 5) * 8 + e;
    const float* k = W + ((size_t)co * c_in + ci) * 9;
    const int ra = xi >> 2, rb = xi & 3;
    float col[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float k0 = k[0 * 3 + c], k1 = k[1 * 3 + c], k2 = k[2 * 3 + c];
      col[c] = (ra == 0) ? k0 : ((ra == 1) ? 0.5f * (k0 + k1 + k2) : ((ra == 2) ? 0.5f * (k0 - k1 + k2) : k2));
    }
    const float u = (rb == 0) ? col[0]
                              : ((rb == 1) ? 0.5f * (col[0] + col[1] + col[2])
                                           : ((rb == 2) ? 0.5f * (col[0] - col[1] + col[2]) : col[2]));
    // the same truncating split as the kernel applies to V
    const float hi_f = __uint_as_float(__float_as_uint(u) & 0xFFFFE000u);
    const _Float16 hi = (_Float16)hi_f;  // exact: 11 significant bits (values below the fp16 normal range lose bits: < 6e-8 abs)
    const _Float16 lo = (_Float16)(u - (float)hi);
    const size_t frag = (((size_t)cb * groups + g) * 16 + xi) * 2;
    union {
      _Float16 h;
      uint16_t u16;
    } cv;
    cv.h = hi;
    packed[(frag + 0) * kFragHalves + lane * 8 + e] = cv.u16;
    cv.h = lo;
    packed[(frag + 1) * kFragHalves + lane * 8 + e] = cv.u16;
  }
}

}  // namespace ws
}  // namespace dt

using namespace dt;

extern "C" {

int64_t dt_conv_wino_split_pack_halves(int c_out, int c_in) { return (int64_t)c_out * c_in * 16 * 2; }

int dt_conv_wino_split_pack_f16(const float* W, uint16_t* packed, int c_out, int c_in, dt_stream_t s) {
  DT_REQUIRE(W && packed, "dt_conv_wino_split_pack_f16: null pointer");
  DT_REQUIRE(c_out > 0 && c_out % 32 == 0, "dt_conv_wino_split_pack_f16: c_out=%d must be a multiple of 32", c_out);
  DT_REQUIRE(c_in > 0 && c_in % 16 == 0, "dt_conv_wino_split_pack_f16: c_in=%d must be a multiple of 16", c_in);
  const size_t total = (size_t)c_out * c_in * 16;
  const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
  DT_LAUNCH(ws::conv_wino_split_pack_kernel, dim3(blocks), dim3(256), 0, to_stream(s), W, packed, c_out, c_in);
  return check_launch("dt_conv_wino_split_pack_f16");
}

int dt_conv2d_wino_split_supported(const dt_conv_desc* d) {
  if (!d || d->ksize != 3 || d->stride != 1 || d->nsrc < 1 || d->nsrc > 3 || d->c_out <= 0 || d->c_out % 32 != 0) return 0;
  for (int s = 0; s < d->nsrc; ++s)
    if (d->c[s] <= 0 || d->c[s] % 16 != 0) return 0;
  return 1;
}

int dt_conv2d_wino_split_f32(const dt_conv_desc* d, const float* in0, const float* in1, const float* in2,
                             const uint16_t* packed_w, const float* bias, const float* residual, float* out, dt_stream_t s) {
  DT_REQUIRE(d != nullptr, "dt_conv2d_wino_split_f32: null descriptor");
  DT_REQUIRE(dt_conv2d_wino_split_supported(d), "dt_conv2d_wino_split_f32: needs a 3x3 stride-1 convolution with c_out %% 32 == 0 "
                                                "and every source a multiple of 16 channels");
  DT_REQUIRE(d->n > 0 && d->h_out > 0 && d->w_out > 0 && d->h_in == d->h_out && d->w_in == d->w_out,
             "dt_conv2d_wino_split_f32: bad extents");
  DT_REQUIRE(d->act >= 0 && d->act <= 3 && (d->pad_mode == 0 || d->pad_mode == 1) && d->transposed == 0,
             "dt_conv2d_wino_split_f32: bad act / pad_mode / transposed");
  DT_REQUIRE(packed_w && out, "dt_conv2d_wino_split_f32: null pointer");
  const float* ins[3] = {in0, in1, in2};
  ws::Args a;
  a.groups = 0;
  for (int i = 0; i < 3; ++i) {
    a.src[i] = nullptr;
    a.c[i] = 0;
    a.up[i] = 0;
  }
  for (int i = 0; i < d->nsrc; ++i) {
    DT_REQUIRE(ins[i] != nullptr, "dt_conv2d_wino_split_f32: source %d is null", i);
    DT_REQUIRE(!d->up[i] || (d->h_in % 2 == 0 && d->w_in % 2 == 0), "dt_conv2d_wino_split_f32: upsampled source needs even extents");
    a.src[i] = ins[i];
    a.c[i] = d->c[i];
    a.up[i] = d->up[i] ? 1 : 0;
    a.groups += d->c[i] >> 4;
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
  const long wt_x = (a.w_out + 2 * ws::kTW - 1) / (2 * ws::kTW), wt_y = (a.h_out + 2 * ws::kTH - 1) / (2 * ws::kTH);
  const long blocks = (long)a.n * wt_y * wt_x * a.co_blocks;
  DT_REQUIRE(blocks < 2147483647L, "dt_conv2d_wino_split_f32: grid too large");
  DT_LAUNCH(ws::conv_wino_split_kernel, dim3((unsigned)blocks), dim3(256), 0, to_stream(s), a);
  return check_launch("dt_conv2d_wino_split_f32");
}

}  // extern "C"
