// OPT-IN split-precision variant of the fused plane-sweep + matching-MLP (+ hint-MLP) volume, gfx950.
//
// Same function as csrc/cv_mlp_mfma.hip (reference modules/feature_volume.py:81-356, modules/mesh_hint_volume.py:84-393,
// Fast variant :679-928), same geometry / gather / hint code, but the two dense contractions (Cin -> 128 -> 128) run on
// the fp16 matrix pipe, which is 16x the fp32 MFMA rate on CDNA4, with every operand split into two halves
//     x = x_hi + x_lo,   x_hi = fp16(x) (top 11 significant bits),   x_lo = fp16(x - x_hi)
//     x * w  ~=  x_hi*w_hi + x_lo*w_hi + x_hi*w_lo          (fp32 accumulation inside v_mfma_f32_32x32x16_f16)
// The dropped x_lo*w_lo term is 2^-22 relative, i.e. fp32-class accuracy at 3/16 of the fp32 matrix time.  It is NOT the
// default: the headline path (dt_cv_mlp_hint_f32) stays exact fp32.  Range assumption: |inputs|, |weights|, |hidden
// activations| < 65504 (fp16 max) -- true for instance-normalised matching features, metric depths and the released
// checkpoints; values below 6e-5 keep an absolute error of 6e-8.  Parity budget: tests/test_volume_gpu.py compares with
// the same reference goldens as the fp32 kernel (2e-4 instead of 5e-5).
//
// Mapping onto v_mfma_f32_32x32x16_f16 (D[i][j] += A[i][k] B[k][j], k = 0..15; layout verified on hardware by
// scripts/mfma_f16_layout_check.hip): lane l = (row/col l & 31, k-block l >> 5); the A operand of a lane is
// A[i][8*kb .. 8*kb+7], the B operand B[8*kb .. 8*kb+7][j], D as in the fp32 kernel (register r of lane (j, kb) = feature
// (r&3) + 8*(r>>2) + 4*kb).  So a lane half feeds EIGHT k-slots per step where the fp32 kernel feeds one:
//   * "F" step of view k: slot (kb, e) = warped channel 8*kb + e -- exactly the 8 channels the lane half gathers;
//   * "M" step of a view PAIR (2m, 2m+1): slots 0..3 = view 2m's metadata of this half, 4..7 = view 2m+1's
//     (half 0: mask, dot*mask, ray.x, ray.z;  half 1: z', angle, ray.y, plane depth for view 0 / unused);
//   * layer 2, step (block i, q): slot (kb, e) = layer-1 accumulator register 8q + e of block i in that lane, i.e. the
//     activations are already where the next layer needs them (no transpose, no LDS round trip), as in the fp32 kernel;
//   * plane-independent columns (current features, current ray, bias, pose metrics) are contracted once per pixel tile.
// The K order of a GEMM is a free permutation: doubletake_amd/modules/mlp_pack.py (pack_mlp_split) builds the weight
// fragments for exactly these slot tables, as fp16 hi and lo parts.
// LDS per workgroup (K = 7): (7 F + 4 M) steps x 8 KB + layer 2 64 KB + tail/hint 2 KB + store staging 4 KB = 158 KB.
#include <cstdlib>

#include "common.hpp"
#include "cv_geometry.hpp"

namespace dt {
namespace sp {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));

#ifndef DT_SPLIT_RTZ
#define DT_SPLIT_RTZ 1  // operand split by truncation (3 VALU per value) instead of round-to-nearest (4)
#endif

constexpr int kF = 16;
constexpr int kFragHalves = 64 * 8;         // one A fragment: 64 lanes x 8 halves = 1 KB
constexpr int kStepHalves = 2 * 4 * kFragHalves;  // [part hi|lo][4 output blocks]
constexpr int kW2Steps = 8;
constexpr int kTailFloats = 260;   // b2r[128], w3r[128], b3, pad[3]
constexpr int kHintFloats = 220;
constexpr int kStageFloats = 128;  // per wave: 32 pixels x 4 planes of finished scores awaiting a 16-byte store
constexpr int kMaxSrc = 7;

__host__ __device__ inline int dyn_steps(int K) { return K + (K + 1) / 2; }
__host__ __device__ inline int pix_steps(int K) { return 1 + (4 + 3 * K + 15) / 16; }

struct Args {
  const float* cur;
  const float* src;
  const float* params;
  const uint16_t* w1dyn;  // [dyn_steps][2][4][64][8] halves
  const uint16_t* w1pix;  // [pix_steps][2][4][64][8] halves (read from global)
  const uint16_t* w2;     // [8][2][4][64][8] halves
  const float* tail;
  const float* hint_mlp;
  const float* hint_d;
  const float* hint_w;
  const float* hint_m;
  float* vol;
  int hint_h, hint_w2;
  int out_nhwc;
  int B, K, h, w, D;
  int num_tiles;
  long total_units;
};
}  // namespace sp
DT_ARG_POINTERS(sp::Args, offsetof(sp::Args, cur), offsetof(sp::Args, src), offsetof(sp::Args, params), offsetof(sp::Args, w1dyn),
                offsetof(sp::Args, w1pix), offsetof(sp::Args, w2), offsetof(sp::Args, tail), offsetof(sp::Args, hint_mlp),
                offsetof(sp::Args, hint_d), offsetof(sp::Args, hint_w), offsetof(sp::Args, hint_m), offsetof(sp::Args, vol));
namespace sp {

struct ViewData {
  float4 t00a, t00b, t01a, t01b, t10a, t10b, t11a, t11b;
  float w00, w01, w10, w11;
  float z, sx, sy, sz, ang;
};

__device__ __forceinline__ void issue_view(ViewData& v, cfloat_ptr vp, const float* __restrict__ src_view, float X, float Y,
                                           float Z, float crx, float cry, float crz, int h, int w, float inv_w, float inv_h,
                                           int half) {
  const ViewProj q = project_view(vp, X, Y, Z);
  const Taps t = bilinear_taps(q.u, q.v, h, w, inv_w, inv_h);
  const float* p00 = src_view + ((size_t)t.y0 * w + t.x0) * kF + half * 8;
  const float* p01 = src_view + ((size_t)t.y0 * w + t.x1) * kF + half * 8;
  const float* p10 = src_view + ((size_t)t.y1 * w + t.x0) * kF + half * 8;
  const float* p11 = src_view + ((size_t)t.y1 * w + t.x1) * kF + half * 8;
  v.t00a = reinterpret_cast<const float4*>(p00)[0];
  v.t00b = reinterpret_cast<const float4*>(p00)[1];
  v.t01a = reinterpret_cast<const float4*>(p01)[0];
  v.t01b = reinterpret_cast<const float4*>(p01)[1];
  v.t10a = reinterpret_cast<const float4*>(p10)[0];
  v.t10b = reinterpret_cast<const float4*>(p10)[1];
  v.t11a = reinterpret_cast<const float4*>(p11)[0];
  v.t11b = reinterpret_cast<const float4*>(p11)[1];
  v.w00 = t.w00;
  v.w01 = t.w01;
  v.w10 = t.w10;
  v.w11 = t.w11;
  v.z = q.z;
  const float sx = X - vp[12], sy = Y - vp[13], sz = Z - vp[14];
  const float inv = rsqrtf(fmaxf(sx * sx + sy * sy + sz * sz, 1e-24f));
  v.sx = sx * inv;
  v.sy = sy * inv;
  v.sz = sz * inv;
  v.ang = crx * v.sx + cry * v.sy + crz * v.sz;
}

// hint MLP [3,12,12,1] from LDS, hidden units split between the two lane halves (same scheme as the fp32 kernel)
__device__ __forceinline__ float hint_mlp_eval_lds(const float* hm, float s, float hint, float hw, int half) {
  const int m0 = half * 6, o0 = 6 - m0;
  float own[6], oth[6];
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    const int m = m0 + j;
    own[j] = lrelu(hm[m * 3 + 0] * s + hm[m * 3 + 1] * hint + hm[m * 3 + 2] * hw + hm[36 + m], 0.01f);
  }
#pragma unroll
  for (int j = 0; j < 6; ++j) oth[j] = __shfl_xor(own[j], 32, 64);
  float part = 0.f;
#pragma unroll 1
  for (int j = 0; j < 6; ++j) {
    const int n = m0 + j;
    const float2* ro = reinterpret_cast<const float2*>(hm + 48 + n * 12 + m0);
    const float2* rx = reinterpret_cast<const float2*>(hm + 48 + n * 12 + o0);
    const float2 a0 = ro[0], a1 = ro[1], a2 = ro[2], b0 = rx[0], b1 = rx[1], b2 = rx[2];
    float acc = hm[192 + n];
    acc += a0.x * own[0] + a0.y * own[1] + a1.x * own[2] + a1.y * own[3] + a2.x * own[4] + a2.y * own[5];
    acc += b0.x * oth[0] + b0.y * oth[1] + b1.x * oth[2] + b1.y * oth[3] + b2.x * oth[4] + b2.y * oth[5];
    part += hm[204 + n] * lrelu(acc, 0.01f);
  }
  return part + __shfl_xor(part, 32, 64) + hm[216];
}

// x[0..7] (fp32) -> hi / lo fp16 operands: hi = fp16(x), lo = fp16(x - hi) (the difference is exact in fp32).
struct SplitB {
  half8 hi, lo;
};
__device__ __forceinline__ SplitB split8(const float (&x)[8]) {
#if DT_SPLIT_RTZ
  // hi = x truncated to 11 significant bits (one v_cvt_pkrtz_f16_f32 per pair); its fp32 value is x with the low 13
  // mantissa bits cleared (one v_and), so the remainder costs one v_sub and is exact: 3 instructions per value
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
#else
  SplitB s;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const _Float16 hi = (_Float16)x[i];          // round to nearest: |x - hi| <= 2^-11 |x|
    s.hi[i] = hi;
    s.lo[i] = (_Float16)(x[i] - (float)hi);      // exact difference, rounded once
  }
  return s;
#endif
}

// Issue order: the three partial products of one output block are dependent MFMAs (same accumulator, ~50-cycle latency),
// so the loop nest is part-major: four independent chains in flight at any time.
#define DT_SPLIT_STEP(ACC, WSTEP, B)                                                                              \
  do {                                                                                                            \
    half8 ah_[4], al_[4];                                                                                         \
    _Pragma("unroll") for (int cb_ = 0; cb_ < 4; ++cb_) {                                                         \
      ah_[cb_] = *reinterpret_cast<const half8*>((WSTEP) + (0 * 4 + cb_) * kFragHalves + lane * 8);                \
      al_[cb_] = *reinterpret_cast<const half8*>((WSTEP) + (1 * 4 + cb_) * kFragHalves + lane * 8);                \
    }                                                                                                             \
    _Pragma("unroll") for (int cb_ = 0; cb_ < 4; ++cb_)                                                           \
      ACC[cb_] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al_[cb_], (B).hi, ACC[cb_], 0, 0, 0);                     \
    _Pragma("unroll") for (int cb_ = 0; cb_ < 4; ++cb_)                                                           \
      ACC[cb_] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah_[cb_], (B).lo, ACC[cb_], 0, 0, 0);                     \
    _Pragma("unroll") for (int cb_ = 0; cb_ < 4; ++cb_)                                                           \
      ACC[cb_] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah_[cb_], (B).hi, ACC[cb_], 0, 0, 0);                     \
  } while (0)

template <bool HINT>
__global__ __launch_bounds__(512, 2) void cv_mlp_split_kernel(const Args a) {
  constexpr int NW = 8, NT = NW * 64;
  extern __shared__ __attribute__((aligned(16))) uint16_t lds_h[];
  const int K = a.K, D = a.D, h = a.h, w = a.w;
  const int n_dyn = dyn_steps(K) * kStepHalves;
  const _Float16* lds_w1 = reinterpret_cast<const _Float16*>(lds_h);
  const _Float16* lds_w2 = lds_w1 + n_dyn;
  float* lds_tail = reinterpret_cast<float*>(lds_h + n_dyn + kW2Steps * kStepHalves);
  float* lds_stage = lds_tail + kTailFloats + kHintFloats + (threadIdx.x >> 6) * kStageFloats;

  {
    const uint4* g1 = reinterpret_cast<const uint4*>(a.w1dyn);
    uint4* l1 = reinterpret_cast<uint4*>(lds_h);
    for (int i = threadIdx.x; i < n_dyn / 8; i += NT) l1[i] = g1[i];
    const uint4* g2 = reinterpret_cast<const uint4*>(a.w2);
    uint4* l2 = reinterpret_cast<uint4*>(lds_h + n_dyn);
    for (int i = threadIdx.x; i < kW2Steps * kStepHalves / 8; i += NT) l2[i] = g2[i];
    const float4* g3 = reinterpret_cast<const float4*>(a.tail);
    float4* l3 = reinterpret_cast<float4*>(lds_tail);
    for (int i = threadIdx.x; i < kTailFloats / 4; i += NT) l3[i] = g3[i];
    if (HINT)
      for (int i = threadIdx.x; i < 217; i += NT) lds_tail[kTailFloats + i] = a.hint_mlp[i];
  }
  __syncthreads();

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int half = lane >> 5, pl = lane & 31;
  const size_t hw = (size_t)h * w;
  const float inv_w = 1.0f / (float)w, inv_h = 1.0f / (float)h;
  const long waves_total = (long)gridDim.x * NW;
  const float b3 = lds_tail[256];
  const bool stage_ok = (D % 4 == 0);

  // balanced span partition in an XCD-aware order (as in the fp32 kernel)
  const int nblk = gridDim.x;
  const int lbid = (nblk % 8 == 0) ? (int)(blockIdx.x % 8) * (nblk / 8) + (int)(blockIdx.x / 8) : (int)blockIdx.x;
  const long wid = (long)lbid * NW + wave;
  long u = wid * a.total_units / waves_total;
  const long u_end = (wid + 1) * a.total_units / waves_total;
  while (u < u_end) {
    const long tile_global = u / D;
    const int d0 = __builtin_amdgcn_readfirstlane((int)(u - tile_global * D));
    const int d1 = __builtin_amdgcn_readfirstlane((int)min((long)D, d0 + (u_end - u)));
    const int tile = __builtin_amdgcn_readfirstlane((int)(tile_global % a.num_tiles));
    const int b = __builtin_amdgcn_readfirstlane((int)(tile_global / a.num_tiles));
    u += d1 - d0;
    const cfloat_ptr p = as_const(a.params + (size_t)b * cv_params_floats(D, K));
    const float* src_b = a.src + (size_t)b * K * hw * kF;

    const size_t pixi = (size_t)tile * 32 + pl;
    const bool live = pixi < hw;
    const size_t pc = live ? pixi : hw - 1;
    const int y = (int)(pc / w), x = (int)(pc % w);

    float cur8[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) cur8[j] = a.cur[((size_t)b * kF + half * 8 + j) * hw + pc];
    float rx, ry, rz;
    pixel_ray(p + kCvInvK, x, y, rx, ry, rz);
    float crx = rx, cry = ry, crz = rz;
    normalize3(crx, cry, crz);

    // ---- plane-independent part: step 0 = current features, then the column list
    //      [ray.x, ray.y, ray.z, 1 (bias), pd_0, R_0, t_0, pd_1, R_1, t_1, ...] 16 per step ------------------------------
    f32x16 accp[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) accp[i][r] = 0.f;
    {
      const _Float16* wp = reinterpret_cast<const _Float16*>(a.w1pix);
      {
        const SplitB bq = split8(cur8);
        DT_SPLIT_STEP(accp, wp, bq);
      }
      const int nps = pix_steps(K);
      for (int st = 1; st < nps; ++st) {
        float xv[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int n = (st - 1) * 16 + half * 8 + e;  // index into the column list
          float v = 0.f;
          if (n == 0) v = crx;
          else if (n == 1) v = cry;
          else if (n == 2) v = crz;
          else if (n == 3) v = 1.0f;
          else if (n < 4 + 3 * K) v = a.params[(size_t)b * cv_params_floats(D, K) + cv_view_off(D, (n - 4) / 3) + 15 + (n - 4) % 3];
          xv[e] = v;
        }
        const SplitB bq = split8(xv);
        DT_SPLIT_STEP(accp, wp + (size_t)st * kStepHalves, bq);
      }
    }

    bool hmask = false;
    float hdepth = 0.f, hweight = 0.f;
    if (HINT) {
      const int sy = nearest_src(y, a.hint_h, h), sx = nearest_src(x, a.hint_w2, w);
      const size_t hi = ((size_t)b * a.hint_h + sy) * a.hint_w2 + sx;
      hmask = a.hint_m[hi] != 0.f;
      hdepth = a.hint_d[hi];
      hweight = hmask ? a.hint_w[hi] : 0.f;
    }

    ViewData v;
    {
      const float depth = p[kCvPlanes + d0];
      issue_view(v, p + cv_view_off(D, 0), src_b, depth * rx, depth * ry, depth * rz, crx, cry, crz, h, w, inv_w, inv_h, half);
    }
    for (int d = d0; d < d1; ++d) {
      const float depth = p[kCvPlanes + d];
      f32x16 acc1[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) acc1[i] = accp[i];

      float meta[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) meta[e] = 0.f;
      for (int k = 0; k < K; ++k) {
        float f[8];
        f[0] = v.t00a.x * v.w00 + v.t01a.x * v.w01 + v.t10a.x * v.w10 + v.t11a.x * v.w11;
        f[1] = v.t00a.y * v.w00 + v.t01a.y * v.w01 + v.t10a.y * v.w10 + v.t11a.y * v.w11;
        f[2] = v.t00a.z * v.w00 + v.t01a.z * v.w01 + v.t10a.z * v.w10 + v.t11a.z * v.w11;
        f[3] = v.t00a.w * v.w00 + v.t01a.w * v.w01 + v.t10a.w * v.w10 + v.t11a.w * v.w11;
        f[4] = v.t00b.x * v.w00 + v.t01b.x * v.w01 + v.t10b.x * v.w10 + v.t11b.x * v.w11;
        f[5] = v.t00b.y * v.w00 + v.t01b.y * v.w01 + v.t10b.y * v.w10 + v.t11b.y * v.w11;
        f[6] = v.t00b.z * v.w00 + v.t01b.z * v.w01 + v.t10b.z * v.w10 + v.t11b.z * v.w11;
        f[7] = v.t00b.w * v.w00 + v.t01b.w * v.w01 + v.t10b.w * v.w10 + v.t11b.w * v.w11;
        const float vz = v.z, vang = v.ang, vsx = v.sx, vsy = v.sy, vsz = v.sz;
        {
          int nk = k + 1, nd = d;
          if (nk == K) {
            nk = 0;
            nd = d + 1;
          }
          nd = min(nd, d1 - 1);
          const float ndepth = p[kCvPlanes + nd];
          issue_view(v, p + cv_view_off(D, nk), src_b + (size_t)nk * hw * kF, ndepth * rx, ndepth * ry, ndepth * rz, crx, cry,
                     crz, h, w, inv_w, inv_h, half);
        }
        float dotp = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) dotp += f[j] * cur8[j];
        const float dot = dotp + __shfl_xor(dotp, 32, 64);
        const float m = (vz > 0.f) ? 1.f : 0.f;
        // this half's four metadata values of view k -> slots 0..3 (even k) / 4..7 (odd k) of the pair's M step
        const float m0 = half ? vz : m, m1 = half ? vang : dot * m, m2 = half ? vsy : vsx,
                    m3 = half ? ((k == 0) ? depth : 0.f) : vsz;
        if (k & 1) {
          meta[4] = m0; meta[5] = m1; meta[6] = m2; meta[7] = m3;
        } else {
          meta[0] = m0; meta[1] = m1; meta[2] = m2; meta[3] = m3;
          meta[4] = 0.f; meta[5] = 0.f; meta[6] = 0.f; meta[7] = 0.f;
        }
        {
          const SplitB bq = split8(f);
          DT_SPLIT_STEP(acc1, lds_w1 + (size_t)k * kStepHalves, bq);
        }
        if ((k & 1) || k == K - 1) {
          const SplitB bq = split8(meta);
          DT_SPLIT_STEP(acc1, lds_w1 + (size_t)(K + (k >> 1)) * kStepHalves, bq);
        }
      }

      // ---- layer-1 activation, then split the 64 activations of this lane into the 8 layer-2 B operands --------------
      SplitB h1[8];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          float t8[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const float z1 = acc1[i][8 * q + e];
            t8[e] = fmaxf(z1, 0.01f * z1);
          }
          h1[i * 2 + q] = split8(t8);
        }

      // ---- layer 2 in four passes of 32 output features; inside a pass the main (hi*hi) and the correction (lo*hi + hi*lo)
      //      products accumulate in two separate registers sets = two independent MFMA chains (a dependent 32x32x16 MFMA
      //      waits ~50 cycles, an independent one issues every ~40), summed before the activation.  Layer 3 on the VALU.
      float s = 0.f;
#pragma unroll
      for (int pass = 0; pass < 4; ++pass) {
        f32x16 accm, accc;
        const float* bl = lds_tail + half * 64 + pass * 16;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          accm[r] = bl[r];
          accc[r] = 0.f;
        }
#pragma unroll
        for (int t = 0; t < kW2Steps; ++t) {
          const _Float16* ws = lds_w2 + (size_t)t * kStepHalves;
          const half8 ah = *reinterpret_cast<const half8*>(ws + (0 * 4 + pass) * kFragHalves + lane * 8);
          const half8 al = *reinterpret_cast<const half8*>(ws + (1 * 4 + pass) * kFragHalves + lane * 8);
          accm = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, h1[t].hi, accm, 0, 0, 0);
          accc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, h1[t].hi, accc, 0, 0, 0);
          accc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, h1[t].lo, accc, 0, 0, 0);
        }
        const float* w3 = lds_tail + 128 + half * 64 + pass * 16;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float v2 = accm[r] + accc[r];
          s += w3[r] * fmaxf(v2, 0.01f * v2);
        }
      }
      s += __shfl_xor(s, 32, 64);
      s += b3;
      if (HINT) {
        const float hint = hmask ? fabsf(hdepth - depth) : -1.f;
        asm volatile("" ::: "memory");
        s = hint_mlp_eval_lds(lds_tail + kTailFloats, s, hint, hweight, half);
      }
      if (!a.out_nhwc) {
        if (live && half == 0) a.vol[((size_t)b * D + d) * hw + pixi] = s;
      } else if (!stage_ok) {
        if (live && half == 0) a.vol[((size_t)b * hw + pixi) * D + d] = s;
      } else {
        // NHWC volume: park the scores of up to 4 consecutive planes in LDS, write 16 contiguous bytes per pixel
        if (half == 0) lds_stage[pl * 4 + (d & 3)] = s;
        if ((d & 3) == 3 || d == d1 - 1) {
          __builtin_amdgcn_wave_barrier();
          const int cbase = d & ~3;
          const int lo = max(d0, cbase) - cbase, hi = d - cbase;
          if (lane < 32) {
            const float4 v4 = *reinterpret_cast<const float4*>(lds_stage + lane * 4);
            const size_t spix = (size_t)tile * 32 + lane;
            if (spix < hw) {
              float* dst = a.vol + ((size_t)b * hw + spix) * D + cbase;
              if (lo == 0 && hi == 3) {
                *reinterpret_cast<float4*>(dst) = v4;
              } else {
                if (lo <= 0 && hi >= 0) dst[0] = v4.x;
                if (lo <= 1 && hi >= 1) dst[1] = v4.y;
                if (lo <= 2 && hi >= 2) dst[2] = v4.z;
                if (lo <= 3 && hi >= 3) dst[3] = v4.w;
              }
            }
          }
          __builtin_amdgcn_wave_barrier();
        }
      }
    }
  }
}

static int num_cus() { return dt::device_cu_count(); }

}  // namespace sp
}  // namespace dt

using namespace dt;

extern "C" {

int dt_cv_mlp_split_pack_halves(int num_src, int* w1dyn, int* w1pix, int* w2) {
  DT_REQUIRE(num_src > 0 && num_src <= sp::kMaxSrc, "dt_cv_mlp_split_pack_halves: num_src=%d not in 1..%d", num_src, sp::kMaxSrc);
  if (w1dyn) *w1dyn = sp::dyn_steps(num_src) * sp::kStepHalves;
  if (w1pix) *w1pix = sp::pix_steps(num_src) * sp::kStepHalves;
  if (w2) *w2 = sp::kW2Steps * sp::kStepHalves;
  return 0;
}

int dt_cv_mlp_hint_split_f32(const float* cur, const float* src, const float* params, const uint16_t* w1dyn,
                             const uint16_t* w1pix, const uint16_t* w2, const float* tail, const float* hint_mlp,
                             const float* depth_hint, const float* hint_weights, const float* hint_mask, int hint_h, int hint_w,
                             float* volume, int out_nhwc, int batch, int num_src, int h, int w, int num_planes, dt_stream_t s) {
  DT_REQUIRE(batch > 0 && h > 0 && w > 0 && num_planes > 0, "dt_cv_mlp_hint_split_f32: bad extents");
  DT_REQUIRE(num_src > 0 && num_src <= sp::kMaxSrc, "dt_cv_mlp_hint_split_f32: num_src=%d not in 1..%d", num_src, sp::kMaxSrc);
  DT_REQUIRE(cur && src && params && w1dyn && w1pix && w2 && tail && volume, "dt_cv_mlp_hint_split_f32: null pointer");
  DT_REQUIRE(hint_mlp == nullptr || (depth_hint && hint_weights && hint_mask && hint_h > 0 && hint_w > 0),
             "dt_cv_mlp_hint_split_f32: hint MLP given without hint maps");
  sp::Args a;
  a.cur = cur; a.src = src; a.params = params; a.w1dyn = w1dyn; a.w1pix = w1pix; a.w2 = w2; a.tail = tail;
  a.hint_mlp = hint_mlp; a.hint_d = depth_hint; a.hint_w = hint_weights; a.hint_m = hint_mask;
  a.vol = volume; a.hint_h = hint_h; a.hint_w2 = hint_w; a.out_nhwc = out_nhwc;
  a.B = batch; a.K = num_src; a.h = h; a.w = w; a.D = num_planes;
  const long hw = (long)h * w;
  a.num_tiles = (int)((hw + 31) / 32);
  a.total_units = (long)batch * a.num_tiles * num_planes;
  const int cus = sp::num_cus();
  const size_t lds_bytes = (size_t)(sp::dyn_steps(num_src) + sp::kW2Steps) * sp::kStepHalves * 2 +
                           (size_t)(sp::kTailFloats + sp::kHintFloats + 8 * sp::kStageFloats) * sizeof(float);
  const long want = (a.total_units + 7) / 8;
  const int blocks = (int)(want < cus ? want : cus);
#define DT_LAUNCH_SPLIT(HINT_)                                                                                       \
  do {                                                                                                               \
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&sp::cv_mlp_split_kernel<HINT_>),                  \
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);                  \
    if (e != hipSuccess) {                                                                                           \
      (void)hipGetLastError();                                                                                       \
      return fail("dt_cv_mlp_hint_split_f32: cannot reserve %zu B of LDS: %s", lds_bytes, hipGetErrorString(e));      \
    }                                                                                                                \
    DT_LAUNCH((sp::cv_mlp_split_kernel<HINT_>), dim3(blocks), dim3(512), lds_bytes, to_stream(s), a);         \
  } while (0)
  if (hint_mlp) DT_LAUNCH_SPLIT(true); else DT_LAUNCH_SPLIT(false);
#undef DT_LAUNCH_SPLIT
  return check_launch("dt_cv_mlp_hint_split_f32");
}

}  // extern "C"
