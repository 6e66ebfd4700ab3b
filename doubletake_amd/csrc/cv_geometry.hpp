// Per-pixel plane-sweep geometry shared by the cost-volume kernels.
//
// Restates (does not copy) the reference arithmetic:
//   BackprojectDepth      utils/geometry_utils.py:34-39,60-63   pix=(x+.5,y+.5,1); X = d*(invK3@pix)
//   Project3D             utils/geometry_utils.py:82-93         q = P@(X,1); z' = q.z+eps; uv = q.xy*(|q.z|>eps ? 1/z' : 1)
//   grid normalisation    modules/cost_volume.py:186            g = 2*uv*(1/w,1/h) - 1
//   F.grid_sample         bilinear / zeros / align_corners=False: idx = ((g+1)*size-1)/2, taps floor/floor+1
#pragma once
#include "common.hpp"

namespace dt {

// Read-only, wave-uniform parameter blocks are read through the constant address space so that the
// compiler emits scalar loads (s_load_dword -> SGPR operands) even in kernels that also store
// through unrelated global pointers.
typedef const float __attribute__((address_space(4))) * cfloat_ptr;
__device__ __forceinline__ cfloat_ptr as_const(const float* p) { return (cfloat_ptr)(uintptr_t)p; }

struct ViewProj {
  float u, v, z;  // source pixel coords (pixel centres at +0.5) and z' = depth + eps
};

// cam ray r = invK3 @ (x+0.5, y+0.5, 1)
template <class PT>
__device__ __forceinline__ void pixel_ray(PT invK3, int x, int y, float& rx, float& ry, float& rz) {
  const float px = (float)x + 0.5f, py = (float)y + 0.5f;
  rx = invK3[0] * px + invK3[1] * py + invK3[2];
  ry = invK3[3] * px + invK3[4] * py + invK3[5];
  rz = invK3[6] * px + invK3[7] * py + invK3[8];
}

template <class PT>
__device__ __forceinline__ ViewProj project_view(PT P, float X, float Y, float Z) {
  const float qx = P[0] * X + P[1] * Y + P[2] * Z + P[3];
  const float qy = P[4] * X + P[5] * Y + P[6] * Z + P[7];
  const float qz = P[8] * X + P[9] * Y + P[10] * Z + P[11];
  ViewProj r;
  r.z = qz + 1e-8f;
  const float s = (fabsf(qz) > 1e-8f) ? (1.0f / r.z) : 1.0f;
  r.u = qx * s;
  r.v = qy * s;
  return r;
}

// ---- round 5: the same geometry with fewer vector instructions (the tuned dot-product and MLP volume kernels) -----------------
// On gfx950 every vector instruction beside fp32 MFMAs costs matrix time, and the dot-product kernel is bound by their issue
// rate (bench.py: roofline_warp_match_dot.bound_by).  Two algebraic shortcuts, both far inside the parity budget (numpy
// oracle with the same substitution on a golden-sized case: matching-MLP volume changes by <= 3.3e-6 against its 5e-5 tolerance,
// dot-product volume by 5.5e-5 on a range of +-20):
//   * 1 / z' as v_rcp_f32 + one Newton step (<= 1 ulp; IEEE division is ~10 instructions, this is 3);
//   * grid_sample's index ((2 u / w - 1 + 1) w - 1) / 2 is u - 0.5 (8 instructions -> 2; differs by rounding only).
// The stand-alone warp_features kernel, the masks and the general (one thread per output) kernels keep the exact forms above.
#ifndef DT_FAST_GEOM
#define DT_FAST_GEOM 1
#endif
__device__ __forceinline__ float fast_rcp(float x) {
  const float r = __builtin_amdgcn_rcpf(x);
  return __builtin_fmaf(__builtin_fmaf(-x, r, 1.0f), r, r);
}
__device__ __forceinline__ void project_scale(float qx, float qy, float qz, ViewProj& r) {
  r.z = qz + 1e-8f;
  const float s = (fabsf(qz) > 1e-8f) ? (DT_FAST_GEOM ? fast_rcp(r.z) : 1.0f / r.z) : 1.0f;
  r.u = qx * s;
  r.v = qy * s;
}
template <class PT>
__device__ __forceinline__ ViewProj project_view_fast(PT P, float X, float Y, float Z) {
  const float qx = P[0] * X + P[1] * Y + P[2] * Z + P[3];
  const float qy = P[4] * X + P[5] * Y + P[6] * Z + P[7];
  const float qz = P[8] * X + P[9] * Y + P[10] * Z + P[11];
  ViewProj r;
  project_scale(qx, qy, qz, r);
  return r;
}
// pixel index of grid_sample(align_corners=False) for source pixel coordinate u (pixel centres at +0.5)
__device__ __forceinline__ float sample_index(float u, float size, float inv_size) {
  if (DT_FAST_GEOM) return u - 0.5f;
  const float g = 2.0f * u * inv_size - 1.0f;
  return ((g + 1.0f) * size - 1.0f) * 0.5f;
}

// Bilinear tap set for one sample: base texel (x0,y0), 4 weights already zeroed for
// out-of-bounds taps, and clamped in-bounds addresses so that loads are always legal.
struct Taps {
  int x0, y0, x1, y1;         // clamped to the image
  float w00, w01, w10, w11;   // (y0,x0) (y0,x1) (y1,x0) (y1,x1); 0 where the tap is outside
};

template <bool FAST = false>
__device__ __forceinline__ Taps bilinear_taps(float u, float v, int h, int w, float inv_w, float inv_h) {
  float ix, iy;
  if (FAST) {
    ix = sample_index(u, (float)w, inv_w);
    iy = sample_index(v, (float)h, inv_h);
  } else {
    const float gx = 2.0f * u * inv_w - 1.0f;
    const float gy = 2.0f * v * inv_h - 1.0f;
    ix = ((gx + 1.0f) * (float)w - 1.0f) * 0.5f;
    iy = ((gy + 1.0f) * (float)h - 1.0f) * 0.5f;
  }
  Taps t;
  // Anything that cannot touch the image (incl. NaN/inf) samples zero: such a sample gets the base texel -2, for which every
  // tap fails the unsigned range tests below.  The 1-D weights are zeroed per axis BEFORE the four products (0 * finite == +0,
  // the same value as selecting 0 after the product): same results as the select-per-tap form with a third fewer vector
  // instructions (round 4: they cost fp32-MFMA time on gfx950, and the dot-product kernel is made of them).
  const bool any = (ix > -1.0f) & (ix < (float)w) & (iy > -1.0f) & (iy < (float)h);
  const float fx = floorf(ix), fy = floorf(iy);
  const int x0 = any ? (int)fx : -2, y0 = any ? (int)fy : -2;
  const float ax = ix - fx, ay = iy - fy;  // (garbage under the sentinel: selected away)
  const float wx0 = (unsigned)x0 < (unsigned)w ? 1.0f - ax : 0.0f;
  const float wx1 = (unsigned)(x0 + 1) < (unsigned)w ? ax : 0.0f;
  const float wy0 = (unsigned)y0 < (unsigned)h ? 1.0f - ay : 0.0f;
  const float wy1 = (unsigned)(y0 + 1) < (unsigned)h ? ay : 0.0f;
  t.w00 = wx0 * wy0;
  t.w01 = wx1 * wy0;
  t.w10 = wx0 * wy1;
  t.w11 = wx1 * wy1;
  t.x0 = min(max(x0, 0), w - 1);
  t.x1 = min(max(x0 + 1, 0), w - 1);
  t.y0 = min(max(y0, 0), h - 1);
  t.y1 = min(max(y0 + 1, 0), h - 1);
  return t;
}

// F.normalize(dim): v / max(||v||, 1e-12)
__device__ __forceinline__ void normalize3(float& x, float& y, float& z) {
  const float n = sqrtf(x * x + y * y + z * z);
  const float inv = 1.0f / fmaxf(n, 1e-12f);
  x *= inv;
  y *= inv;
  z *= inv;
}

// F.cosine_similarity(eps=1e-5) of two (already normalised) rays
__device__ __forceinline__ float cos_sim3(float ax, float ay, float az, float bx, float by, float bz) {
  const float w12 = ax * bx + ay * by + az * bz;
  const float n1 = fmaxf(sqrtf(ax * ax + ay * ay + az * az), 1e-5f);
  const float n2 = fmaxf(sqrtf(bx * bx + by * by + bz * bz), 1e-5f);
  return w12 / (n1 * n2);
}

// nearest-neighbour source index of F.interpolate(mode="nearest"): floor(dst * in/out)
__device__ __forceinline__ int nearest_src(int dst, int in_size, int out_size) {
  const float scale = (float)in_size / (float)out_size;
  return min((int)floorf((float)dst * scale), in_size - 1);
}

// hint MLP 3 -> 12 -> 12 -> 1, LeakyReLU(0.01) (modules/mesh_hint_volume.py:78-79,373-386)
__device__ __forceinline__ float hint_mlp_eval(const float* __restrict__ hm, float s, float hint, float hw) {
  // hm: V1[12x3], c1[12], V2[12x12], c2[12], V3[12], c3
  const float* V1 = hm;
  const float* c1 = hm + 36;
  const float* V2 = hm + 48;
  const float* c2 = hm + 192;
  const float* V3 = hm + 204;
  const float c3 = hm[216];
  float a[12];
#pragma unroll
  for (int m = 0; m < 12; ++m) a[m] = lrelu(V1[m * 3 + 0] * s + V1[m * 3 + 1] * hint + V1[m * 3 + 2] * hw + c1[m], 0.01f);
  float out = c3;
#pragma unroll
  for (int n = 0; n < 12; ++n) {
    float acc = c2[n];
#pragma unroll
    for (int m = 0; m < 12; ++m) acc += V2[n * 12 + m] * a[m];
    out += V3[n] * lrelu(acc, 0.01f);
  }
  return out;
}


}  // namespace dt
