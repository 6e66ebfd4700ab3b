// Plane-sweep cost volume: setup, dot-product volume, simple (cross-check) MLP volume,
// lowest-cost and overall-mask kernels.  gfx950 only.
//
// Reference behaviour restated here (paths relative to /root/reference/src/doubletake/):
//   CostVolumeManager.generate_depth_planes   modules/cost_volume.py:96-130
//   CostVolumeManager.warp_features           modules/cost_volume.py:132-217
//   CostVolumeManager.build_cost_volume       modules/cost_volume.py:219-315
//   CostVolumeManager.forward (argmax/gather) modules/cost_volume.py:355-361
//   CostVolumeManager.get_mask                modules/cost_volume.py:73-94
//   pose_distance                             utils/geometry_utils.py:187-199
//   FeatureMeshHintVolumeManager.build_cost_volume modules/mesh_hint_volume.py:84-393
#include "common.hpp"
#include "cv_geometry.hpp"

namespace dt {

// ------------------------------------------------------------------------------------------
// relative poses of a keyframe tuple: src_cam_T_cur_cam = src_cam_T_world @ cur_world_T_cam and
// cur_cam_T_src_cam = cur_cam_T_world @ src_world_T_cam (doubletake_model.py:330-339).  The reference issues two
// torch.matmul calls on [b,K,4,4] tensors; on ROCm each is a hipBLASLt launch with ~100 us of host-side set-up, which sat
// on the per-frame critical path of the incremental loop (profiles/r3q_incremental_frame_timeline.txt).  One thread per
// output element here.
// ------------------------------------------------------------------------------------------
__global__ void cv_relative_poses_kernel(const float* __restrict__ src_cTw, const float* __restrict__ src_wTc,
                                         const float* __restrict__ cur_cTw, const float* __restrict__ cur_wTc, int B, int K,
                                         float* __restrict__ ext, float* __restrict__ poses) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= B * K * 16) return;
  const int e = idx & 15, bk = idx >> 4, b = bk / K;
  const int i = e >> 2, j = e & 3;
  const float* A = src_cTw + (size_t)bk * 16;  // ext = A @ Bm
  const float* Bm = cur_wTc + (size_t)b * 16;
  ext[idx] = A[i * 4 + 0] * Bm[0 * 4 + j] + A[i * 4 + 1] * Bm[1 * 4 + j] + A[i * 4 + 2] * Bm[2 * 4 + j] + A[i * 4 + 3] * Bm[3 * 4 + j];
  const float* Cm = cur_cTw + (size_t)b * 16;  // poses = Cm @ Dm
  const float* Dm = src_wTc + (size_t)bk * 16;
  poses[idx] = Cm[i * 4 + 0] * Dm[0 * 4 + j] + Cm[i * 4 + 1] * Dm[1 * 4 + j] + Cm[i * 4 + 2] * Dm[2 * 4 + j] + Cm[i * 4 + 3] * Dm[3 * 4 + j];
}

// ------------------------------------------------------------------------------------------
// setup: one block per batch element, thread k handles source view k; thread-strided planes
// ------------------------------------------------------------------------------------------
__global__ void cv_setup_kernel(const float* __restrict__ src_Ks, const float* __restrict__ src_ext,
                                const float* __restrict__ src_poses, const float* __restrict__ cur_invK,
                                const float* __restrict__ min_depth, const float* __restrict__ max_depth,
                                int K, int D, float* __restrict__ params) {
  const int b = blockIdx.x;
  float* p = params + (size_t)b * cv_params_floats(D, K);
  const int t = threadIdx.x;
  if (t < 12) {
    // invK[:3,:3] row-major (pad with zeros)
    float v = 0.f;
    if (t < 9) v = cur_invK[b * 16 + (t / 3) * 4 + (t % 3)];
    p[kCvInvK + t] = v;
  }
  const float mn = min_depth[b], mx = max_depth[b];
  const float lmn = logf(mn), lr = logf(mx / mn);
  for (int d = t; d < D; d += blockDim.x) {
    const float ramp = (D > 1) ? (float)d / (float)(D - 1) : 0.f;
    p[kCvPlanes + d] = expf(lmn + lr * ramp);
  }
  for (int k = t; k < K; k += blockDim.x) {
    const float* Km = src_Ks + ((size_t)b * K + k) * 16;
    const float* E = src_ext + ((size_t)b * K + k) * 16;
    const float* T = src_poses + ((size_t)b * K + k) * 16;
    float* v = p + cv_view_off(D, k);
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 4; ++j) {
        float acc = 0.f;
        for (int m = 0; m < 4; ++m) acc += Km[i * 4 + m] * E[m * 4 + j];
        v[i * 4 + j] = acc;
      }
    const float tx = T[3], ty = T[7], tz = T[11];
    v[12] = tx;
    v[13] = ty;
    v[14] = tz;
    const float tr = T[0] + T[5] + T[10];
    const float Rm = sqrtf(2.f * (1.f - fminf(3.f, tr) / 3.f));
    const float tm = sqrtf(tx * tx + ty * ty + tz * tz);
    v[15] = sqrtf(tm * tm + Rm * Rm);
    v[16] = Rm;
    v[17] = tm;
    v[18] = 0.f;
    v[19] = 0.f;
  }
}

// ------------------------------------------------------------------------------------------
// warp_features (cost_volume.py:132-217) as a stand-alone op: ONE depth map per batch element (any value per
// pixel), all source views.  One thread per (batch, view, pixel); NCHW in, NCHW out like the reference.
// The fused volume kernels do not call this -- it exists for callers of the public method.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void cv_warp_kernel(const float* __restrict__ src_bkchw, const float* __restrict__ params,
                                                     const float* __restrict__ depth_bhw, int K, int C, int h, int w,
                                                     int D, float* __restrict__ world_B4N, float* __restrict__ depths_bkhw,
                                                     float* __restrict__ warped_bkchw, float* __restrict__ mask_bkhw) {
  const size_t hw = (size_t)h * w;
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int b = blockIdx.z, k = blockIdx.y;
  if (idx >= hw) return;
  const int y = (int)(idx / w), x = (int)(idx % w);
  const float* p = params + (size_t)b * cv_params_floats(D, K);
  float rx, ry, rz;
  pixel_ray(p + kCvInvK, x, y, rx, ry, rz);
  const float depth = depth_bhw[(size_t)b * hw + idx];
  const float X = depth * rx, Y = depth * ry, Z = depth * rz;
  const size_t B = (size_t)b * K + k;
  float* wp = world_B4N + B * 4 * hw + idx;
  wp[0] = X;
  wp[hw] = Y;
  wp[2 * hw] = Z;
  wp[3 * hw] = 1.0f;
  const ViewProj q = project_view(p + cv_view_off(D, k), X, Y, Z);
  depths_bkhw[B * hw + idx] = q.z;
  mask_bkhw[B * hw + idx] = (q.z > 0.f) ? 1.0f : 0.0f;
  const Taps t = bilinear_taps(q.u, q.v, h, w, 1.0f / (float)w, 1.0f / (float)h);
  const float* sb = src_bkchw + B * C * hw;
  float* ob = warped_bkchw + B * C * hw + idx;
  const size_t o00 = (size_t)t.y0 * w + t.x0, o01 = (size_t)t.y0 * w + t.x1, o10 = (size_t)t.y1 * w + t.x0,
               o11 = (size_t)t.y1 * w + t.x1;
  for (int c = 0; c < C; ++c) {
    const float* sc = sb + (size_t)c * hw;
    ob[(size_t)c * hw] = sc[o00] * t.w00 + sc[o01] * t.w01 + sc[o10] * t.w10 + sc[o11] * t.w11;
  }
}

// (the dot-product volume lives in cv_dot_lds.hip)

// ------------------------------------------------------------------------------------------
// simple MLP / hint volume: one thread per (pixel, plane); everything in plain fp32 loops.
// Input-vector channel order follows modules/mesh_hint_volume.py:353-370.
// ------------------------------------------------------------------------------------------
constexpr int kMaxSrc = 16;
constexpr int kMaxFeatSimple = 32;  // matching_dim_size accepted by the one-thread-per-sample kernels
constexpr int kHidden = 128;

// depth of plane d at a pixel: the per-batch plane list of the parameter block, or -- when the caller handed its own
// depth_planes_bdhw (modules/cost_volume.py:249-250, feature_volume.py:145-146, mesh_hint_volume.py:149-150) -- that tensor
__device__ __forceinline__ float plane_depth(const float* __restrict__ p, const float* __restrict__ planes_bdhw, int b,
                                             int d, int D, size_t hw, size_t pix) {
  return planes_bdhw ? planes_bdhw[((size_t)b * D + d) * hw + pix] : p[kCvPlanes + d];
}

__global__ __launch_bounds__(128) void cv_mlp_simple_kernel(
    const float* __restrict__ cur_bchw, const float* __restrict__ src_bkhwc, const float* __restrict__ params,
    const float* __restrict__ planes_bdhw, const float* __restrict__ W1, const float* __restrict__ b1,
    const float* __restrict__ W2, const float* __restrict__ b2, const float* __restrict__ W3, const float* __restrict__ b3,
    const float* __restrict__ hint_mlp, const float* __restrict__ depth_hint, const float* __restrict__ hint_w,
    const float* __restrict__ hint_m, int hint_h, int hint_w2, float* __restrict__ vol, int K, int nfeat, int h, int w,
    int D) {
  const int b = blockIdx.z;
  const int d = blockIdx.y;
  const size_t hw = (size_t)h * w;
  const size_t pix = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (pix >= hw) return;
  const int y = (int)(pix / w), x = (int)(pix % w);
  const float* p = params + (size_t)b * cv_params_floats(D, K);
  const int Cin = (nfeat + 4) * (K + 1) + 6 * K;

  float in[(kMaxFeatSimple + 4) * (kMaxSrc + 1) + 6 * kMaxSrc];
  float rx, ry, rz;
  pixel_ray(p + kCvInvK, x, y, rx, ry, rz);
  const float depth = plane_depth(p, planes_bdhw, b, d, D, hw, pix);
  const float X = depth * rx, Y = depth * ry, Z = depth * rz;
  float cx = X, cy = Y, cz = Z;
  normalize3(cx, cy, cz);
  const float inv_w = 1.0f / (float)w, inv_h = 1.0f / (float)h;

  const int o_cur = nfeat * K, o_mask = o_cur + nfeat, o_z = o_mask + K, o_plane = o_z + K;
  const int o_dot = o_plane + 1, o_ang = o_dot + K, o_cray = o_ang + K, o_sray = o_cray + 3;
  const int o_pd = o_sray + 3 * K, o_R = o_pd + K, o_t = o_R + K;

  for (int c = 0; c < nfeat; ++c) in[o_cur + c] = cur_bchw[((size_t)b * nfeat + c) * hw + pix];
  in[o_plane] = depth;
  in[o_cray + 0] = cx;
  in[o_cray + 1] = cy;
  in[o_cray + 2] = cz;
  for (int k = 0; k < K; ++k) {
    const float* vp = p + cv_view_off(D, k);
    const ViewProj q = project_view(vp, X, Y, Z);
    const Taps t = bilinear_taps(q.u, q.v, h, w, inv_w, inv_h);
    const float* base = src_bkhwc + ((size_t)b * K + k) * hw * nfeat;
    const float* p00 = base + ((size_t)t.y0 * w + t.x0) * nfeat;
    const float* p01 = base + ((size_t)t.y0 * w + t.x1) * nfeat;
    const float* p10 = base + ((size_t)t.y1 * w + t.x0) * nfeat;
    const float* p11 = base + ((size_t)t.y1 * w + t.x1) * nfeat;
    float dot = 0.f;
    for (int c = 0; c < nfeat; ++c) {
      const float f = p00[c] * t.w00 + p01[c] * t.w01 + p10[c] * t.w10 + p11[c] * t.w11;
      in[k * nfeat + c] = f;
      dot += f * in[o_cur + c];
    }
    const float m = (q.z > 0.f) ? 1.f : 0.f;
    in[o_mask + k] = m;
    in[o_z + k] = q.z;
    in[o_dot + k] = dot * m;
    float sx = X - vp[12], sy = Y - vp[13], sz = Z - vp[14];
    normalize3(sx, sy, sz);
    in[o_ang + k] = cos_sim3(cx, cy, cz, sx, sy, sz);
    in[o_sray + 3 * k + 0] = sx;
    in[o_sray + 3 * k + 1] = sy;
    in[o_sray + 3 * k + 2] = sz;
    in[o_pd + k] = vp[15];
    in[o_R + k] = vp[16];
    in[o_t + k] = vp[17];
  }

  float h1[kHidden];
  for (int j = 0; j < kHidden; ++j) {
    float acc = b1[j];
    const float* wr = W1 + (size_t)j * Cin;
    for (int c = 0; c < Cin; ++c) acc += wr[c] * in[c];
    h1[j] = lrelu(acc, 0.01f);
  }
  float s = b3[0];
  for (int j = 0; j < kHidden; ++j) {
    float acc = b2[j];
    const float* wr = W2 + (size_t)j * kHidden;
    for (int c = 0; c < kHidden; ++c) acc += wr[c] * h1[c];
    s += W3[j] * lrelu(acc, 0.01f);
  }
  if (hint_mlp != nullptr) {
    const int sy = nearest_src(y, hint_h, h), sx = nearest_src(x, hint_w2, w);
    const size_t hi = ((size_t)b * hint_h + sy) * hint_w2 + sx;
    const bool m = hint_m[hi] != 0.f;
    const float hint = m ? fabsf(depth_hint[hi] - depth) : -1.f;
    const float hwt = m ? hint_w[hi] : 0.f;
    s = hint_mlp_eval(hint_mlp, s, hint, hwt);
  }
  vol[((size_t)b * D + d) * hw + pix] = s;
}

// ------------------------------------------------------------------------------------------
// simple dot-product volume: one thread per (pixel, plane), any channel count, optional per-pixel planes.
// modules/cost_volume.py:219-315 for the shapes the tuned kernel (cv_dot_lds.hip: 16 channels, one plane list per
// batch element) does not take.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(128) void cv_dot_simple_kernel(const float* __restrict__ cur_bchw,
                                                             const float* __restrict__ src_bkhwc,
                                                             const float* __restrict__ params,
                                                             const float* __restrict__ planes_bdhw, float* __restrict__ vol,
                                                             int K, int C, int h, int w, int D) {
  const int b = blockIdx.z;
  const int d = blockIdx.y;
  const size_t hw = (size_t)h * w;
  const size_t pix = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (pix >= hw) return;
  const int y = (int)(pix / w), x = (int)(pix % w);
  const float* p = params + (size_t)b * cv_params_floats(D, K);
  float rx, ry, rz;
  pixel_ray(p + kCvInvK, x, y, rx, ry, rz);
  const float depth = plane_depth(p, planes_bdhw, b, d, D, hw, pix);
  const float X = depth * rx, Y = depth * ry, Z = depth * rz;
  const float inv_w = 1.0f / (float)w, inv_h = 1.0f / (float)h;
  float sum = 0.f;
  for (int k = 0; k < K; ++k) {
    const ViewProj q = project_view(p + cv_view_off(D, k), X, Y, Z);
    const Taps t = bilinear_taps(q.u, q.v, h, w, inv_w, inv_h);
    const float* base = src_bkhwc + ((size_t)b * K + k) * hw * C;
    const float* p00 = base + ((size_t)t.y0 * w + t.x0) * C;
    const float* p01 = base + ((size_t)t.y0 * w + t.x1) * C;
    const float* p10 = base + ((size_t)t.y1 * w + t.x0) * C;
    const float* p11 = base + ((size_t)t.y1 * w + t.x1) * C;
    float dot = 0.f;
    for (int c = 0; c < C; ++c) {
      const float f = p00[c] * t.w00 + p01[c] * t.w01 + p10[c] * t.w10 + p11[c] * t.w11;
      dot += f * cur_bchw[((size_t)b * C + c) * hw + pix];
    }
    sum += (q.z > 0.f) ? dot : 0.f;
  }
  vol[((size_t)b * D + d) * hw + pix] = sum;
}

// ------------------------------------------------------------------------------------------
// lowest cost: plane depth at the first maximum over d
// ------------------------------------------------------------------------------------------
__global__ void cv_lowest_cost_kernel(const float* __restrict__ vol, const float* __restrict__ params,
                                      const float* __restrict__ planes_bdhw, float* __restrict__ out, int nhwc, int K,
                                      size_t hw, int D) {
  const int b = blockIdx.y;
  const float* p = params + (size_t)b * cv_params_floats(D, K);
  if (nhwc) {
    // [pixel][D] layout: 16 lanes share one pixel (coalesced float4 reads of its D contiguous costs),
    // then a 16-lane butterfly picks the maximum, lowest plane index winning ties (= first maximum)
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t pix = t >> 4;
    const int sub = (int)(t & 15);
    float best = -INFINITY;
    int bi = 0x7fffffff;
    bool nan_seen = false;
    if (pix < hw) {
      const float* row = vol + ((size_t)b * hw + pix) * D;
      for (int d = sub * 4; d < D; d += 64) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if (d + j >= D) break;
          const float v = row[d + j];
          if (v != v) {  // torch.argmax treats NaN as the maximum (first NaN wins)
            if (!nan_seen) {
              nan_seen = true;
              bi = d + j;
            }
          } else if (!nan_seen && (bi == 0x7fffffff || v > best)) {
            best = v;
            bi = d + j;
          }
        }
      }
    }
#pragma unroll
    for (int m = 1; m < 16; m <<= 1) {
      const float ob = __shfl_xor(best, m, 64);
      const int oi = __shfl_xor(bi, m, 64);
      const int on = __shfl_xor((int)nan_seen, m, 64);
      bool take;
      if (on != (int)nan_seen) take = on != 0;          // a NaN beats any number
      else if (nan_seen) take = oi < bi;                 // both NaN: first index
      else take = (ob > best) || (ob == best && oi < bi);
      if (take) {
        best = ob;
        bi = oi;
        nan_seen = on != 0;
      }
    }
    if (pix < hw && sub == 0) out[(size_t)b * hw + pix] = plane_depth(p, planes_bdhw, b, bi == 0x7fffffff ? 0 : bi, D, hw, pix);
    return;
  }
  const size_t pix = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (pix >= hw) return;
  float best = -INFINITY;
  int bi = 0;
  bool seen_nan = false;
  for (int d = 0; d < D; ++d) {
    const float v = vol[((size_t)b * D + d) * hw + pix];
    if (!seen_nan && (v != v)) {
      seen_nan = true;
      bi = d;
    }
    if (!seen_nan && v > best) {
      best = v;
      bi = d;
    }
  }
  out[(size_t)b * hw + pix] = plane_depth(p, planes_bdhw, b, bi, D, hw, pix);
}

// ------------------------------------------------------------------------------------------
// overall mask at the last plane
// ------------------------------------------------------------------------------------------
__global__ void cv_mask_kernel(const float* __restrict__ params, const float* __restrict__ planes_bdhw,
                               uint8_t* __restrict__ out, int per_view, int K, int h, int w, int D) {
  const int b = blockIdx.y;
  const size_t hw = (size_t)h * w;
  const size_t pix = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (pix >= hw) return;
  const int y = (int)(pix / w), x = (int)(pix % w);
  const float* p = params + (size_t)b * cv_params_floats(D, K);
  float rx, ry, rz;
  pixel_ray(p + kCvInvK, x, y, rx, ry, rz);
  const float depth = plane_depth(p, planes_bdhw, b, D - 1, D, hw, pix);
  const float X = depth * rx, Y = depth * ry, Z = depth * rz;
  bool any_d = false, any_b = false;
  for (int k = 0; k < K; ++k) {
    const ViewProj q = project_view(p + cv_view_off(D, k), X, Y, Z);
    const bool dm = q.z > 0.f;
    const bool bm = (q.u > 2.f) && (q.u < (float)(w - 2)) && (q.v > 2.f) && (q.v < (float)(h - 2));
    any_d |= dm;
    any_b |= bm;
    if (per_view) out[((size_t)b * K + k) * hw + pix] = (dm && bm) ? 1 : 0;
  }
  if (!per_view) out[(size_t)b * hw + pix] = (any_d && any_b) ? 1 : 0;
}

}  // namespace dt

using namespace dt;

extern "C" {

int dt_cv_params_floats(int num_planes, int num_src) { return cv_params_floats(num_planes, num_src); }

int dt_cv_setup_f32(const float* src_Ks, const float* src_ext, const float* src_poses, const float* cur_invK,
                    const float* min_depth, const float* max_depth, int batch, int num_src, int num_planes,
                    float* params_out, dt_stream_t s) {
  DT_REQUIRE(batch > 0 && num_src > 0 && num_planes > 0, "dt_cv_setup_f32: bad extents b=%d k=%d D=%d", batch,
             num_src, num_planes);
  DT_REQUIRE(src_Ks && src_ext && src_poses && cur_invK && min_depth && max_depth && params_out,
             "dt_cv_setup_f32: null pointer");
  DT_LAUNCH(cv_setup_kernel, dim3(batch), dim3(64), 0, to_stream(s), src_Ks, src_ext, src_poses, cur_invK,
                     min_depth, max_depth, num_src, num_planes, params_out);
  return check_launch("dt_cv_setup_f32");
}

int dt_cv_relative_poses_f32(const float* src_cam_T_world_bk44, const float* src_world_T_cam_bk44,
                             const float* cur_cam_T_world_b44, const float* cur_world_T_cam_b44, int batch, int num_src,
                             float* src_cam_T_cur_cam_bk44, float* cur_cam_T_src_cam_bk44, dt_stream_t s) {
  DT_REQUIRE(batch > 0 && num_src > 0, "dt_cv_relative_poses_f32: bad extents");
  DT_REQUIRE(src_cam_T_world_bk44 && src_world_T_cam_bk44 && cur_cam_T_world_b44 && cur_world_T_cam_b44 && src_cam_T_cur_cam_bk44 &&
                 cur_cam_T_src_cam_bk44,
             "dt_cv_relative_poses_f32: null pointer");
  const int n = batch * num_src * 16;
  DT_LAUNCH(cv_relative_poses_kernel, dim3((n + 255) / 256), dim3(256), 0, to_stream(s), src_cam_T_world_bk44, src_world_T_cam_bk44,
            cur_cam_T_world_b44, cur_world_T_cam_b44, batch, num_src, src_cam_T_cur_cam_bk44, cur_cam_T_src_cam_bk44);
  return check_launch("dt_cv_relative_poses_f32");
}

int dt_cv_warp_f32(const float* src_bkchw, const float* params, const float* depth_bhw, int batch, int num_src,
                   int channels, int h, int w, int num_planes, float* world_points_B4N, float* depths_bkhw,
                   float* warped_bkchw, float* mask_bkhw, dt_stream_t s) {
  DT_REQUIRE(batch > 0 && num_src > 0 && channels > 0 && h > 0 && w > 0 && num_planes > 0, "dt_cv_warp_f32: bad extents");
  DT_REQUIRE(src_bkchw && params && depth_bhw && world_points_B4N && depths_bkhw && warped_bkchw && mask_bkhw,
             "dt_cv_warp_f32: null pointer");
  const size_t hw = (size_t)h * w;
  dim3 grid((unsigned)((hw + 255) / 256), num_src, batch);
  DT_LAUNCH(cv_warp_kernel, grid, dim3(256), 0, to_stream(s), src_bkchw, params, depth_bhw, num_src, channels, h, w,
                     num_planes, world_points_B4N, depths_bkhw, warped_bkchw, mask_bkhw);
  return check_launch("dt_cv_warp_f32");
}

int dt_cv_dot_simple_f32(const float* cur, const float* src, const float* params, const float* depth_planes_bdhw,
                         float* vol, int batch, int num_src, int channels, int h, int w, int num_planes, dt_stream_t s) {
  DT_REQUIRE(batch > 0 && num_src > 0 && channels > 0 && h > 0 && w > 0 && num_planes > 0, "dt_cv_dot_simple_f32: bad extents");
  DT_REQUIRE(cur && src && params && vol, "dt_cv_dot_simple_f32: null pointer");
  const size_t hw = (size_t)h * w;
  dim3 grid((unsigned)((hw + 127) / 128), num_planes, batch);
  DT_LAUNCH(cv_dot_simple_kernel, grid, dim3(128), 0, to_stream(s), cur, src, params, depth_planes_bdhw, vol, num_src,
            channels, h, w, num_planes);
  return check_launch("dt_cv_dot_simple_f32");
}

int dt_cv_mlp_hint_simple_f32(const float* cur, const float* src, const float* params, const float* depth_planes_bdhw,
                              const float* W1, const float* b1, const float* W2, const float* b2, const float* W3,
                              const float* b3, const float* hint_mlp, const float* depth_hint, const float* hint_w,
                              const float* hint_m, int hint_h, int hint_w2, float* vol, int batch, int num_src, int channels,
                              int h, int w, int num_planes, dt_stream_t s) {
  DT_REQUIRE(batch > 0 && h > 0 && w > 0 && num_planes > 0, "dt_cv_mlp_hint_simple_f32: bad extents");
  DT_REQUIRE(num_src > 0 && num_src <= kMaxSrc, "dt_cv_mlp_hint_simple_f32: num_src=%d not in 1..%d", num_src,
             kMaxSrc);
  DT_REQUIRE(channels > 0 && channels <= kMaxFeatSimple, "dt_cv_mlp_hint_simple_f32: channels=%d not in 1..%d", channels,
             kMaxFeatSimple);
  DT_REQUIRE(cur && src && params && W1 && b1 && W2 && b2 && W3 && b3 && vol, "dt_cv_mlp_hint_simple_f32: null pointer");
  DT_REQUIRE(hint_mlp == nullptr || (depth_hint && hint_w && hint_m && hint_h > 0 && hint_w2 > 0),
             "dt_cv_mlp_hint_simple_f32: hint MLP given without hint maps");
  const size_t hw = (size_t)h * w;
  dim3 grid((unsigned)((hw + 127) / 128), num_planes, batch);
  DT_LAUNCH(cv_mlp_simple_kernel, grid, dim3(128), 0, to_stream(s), cur, src, params, depth_planes_bdhw, W1, b1, W2, b2, W3,
            b3, hint_mlp, depth_hint, hint_w, hint_m, hint_h, hint_w2, vol, num_src, channels, h, w, num_planes);
  return check_launch("dt_cv_mlp_hint_simple_f32");
}

int dt_cv_lowest_cost_f32(const float* volume, const float* params, const float* depth_planes_bdhw, float* lowest, int nhwc,
                          int batch, int num_src, int h, int w, int num_planes, dt_stream_t s) {
  DT_REQUIRE(batch > 0 && h > 0 && w > 0 && num_planes > 0 && num_src > 0, "dt_cv_lowest_cost_f32: bad extents");
  DT_REQUIRE(volume && params && lowest, "dt_cv_lowest_cost_f32: null pointer");
  const size_t hw = (size_t)h * w;
  const size_t threads = nhwc ? hw * 16 : hw;
  dim3 grid((unsigned)((threads + 255) / 256), batch);
  DT_LAUNCH(cv_lowest_cost_kernel, grid, dim3(256), 0, to_stream(s), volume, params, depth_planes_bdhw, lowest, nhwc,
            num_src, hw, num_planes);
  return check_launch("dt_cv_lowest_cost_f32");
}

int dt_cv_overall_mask_u8(const float* params, const float* depth_planes_bdhw, uint8_t* mask_out, int per_view, int batch,
                          int num_src, int h, int w, int num_planes, dt_stream_t s) {
  DT_REQUIRE(batch > 0 && h > 0 && w > 0 && num_planes > 0 && num_src > 0, "dt_cv_overall_mask_u8: bad extents");
  DT_REQUIRE(params && mask_out, "dt_cv_overall_mask_u8: null pointer");
  const size_t hw = (size_t)h * w;
  dim3 grid((unsigned)((hw + 255) / 256), batch);
  DT_LAUNCH(cv_mask_kernel, grid, dim3(256), 0, to_stream(s), params, depth_planes_bdhw, mask_out, per_view, num_src, h, w,
            num_planes);
  return check_launch("dt_cv_overall_mask_u8");
}

}  // extern "C"
