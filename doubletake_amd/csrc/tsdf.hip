// fp16 TSDF fusion for gfx950: frame setup, integrate, trilinear sampling.
//
// Replaces (paths relative to /root/reference/src/doubletake/):
//   TSDFFuser.integrate_depth      tools/tsdf.py:414-558   (torch ops on half tensors)
//   get_frustum_bounds             tools/tsdf.py:15-50
//   TSDFFuser.project_to_camera    tools/tsdf.py:401-412
//   TSDF.sample_tsdf               tools/tsdf.py:277-339
//   the open3d HashSet of active voxel keys (tools/tsdf.py:79-84,530-538) -> a bitmap
//
// fp16 faithfulness: the reference runs every op on half tensors, i.e. fp32 arithmetic with
// ONE rounding to half per torch op.  The kernels restate that op by op (`rh()` marks each
// reference op boundary); FMA contraction is disabled in this file because fusing two reference
// ops would remove a rounding.  The integer outcomes (which voxels are updated / active) are
// therefore those of the reference's half pipeline, pinned against its CPU-half run.
//
// HBM-bound streaming pass: one thread per voxel, Z fastest (coalesced half loads/stores of the
// touched voxels; voxels outside the frustum AABB exit after recomputing their coordinate from
// the index -- no 3x fp16 coordinate volume is read, unlike the reference).
#pragma clang fp contract(off)
#include "common.hpp"

namespace dt {

typedef _Float16 half_t;

__device__ __forceinline__ float rh(float x) { return (float)(half_t)x; }  // round to half, back to f32
__device__ __forceinline__ float h2f(uint16_t b) {
  half_t h;
  __builtin_memcpy(&h, &b, 2);
  return (float)h;
}
__device__ __forceinline__ uint16_t f2h(float x) {
  half_t h = (half_t)x;
  uint16_t b;
  __builtin_memcpy(&b, &h, 2);
  return b;
}

// frame_params layout (floats, all values already rounded to half unless noted)
//   [0..11]  P = (K @ cam_T_world)[:3,:4]
//   [12..14] frustum AABB min, [15..17] AABB max
constexpr int kFrameParams = 32;

// 4x4 inverse in fp32 (Gauss-Jordan, partial pivoting); result rounded to half like
// torch.inverse(M.float()).half() (tools/tsdf.py:452-453)
__device__ void inv4_half(const float* __restrict__ M, float* __restrict__ out) {
  float a[4][8];
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) {
      a[i][j] = M[i * 4 + j];
      a[i][4 + j] = (i == j) ? 1.f : 0.f;
    }
  for (int c = 0; c < 4; ++c) {
    int piv = c;
    float best = fabsf(a[c][c]);
    for (int r = c + 1; r < 4; ++r)
      if (fabsf(a[r][c]) > best) {
        best = fabsf(a[r][c]);
        piv = r;
      }
    if (piv != c)
      for (int j = 0; j < 8; ++j) {
        const float t = a[c][j];
        a[c][j] = a[piv][j];
        a[piv][j] = t;
      }
    const float inv = 1.f / a[c][c];
    for (int j = 0; j < 8; ++j) a[c][j] *= inv;
    for (int r = 0; r < 4; ++r) {
      if (r == c) continue;
      const float f = a[r][c];
      for (int j = 0; j < 8; ++j) a[r][j] -= f * a[c][j];
    }
  }
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) out[i * 4 + j] = rh(a[i][4 + j]);
}

// C = half(A @ B) with fp32 accumulation, A [ra x 4], B [4 x cb], all values are halves in f32
__device__ void matmul_half(const float* A, const float* B, float* C, int ra, int cb) {
  for (int i = 0; i < ra; ++i)
    for (int j = 0; j < cb; ++j) {
      float acc = 0.f;
      for (int k = 0; k < 4; ++k) acc += A[i * 4 + k] * B[k * cb + j];
      C[i * cb + j] = rh(acc);
    }
}

__global__ void tsdf_frame_setup_kernel(const uint16_t* __restrict__ K16, const uint16_t* __restrict__ T16, int img_h,
                                        int img_w, float depth_min, float depth_max, float* __restrict__ fp) {
  if (threadIdx.x != 0) return;
  // one workgroup per frame: frame f reads K16 + 16 f, T16 + 16 f and writes fp + kFrameParams f
  K16 += (size_t)blockIdx.x * 16;
  T16 += (size_t)blockIdx.x * 16;
  fp += (size_t)blockIdx.x * kFrameParams;
  float K[16], T[16], invK[16], pose[16], P[16];
  for (int i = 0; i < 16; ++i) {
    K[i] = h2f(K16[i]);
    T[i] = h2f(T16[i]);
  }
  inv4_half(K, invK);
  inv4_half(T, pose);
  matmul_half(K, T, P, 4, 4);
  for (int i = 0; i < 12; ++i) fp[i] = P[i];
  // get_frustum_bounds: corners (0|W, 0|H, 1, 1) as columns
  const float W = rh((float)img_w), H = rh((float)img_h);
  const float corners[16] = {0.f, W, 0.f, W, 0.f, 0.f, H, H, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f};  // [4 rows][4 cols]
  float cp[16];
  matmul_half(invK, corners, cp, 4, 4);
  float c8[32];  // [4][8]
  for (int r = 0; r < 4; ++r)
    for (int j = 0; j < 4; ++j) {
      const float v = cp[r * 4 + j];
      c8[r * 8 + j] = (r < 3) ? rh(v * depth_min) : v;
      c8[r * 8 + 4 + j] = (r < 3) ? rh(v * depth_max) : v;
    }
  float w8[32];
  matmul_half(pose, c8, w8, 4, 8);
  for (int r = 0; r < 3; ++r) {
    float mn = w8[r * 8], mx = w8[r * 8];
    for (int j = 1; j < 8; ++j) {
      mn = fminf(mn, w8[r * 8 + j]);
      mx = fmaxf(mx, w8[r * 8 + j]);
    }
    fp[12 + r] = mn;
    fp[15 + r] = mx;
  }
  for (int i = 18; i < kFrameParams; ++i) fp[i] = 0.f;
}

struct TsdfConsts {
  float origin[3];
  float voxel_size;
  float trunc;           // fp32(3 * voxel_size): divisor of dist
  float thr_neg;         // half(-trunc [*1.5]) as float
  float thr_pos;         // half(trunc) as float
  float max_depth_h;     // half(max_depth) as float
  float min_depth;       // fp32(min_depth)
  float depth_range;     // fp32(max_depth - min_depth)
  float img_w_h, img_h_h;  // half(W), half(H) as float
};
DT_ARG_NO_POINTERS(TsdfConsts);

// DEPTH32: the depth maps are fp32 and rounded to half on the fly (what OurFuser.fuse_frames's .half() does,
// tools/fusers_helper.py:67-73, without a converting copy kernel in front of every integration)
constexpr int kTsdfBallotMinFrames = 3;  // multi-frame calls of at least this many frames pre-select their frames per wave (below)
template <bool DEPTH32, bool PRESELECT>
__global__ __launch_bounds__(256) void tsdf_integrate_kernel(uint16_t* __restrict__ values, uint16_t* __restrict__ weights,
                                                            uint32_t* __restrict__ active, int X, int Y, int Z,
                                                            const void* __restrict__ depth_any, int img_h, int img_w,
                                                            const float* __restrict__ fp_all, int num_frames,
                                                            const TsdfConsts c, int x_begin) {
  // num_frames frames are integrated IN ORDER per voxel (the update is order dependent through the half
  // running mean and the weight clamp, tools/tsdf.py:553-558); the voxel's value/weight stay in registers
  // between frames, rounded to half exactly where the reference stores them.
  // grid: x over one (j,k) slab of Y*Z voxels (a multiple of 64, so waves never straddle slabs and the
  // bitmap words stay wave-aligned), y = i.  A slab whose x coordinate lies outside every frame's
  // frustum box leaves after one scalar test -- on the default +-10 m volume that is most of the grid.
  const size_t total = (size_t)X * Y * Z;
  const int i = blockIdx.y + x_begin;  // (x_begin > 0: this launch integrates an x-slab [x_begin, x_begin + gridDim.y) only)
  const unsigned slab = (unsigned)Y * (unsigned)Z;
  const unsigned pidx = blockIdx.x * blockDim.x + threadIdx.x;
  const float cx = rh(c.origin[0] + (float)i * c.voxel_size);
  const int lane = threadIdx.x & 63;
  // Which frames can touch this wave at all?  Frame f0 + l is tested in lane l, one ballot per 64 frames; the frame loop below
  // then visits the set bits only (ascending = frame order).  A multi-frame call used to repeat the box test of EVERY frame in
  // every wave of a slab inside any frame's x-range -- half of the launch's instructions (round 4: 243 M vector + 340 M scalar
  // wave-instructions for 16 frames at 0.02 m).  First the x test alone, before any per-lane index arithmetic:
  // (PRESELECT = false: one or two frames -- the per-keyframe call of the drivers -- where the plain scalar walk over the frames
  //  is cheaper than the pre-pass; the host picks the instantiation)
  constexpr bool few = !PRESELECT;
  bool any_x = false;
  if (few) {
    for (int f = 0; f < num_frames; ++f) {
      const float* fp = fp_all + (size_t)f * kFrameParams;
      any_x |= (cx > fp[12] && cx < fp[15]);
    }
  } else {
    for (int f0 = 0; f0 < num_frames; f0 += 64) {
      const int f = f0 + lane;
      bool hx = false;
      if (f < num_frames) {
        const float* fp = fp_all + (size_t)f * kFrameParams;
        hx = cx > fp[12] && cx < fp[15];
      }
      any_x |= __ballot(hx) != 0ull;
    }
  }
  if (!any_x) return;
  const size_t id = (size_t)i * slab + pidx;
  bool is_active = false;
  if (pidx < slab) {  // (slab is a multiple of 64: a wave is inside it with all 64 lanes or not at all)
    const int j = (int)(pidx / (unsigned)Z);
    const int k = (int)(pidx - (unsigned)j * (unsigned)Z);
    // voxel centre: half(fp32(origin) + idx * vs)   (tools/tsdf.py:144-148,164)
    const float cy = rh(c.origin[1] + (float)j * c.voxel_size);
    const float cz = rh(c.origin[2] + (float)k * c.voxel_size);
    // the wave's own extent in y and z (pidx runs z-fastest: j is non-decreasing over the lanes; a wave that spans two rows
    // may hold any z), a conservative stand-in for the per-voxel test that every visited frame still makes exactly
    float cy_lo = 0.f, cy_hi = 0.f, cz_lo = 0.f, cz_hi = 0.f;
    if (!few) {
      cy_lo = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, cy)));
      cy_hi = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, cy), 63));
      const bool one_row = __builtin_amdgcn_readfirstlane(j) == __builtin_amdgcn_readlane(j, 63);
      cz_lo = one_row ? __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, cz))) : rh(c.origin[2]);
      cz_hi = one_row ? __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, cz), 63))
                      : rh(c.origin[2] + (float)(Z - 1) * c.voxel_size);
    }
    bool loaded = false, dirty = false;
    float cur_v = 0.f, cur_w = 0.f;
    auto integrate_frame = [&](const int f) {
      const float* fp = fp_all + (size_t)f * kFrameParams;
      const bool inside = cx > fp[12] && cx < fp[15] && cy > fp[13] && cy < fp[16] && cz > fp[14] && cz < fp[17];
      if (!inside) return;
      float q[3];
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        float acc = 0.f;
        acc += fp[r * 4 + 0] * cx;
        acc += fp[r * 4 + 1] * cy;
        acc += fp[r * 4 + 2] * cz;
        acc += fp[r * 4 + 3] * 1.0f;
        q[r] = rh(acc);
      }
      const float u = rh(q[0] / q[2]), v = rh(q[1] / q[2]);
      const float gx = rh(rh(rh(2.0f * u) / c.img_w_h) - 1.0f);
      const float gy = rh(rh(rh(2.0f * v) / c.img_h_h) - 1.0f);
      // grid_sample(nearest, zeros, align_corners=False) in half: ((g+1)*size-1)/2 per-op rounded
      const float ix = rh(rh(rh(rh(gx + 1.0f) * c.img_w_h) - 1.0f) / 2.0f);
      const float iy = rh(rh(rh(rh(gy + 1.0f) * c.img_h_h) - 1.0f) / 2.0f);
      const float xn = rintf(ix), yn = rintf(iy);
      float sd = 0.f;
      if (xn >= 0.f && xn < (float)img_w && yn >= 0.f && yn < (float)img_h) {  // false for NaN/inf
        const size_t di = ((size_t)f * img_h + (int)yn) * img_w + (int)xn;
        sd = DEPTH32 ? rh(reinterpret_cast<const float*>(depth_any)[di]) : h2f(reinterpret_cast<const uint16_t*>(depth_any)[di]);
      }
      const float vd = q[2];
      float t = rh(sd - c.min_depth);
      t = rh(t / c.depth_range);
      t = rh(1.0f - t);
      t = fminf(fmaxf(t, 0.25f), 1.0f);
      const float conf = rh(t * t);
      const float dist = rh(sd - vd);
      const float tsdf = fminf(fmaxf(rh(dist / c.trunc), -1.0f), 1.0f);
      const bool valid = (vd > 0.f) && (dist > c.thr_neg) && (sd > 0.f) && (vd < c.max_depth_h) && (conf > 0.f);
      if (valid) {
        is_active |= dist < c.thr_pos;
        if (!loaded) {
          cur_v = h2f(values[id]);
          cur_w = h2f(weights[id]);
          loaded = true;
        }
        const float new_w = rh(rh(conf * 2.5f) / 100.0f);
        const float tot = rh(cur_w + new_w);
        const float num = rh(rh(cur_v * cur_w) + rh(tsdf * new_w));
        cur_v = rh(num / tot);
        cur_w = rh(fminf(tot, 1.0f));
        dirty = true;
      }
    };
    if (few) {
      for (int f = 0; f < num_frames; ++f) integrate_frame(f);
    } else {
      for (int f0 = 0; f0 < num_frames; f0 += 64) {
        bool hit = false;
        if (f0 + lane < num_frames) {
          const float* fpl = fp_all + (size_t)(f0 + lane) * kFrameParams;
          hit = cx > fpl[12] && cx < fpl[15] && cy_hi > fpl[13] && cy_lo < fpl[16] && cz_hi > fpl[14] && cz_lo < fpl[17];
        }
        for (unsigned long long todo = __ballot(hit); todo != 0ull; todo &= todo - 1ull) integrate_frame(f0 + (int)__builtin_ctzll(todo));
      }
    }
    if (dirty) {
      values[id] = f2h(cur_v);
      weights[id] = f2h(cur_w);
    }
  }
  // active bitmap: bit id of word id>>5.  A wave covers 64 consecutive ids = two whole words
  // (blockDim is a multiple of 64 and the grid is linear), so one lane per word does a plain OR.
  const unsigned long long bal = __ballot(is_active);
  if (bal != 0ull && (lane == 0 || lane == 32)) {
    const uint32_t bits = (lane == 0) ? (uint32_t)(bal & 0xffffffffull) : (uint32_t)(bal >> 32);
    const size_t word = (id >> 5);
    if (bits != 0u && (word << 5) < total) active[word] |= bits;
  }
}

// trilinear sample (align_corners=True) of a half volume at one world point (tools/tsdf.py:277-339)
__device__ __forceinline__ float sample_trilinear(const uint16_t* __restrict__ vol, float ox, float oy, float oz, float vs,
                                                  int X, int Y, int Z, float wx_, float wy_, float wz_, int fp16_math) {
  const float dims[3] = {(float)X, (float)Y, (float)Z};
  const float org[3] = {ox, oy, oz};
  const float pt[3] = {wx_, wy_, wz_};
  float idx[3];
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    // world -> [-1,1] (tools/tsdf.py:306-319), then align_corners=True unnormalise
    float vc = pt[a] - org[a];
    vc = vc / vs;
    vc = vc / (dims[a] - 1.0f);
    vc = vc * 2.0f - 1.0f;
    if (fp16_math) vc = rh(vc);
    float v = ((vc + 1.0f) / 2.0f) * (dims[a] - 1.0f);
    if (fp16_math) v = rh(v);
    idx[a] = v;
  }
  float acc = 0.f;
  const bool fin = (idx[0] == idx[0]) && (idx[1] == idx[1]) && (idx[2] == idx[2]) && fabsf(idx[0]) < 1e9f &&
                   fabsf(idx[1]) < 1e9f && fabsf(idx[2]) < 1e9f;
  if (fin) {
    const float fx0 = floorf(idx[0]), fy0 = floorf(idx[1]), fz0 = floorf(idx[2]);
    const float tx = idx[0] - fx0, ty = idx[1] - fy0, tz = idx[2] - fz0;
    const int x0 = (int)fx0, y0 = (int)fy0, z0 = (int)fz0;
    for (int dx = 0; dx < 2; ++dx)
      for (int dy = 0; dy < 2; ++dy)
        for (int dz = 0; dz < 2; ++dz) {
          const int xi = x0 + dx, yi = y0 + dy, zi = z0 + dz;
          if (xi < 0 || xi >= X || yi < 0 || yi >= Y || zi < 0 || zi >= Z) continue;
          const float wx = dx ? tx : 1.0f - tx, wy = dy ? ty : 1.0f - ty, wz = dz ? tz : 1.0f - tz;
          acc += h2f(vol[((size_t)xi * Y + yi) * Z + zi]) * (wx * wy * wz);
        }
  }
  return fp16_math ? rh(acc) : acc;
}

__global__ void tsdf_sample_kernel(const uint16_t* __restrict__ vol, float ox, float oy, float oz, float vs, int X, int Y,
                                   int Z, const float* __restrict__ pts, float* __restrict__ out, int64_t n,
                                   int fp16_math) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
  out[t] = sample_trilinear(vol, ox, oy, oz, vs, X, Y, Z, pts[t * 3 + 0], pts[t * 3 + 1], pts[t * 3 + 2], fp16_math);
}

// Hint maps from a rendered depth (reference test_incremental.py:204-258 in one pass): back-project the
// pixel centre (x+0.5, y+0.5) with invK and the camera pose, sample the fused WEIGHT volume there, and keep
// the depth as a hint where a surface was rendered (depth != -1) and the sampled weight reaches `thr`.
__global__ __launch_bounds__(256) void hint_from_depth_kernel(const float* __restrict__ depth, const uint16_t* __restrict__ wvol,
                                                             float ox, float oy, float oz, float vs, int X, int Y, int Z,
                                                             const float* __restrict__ invK, const float* __restrict__ pose,
                                                             float thr, int h, int w, float* __restrict__ hint,
                                                             float* __restrict__ mask_f, uint8_t* __restrict__ mask_b,
                                                             float* __restrict__ weights, int fp16_math) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= h * w) return;
  const int y = i / w, x = i - y * w;
  const float d = depth[i];
  const float px = (float)x + 0.5f, py = (float)y + 0.5f;
  // (invK @ pix) * depth, then world_T_cam @ (cam, 1): same association as the reference's matmuls
  // invK / pose: row-major 4x4
  const float cx = (invK[0] * px + invK[1] * py + invK[2]) * d;
  const float cy = (invK[4] * px + invK[5] * py + invK[6]) * d;
  const float cz = (invK[8] * px + invK[9] * py + invK[10]) * d;
  const float wx = pose[0] * cx + pose[1] * cy + pose[2] * cz + pose[3];
  const float wy = pose[4] * cx + pose[5] * cy + pose[6] * cz + pose[7];
  const float wz = pose[8] * cx + pose[9] * cy + pose[10] * cz + pose[11];
  // fp16_math: grid and result rounded to half and the cut compared in half, as the reference's device branch would
  // (tools/tsdf.py:327-330: a half volume on the GPU keeps grid_sample in half; `sampled < 0.025` on a half tensor)
  const float sw = sample_trilinear(wvol, ox, oy, oz, vs, X, Y, Z, wx, wy, wz, fp16_math);
  const float cut = (fp16_math && thr == thr && fabsf(thr) < 65504.f) ? rh(thr) : thr;
  const bool keep = (d != -1.0f) && !(sw < cut) && (d == d);
  hint[i] = keep ? d : __builtin_nanf("");
  mask_f[i] = keep ? 1.0f : 0.0f;
  mask_b[i] = keep ? 1 : 0;
  weights[i] = keep ? sw : 0.0f;
}

}  // namespace dt

using namespace dt;

extern "C" {

int dt_tsdf_frame_params_floats(void) { return kFrameParams; }

int dt_tsdf_frames_setup_f16(const uint16_t* K16, const uint16_t* T16, int num_frames, int img_h, int img_w,
                             float depth_min, float depth_max, float* frame_params, dt_stream_t s) {
  DT_REQUIRE(K16 && T16 && frame_params, "dt_tsdf_frames_setup_f16: null pointer");
  DT_REQUIRE(img_h > 0 && img_w > 0 && num_frames > 0, "dt_tsdf_frames_setup_f16: bad extents");
  DT_LAUNCH(tsdf_frame_setup_kernel, dim3(num_frames), dim3(64), 0, to_stream(s), K16, T16, img_h, img_w, depth_min,
                     depth_max, frame_params);
  return check_launch("dt_tsdf_frames_setup_f16");
}

int dt_tsdf_frame_setup_f16(const uint16_t* K16, const uint16_t* T16, int img_h, int img_w, float depth_min,
                            float depth_max, float* frame_params, dt_stream_t s) {
  return dt_tsdf_frames_setup_f16(K16, T16, 1, img_h, img_w, depth_min, depth_max, frame_params, s);
}

int dt_tsdf_integrate_f16(uint16_t* values, uint16_t* weights, uint32_t* active, const float* origin3, float voxel_size,
                          int X, int Y, int Z, const uint16_t* depth, int img_h, int img_w, const float* frame_params,
                          const dt_tsdf_thresholds* th, dt_stream_t s) {
  return dt_tsdf_integrate_frames_f16(values, weights, active, origin3, voxel_size, X, Y, Z, depth, 1, img_h, img_w,
                                      frame_params, th, s);
}

static int integrate_frames(uint16_t* values, uint16_t* weights, uint32_t* active, const float* origin3, float voxel_size, int X,
                            int Y, int Z, const void* depth, bool depth32, int num_frames, int img_h, int img_w,
                            const float* frame_params, const dt_tsdf_thresholds* th, dt_stream_t s, int x_begin = 0,
                            int x_count = -1) {
  DT_REQUIRE(values && weights && active && origin3 && depth && frame_params && th, "dt_tsdf_integrate_f16: null pointer");
  DT_REQUIRE(X > 0 && Y > 0 && Z > 0 && img_h > 0 && img_w > 0 && voxel_size > 0.f && num_frames > 0,
             "dt_tsdf_integrate_f16: bad extents");
  const size_t total = (size_t)X * Y * Z;
  DT_REQUIRE(total % 32 == 0, "dt_tsdf_integrate_f16: voxel count must be a multiple of 32 (dims are multiples of 8)");
  TsdfConsts c;
  c.origin[0] = origin3[0];
  c.origin[1] = origin3[1];
  c.origin[2] = origin3[2];
  c.voxel_size = voxel_size;
  c.trunc = th->trunc;
  c.thr_neg = th->thr_neg;
  c.thr_pos = th->thr_pos;
  c.max_depth_h = th->max_depth_h;
  c.min_depth = th->min_depth;
  c.depth_range = th->depth_range;
  c.img_w_h = (float)img_w;  // image extents are exactly representable in half (< 2048)
  c.img_h_h = (float)img_h;
  DT_REQUIRE(img_w <= 2048 && img_h <= 2048, "dt_tsdf_integrate_f16: image extent above 2048 is not exact in half");
  const size_t slab = (size_t)Y * Z;
  DT_REQUIRE(slab % 64 == 0, "dt_tsdf_integrate_f16: Y*Z must be a multiple of 64 (dims are multiples of 8)");
  DT_REQUIRE(X <= 65535 && slab < 4294967040ull, "dt_tsdf_integrate_f16: volume too large for one launch");
  if (x_count < 0) x_count = X - x_begin;
  DT_REQUIRE(x_begin >= 0 && x_count >= 0 && x_begin + x_count <= X, "dt_tsdf_integrate_f16: x-slab [%d, %d) outside 0..%d", x_begin,
             x_begin + x_count, X);
  if (x_count == 0) return 0;
  const dim3 grid((unsigned)((slab + 255) / 256), (unsigned)x_count);
  const bool preselect = num_frames >= kTsdfBallotMinFrames;
#define DT_LAUNCH_TSDF(D32_, PRE_)                                                                                          \
  DT_LAUNCH((tsdf_integrate_kernel<D32_, PRE_>), grid, dim3(256), 0, to_stream(s), values, weights, active, X, Y, Z, depth, \
            img_h, img_w, frame_params, num_frames, c, x_begin)
  if (depth32) {
    if (preselect) DT_LAUNCH_TSDF(true, true); else DT_LAUNCH_TSDF(true, false);
  } else {
    if (preselect) DT_LAUNCH_TSDF(false, true); else DT_LAUNCH_TSDF(false, false);
  }
#undef DT_LAUNCH_TSDF
  return check_launch("dt_tsdf_integrate_f16");
}

int dt_tsdf_integrate_frames_f16(uint16_t* values, uint16_t* weights, uint32_t* active, const float* origin3,
                                 float voxel_size, int X, int Y, int Z, const uint16_t* depth, int num_frames, int img_h,
                                 int img_w, const float* frame_params, const dt_tsdf_thresholds* th, dt_stream_t s) {
  return integrate_frames(values, weights, active, origin3, voxel_size, X, Y, Z, depth, false, num_frames, img_h, img_w, frame_params,
                          th, s);
}

int dt_tsdf_integrate_frames_f32depth_f16(uint16_t* values, uint16_t* weights, uint32_t* active, const float* origin3,
                                          float voxel_size, int X, int Y, int Z, const float* depth_f32, int num_frames, int img_h,
                                          int img_w, const float* frame_params, const dt_tsdf_thresholds* th, dt_stream_t s) {
  return integrate_frames(values, weights, active, origin3, voxel_size, X, Y, Z, depth_f32, true, num_frames, img_h, img_w,
                          frame_params, th, s);
}

int dt_tsdf_integrate_frames_xslab_f16(uint16_t* values, uint16_t* weights, uint32_t* active, const float* origin3,
                                       float voxel_size, int X, int Y, int Z, int x_begin, int x_count, const void* depth,
                                       int depth_is_f32, int num_frames, int img_h, int img_w, const float* frame_params,
                                       const dt_tsdf_thresholds* th, dt_stream_t s) {
  return integrate_frames(values, weights, active, origin3, voxel_size, X, Y, Z, depth, depth_is_f32 != 0, num_frames, img_h,
                          img_w, frame_params, th, s, x_begin, x_count);
}

int dt_tsdf_sample_f16(const uint16_t* volume, const float* origin3, float voxel_size, int X, int Y, int Z,
                       const float* points_n3, float* out_n, int64_t n, int fp16_math, dt_stream_t s) {
  DT_REQUIRE(volume && origin3 && points_n3 && out_n, "dt_tsdf_sample_f16: null pointer");
  DT_REQUIRE(X > 1 && Y > 1 && Z > 1 && n >= 0 && voxel_size > 0.f, "dt_tsdf_sample_f16: bad extents");
  if (n == 0) return 0;
  DT_LAUNCH(tsdf_sample_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, to_stream(s), volume, origin3[0],
                     origin3[1], origin3[2], voxel_size, X, Y, Z, points_n3, out_n, n, fp16_math);
  return check_launch("dt_tsdf_sample_f16");
}

int dt_hint_from_depth_f32(const float* depth_hw, const uint16_t* weights_vol_f16, const float* origin3, float voxel_size,
                           int X, int Y, int Z, const float* invK_44, const float* world_T_cam_44, float threshold,
                           int h, int w, float* hint_hw, float* mask_hw, uint8_t* mask_b_hw, float* sampled_weights_hw,
                           int fp16_math, dt_stream_t s) {
  DT_REQUIRE(depth_hw && weights_vol_f16 && origin3 && invK_44 && world_T_cam_44 && hint_hw && mask_hw &&
                 mask_b_hw && sampled_weights_hw,
             "dt_hint_from_depth_f32: null pointer");
  DT_REQUIRE(X > 1 && Y > 1 && Z > 1 && h > 0 && w > 0 && voxel_size > 0.f, "dt_hint_from_depth_f32: bad extents");
  DT_LAUNCH(hint_from_depth_kernel, dim3((unsigned)((h * w + 255) / 256)), dim3(256), 0, to_stream(s), depth_hw,
                     weights_vol_f16, origin3[0], origin3[1], origin3[2], voxel_size, X, Y, Z, invK_44, world_T_cam_44, threshold, h, w, hint_hw,
                     mask_hw, mask_b_hw, sampled_weights_hw, fp16_math);
  return check_launch("dt_hint_from_depth_f32");
}

}  // extern "C"
