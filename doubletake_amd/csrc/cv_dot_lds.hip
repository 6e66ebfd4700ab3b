// Dot-product plane-sweep volume with the source-feature footprint staged in LDS.  gfx950 only.
//
// Replaces CostVolumeManager.build_cost_volume (reference modules/cost_volume.py:219-315) =
// warp_features (:132-217: back-project, project, grid_sample bilinear/zeros) + channel dot + (z' > 0) mask
// + sum over source views, for every depth plane.
//
// Why LDS: one (pixel, plane, view) sample reads 4 bilinear taps x 64 B.  At 640x480 / 7 views / 64 planes that is
// 2.2 GB of taps against 14.75 MB of compulsory HBM traffic; pulled through the texture path (the first version of
// this kernel, still available as dt_cv_dot_direct_f32) it runs at the L1 gather rate (0.148 ms).
//
// Structure (wave64-native, no workgroup barrier anywhere): a WAVE owns an 8x8-pixel tile (one pixel per lane) and a
// group of up to 8 consecutive planes; the four waves of a workgroup are four neighbouring tiles of the same plane
// group and never synchronise.  Per source view the wave
//   1. bounds everything its tile can touch in that view for its plane range: the tile maps through a homography
//      per plane (convex quadrilateral) and every pixel slides along its epipolar line between the first and the
//      last plane, so the 4 tile corners at the two end planes bound all taps.  All views are bounded in ONE pass
//      (lane 8v+4e+c projects corner c at end e into view v; three xor-shuffles reduce each view's eight lanes);
//   2. copies that box of NHWC texels into its private 13 KB LDS slab with LDS-DMA (global_load_lds_dwordx4: 64 x 16 B
//      per wave-instruction at a wave-uniform LDS base = the row-major box layout; nothing passes through VGPRs);
//   3. samples every plane of the range from LDS: 4 taps x 4 ds_read_b128 per lane, blend + dot in packed fp32.
// If a box does not fit, the plane range is halved (wave-uniform); a box that would serve a single plane is not
// staged at all (it moves more bytes into LDS than its samples read back: measured) and that (view, plane) is
// sampled from global memory; a view the tile cannot see is skipped (it adds exactly 0).
//
// Bank layout: a texel is 64 B = 16 banks, so 16 lanes reading the same 16-byte quad of 16 neighbouring texels
// would hit 4 bank groups four times over.  Lane l therefore walks the four quads of a texel in the rotated order
// (s + (l >> 2)) & 3 (its current-view feature quads are pre-rotated the same way, so no register is indexed
// dynamically): the four lane quads that ds_read_b128 services together (MI355X_MICROARCH.md LDS table) then
// carry four different rotations and a unit-stride warp is conflict free.
//
// Robustness: the box logic is an optimisation, never a correctness assumption.  A view whose corner projections
// come within 1e-3 of the camera plane, a box that does not fit in LDS even for a single plane, or an individual
// tap that falls outside the staged box (cannot happen in exact arithmetic; counted by the _stats_ entry point)
// all take the direct global-memory path, which evaluates the same expressions in the same order: the staged and
// the direct kernel produce bit-identical volumes (tests/test_volume_gpu.py).
//
// Measured (profiles/): cfg2 (B=1) 0.148 -> 0.087 ms; cfg3 (B=8, 512x384) 0.61 -> 0.32 ms.
// Ablations at cfg2: sampling only (no staging) 0.077 ms, staging only 0.099 ms -- at B=1 both are latency/tail
// bound (600 workgroups, the near-plane groups split into many small boxes); at B=8 staging runs at the L2->LDS
// rate (1.5 GB of boxes in 0.16 ms = 9 TB/s).
#include <cstdlib>

#include "common.hpp"
#include "cv_geometry.hpp"

namespace dt {

#ifdef DT_DOT_TIMING
// experiment builds only (scripts/dot_wave_times.py): per-wave (start, end) realtime stamps (100 MHz) of the last launch
__device__ unsigned long long g_dot_times[16384 * 3];
#endif

constexpr int kDotTile = 8;            // a wave owns an 8 x 8 pixel tile, one pixel per lane
#ifndef DT_DOT_CAP
#define DT_DOT_CAP 208
#endif
#ifndef DT_DOT_OCC
#define DT_DOT_OCC 3
#endif
constexpr int kDotCapTexels = DT_DOT_CAP;  // per-wave staged box: 208 texels x 64 B = 13 KB -> 52 KB per 4-wave workgroup, 3 per CU
constexpr int kDotMaxGroup = 8;        // planes per wave (accumulators per lane)
constexpr int kDotWaves = 4;           // independent waves per workgroup (no workgroup barrier anywhere)

typedef float v2f __attribute__((ext_vector_type(2)));

#ifndef DT_DOT_MIN_PLANES
#define DT_DOT_MIN_PLANES 2  // stage a box only when it serves at least this many planes (measured: 1 -> 0.137 ms, 2 -> 0.106 ms, 3 -> 0.111 ms at cfg2)
#endif
#ifndef DT_DOT_ABL
#define DT_DOT_ABL 0  // ablation switches (timing experiments only): 1 = skip the LDS-DMA staging, 2 = skip the sampling
#endif

// per-sample state that survives from the address phase to the arithmetic phase
struct DotSample {
  float w00, w01, w10, w11;
  float z;
  int o00, o01, o10, o11;  // LDS float offsets of the four taps (staged) / texel indices y*w+x (direct)
  bool need, stray;
};

// One quad step of the blend + dot, two channels per instruction.  acc2 carries the even/odd channel partial sums.
__device__ __forceinline__ void blend_dot(const float4& a, const float4& b, const float4& c, const float4& d, const DotSample& sm,
                                          const float* cq, v2f& acc2) {
  v2f f0 = v2f{a.x, a.y} * sm.w00 + v2f{b.x, b.y} * sm.w01 + v2f{c.x, c.y} * sm.w10 + v2f{d.x, d.y} * sm.w11;
  v2f f1 = v2f{a.z, a.w} * sm.w00 + v2f{b.z, b.w} * sm.w01 + v2f{c.z, c.w} * sm.w10 + v2f{d.z, d.w} * sm.w11;
  acc2 += f0 * v2f{cq[0], cq[1]};
  acc2 += f1 * v2f{cq[2], cq[3]};
}

__device__ __forceinline__ float xor_min(float v, int m) { return fminf(v, __shfl_xor(v, m, 64)); }
__device__ __forceinline__ float xor_max(float v, int m) { return fmaxf(v, __shfl_xor(v, m, 64)); }

// MODE 0: LDS staging with direct fallback; MODE 1: direct path only (parity / ablation entry point)
template <int MODE>
__global__ __launch_bounds__(64 * kDotWaves, DT_DOT_OCC) void cv_dot_lds_kernel(const float* __restrict__ cur_bchw,
                                                                       const float* __restrict__ src_bkhwc,
                                                                       const float* __restrict__ params,
                                                                       float* __restrict__ vol, int K, int h, int w, int D,
                                                                       int group, int* __restrict__ stats) {
  constexpr int C = 16;
  __shared__ __attribute__((aligned(16))) float box_all[kDotWaves * kDotCapTexels * C];

#ifdef DT_DOT_TIMING
  const unsigned long long t_begin = __builtin_amdgcn_s_memrealtime();
#endif
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  float* box = box_all + wave * (kDotCapTexels * C);
  const int b = blockIdx.z;
  const int tiles_x = (w + kDotTile - 1) / kDotTile, tiles_y = (h + kDotTile - 1) / kDotTile;
  // the four waves of a workgroup take four neighbouring tiles of the SAME plane group: similar cost, one release
  const int tile = blockIdx.x * kDotWaves + wave;
  if (tile >= tiles_x * tiles_y) return;  // (no barriers in this kernel)
  const int tx0 = (tile % tiles_x) * kDotTile, ty0 = (tile / tiles_x) * kDotTile;
  const int x = tx0 + (lane & 7), y = ty0 + (lane >> 3);
  const bool live = (x < w) && (y < h);
  const int xc = min(x, w - 1), yc = min(y, h - 1);
  cfloat_ptr p = as_const(params + (size_t)b * cv_params_floats(D, K));
  const size_t hw = (size_t)h * w;

  // current-view features, quads pre-rotated by this lane's rotation
  const int rot = (lane >> 2) & 3;
  float cur[C];
  {
    float raw[C];
#pragma unroll
    for (int c = 0; c < C; ++c) raw[c] = cur_bchw[((size_t)b * C + c) * hw + (size_t)yc * w + xc];
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float v0 = raw[((s + 0) & 3) * 4 + j], v1 = raw[((s + 1) & 3) * 4 + j], v2 = raw[((s + 2) & 3) * 4 + j],
                    v3 = raw[((s + 3) & 3) * 4 + j];
        cur[s * 4 + j] = (rot == 0) ? v0 : ((rot == 1) ? v1 : ((rot == 2) ? v2 : v3));
      }
  }
  // float offsets of the four quads of a texel in this lane's order
  const int qd0 = ((0 + rot) & 3) * 4, qd1 = ((1 + rot) & 3) * 4, qd2 = ((2 + rot) & 3) * 4, qd3 = ((3 + rot) & 3) * 4;
  float rx, ry, rz;
  pixel_ray(p + kCvInvK, xc, yc, rx, ry, rz);
  const float inv_w = 1.0f / (float)w, inv_h = 1.0f / (float)h;

  // ray of the tile corner this lane evaluates in the box pass: lane = 8*view + 4*end + corner
  float crx, cry, crz;
  {
    const int c = lane & 3;
    const int cx = min(tx0 + ((c & 1) ? kDotTile - 1 : 0), w - 1), cy = min(ty0 + ((c & 2) ? kDotTile - 1 : 0), h - 1);
    pixel_ray(p + kCvInvK, cx, cy, crx, cry, crz);
  }

  // (round 4: handing the plane groups out far planes first -- the expensive ones, so that the partial second round of
  //  workgroups would be the cheap near groups -- measured slower, 0.0845 -> 0.0885 ms at cfg2, 0.303 -> 0.314 at B=8)
  const int pg = (int)blockIdx.y;
  const int ds = pg * group, de = min(ds + group, D);
  int d0 = ds;
  while (d0 < de) {
    int d1 = de;
    float acc[kDotMaxGroup];
#pragma unroll
    for (int j = 0; j < kDotMaxGroup; ++j) acc[j] = 0.f;

    for (int kb = 0; kb < K; kb += 8) {
      // ---- boxes of views kb..kb+7 for planes [d0, d1), all in one pass: lane 8v+4e+c projects tile corner c at the
      // first (e=0) / last (e=1) plane of the range into view kb+v; three xor-shuffles reduce each view's 8 lanes.
      // (The tile maps through a homography per plane -- a convex quadrilateral -- and every pixel slides along its
      // epipolar line between the two end planes, so these 8 points bound every tap.)  Shrink the range until all fit.
      int vx0 = 0, vy0 = 0, vbw = 0, vbh = 0, vdirect = 1;  // per-lane copy of "its" view's box (lanes of view v agree)
      if (MODE == 0) {
        for (;;) {
          const int v = min(kb + (lane >> 3), K - 1);
          const float depth = p[kCvPlanes + ((lane & 4) ? d1 - 1 : d0)];
          const ViewProj q = project_view(params + (size_t)b * cv_params_floats(D, K) + cv_view_off(D, v), depth * crx,
                                          depth * cry, depth * crz);
          int bad = !(q.z > 1e-3f) || !(fabsf(q.u) < 1.0e6f) || !(fabsf(q.v) < 1.0e6f);
          float lo_x = q.u, hi_x = q.u, lo_y = q.v, hi_y = q.v;
#pragma unroll
          for (int m = 1; m < 8; m <<= 1) {
            lo_x = xor_min(lo_x, m);
            hi_x = xor_max(hi_x, m);
            lo_y = xor_min(lo_y, m);
            hi_y = xor_max(hi_y, m);
            bad |= __shfl_xor(bad, m, 64);
          }
          // sample index = u - 0.5 (grid_sample, align_corners=False); taps floor(i), floor(i)+1; one texel of slack
          const int xa = max((int)floorf(lo_x - 0.5f) - 1, 0), xb = min((int)floorf(hi_x - 0.5f) + 2, w - 1);
          const int ya = max((int)floorf(lo_y - 0.5f) - 1, 0), yb = min((int)floorf(hi_y - 0.5f) + 2, h - 1);
          vx0 = xa;
          vy0 = ya;
          vbw = bad ? 0 : max(xb - xa + 1, 0);
          vbh = bad ? 0 : max(yb - ya + 1, 0);
          if (vbw == 0 || vbh == 0) vbw = vbh = 0;  // the whole footprint misses the image: every tap weighs zero
          vdirect = bad;
          const bool fits = vdirect || vbw * vbh <= kDotCapTexels;
          if (__all(fits) || d1 - d0 == 1) break;
          d1 = d0 + (d1 - d0 + 1) / 2;
        }
      }

      // plane depths of the range: read once, so that no scalar-memory load (which shares the LDS wait counter and
      // returns out of order) sits inside the sampling loop
      float pd[kDotMaxGroup];
#pragma unroll
      for (int j = 0; j < kDotMaxGroup; ++j) pd[j] = p[kCvPlanes + min(d0 + j, D - 1)];

      for (int k = kb; k < min(kb + 8, K); ++k) {
        // q(depth) = P [depth r; 1] = depth (P3 r) + P[:,3]: the 3x3 part is applied to this lane's ray ONCE per view, a
        // plane then costs three FMAs instead of a 3x4 matrix-vector product.  Staged and direct path share the expression.
        float pa[3], pt[3];
        {
          cfloat_ptr vpc = p + cv_view_off(D, k);
#pragma unroll
          for (int i = 0; i < 3; ++i) {
            pa[i] = vpc[4 * i + 0] * rx + vpc[4 * i + 1] * ry + vpc[4 * i + 2] * rz;
            pt[i] = vpc[4 * i + 3];
          }
        }
        const float* base = src_bkhwc + ((size_t)b * K + k) * hw * C;
        int bx0 = 0, by0 = 0, bw = 0, bh = 0;
        bool direct = true;
        if (MODE == 0) {
          const int src_lane = (k - kb) * 8;
          bx0 = __builtin_amdgcn_readlane(vx0, src_lane);
          by0 = __builtin_amdgcn_readlane(vy0, src_lane);
          bw = __builtin_amdgcn_readlane(vbw, src_lane);
          bh = __builtin_amdgcn_readlane(vbh, src_lane);
          direct = __builtin_amdgcn_readlane(vdirect, src_lane) != 0 || bw * bh > kDotCapTexels;  // one plane that does not fit
          const bool empty = !direct && bw == 0;
          // a box that serves a single plane moves more bytes into LDS than its samples read back out of it
          direct = direct || (!empty && (d1 - d0) < DT_DOT_MIN_PLANES);
          if (stats && lane == 0) atomicAdd(stats + (direct ? 1 : (empty ? 2 : 0)), 1);
          if (empty) continue;  // nothing of this view is visible from the tile: adds 0
        }

        if (MODE == 0 && !direct) {
          // ---- stage the box with LDS-DMA (global_load_lds_dwordx4): each wave-instruction drops 64 x 16 B at a
          // wave-uniform LDS base, which is exactly the row-major box layout; nothing passes through VGPRs and all
          // copies are in flight at once.  The wave's own earlier ds_reads of the slab have all returned (their values
          // were consumed), so overwriting it needs no barrier.
          const int row4 = bw * 4;  // float4 per box row
          const int n4 = row4 * bh;
          const float inv_row4 = 1.0f / (float)row4;
          for (int i0 = 0; i0 < n4; i0 += 64) {
            const int i = i0 + lane;
            int r = (int)((float)i * inv_row4);
            int c4 = i - r * row4;
            if (c4 < 0) { --r; c4 += row4; }          // float reciprocal may be one off
            if (c4 >= row4) { ++r; c4 -= row4; }
            const float* gsrc = base + ((size_t)(by0 + r) * w + bx0) * C + c4 * 4;
            float* ldst = box + (size_t)i0 * 4;
            if (!(DT_DOT_ABL & 1) && i < n4)
              __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                               (__attribute__((address_space(3))) void*)ldst, 16, 0, 0);
          }
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }

        // ---- address phase of one plane: projection, taps, LDS offsets ------------------------------------------------
        auto prepare = [&](const float depth, DotSample& sm, const bool staged) {
          ViewProj q;
          {
            const float qx = depth * pa[0] + pt[0], qy = depth * pa[1] + pt[1], qz = depth * pa[2] + pt[2];
            project_scale(qx, qy, qz, q);  // Project3D (utils/geometry_utils.py:82-93), reciprocal by v_rcp + Newton
          }
          const Taps t = bilinear_taps<true>(q.u, q.v, h, w, inv_w, inv_h);
          sm.w00 = t.w00; sm.w01 = t.w01; sm.w10 = t.w10; sm.w11 = t.w11;
          sm.z = q.z;
          sm.need = (t.w00 != 0.f) || (t.w01 != 0.f) || (t.w10 != 0.f) || (t.w11 != 0.f);
          if (staged) {
            const bool in_box = t.x0 >= bx0 && t.x1 < bx0 + bw && t.y0 >= by0 && t.y1 < by0 + bh;
            sm.stray = sm.need && !in_box;
            const bool ok = sm.need && in_box;  // otherwise read texel 0 of the box (weights are zero / lane is redone)
            const int r0 = (t.y0 - by0) * bw - bx0, r1 = (t.y1 - by0) * bw - bx0;
            sm.o00 = ok ? (r0 + t.x0) * C : 0;
            sm.o01 = ok ? (r0 + t.x1) * C : 0;
            sm.o10 = ok ? (r1 + t.x0) * C : 0;
            sm.o11 = ok ? (r1 + t.x1) * C : 0;
          } else {
            sm.stray = false;
            sm.o00 = t.y0 * w + t.x0;
            sm.o01 = t.y0 * w + t.x1;
            sm.o10 = t.y1 * w + t.x0;
            sm.o11 = t.y1 * w + t.x1;
          }
        };
        // same expressions from global memory: whole views without a box, or a stray tap
        auto direct_dot = [&](const DotSample& sm) -> float {
          const float* g00 = base + (size_t)sm.o00 * C;
          const float* g01 = base + (size_t)sm.o01 * C;
          const float* g10 = base + (size_t)sm.o10 * C;
          const float* g11 = base + (size_t)sm.o11 * C;
          v2f a2 = v2f{0.f, 0.f};
#define DT_DOT_GSTEP(QD, S)                                                                                              \
          blend_dot(*reinterpret_cast<const float4*>(g00 + QD), *reinterpret_cast<const float4*>(g01 + QD),               \
                    *reinterpret_cast<const float4*>(g10 + QD), *reinterpret_cast<const float4*>(g11 + QD), sm, cur + S * 4, a2)
          DT_DOT_GSTEP(qd0, 0);
          DT_DOT_GSTEP(qd1, 1);
          DT_DOT_GSTEP(qd2, 2);
          DT_DOT_GSTEP(qd3, 3);
#undef DT_DOT_GSTEP
          return a2.x + a2.y;
        };

        if (MODE == 0 && !direct && (DT_DOT_ABL & 2)) {
          acc[0] += box[lane * 4];  // ablation: staging only
        } else if (MODE == 0 && !direct) {
          unsigned stray_mask = 0u;
#pragma unroll
          for (int j = 0; j < kDotMaxGroup; ++j) {
            if (d0 + j < d1) {
              DotSample sm;
              prepare(pd[j], sm, true);
              const float* q00 = box + sm.o00;
              const float* q01 = box + sm.o01;
              const float* q10 = box + sm.o10;
              const float* q11 = box + sm.o11;
              v2f a2 = v2f{0.f, 0.f};
#define DT_DOT_LSTEP(QD, S)                                                                                              \
              blend_dot(*reinterpret_cast<const float4*>(q00 + QD), *reinterpret_cast<const float4*>(q01 + QD),           \
                        *reinterpret_cast<const float4*>(q10 + QD), *reinterpret_cast<const float4*>(q11 + QD), sm, cur + S * 4, a2)
              DT_DOT_LSTEP(qd0, 0);
              DT_DOT_LSTEP(qd1, 1);
              DT_DOT_LSTEP(qd2, 2);
              DT_DOT_LSTEP(qd3, 3);
#undef DT_DOT_LSTEP
              const float dot = (sm.need && !sm.stray) ? a2.x + a2.y : 0.f;
              stray_mask |= sm.stray ? (1u << j) : 0u;
              acc[j] += (sm.z > 0.f) ? dot : 0.f;
            }
          }
          if (__builtin_expect(stray_mask != 0u, 0)) {
            // a tap of this lane missed the staged box (not expected): redo those planes from global memory
#pragma unroll
            for (int j = 0; j < kDotMaxGroup; ++j) {
              if ((stray_mask >> j) & 1u) {
                DotSample sm;
                prepare(pd[j], sm, false);
                const float dot = direct_dot(sm);
                acc[j] += (sm.z > 0.f) ? dot : 0.f;
                if (stats) atomicAdd(stats + 3, 1);
              }
            }
          }
        } else {
#pragma unroll
          for (int j = 0; j < kDotMaxGroup; ++j) {
            if (d0 + j < d1) {
              DotSample sm;
              prepare(pd[j], sm, false);
              const float dot = sm.need ? direct_dot(sm) : 0.f;
              acc[j] += (sm.z > 0.f) ? dot : 0.f;
            }
          }
        }
      }
      // a wave whose plane range was shrunk while looking at views kb.. keeps the shrunk range for the remaining views
      // (K <= 8 in every configuration of the reference, so this loop runs once)
    }
    if (live) {
#pragma unroll
      for (int j = 0; j < kDotMaxGroup; ++j)
        if (d0 + j < d1) vol[((size_t)b * D + d0 + j) * hw + (size_t)y * w + x] = acc[j];
    }
    d0 = d1;
  }
#ifdef DT_DOT_TIMING
  if (lane == 0) {
    const unsigned slot = ((blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * kDotWaves + wave;
    if (slot < 16384) {
      g_dot_times[slot * 3 + 0] = t_begin;
      g_dot_times[slot * 3 + 1] = __builtin_amdgcn_s_memrealtime();
      g_dot_times[slot * 3 + 2] = ((unsigned long long)pg << 32) | (unsigned)(blockIdx.x * kDotWaves + wave);
    }
  }
#endif
}

}  // namespace dt

using namespace dt;

extern "C" {

#ifdef DT_DOT_TIMING
int dt_debug_dot_times(unsigned long long* out, int n) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_dot_times), sizeof(unsigned long long) * n) == hipSuccess ? 0 : 1;
}
#endif

static int dot_launch(int mode, const float* cur, const float* src, const float* params, float* vol, int batch, int num_src,
                      int channels, int h, int w, int num_planes, int* stats, dt_stream_t s, const char* what) {
  DT_REQUIRE(batch > 0 && num_src > 0 && h > 0 && w > 0 && num_planes > 0, "%s: bad extents", what);
  DT_REQUIRE(channels == 16, "%s: channels=%d unsupported (matching_feature_dims must be 16)", what, channels);
  DT_REQUIRE(cur && src && params && vol, "%s: null pointer", what);
  // a plane range that was shrunk for views 0..7 must also be valid for later views: only K <= 8 keeps the
  // "one pass over the views per plane range" structure; more views take the direct kernel
  if (num_src > 8) mode = 1;
  const int tiles = ((w + kDotTile - 1) / kDotTile) * ((h + kDotTile - 1) / kDotTile);
  const int wgs = (tiles + kDotWaves - 1) / kDotWaves;
  // planes per wave: up to 8, fewer while the grid has fewer than ~4 workgroups per CU -- the units differ a lot in cost
  // (near planes split into many small boxes), so a B=1 frame needs the finer grain to balance: measured at cfg2
  // 8 planes 0.105 ms, 4 planes 0.087 ms, 2 planes 0.091 ms; at B=8 (3072 workgroups with 8 planes) 8 planes stay best
  // Round 4 (scripts/dot_wave_times.py -> profiles/r4u_dot_wave_times.txt): at cfg2 every wave is busy for 21-44 us whatever
  // its planes (vector-instruction throughput: 121 plain + 40 packed per wave-sample), so the launch lasts two rounds of the 768
  // resident workgroup slots, the second 56 % full.  Tried against that: one balanced span of 6-7 planes per wave, all resident
  // at once (0.109 ms against 0.089: ranges of more than 4 planes stage worse, as with the fixed 8-plane grid); half-size groups
  // at the far end of the plane list cannot shorten the schedule either (1.56 rounds of work need quarter-size units to beat 2.0,
  // and single-plane ranges are not worth staging).
  int group = kDotMaxGroup;
  while (group > 2 && (long)wgs * batch * ((num_planes + group - 1) / group) < 1024) group >>= 1;
  static const int force_group = [] { const char* e = getenv("DT_DOT_GROUP"); return e ? atoi(e) : 0; }();  // tuning hook
  if (force_group >= 1 && force_group <= kDotMaxGroup) group = force_group;
  dim3 grid(wgs, (num_planes + group - 1) / group, batch);
  if (mode == 0)
    DT_LAUNCH(cv_dot_lds_kernel<0>, grid, dim3(64 * kDotWaves), 0, to_stream(s), cur, src, params, vol, num_src, h, w,
                       num_planes, group, stats);
  else
    DT_LAUNCH(cv_dot_lds_kernel<1>, grid, dim3(64 * kDotWaves), 0, to_stream(s), cur, src, params, vol, num_src, h, w,
                       num_planes, group, stats);
  return check_launch(what);
}

int dt_cv_dot_f32(const float* cur, const float* src, const float* params, float* vol, int batch, int num_src, int channels,
                  int h, int w, int num_planes, dt_stream_t s) {
  return dot_launch(0, cur, src, params, vol, batch, num_src, channels, h, w, num_planes, nullptr, s, "dt_cv_dot_f32");
}

int dt_cv_dot_stats_f32(const float* cur, const float* src, const float* params, float* vol, int batch, int num_src,
                        int channels, int h, int w, int num_planes, int* stats4, dt_stream_t s) {
  DT_REQUIRE(stats4, "dt_cv_dot_stats_f32: null stats pointer");
  return dot_launch(0, cur, src, params, vol, batch, num_src, channels, h, w, num_planes, stats4, s, "dt_cv_dot_stats_f32");
}

int dt_cv_dot_direct_f32(const float* cur, const float* src, const float* params, float* vol, int batch, int num_src,
                         int channels, int h, int w, int num_planes, dt_stream_t s) {
  return dot_launch(1, cur, src, params, vol, batch, num_src, channels, h, w, num_planes, nullptr, s, "dt_cv_dot_direct_f32");
}

}  // extern "C"
