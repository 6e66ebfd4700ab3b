// Fused plane-sweep + matching-MLP (+ hint-MLP) cost volume for gfx950, fp32 MFMA.
//
// Replaces (paths relative to /root/reference/src/doubletake/):
//   FeatureVolumeManager.build_cost_volume            modules/feature_volume.py:81-356
//   FeatureMeshHintVolumeManager.build_cost_volume    modules/mesh_hint_volume.py:84-393
//   FastFeatureMeshHintVolumeManager.build_cost_volume modules/mesh_hint_volume.py:679-928
// i.e. per (pixel, plane): backproject -> project into K source views -> bilinear warp ->
// metadata -> MLP [Cin,128,128,1] (LeakyReLU 0.01) -> hint MLP [3,12,12,1].  The reference
// materialises a [pairs, Cin] matrix (993 MB at 640x480/K7/D64 in the Fast variant); here the
// input vector only ever exists as MFMA B-operands in registers.
//
// Mapping onto v_mfma_f32_32x32x2_f32 (D[i][j] += A[i][k] B[k][j], exact fp32):
//   i = output feature (A = weights, read from LDS),  j = pixel (B = inputs, in registers),
//   lane l = (pixel l&31, half l>>5); half h supplies k-slot h of every MFMA step.
//   C/D layout: lane (p,h) holds features (r&3)+8*(r>>2)+4h of each 32-feature block, so the
//   layer-1 accumulators are, after bias+LeakyReLU, exactly the B-operands layer 2 needs for
//   the same lane: no transpose, no LDS round trip between the layers.
//   The K order of each GEMM is a free permutation; the host packs the weights to match
//   (doubletake_amd/modules/mlp_pack.py mirrors the step tables below).
//
// Work decomposition: a wave owns 32 consecutive pixels x PG planes.  The 16+3+3K+1 input
// columns that do not depend on the plane (cur features, cur ray, pose metrics, bias) are
// contracted once per task into `accp` and reused as the initial layer-1 accumulator of every
// plane.  One 256-thread workgroup per CU keeps the plane-dependent layer-1 weights, layer-2
// weights and the tail (b2, W3, b3) resident in LDS (152.6 KB for K=7).
#include <cstdlib>
#include <algorithm>
#include <atomic>
#include <map>
#include <mutex>
#include <utility>
#include <vector>

#include "common.hpp"
#include "cv_geometry.hpp"

namespace dt {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kF = 16;             // matching feature channels
constexpr int kStepsPerView = 12;  // 8 feature steps + 4 metadata steps
constexpr int kPixFixed = 10;      // 8 cur-feature steps + (ray.x|ray.y) + (ray.z|bias)
constexpr int kStepFloats = 256;   // [2 halves][32 lanes][4 blocks]
constexpr int kW2Steps = 64;
constexpr int kTailFloats = 260;   // b2r[128], w3r[128], b3, pad[3]
constexpr int kHintFloats = 220;   // hint MLP (217 floats) staged in LDS behind the tail
constexpr int kStageFloats = 256;  // per wave: 32 pixels x 8 planes of finished scores awaiting a 32-byte store
constexpr int kMaxSrcMfma = 7;     // LDS budget: 12*K + 64 + ~1 KB <= 160 KB
constexpr int kMaxSrcStream = 15;  // with the views beyond the seventh streamed from global memory (STREAM instantiation)

// First plane-dependent layer-1 step of source view k (mirrors mlp_pack.view_step_base).  Round 5: a view has 7 plane-dependent
// metadata inputs (mask, z', dot, angle, source ray xyz) = 3.5 two-slot K steps.  View 0 keeps four steps (its spare slot
// carries the plane depth); the further views are PAIRED (1,2), (3,4), ...: the first of a pair runs four steps whose last slot
// carries its partner's mask -- known by then: the next view has already been projected for the prefetch --, the second three.
// 25 instead of 28 metadata steps at K = 7: 12 MFMAs less per (tile, plane), 3 KB less LDS.  An unpaired last view keeps four.
#ifndef DT_MLP_PAIR_META
#define DT_MLP_PAIR_META 1  // (0: four metadata steps for every view, the layout of rounds 1-4; A/B switch -- the host packs for
#endif                      //  whichever layout dt_cv_mlp_pack_floats reports)
__host__ __device__ inline int mlp_view_step_base(int k) {
  if (!DT_MLP_PAIR_META) return k * kStepsPerView;
  return k <= 0 ? 0 : kStepsPerView + (2 * kStepsPerView - 1) * ((k - 1) / 2) + (((k - 1) & 1) ? kStepsPerView : 0);
}
__host__ __device__ inline int mlp_w1dyn_floats(int K) { return mlp_view_step_base(K) * kStepFloats; }
__host__ __device__ inline int mlp_w1pix_floats(int K) { return (kPixFixed + 2 * K) * kStepFloats; }
constexpr int kW2Floats = kW2Steps * kStepFloats;

#ifdef DT_MLP_TIMING
// experiment builds only (scripts/mlp_wave_times.py): per-wave (start, end, first-task end, id) realtime stamps of the last
// launch (100 MHz counter) and the time at the end of each of a wave's first 24 planes
__device__ unsigned long long g_mlp_times[4096 * 4];
__device__ unsigned long long g_mlp_prog[2048 * 24];
#endif

struct MlpArgs {
  const float* cur;       // [b,16,h,w]
  const float* src;       // [b,K,h,w,16]
  const float* params;    // dt_cv_setup_f32 block
  const float* w1dyn;     // packed
  const float* w1pix;     // packed (read from global)
  const float* w2p;       // packed
  const float* tail;      // packed
  const float* hint_mlp;  // 217 floats or null
  const float* hint_d;    // [b,1,H2,W2]
  const float* hint_w;
  const float* hint_m;
  float* vol;
  int hint_h, hint_w2;
  int out_nhwc;
  int B, K, h, w, D;
  int num_tiles;          // 32-pixel tiles per batch element
  const int* tile_order;  // [num_tiles] position in the processing order -> tile, or null (row-major order)
  long total_units;       // B * num_tiles * D (tile, plane) units, split over the resident waves
  const int* span_bounds; // [gridDim.x * NWAVES + 1] first unit of every wave's span from the cost-aware plan, or null (even / weighted split)
  int old_share_q16;      // share (x 65536) of a workgroup's units that its older four waves take (32768 = even split)
};
DT_ARG_POINTERS(MlpArgs, offsetof(MlpArgs, cur), offsetof(MlpArgs, src), offsetof(MlpArgs, params), offsetof(MlpArgs, w1dyn),
                offsetof(MlpArgs, w1pix), offsetof(MlpArgs, w2p), offsetof(MlpArgs, tail), offsetof(MlpArgs, hint_mlp),
                offsetof(MlpArgs, hint_d), offsetof(MlpArgs, hint_w), offsetof(MlpArgs, hint_m), offsetof(MlpArgs, vol),
                offsetof(MlpArgs, tile_order), offsetof(MlpArgs, span_bounds));

// LeakyReLU(0.01) in two vector instructions: max(x, 0.01 x) == med3(x, 0.01 x, +inf) exactly; fmaxf() costs a third one
// (it canonicalises x first).  pinf must be an OPAQUE +inf (see the kernel): a literal is folded back into fmaxf().
// NaN: v_med3_f32 with a NaN operand falls back to min3, which returns the non-NaN operand -- a NaN activation leaves this
// function as +inf, where the reference's leaky_relu (and the fmaxf form) returns NaN.  Finite inputs cannot produce a NaN
// here (weights and features are finite, masked hints never enter as NaN: mesh_hint_volume.py:186-214), and a +inf still
// poisons the pixel's score visibly; noted because it changes what a NaN diagnostic would show.
__device__ __forceinline__ float lrelu01(float x, float pinf) { return __builtin_amdgcn_fmed3f(x, 0.01f * x, pinf); }

// one source view's gathered taps + metadata, produced by issue_view(), consumed later
struct ViewData {
  float4 t00a, t00b, t01a, t01b, t10a, t10b, t11a, t11b;  // this half's 8 channels of the 4 taps
  float w00, w01, w10, w11;
  float z, sx, sy, sz, ang;
};

// (rx, ry, rz) = un-normalised current ray, rinv = 1 / its norm: the cosine below needs the unit ray only inside one dot
// product, so the three normalised components are not kept (two registers less across the plane loop)
__device__ __forceinline__ void issue_view(ViewData& v, cfloat_ptr vp,
                                           const float* __restrict__ src_view, float X, float Y, float Z,
                                           float rx, float ry, float rz, float rinv, int h, int w, float inv_w,
                                           float inv_h, int half) {
#ifndef DT_MABL
#define DT_MABL 0
#endif
  const ViewProj q = project_view_fast(vp, X, Y, Z);
  // Tap logic of bilinear_taps() (cv_geometry.hpp) with fewer vector instructions -- on gfx950 every vector instruction
  // beside the fp32 MFMAs costs matrix time (scripts/mfma_filler_bench.hip: 3-5.5 cycles each; the fp32 matrix and vector
  // pipes are not independent), so this kernel counts them.  Same results: a sample that cannot touch the image (incl.
  // NaN / inf) gets the base texel -2, for which every tap fails the unsigned range test below; 1-D weights are zeroed
  // per axis BEFORE the four products (0 * finite == +0 == the reference's zeroed tap).
  const float ix = sample_index(q.u, (float)w, inv_w);  // (round 5: u - 0.5, see cv_geometry.hpp)
  const float iy = sample_index(q.v, (float)h, inv_h);
  const bool any = (ix > -1.0f) & (ix < (float)w) & (iy > -1.0f) & (iy < (float)h);
  const float fx = floorf(ix), fy = floorf(iy);
  int x0 = any ? (int)fx : -2, y0 = any ? (int)fy : -2;
  const float ax = ix - fx, ay = iy - fy;  // (garbage under the sentinel: selected away)
  const float wx0 = (unsigned)x0 < (unsigned)w ? 1.0f - ax : 0.0f;
  const float wx1 = (unsigned)(x0 + 1) < (unsigned)w ? ax : 0.0f;
  const float wy0 = (unsigned)y0 < (unsigned)h ? 1.0f - ay : 0.0f;
  const float wy1 = (unsigned)(y0 + 1) < (unsigned)h ? ay : 0.0f;
  if (DT_MABL & 1) {  // ablation: every lane gathers the same four texels (L1 hits)
    x0 = 0;
    y0 = 0;
  }
  // clamped (always legal) tap addresses as 32-bit byte offsets from the wave-uniform view base: scalar base + vector
  // offset addressing, no 64-bit vector arithmetic
  const int cx0 = min(max(x0, 0), w - 1), cx1 = min(max(x0 + 1, 0), w - 1);
  const int cy0 = min(max(y0, 0), h - 1), cy1 = min(max(y0 + 1, 0), h - 1);
  const unsigned r0 = (unsigned)(cy0 * w), r1 = (unsigned)(cy1 * w), hb = (unsigned)half * 32u;
  const unsigned o00 = (r0 + (unsigned)cx0) * 64u + hb, o01 = (r0 + (unsigned)cx1) * 64u + hb;
  const unsigned o10 = (r1 + (unsigned)cx0) * 64u + hb, o11 = (r1 + (unsigned)cx1) * 64u + hb;
  const char* base = reinterpret_cast<const char*>(src_view);
  v.t00a = reinterpret_cast<const float4*>(base + o00)[0];
  v.t00b = reinterpret_cast<const float4*>(base + o00)[1];
  v.t01a = reinterpret_cast<const float4*>(base + o01)[0];
  v.t01b = reinterpret_cast<const float4*>(base + o01)[1];
  v.t10a = reinterpret_cast<const float4*>(base + o10)[0];
  v.t10b = reinterpret_cast<const float4*>(base + o10)[1];
  v.t11a = reinterpret_cast<const float4*>(base + o11)[0];
  v.t11b = reinterpret_cast<const float4*>(base + o11)[1];
  v.w00 = wx0 * wy0;
  v.w01 = wx1 * wy0;
  v.w10 = wx0 * wy1;
  v.w11 = wx1 * wy1;
  v.z = q.z;
  // source ray = normalize(X - t_src); angle = cos between the two unit rays.  One v_rsq_f32
  // instead of sqrt + divide, and the cosine of two unit vectors is their dot product (the
  // reference's clamp_min(eps) on the norms is inactive): differences ~1e-7, far below tolerance.
  const float sx = X - vp[12], sy = Y - vp[13], sz = Z - vp[14];
  const float inv = rsqrtf(fmaxf(sx * sx + sy * sy + sz * sz, 1e-24f));
  v.sx = sx * inv;
  v.sy = sy * inv;
  v.sz = sz * inv;
  v.ang = (rx * v.sx + ry * v.sy + rz * v.sz) * rinv;
}

// hint MLP [3,12,12,1] evaluated from LDS.  Both lane halves of a wave hold the same pixel, so the 12 hidden
// units of each layer are split between them (half h owns units 6h..6h+5): layer-1 activations are exchanged
// with six lane^32 shuffles, layer-2 partial sums with one.  The outer loop stays rolled: fully unrolled, the
// weight reads are all issued up front and spill ~100 registers in the two-waves-per-SIMD variant.
// hm: V1[12x3] @0, c1[12] @36, V2[12x12] @48, c2[12] @192, V3[12] @204, c3 @216.
__device__ __forceinline__ float hint_mlp_eval_lds(const float* hm, float s, float hint, float hw, int half, float pinf) {
  const int m0 = half * 6, o0 = 6 - m0;
  float own[6], oth[6];
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    const int m = m0 + j;
    own[j] = lrelu01(hm[m * 3 + 0] * s + hm[m * 3 + 1] * hint + hm[m * 3 + 2] * hw + hm[36 + m], pinf);
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
    part += hm[204 + n] * lrelu01(acc, pinf);
  }
  return part + __shfl_xor(part, 32, 64) + hm[216];
}

// skip the feature MFMAs of (tile, plane, view) triples whose footprint misses the source image (A/B switch)
#ifndef DT_MLP_SKIP_EMPTY
#define DT_MLP_SKIP_EMPTY 1
#endif

#define DT_MFMA4(ACC, A4, BVAL)                                                   \
  do {                                                                            \
    ACC[0] = __builtin_amdgcn_mfma_f32_32x32x2f32((A4).x, (BVAL), ACC[0], 0, 0, 0); \
    ACC[1] = __builtin_amdgcn_mfma_f32_32x32x2f32((A4).y, (BVAL), ACC[1], 0, 0, 0); \
    ACC[2] = __builtin_amdgcn_mfma_f32_32x32x2f32((A4).z, (BVAL), ACC[2], 0, 0, 0); \
    ACC[3] = __builtin_amdgcn_mfma_f32_32x32x2f32((A4).w, (BVAL), ACC[3], 0, 0, 0); \
  } while (0)

// NWAVES = 4: one wave per SIMD, up to 512 registers, four accumulator chains in both layers.
// NWAVES = 8: two waves per SIMD (<= 256 registers each); layer 2 runs in four passes of 32 output
//             features (one accumulator chain) to fit the register budget.
// A chain of dependent v_mfma_f32_32x32x2_f32 issues at ~1/4 of the pipe rate, so >= 4 independent
// accumulators must be in flight per SIMD to keep the matrix pipe full.
// STREAM = false: K <= 7 source views, every view's layer-1 fragments resident in LDS (the reference default and every
//                 released checkpoint; the tuned headline path -- its code is untouched by the other instantiation).
// STREAM = true : 7 < K <= 15: the first seven views as above, the fragments of the further views are read from global
//                 memory (L2) inside the view loop -- 12 KB per view and wave-pass, behind the 48 MFMAs of the view.
template <bool HINT, int NWAVES, bool STREAM = false>
__global__ __launch_bounds__(NWAVES * 64, NWAVES / 4) void cv_mlp_mfma_kernel(const MlpArgs a) {
  constexpr int NT = NWAVES * 64;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int K = a.K, D = a.D, h = a.h, w = a.w;
  const int K_lds = STREAM ? min(K, kMaxSrcMfma) : K;  // views whose layer-1 weights live in LDS
  const int n_dyn = mlp_w1dyn_floats(K_lds);
  float* lds_w1 = lds;
  float* lds_w2 = lds + n_dyn;
  float* lds_tail = lds_w2 + kW2Floats;

  // ---- stage the weights once per workgroup ------------------------------------------------
  {
    const float4* g1 = reinterpret_cast<const float4*>(a.w1dyn);
    float4* l1 = reinterpret_cast<float4*>(lds_w1);
    for (int i = threadIdx.x; i < n_dyn / 4; i += NT) l1[i] = g1[i];
    const float4* g2 = reinterpret_cast<const float4*>(a.w2p);
    float4* l2 = reinterpret_cast<float4*>(lds_w2);
    if (NWAVES == 4) {
      for (int i = threadIdx.x; i < kW2Floats / 4; i += NT) l2[i] = g2[i];
    } else {
      // pass-major layout for the four single-chain layer-2 passes: [pass][step/4][lane][step%4], so that a
      // lane reads four consecutive K steps of its pass with one conflict-free ds_read_b128 (reading one
      // component of the packed [step][lane][4 blocks] rows put 64 lanes on 16 LDS banks)
      for (int i = threadIdx.x; i < kW2Floats / 4; i += NT) {
        const float4 v = g2[i];
        const int t = i >> 6, ln = i & 63;  // K step, lane slot
        float* d = lds_w2 + (t >> 2) * 256 + ln * 4 + (t & 3);
        d[0 * 16 * 256] = v.x;
        d[1 * 16 * 256] = v.y;
        d[2 * 16 * 256] = v.z;
        d[3 * 16 * 256] = v.w;
      }
    }
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
  // wave-uniform floats computed on the VALU (division): move them to SGPRs instead of holding a VGPR each
  const float inv_w = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(1.0f / (float)w)));
  const float inv_h = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(1.0f / (float)h)));
  const int lane_off = (half * 32 + pl) * 4;  // float offset of this lane inside a step block
  float* lds_stage = lds_tail + kTailFloats + kHintFloats + wave * kStageFloats;  // (wave-uniform: stays in SGPRs)
  const long waves_total = (long)gridDim.x * NWAVES;
  const float b3 = lds_tail[256];
  const bool stage_ok = (D % 8 == 0);  // float4 alignment of the staged NHWC stores
  float pinf = __builtin_inff();  // opaque +inf for lrelu01()
  asm volatile("" : "+s"(pinf));

  // Balanced static partition: the (batch, tile, plane) units are flattened (plane fastest) and every
  // resident wave takes one contiguous span of floor/ceil(total / waves) units, i.e. one or two
  // partial tiles.  (With a fixed planes-per-task granularity the B=1 case left up to 25 % of the
  // waves one task short of the others.)
  // XCD-aware order (blocks are dealt round-robin to the 8 XCDs, each with a private 4 MiB L2): give
  // every XCD one contiguous eighth of the span space so that the source-feature footprint of an XCD
  // fits its L2 instead of every XCD streaming all 7 views.  The span space runs over the tiles in
  // a.tile_order (column strips walked boustrophedon, see mlp_tile_order below), so an eighth is a
  // compact block of the image, not a band of rows.
  const int nblk = gridDim.x;
  const int lbid = (nblk % 8 == 0) ? (int)(blockIdx.x % 8) * (nblk / 8) + (int)(blockIdx.x / 8) : (int)blockIdx.x;
  const long wid = (long)lbid * NWAVES + wave;
  // The span [u, u_end) is decomposed into (batch, tile, first plane) ONCE; the task loop then steps through it with
  // integer compares only.  (Divisions inside the loop made the compiler hoist their reciprocal constants -- wave-uniform
  // VALU results, i.e. VGPRs -- out of the loop and carry them, spilled, through every plane loop.)  Integer division runs
  // on the VALU, which also makes its results "divergent" to the compiler; readfirstlane restores wave-uniformity so that
  // the per-view parameter reads below are scalar loads (SGPR operands, scalar cache).
  int remaining, d0, otile, b;  // otile: position in the tile order
  {
    // Weighted split inside a workgroup (round 4).  With two waves per SIMD the older wave of a pair (waves 0-3) wins the issue
    // arbitration and runs ~32 us per plane against ~43 for its partner (scripts/mlp_wave_times.py): with equal spans it
    // finished ~130 us early and the partner ran the rest alone, at a lower pipe rate.  The workgroup's contiguous share of the
    // span space is therefore cut in a.old_share : (1 - a.old_share) between the older and the younger half of its waves, so
    // that partners finish together.  NWAVES = 4 (one wave per SIMD) keeps the even split.
    long u, u_end;
    if (a.span_bounds) {  // cost-aware plan (mlp_plan_* below): spans of equal estimated work
      const int* sb = a.span_bounds + (long)lbid * NWAVES + wave;
      // (clamped: a scratch that dt_cv_mlp_plan_f32 never filled yields a wrong volume, never an access outside it)
      u = min(max((long)sb[0], 0L), a.total_units);
      u_end = min(max((long)sb[1], u), a.total_units);
    } else if (NWAVES == 8 && a.old_share_q16 != 32768) {
      const long g0 = (long)lbid * a.total_units / nblk, g1 = ((long)lbid + 1) * a.total_units / nblk;  // this workgroup's units
      const long cut = g0 + (g1 - g0) * a.old_share_q16 / 65536;                                        // older half | younger half
      const long lo = (wave < 4) ? g0 : cut, hi = (wave < 4) ? cut : g1;
      const int wq = wave & 3;
      u = lo + (hi - lo) * wq / 4;
      u_end = lo + (hi - lo) * (wq + 1) / 4;
    } else {
      u = wid * a.total_units / waves_total;
      u_end = (wid + 1) * a.total_units / waves_total;
    }
    const long tile_global = u / D;
    d0 = __builtin_amdgcn_readfirstlane((int)(u - tile_global * D));
    otile = __builtin_amdgcn_readfirstlane((int)(tile_global % a.num_tiles));
    b = __builtin_amdgcn_readfirstlane((int)(tile_global / a.num_tiles));
    remaining = __builtin_amdgcn_readfirstlane((int)(u_end - u));
  }
#ifdef DT_MLP_TIMING
  const unsigned long long t_begin = __builtin_amdgcn_s_memrealtime();
  unsigned long long t_first = 0;
  int planes_done = 0;
#endif
  for (; remaining > 0;) {
    const int d1 = min(D, d0 + remaining);
    const int tile = a.tile_order ? ((const int __attribute__((address_space(4)))*)(uintptr_t)a.tile_order)[otile] : otile;  // (wave-uniform: one scalar load per task)
    const cfloat_ptr p = as_const(a.params + (size_t)b * cv_params_floats(D, K));
    const float* src_b = a.src + (size_t)b * K * hw * kF;

    // Values that only the per-task prologue needs (the ten 64-bit addresses of this lane's w1pix fragments, the hint
    // index scale, ...) are loop invariant, so the compiler hoisted them out of the TASK loop and kept them alive -- i.e.
    // spilled: 23 VGPRs, 96 B of scratch per lane -- across the plane loop (round 2: 12.6 MB of scratch writes per launch).
    // Deriving them from a per-task opaque copy of the lane id keeps them local to the prologue.
    int lane_t = lane;
    asm volatile("" : "+v"(lane_t));
    const int lane_off_t = ((lane_t >> 5) * 32 + (lane_t & 31)) * 4;
    const unsigned pixi = (unsigned)tile * 32u + (unsigned)(lane_t & 31);
    const bool live = pixi < (unsigned)hw;
    const unsigned pc = live ? pixi : (unsigned)hw - 1u;
    // (32-bit division by an opaque copy of w: its reciprocal is recomputed per task instead of living in a VGPR)
    unsigned w_t = (unsigned)w;
    asm volatile("" : "+s"(w_t));
    const int y = (int)(pc / w_t), x = (int)(pc - (unsigned)y * w_t);

    // ---- per-pixel, plane-independent part ------------------------------------------------
    float cur8[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) cur8[j] = a.cur[((size_t)b * kF + (lane_t >> 5) * 8 + j) * hw + pc];
    float rx, ry, rz;
    pixel_ray(p + kCvInvK, x, y, rx, ry, rz);
    // F.normalize: r / max(|r|, 1e-12)
    const float rinv = 1.0f / fmaxf(sqrtf(rx * rx + ry * ry + rz * rz), 1e-12f);

    f32x16 accp[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) accp[i][r] = 0.f;
    {
      const float4* wp = reinterpret_cast<const float4*>(a.w1pix + lane_off_t);
#pragma unroll
      for (int s = 0; s < 8; ++s) {
        const float4 a4 = wp[s * (kStepFloats / 4)];
        DT_MFMA4(accp, a4, cur8[s]);
      }
      {
        const float4 a4 = wp[8 * (kStepFloats / 4)];
        const float bv = (half ? ry : rx) * rinv;
        DT_MFMA4(accp, a4, bv);
      }
      {
        const float4 a4 = wp[9 * (kStepFloats / 4)];
        const float bv = half ? 1.0f : rz * rinv;
        DT_MFMA4(accp, a4, bv);
      }
      for (int k = 0; k < K; ++k) {
        const cfloat_ptr vp = p + cv_view_off(D, k);
        const float4 a4 = wp[(kPixFixed + 2 * k) * (kStepFloats / 4)];
        const float bv = half ? vp[16] : vp[15];
        DT_MFMA4(accp, a4, bv);
        const float4 c4 = wp[(kPixFixed + 2 * k + 1) * (kStepFloats / 4)];
        const float cv = half ? 0.0f : vp[17];
        DT_MFMA4(accp, c4, cv);
      }
    }

    // hint inputs of this pixel
    bool hmask = false;
    float hdepth = 0.f, hweight = 0.f;
    if (HINT) {
      // (opaque copies: the two float scale factors of nearest_src are wave-uniform VALU results that would otherwise
      // be hoisted out of the task loop and occupy / spill two VGPRs for the whole kernel)
      int hh = a.hint_h, hw2 = a.hint_w2;
      asm volatile("" : "+s"(hh), "+s"(hw2));
      const int sy = nearest_src(y, hh, h), sx = nearest_src(x, hw2, w);
      const size_t hi = ((size_t)b * a.hint_h + sy) * a.hint_w2 + sx;
      hmask = a.hint_m[hi] != 0.f;
      hdepth = a.hint_d[hi];
      hweight = hmask ? a.hint_w[hi] : 0.f;
    }

    // ---- planes -------------------------------------------------------------------------
    // Software pipeline with ONE ViewData: consume view k into f[] / scalars, re-issue the gathers
    // of the next (plane, view) into the same registers, then run view k's 48 MFMAs while they fly.
    ViewData v;
    {
      const float depth = p[kCvPlanes + d0];
      issue_view(v, p + cv_view_off(D, 0), src_b, depth * rx, depth * ry, depth * rz, rx, ry, rz, rinv, h, w, inv_w,
                 inv_h, half);
    }
    for (int d = d0; d < d1; ++d) {
      const float depth = p[kCvPlanes + d];
      f32x16 acc1[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) acc1[i] = accp[i];

      int wstep = 0;  // first layer-1 step of view k (wave-uniform; mlp_view_step_base(k) without the division)
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
        // No pixel of this 32-pixel tile sees the view at this plane (every bilinear weight of every lane is zero: the
        // footprint lies outside the source image): the warped features are exactly 0 and their 32 MFMAs add exactly 0 --
        // skip them (wave-uniform).  The metadata steps below still run (depth, angle and rays are defined regardless).
        // (only in the hint instantiation -- DoubleTake's kernel: in the no-hint one the extra branch cost 15 spilled VGPRs)
        const bool tile_sees_view = (DT_MLP_SKIP_EMPTY && HINT) ? __builtin_amdgcn_ballot_w64((v.w00 != 0.f) | (v.w01 != 0.f) | (v.w10 != 0.f) |
                                                                                   (v.w11 != 0.f)) != 0ull
                                                      : true;
        {
          // branch-free: the loop body stays ONE basic block, so the scheduler can place the next
          // view's projection / address / gather instructions between this view's MFMAs.  The
          // very last prefetch of a task re-reads its final view (harmless).
          int nk = k + 1, nd = d;
          if (nk == K) {
            nk = 0;
            nd = d + 1;
          }
          nd = min(nd, d1 - 1);
          const float ndepth = p[kCvPlanes + nd];
          issue_view(v, p + cv_view_off(D, nk), src_b + (size_t)nk * hw * kF, ndepth * rx, ndepth * ry, ndepth * rz, rx,
                     ry, rz, rinv, h, w, inv_w, inv_h, half);
        }
        float dotp = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) dotp += f[j] * cur8[j];
        const float dot = dotp + __shfl_xor(dotp, 32, 64);
        const float m = (vz > 0.f) ? 1.f : 0.f;

        // the K steps of view k against its layer-1 fragments (WL: LDS for resident views, global memory for streamed ones):
        // 8 feature steps, three metadata steps (z'|dot, angle|ray.x, ray.y|ray.z) and -- except for the second view of a
        // metadata pair (2, 4, 6, ...) -- a fourth one carrying the view's mask beside the plane depth (view 0), the NEXT
        // view's mask (first of a pair: v already holds view k + 1 of this plane) or nothing (unpaired last view).
        // (the pair flags come from an opaque copy of k: derived from k itself the compiler clones the whole view loop per
        //  parity and spills 69 registers)
        int kq = k;
        asm volatile("" : "+s"(kq));
        const bool pair_second = DT_MLP_PAIR_META && (kq >= 2) && !(kq & 1);
        const bool pair_first = DT_MLP_PAIR_META && (kq & 1) && (kq + 1 < K);
        const float m_next = (v.z > 0.f) ? 1.f : 0.f;
        const float spare = (kq == 0) ? depth : (pair_first ? m_next : 0.f);
#define DT_L1_VIEW(WL)                                                              \
  do {                                                                              \
    if (tile_sees_view) {                                                           \
      _Pragma("unroll") for (int s = 0; s < 8; ++s) {                               \
        const float4 a4 = (WL)[s * (kStepFloats / 4)];                              \
        DT_MFMA4(acc1, a4, f[s]);                                                   \
      }                                                                             \
    }                                                                               \
    {                                                                               \
      const float4 a4 = (WL)[8 * (kStepFloats / 4)];                                \
      const float bv = half ? dot * m : vz;                                         \
      DT_MFMA4(acc1, a4, bv);                                                       \
    }                                                                               \
    {                                                                               \
      const float4 a4 = (WL)[9 * (kStepFloats / 4)];                                \
      const float bv = half ? vsx : vang;                                           \
      DT_MFMA4(acc1, a4, bv);                                                       \
    }                                                                               \
    {                                                                               \
      const float4 a4 = (WL)[10 * (kStepFloats / 4)];                               \
      const float bv = half ? vsz : vsy;                                            \
      DT_MFMA4(acc1, a4, bv);                                                       \
    }                                                                               \
    if (!pair_second) {                                                             \
      const float4 a4 = (WL)[11 * (kStepFloats / 4)];                               \
      const float bv = half ? spare : m;                                            \
      DT_MFMA4(acc1, a4, bv);                                                       \
    }                                                                               \
  } while (0)
        if (!STREAM || k < K_lds) {
          const float4* wl = reinterpret_cast<const float4*>(lds_w1 + (size_t)wstep * kStepFloats + lane_off);
          DT_L1_VIEW(wl);
        } else {
          const float4* wg = reinterpret_cast<const float4*>(a.w1dyn + (size_t)wstep * kStepFloats + lane_off);
          DT_L1_VIEW(wg);
        }
        wstep += pair_second ? kStepsPerView - 1 : kStepsPerView;
#undef DT_L1_VIEW
      }

      // ---- layer 1 activation (bias came in through the constant-1 input) ------------------
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc1[i][r] = lrelu01(acc1[i][r], pinf);

      // ---- layer 2: acc2 = b2 + W2 h1, layer 3: s = W3 . lrelu(acc2) ----------------------------
      float s = 0.f;
      if (NWAVES == 4) {
        f32x16 acc2[4];
        {
          const float* bl = lds_tail + half * 64;
#pragma unroll
          for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc2[i][r] = bl[i * 16 + r];
        }
        {
          const float4* wl = reinterpret_cast<const float4*>(lds_w2 + lane_off);
#pragma unroll
          for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const float4 a4 = wl[(i * 16 + r) * (kStepFloats / 4)];
              DT_MFMA4(acc2, a4, acc1[i][r]);
              if ((r & 3) == 3) __builtin_amdgcn_sched_barrier(0);
            }
        }
        {
          const float* wl = lds_tail + 128 + half * 64;
#pragma unroll
          for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              s += wl[i * 16 + r] * lrelu01(acc2[i][r], pinf);
            }
        }
      } else {
        // four passes of 32 output features, one accumulator chain each: 16 registers less than two chains of
        // 64 (13 instead of 28 spilled VGPRs at the 256-register cap) and no slower -- a single dependent chain
        // of 32x32x2 MFMAs already issues back to back (scripts/mfma_chain_bench.hip)
#pragma unroll
        for (int pass = 0; pass < 4; ++pass) {
          f32x16 acc2;
          const float* bl = lds_tail + half * 64 + pass * 16;
#pragma unroll
          for (int r = 0; r < 16; ++r) acc2[r] = bl[r];
          const float4* wl = reinterpret_cast<const float4*>(lds_w2 + pass * 16 * 256 + lane_off);
#pragma unroll
          for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int rq = 0; rq < 4; ++rq) {
              const float4 a4 = wl[(i * 4 + rq) * 64];
              acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.x, acc1[i][rq * 4 + 0], acc2, 0, 0, 0);
              acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.y, acc1[i][rq * 4 + 1], acc2, 0, 0, 0);
              acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.z, acc1[i][rq * 4 + 2], acc2, 0, 0, 0);
              acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.w, acc1[i][rq * 4 + 3], acc2, 0, 0, 0);
              if (rq & 1) __builtin_amdgcn_sched_barrier(0);
            }
          const float* w3 = lds_tail + 128 + half * 64 + pass * 16;
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            s += w3[r] * lrelu01(acc2[r], pinf);
          }
        }
      }
      s += __shfl_xor(s, 32, 64);
      s += b3;
      if (HINT) {
        const float hint = hmask ? fabsf(hdepth - depth) : -1.f;
        // keep the 217 weight reads inside the plane loop (hoisted, they cost > 100 spilled registers)
        asm volatile("" ::: "memory");
        if (!(DT_MABL & 2)) s = hint_mlp_eval_lds(lds_tail + kTailFloats, s, hint, hweight, half, pinf);
        else s += hint * hweight;
      }
      // Store addresses are rebuilt from an opaque copy of the lane id at every store: hoisted out of the plane loop they
      // were three 64-bit per-lane bases that lived (spilled) across the whole loop.  Uniform base + 32-bit lane offset.
      if (!a.out_nhwc) {
        int ls = lane;
        asm volatile("" : "+v"(ls));
        const unsigned px2 = (unsigned)tile * 32u + (unsigned)(ls & 31);
        float* plane_base = a.vol + ((size_t)b * D + d) * hw;
        if (px2 < (unsigned)hw && ls < 32) plane_base[px2] = s;
      } else if (!stage_ok) {
        int ls = lane;
        asm volatile("" : "+v"(ls));
        const unsigned px2 = (unsigned)tile * 32u + (unsigned)(ls & 31);
        float* img_base = a.vol + (size_t)b * hw * D + d;
        if (px2 < (unsigned)hw && ls < 32) img_base[(size_t)px2 * D] = s;
      } else {
        // NHWC volume: a lane-per-pixel store writes 4 bytes every D*4 bytes (11x write amplification
        // at the memory controller).  Park the scores of up to 8 consecutive planes in LDS and write
        // them as one float4 per lane, i.e. 32 contiguous bytes per pixel.
        int ls = lane;
        asm volatile("" : "+v"(ls));
        if (ls < 32) lds_stage[ls * 8 + (d & 7)] = s;
        if ((d & 7) == 7 || d == d1 - 1) {
          __builtin_amdgcn_wave_barrier();
          const int cbase = d & ~7;
          const int lo = max(d0, cbase) - cbase, hi = d - cbase;  // valid planes of this chunk: [lo, hi]
          const int spx = ls >> 1, q = (ls & 1) * 4;
          const float4 v4 = *reinterpret_cast<const float4*>(lds_stage + spx * 8 + q);
          const unsigned spix = (unsigned)tile * 32u + (unsigned)spx;
          if (spix < (unsigned)hw) {
            float* dst = a.vol + (size_t)b * hw * D + cbase + ((size_t)spix * D + q);
            if (lo <= q && hi >= q + 3) {
              *reinterpret_cast<float4*>(dst) = v4;
            } else {
              if (lo <= q + 0 && hi >= q + 0) dst[0] = v4.x;
              if (lo <= q + 1 && hi >= q + 1) dst[1] = v4.y;
              if (lo <= q + 2 && hi >= q + 2) dst[2] = v4.z;
              if (lo <= q + 3 && hi >= q + 3) dst[3] = v4.w;
            }
          }
          __builtin_amdgcn_wave_barrier();
        }
      }
#if defined(DT_MLP_TIMING) && DT_MLP_TIMING + 0 >= 2
      // (level 2 only: a realtime read per plane drains the wave's LDS / scalar queue and slows the launch by ~15 %, which also
      //  shifts the balance between the two waves of a SIMD; level 1 keeps the begin / end stamps alone)
      if (lane == 0 && wid < 2048 && planes_done < 24) g_mlp_prog[wid * 24 + planes_done] = __builtin_amdgcn_s_memrealtime();
      ++planes_done;
#endif
    }
#if defined(DT_MLP_TIMING) && DT_MLP_TIMING + 0 >= 2
    if (t_first == 0) t_first = __builtin_amdgcn_s_memrealtime();
#endif
    // next task of the span: the following planes of the same tile, else plane 0 of the next tile / batch element
    remaining -= d1 - d0;
    d0 = d1;
    if (d0 == D) {
      d0 = 0;
      if (++otile == a.num_tiles) {
        otile = 0;
        ++b;
      }
    }
  }
#ifdef DT_MLP_TIMING
  if (lane == 0 && wid < 4096) {
    g_mlp_times[wid * 4 + 0] = t_begin;
    g_mlp_times[wid * 4 + 1] = __builtin_amdgcn_s_memrealtime();
    g_mlp_times[wid * 4 + 2] = t_first;
    g_mlp_times[wid * 4 + 3] = ((unsigned long long)blockIdx.x << 8) | (unsigned)wave;
  }
#endif
}


// ---- cost-aware span plan (round 4) -------------------------------------------------------------------------------------
// The hint kernel skips the 32 feature MFMAs of every (32-pixel tile, plane, view) whose footprint misses the source image.  On
// the bench frame that is 8.7 % of all MFMAs -- but with spans of equal LENGTH the saving is spread unevenly: near planes and
// border tiles are cheap, and per-wave stamps (scripts/mlp_wave_times.py) showed waves finishing anywhere between 560 and 830 us
// with the launch waiting for the last.  Two small kernels in front of the volume kernel give every wave a span of equal
// estimated WORK instead:
//   mlp_plan_cost:   one thread per (batch, tile, plane) unit: projects the tile's first, middle and last pixel into every view
//                    (same arithmetic as the volume kernel) and prices the unit as kPlanFixed + 27 K + 24 x (views seen)
//                    + 200 on the first plane of a tile (layer-2 MFMAs + metadata steps + the unit's vector work in MFMA
//                    equivalents; the per-view and per-tile prices are fitted to per-unit stamps, see the kernel)
//   mlp_plan_bounds: one workgroup: prefix sum of the prices, then the unit at which every wave's share begins -- a workgroup gets
//                    1 / gridDim of the total, its older four waves old_share of that (see the kernel), in equal parts
// The plan only moves span boundaries: every (pixel, plane) value is computed by the same code whichever wave owns it.
constexpr int kPlanFixed = 290;
#ifndef DT_PLAN_VIEW
#define DT_PLAN_VIEW 24
#endif
#ifndef DT_PLAN_TILE
#define DT_PLAN_TILE 200
#endif
#ifndef DT_PLAN_PAIR_ROUNDING
#define DT_PLAN_PAIR_ROUNDING 1  // younger waves' boundaries compensate the rounding of their SIMD partners' (0 = independent targets)
#endif
#ifndef DT_PLAN_DEPTH
#define DT_PLAN_DEPTH 0  // (a plane-index term, fitted at -22 for the far end, made the launch slower: 0.726 -> 0.732 ms)
#endif
__host__ __device__ inline int mlp_plan_bound_ints(int cus) { return (cus * 8 + 1 + 1) / 2 * 2; }  // (+1 end marker, even count)
__host__ __device__ inline long mlp_plan_groups(long units) { return (units + 255) / 256; }         // workgroups of the pricing kernel

// prices of 256 consecutive units -> their inclusive prefix inside the group (pref) + the group total (gsum)
__global__ __launch_bounds__(256) void mlp_plan_cost_kernel(const float* __restrict__ params, const int* __restrict__ tile_order,
                                                           unsigned* __restrict__ pref, unsigned* __restrict__ gsum, int B, int K,
                                                           int h, int w, int D, int num_tiles, long total_units) {
  __shared__ unsigned wsum[4];
  const long u = (long)blockIdx.x * 256 + threadIdx.x;
  unsigned c = 0;
  if (u < total_units) {
    const long tg = u / D;
    const int d = (int)(u - tg * D), otile = (int)(tg % num_tiles), b = (int)(tg / num_tiles);
    const int tile = tile_order ? tile_order[otile] : otile;
    const float* p = params + (size_t)b * cv_params_floats(D, K);
    const int hw = h * w;
    const float depth = p[kCvPlanes + d];
    const float inv_w = 1.0f / (float)w, inv_h = 1.0f / (float)h;
    int seen = 0;  // bit k: view k touches the image from at least one of the three sample pixels
#pragma unroll
    for (int sp = 0; sp < 3; ++sp) {
      const int pix = min(tile * 32 + sp * 16 - (sp == 2 ? 1 : 0), hw - 1);  // pixels 0, 16, 31 of the tile
      const int y = pix / w, x = pix - y * w;
      float rx, ry, rz;
      pixel_ray(p + kCvInvK, x, y, rx, ry, rz);
      for (int k = 0; k < K; ++k) {
        const ViewProj q = project_view(p + cv_view_off(D, k), depth * rx, depth * ry, depth * rz);
        const float gx = 2.0f * q.u * inv_w - 1.0f, gy = 2.0f * q.v * inv_h - 1.0f;
        const float ix = ((gx + 1.0f) * (float)w - 1.0f) * 0.5f, iy = ((gy + 1.0f) * (float)h - 1.0f) * 0.5f;
        if ((ix > -1.0f) & (ix < (float)w) & (iy > -1.0f) & (iy < (float)h)) seen |= 1 << k;
      }
    }
    // prices fitted to per-unit stamps of the kernel (scripts/mlp_cost_fit_probe.py, four frames, both waves of a SIMD pair):
    // a visible view costs 5 % of an all-empty unit (not the 7 % its MFMA count suggests) and the first plane of a tile 0.4 units
    // more (the plane-invariant contraction)
    c = (unsigned)(kPlanFixed + 27 * K + DT_PLAN_VIEW * __builtin_popcount(seen) + (d == 0 ? DT_PLAN_TILE : 0) - (DT_PLAN_DEPTH * d) / D);
  }
  unsigned incl = c;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const unsigned up = __shfl_up(incl, off, 64);
    if (lane >= off) incl += up;
  }
  if (lane == 63) wsum[wave] = incl;
  __syncthreads();
  unsigned base = 0;
  for (int q = 0; q < wave; ++q) base += wsum[q];
  if (u < total_units) pref[u] = base + incl;
  if (threadIdx.x == 255) gsum[blockIdx.x] = base + incl;
}

// one thread per wave slot j (+ the end marker): bounds[j] = first unit of slot j's span = the number of units whose cumulative
// price stays below the slot's target
__global__ __launch_bounds__(256) void mlp_plan_bounds_kernel(const unsigned* __restrict__ pref, const unsigned* __restrict__ gsum,
                                                             long total_units, int ngroups, int nblk, int nwaves,
                                                             int old_share_q16, int* __restrict__ bounds) {
  extern __shared__ long gpre[];  // [ngroups + 1] exclusive prefix of the group totals
  __shared__ long wtot[4];
  // exclusive scan of gsum by the whole workgroup (every workgroup of this kernel repeats it: ngroups is a few hundred)
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int per = (ngroups + 255) / 256;
  const int g0 = min(t * per, ngroups), g1 = min(g0 + per, ngroups);
  long s = 0;
  for (int g = g0; g < g1; ++g) s += gsum[g];
  long incl = s;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const long up = __shfl_up(incl, off, 64);
    if (lane >= off) incl += up;
  }
  if (lane == 63) wtot[wave] = incl;
  __syncthreads();
  long base = 0, total = 0;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    if (q < wave) base += wtot[q];
    total += wtot[q];
  }
  long run = base + incl - s;
  for (int g = g0; g < g1; ++g) {
    gpre[g] = run;
    run += gsum[g];
  }
  if (t == 0) gpre[ngroups] = total;
  __syncthreads();
  const long nslots = (long)nblk * nwaves;
  const long j = (long)blockIdx.x * 256 + t;
  __shared__ int sout[256];
  const bool have_slot = j <= nslots;
  // number of units whose cumulative price stays below `target` = the unit boundary behind the unit that reaches it
  auto bound_at = [&](long target) -> int {
    if (target <= 0) return 0;
    // group whose cumulative range (gpre[g], gpre[g + 1]] holds the target, then the unit inside it
    int lo = 0, hi = ngroups - 1;
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (gpre[mid + 1] >= target) hi = mid; else lo = mid + 1;
    }
    const long in_group = target - gpre[lo];  // 1 .. group total
    const long ubase = (long)lo * 256;
    int a = 0, c = (int)min(255L, total_units - 1 - ubase);
    while (a < c) {
      const int mid = (a + c) >> 1;
      if ((long)pref[ubase + mid] >= in_group) c = mid; else a = mid + 1;
    }
    return (int)(ubase + a + 1);
  };
  auto price_upto = [&](int i) -> long { return i <= 0 ? 0L : gpre[(i - 1) >> 8] + (long)pref[i - 1]; };  // cumulative price of units [0, i)
  // cumulative price at which slot j begins: workgroup blk gets 1 / nblk of the total (remainder spread over the first ones),
  // inside it the older four waves old_share of that in equal parts, the younger four the rest
  // Two phases, so that no thread runs more than two binary searches one after the other (each is ~16 dependent loads; the
  // pair-compensated slots used to run five: the kernel sits on the single-stream critical path in front of the volume kernel).
  // Phase 1: every slot's own primary boundary -- for the younger waves 5..7 of a workgroup, whose final boundary depends on their
  // partners', the END of the workgroup's share (B8) instead; phase 2 reads the partners' phase-1 results from LDS.
  int out = 0;
  long b0 = 0, b1 = 0, cut = 0;
  int wv = 0;
  bool compensated = false;
  if (!have_slot) {
    // (threads beyond the end marker only take part in the barriers below)
  } else if (j == nslots) {
    out = bound_at(total);
  } else {
    const long blk = j / nwaves;
    wv = (int)(j - blk * nwaves);
    const long share = total / nblk, rem = total - share * nblk;
    b0 = blk * share + min(blk, rem);
    b1 = b0 + share + (blk < rem ? 1 : 0);
    if (nwaves == 8) {
      cut = b0 + (((b1 - b0) * old_share_q16) >> 16);
      compensated = DT_PLAN_PAIR_ROUNDING && wv > 4;
      if (!compensated) out = bound_at((wv < 4) ? b0 + (((cut - b0) * wv) >> 2) : cut + (((b1 - cut) * (wv - 4)) >> 2));
      else out = bound_at(b1);  // B8, for phase 2
    } else {
      out = bound_at(b0 + (b1 - b0) * wv / nwaves);
    }
  }
  sout[t] = out;
  __syncthreads();
  if (compensated) {
    // Units are indivisible (a wave has ~20), so every boundary above is off its target by up to a unit and the summed price of
    // a SIMD pair (waves w and w + 4) scattered by 1.6 % rms -- +5 % for the unluckiest of 1024 pairs, which the launch waits
    // for.  The younger waves' boundaries therefore aim at what their OLDER partners really got: pair w should end up with a
    // quarter of what the workgroup really holds, and the boundary is the unit edge NEAREST to that (round 4, last pass).
    // (slots 0, 4 and wv - 4 of this workgroup are threads t - wv, t - wv + 4 and t - 4: same workgroup of this kernel, 256 % 8 == 0)
    const int B0 = sout[t - wv], B4 = sout[t - wv + 4], B8 = out;
    const int Bw = sout[t - 4];  // first unit of older wave wv - 4
    const long W = price_upto(B8) - price_upto(B0), older_so_far = price_upto(Bw) - price_upto(B0);
    const long target4 = 4 * price_upto(B4) + (long)(wv - 4) * W - 4 * older_so_far;  // 4 x the cumulative price to reach
    int hi_b = bound_at((target4 + 3) >> 2);
    hi_b = min(max(hi_b, B4), B8);
    const int lo_b = max(hi_b - 1, B4);
    out = (4 * price_upto(hi_b) - target4 <= target4 - 4 * price_upto(lo_b)) ? hi_b : lo_b;
  }
  __syncthreads();  // (phase-1 values have been read)
  // The pair-compensated boundaries of waves 5..7 are rounded on their own and can fall before their predecessor's: the volume
  // kernel's clamp keeps coverage complete either way, but two waves would then compute (and write, identically) the same
  // units.  A running maximum over the slots of a workgroup (256 % nwaves == 0: they sit in one workgroup of this kernel)
  // makes the spans disjoint.
  sout[t] = out;
  __syncthreads();
  if (!have_slot) return;
  if (j < nslots) {
    const int wv = (int)(j % nwaves);
    for (int q = 1; q <= wv; ++q) out = max(out, sout[t - q]);
  }
  bounds[j] = out;
}

// Tile processing order.  A wave's span and an XCD's eighth of the span space are contiguous in this order, and every XCD
// fetches the source texels its eighth's epipolar segments touch into its own L2.  Row-major order makes an eighth a band of
// h/8 rows over the full width, whose footprint under a horizontal baseline is fine but under any vertical parallax covers
// most of the source map in all eight L2s.  Column strips of one tile width, walked down / up alternately, make an eighth a
// compact block (about 40 x 60 pixels at 160 x 120): the summed footprint of the eight XCDs on the bench geometry drops from
// 25.5 MB to 16.9 MB (7.9 MB is one copy; scripts/volume_footprint.py).  Results do not depend on the order.
// The table lives in device memory per (device, h, w), built on first use with a blocking copy (visible to every stream
// afterwards); a first use that happens during a stream capture keeps the row-major order instead of allocating.
struct TileOrderKey {
  int dev, h, w;
  bool operator<(const TileOrderKey& o) const { return dev != o.dev ? dev < o.dev : (h != o.h ? h < o.h : w < o.w); }
};
static std::mutex g_order_mutex;
static std::map<TileOrderKey, int*> g_tile_orders;
static const bool g_tile_order_on = [] { const char* e = getenv("DT_MLP_TILE_ORDER"); return !(e && e[0] == '0'); }();

static const int* mlp_tile_order(int h, int w, hipStream_t st) {
  if (!g_tile_order_on) return nullptr;
  const int num_tiles = (int)(((long)h * w + 31) / 32);
  if (num_tiles < 64) return nullptr;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) {
    (void)hipGetLastError();
    return nullptr;
  }
  std::lock_guard<std::mutex> lock(g_order_mutex);
  const TileOrderKey key{dev, h, w};
  auto it = g_tile_orders.find(key);
  if (it != g_tile_orders.end()) return it->second;
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(st, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone) {
    (void)hipGetLastError();
    return nullptr;  // (not cached: the next eager call builds it)
  }
  // sort key of a tile: (column strip of its first pixel, row -- descending in odd strips)
  std::vector<std::pair<long, int>> keyed(num_tiles);
  for (int t = 0; t < num_tiles; ++t) {
    const long p = (long)t * 32;
    const int y = (int)(p / w), strip = (int)(p % w) / 32;
    keyed[t] = std::make_pair((long)strip * (h + 1) + ((strip & 1) ? h - y : y), t);
  }
  std::sort(keyed.begin(), keyed.end());
  std::vector<int> order(num_tiles);
  for (int i = 0; i < num_tiles; ++i) order[i] = keyed[i].second;
  // The allocation and the blocking copy are "unsafe" calls for a stream capture in global mode: one in progress on ANOTHER
  // stream of the process (this stream was checked above) would be invalidated.  For the duration of the two calls this
  // thread therefore runs in relaxed capture mode, which exempts exactly such calls.  The table is small
  // (one int per 32-pixel tile) and lives for the process: one per (device, h, w) ever used.
  int* dptr = nullptr;
  hipStreamCaptureMode mode = hipStreamCaptureModeRelaxed;
  const bool swapped = hipThreadExchangeStreamCaptureMode(&mode) == hipSuccess;
  if (hipMalloc(&dptr, num_tiles * sizeof(int)) != hipSuccess ||
      hipMemcpy(dptr, order.data(), num_tiles * sizeof(int), hipMemcpyHostToDevice) != hipSuccess) {
    (void)hipGetLastError();
    if (dptr) (void)hipFree(dptr);
    dptr = nullptr;
  }
  if (swapped) (void)hipThreadExchangeStreamCaptureMode(&mode);
  g_tile_orders[key] = dptr;  // (a failed allocation is remembered as "row-major")
  return dptr;
}

// waves per workgroup of the fused kernel (4 = one per SIMD, 8 = two per SIMD); DT_MLP_WAVES overrides
static int g_mlp_waves = [] { const char* e = getenv("DT_MLP_WAVES"); return (e && e[0] == '4') ? 4 : 8; }();
// compute-unit budget of the volume kernel (dt_cv_mlp_set_cu_budget; 0 = the whole device); DT_MLP_CUS presets it
static std::atomic<int> g_mlp_cu_budget{[] { const char* e = getenv("DT_MLP_CUS"); return e ? atoi(e) : 0; }()};
int mlp_cu_budget_value() { return g_mlp_cu_budget.load(std::memory_order_relaxed); }
static int num_cus() {
  const int dev = device_cu_count(), b = g_mlp_cu_budget.load(std::memory_order_relaxed);
  return (b > 0 && b < dev) ? std::max(8, b / 8 * 8) : dev;
}

}  // namespace dt

using namespace dt;

extern "C" {

#ifdef DT_MLP_TIMING
// experiment builds only (not in the header): copy the stamps of the last launch to the host
int dt_debug_mlp_times(unsigned long long* out, int n) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_mlp_times), sizeof(unsigned long long) * n) == hipSuccess ? 0 : 1;
}
int dt_debug_mlp_progress(unsigned long long* out, int n) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_mlp_prog), sizeof(unsigned long long) * n) == hipSuccess ? 0 : 1;
}
#endif

int dt_cv_mlp_pack_floats(int num_src, int* w1dyn, int* w1pix, int* w2p, int* tail) {
  DT_REQUIRE(num_src > 0 && num_src <= kMaxSrcStream, "dt_cv_mlp_pack_floats: num_src=%d not in 1..%d", num_src, kMaxSrcStream);
  if (w1dyn) *w1dyn = mlp_w1dyn_floats(num_src);
  if (w1pix) *w1pix = mlp_w1pix_floats(num_src);
  if (w2p) *w2p = kW2Floats;
  if (tail) *tail = kTailFloats;
  return 0;
}

// launch geometry of the volume kernel for one call (shared by the plan and the launch, which must agree on it)
struct MlpGrid {
  int num_tiles, cus, nw, blocks;
  long total_units;
  bool stream, plan;  // plan: the cost-aware span plan applies to the hint kernel of this geometry
};
static MlpGrid mlp_grid(int batch, int num_src, int h, int w, int num_planes) {
  MlpGrid g;
  g.num_tiles = (int)(((long)h * w + 31) / 32);
  g.cus = num_cus();
  g.total_units = (long)batch * g.num_tiles * num_planes;
  g.stream = num_src > kMaxSrcMfma;  // more views than LDS holds: the further views' fragments come from L2
  g.nw = g.stream ? 4 : g_mlp_waves;  // streamed views: one wave per SIMD (register room for the loads in flight)
  const long want = (g.total_units + g.nw - 1) / g.nw;  // at least one unit per wave
  g.blocks = (int)(want < g.cus ? want : g.cus);
  // cost-aware span plan: only where units differ in cost (the hint kernel's empty-view skip) and for the tuned instantiation
  static const bool plan_on = [] { const char* e = getenv("DT_MLP_PLAN"); return !(e && e[0] == '0'); }();
  g.plan = plan_on && !g.stream && g.nw == 8 && DT_MLP_SKIP_EMPTY && g.total_units < 2147483647L && mlp_plan_groups(g.total_units) < 8000;
  return g;
}

static int mlp_hint_launch(const float* cur, const float* src, const float* params, const float* w1dyn,
                           const float* w1pix, const float* w2p, const float* tail, const float* hint_mlp,
                           const float* depth_hint, const float* hint_weights, const float* hint_mask, int hint_h,
                           int hint_w, float* volume, int out_nhwc, int batch, int num_src, int h, int w,
                           int num_planes, void* plan_scratch, dt_stream_t s) {
  DT_REQUIRE(batch > 0 && h > 0 && w > 0 && num_planes > 0, "dt_cv_mlp_hint_f32: bad extents");
  DT_REQUIRE(num_src > 0 && num_src <= kMaxSrcStream, "dt_cv_mlp_hint_f32: num_src=%d not in 1..%d", num_src, kMaxSrcStream);
  DT_REQUIRE(cur && src && params && w1dyn && w1pix && w2p && tail && volume, "dt_cv_mlp_hint_f32: null pointer");
  DT_REQUIRE(hint_mlp == nullptr || (depth_hint && hint_weights && hint_mask && hint_h > 0 && hint_w > 0),
             "dt_cv_mlp_hint_f32: hint MLP given without hint maps");
  const MlpGrid g = mlp_grid(batch, num_src, h, w, num_planes);
  MlpArgs a;
  a.cur = cur; a.src = src; a.params = params; a.w1dyn = w1dyn; a.w1pix = w1pix; a.w2p = w2p; a.tail = tail;
  a.hint_mlp = hint_mlp; a.hint_d = depth_hint; a.hint_w = hint_weights; a.hint_m = hint_mask;
  a.vol = volume; a.hint_h = hint_h; a.hint_w2 = hint_w; a.out_nhwc = out_nhwc;
  a.B = batch; a.K = num_src; a.h = h; a.w = w; a.D = num_planes;
  a.num_tiles = g.num_tiles;
  a.tile_order = mlp_tile_order(h, w, to_stream(s));
  a.total_units = g.total_units;
  static const int old_share = [] { const char* e = getenv("DT_MLP_OLD_SHARE"); const double v = e ? atof(e) : 0.61; return (int)(65536.0 * (v > 0.2 && v < 0.9 ? v : 0.5)); }();
  a.old_share_q16 = old_share;
  // spans of equal estimated work, written by dt_cv_mlp_plan_f32 for this geometry (the same predicate decides there)
  a.span_bounds = (plan_scratch && hint_mlp && g.plan) ? reinterpret_cast<const int*>(plan_scratch) : nullptr;
  const int nw = g.nw, blocks = g.blocks;
  const bool stream = g.stream;
  const size_t lds_bytes = (size_t)(mlp_w1dyn_floats(stream ? kMaxSrcMfma : num_src) + kW2Floats + kTailFloats + kHintFloats +
                                    nw * kStageFloats) * sizeof(float);
#define DT_LAUNCH_MLP(HINT_, NW_, ST_)                                                                             \
  do {                                                                                                             \
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&cv_mlp_mfma_kernel<HINT_, NW_, ST_>),          \
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);                \
    if (e != hipSuccess) {                                                                                         \
      (void)hipGetLastError();                                                                                     \
      return fail("dt_cv_mlp_hint_f32: cannot reserve %zu B of LDS: %s", lds_bytes, hipGetErrorString(e));          \
    }                                                                                                              \
    DT_LAUNCH((cv_mlp_mfma_kernel<HINT_, NW_, ST_>), dim3(blocks), dim3(NW_ * 64), lds_bytes, to_stream(s), a);      \
  } while (0)
  if (stream) {  // (the one-wave-per-SIMD variant only: spill free with the streamed fragments in flight)
    if (hint_mlp) DT_LAUNCH_MLP(true, 4, true); else DT_LAUNCH_MLP(false, 4, true);
  } else if (hint_mlp) {
    if (nw == 8) DT_LAUNCH_MLP(true, 8, false); else DT_LAUNCH_MLP(true, 4, false);
  } else {
    if (nw == 8) DT_LAUNCH_MLP(false, 8, false); else DT_LAUNCH_MLP(false, 4, false);
  }
#undef DT_LAUNCH_MLP
  return check_launch("dt_cv_mlp_hint_f32");
}

int dt_cv_mlp_hint_f32(const float* cur, const float* src, const float* params, const float* w1dyn,
                       const float* w1pix, const float* w2p, const float* tail, const float* hint_mlp,
                       const float* depth_hint, const float* hint_weights, const float* hint_mask, int hint_h,
                       int hint_w, float* volume, int out_nhwc, int batch, int num_src, int h, int w,
                       int num_planes, dt_stream_t s) {
  return mlp_hint_launch(cur, src, params, w1dyn, w1pix, w2p, tail, hint_mlp, depth_hint, hint_weights, hint_mask, hint_h,
                         hint_w, volume, out_nhwc, batch, num_src, h, w, num_planes, nullptr, s);
}

int dt_cv_mlp_set_cu_budget(int cus) {
  g_mlp_cu_budget.store(cus > 0 ? cus : 0, std::memory_order_relaxed);
  return num_cus();
}

int64_t dt_cv_mlp_plan_bytes(int batch, int h, int w, int num_planes) {
  if (batch <= 0 || h <= 0 || w <= 0 || num_planes <= 0) return 0;
  const int64_t units = (int64_t)batch * (((int64_t)h * w + 31) / 32) * num_planes;
  const int64_t ngroups = mlp_plan_groups(units);
  // [span bounds int32][group totals uint32][in-group prefixes uint32 per unit]
  return ((int64_t)mlp_plan_bound_ints(device_cu_count()) + (ngroups + 1) / 2 * 2 + units) * 4;
}

int dt_cv_mlp_plan_f32(const float* params, int batch, int num_src, int h, int w, int num_planes, void* plan_scratch,
                       int64_t plan_scratch_bytes, dt_stream_t s) {
  DT_REQUIRE(params && plan_scratch, "dt_cv_mlp_plan_f32: null pointer");
  DT_REQUIRE(batch > 0 && h > 0 && w > 0 && num_planes > 0, "dt_cv_mlp_plan_f32: bad extents");
  DT_REQUIRE(num_src > 0 && num_src <= kMaxSrcStream, "dt_cv_mlp_plan_f32: num_src=%d not in 1..%d", num_src, kMaxSrcStream);
  DT_REQUIRE(plan_scratch_bytes >= dt_cv_mlp_plan_bytes(batch, h, w, num_planes),
             "dt_cv_mlp_plan_f32: plan scratch of %ld bytes, %ld needed for batch=%d h=%d w=%d planes=%d on this device",
             (long)plan_scratch_bytes, (long)dt_cv_mlp_plan_bytes(batch, h, w, num_planes), batch, h, w, num_planes);
  const MlpGrid g = mlp_grid(batch, num_src, h, w, num_planes);
  if (!g.plan) return 0;  // (streamed views, the one-wave-per-SIMD build, DT_MLP_PLAN=0: the planned call uses equal-length spans)
  const int* tile_order = mlp_tile_order(h, w, to_stream(s));
  static const int old_share = [] { const char* e = getenv("DT_MLP_OLD_SHARE"); const double v = e ? atof(e) : 0.61; return (int)(65536.0 * (v > 0.2 && v < 0.9 ? v : 0.5)); }();
  int* bounds = reinterpret_cast<int*>(plan_scratch);
  const int ngroups = (int)mlp_plan_groups(g.total_units);
  unsigned* gsum = reinterpret_cast<unsigned*>(bounds + mlp_plan_bound_ints(g.cus));
  unsigned* pref = gsum + (ngroups + 1) / 2 * 2;
  DT_LAUNCH(mlp_plan_cost_kernel, dim3((unsigned)ngroups), dim3(256), 0, to_stream(s), params, tile_order, pref, gsum, batch,
            num_src, h, w, num_planes, g.num_tiles, g.total_units);
  const int nslots = g.blocks * g.nw;
  DT_LAUNCH(mlp_plan_bounds_kernel, dim3((unsigned)((nslots + 1 + 255) / 256)), dim3(256), (size_t)(ngroups + 1) * sizeof(long),
            to_stream(s), pref, gsum, g.total_units, ngroups, g.blocks, g.nw, old_share, bounds);
  return check_launch("dt_cv_mlp_plan_f32");
}

int dt_cv_mlp_hint_planned_f32(const float* cur, const float* src, const float* params, const float* w1dyn,
                               const float* w1pix, const float* w2p, const float* tail, const float* hint_mlp,
                               const float* depth_hint, const float* hint_weights, const float* hint_mask, int hint_h,
                               int hint_w, float* volume, int out_nhwc, int batch, int num_src, int h, int w,
                               int num_planes, const void* plan, int64_t plan_bytes, dt_stream_t s) {
  DT_REQUIRE(plan != nullptr, "dt_cv_mlp_hint_planned_f32: null plan scratch (dt_cv_mlp_plan_f32 / dt_cv_mlp_plan_bytes)");
  DT_REQUIRE(plan_bytes >= dt_cv_mlp_plan_bytes(batch, h, w, num_planes),
             "dt_cv_mlp_hint_planned_f32: plan scratch of %ld bytes, %ld needed for batch=%d h=%d w=%d planes=%d on this device",
             (long)plan_bytes, (long)dt_cv_mlp_plan_bytes(batch, h, w, num_planes), batch, h, w, num_planes);
  return mlp_hint_launch(cur, src, params, w1dyn, w1pix, w2p, tail, hint_mlp, depth_hint, hint_weights, hint_mask, hint_h,
                         hint_w, volume, out_nhwc, batch, num_src, h, w, num_planes, const_cast<void*>(plan), s);
}

}  // extern "C"
