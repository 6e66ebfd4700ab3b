// Implicit-GEMM convolution for the cost-volume encoder / depth decoders, gfx950, fp32 MFMA.
//
// Replaces the torch ops composed in (paths relative to /root/reference/src/doubletake/):
//   BasicBlock.forward            modules/layers.py:77-94   conv3x3+bias -> LeakyReLU(0.2) -> conv3x3+bias
//                                                           (+identity | 1x1 | 3x3-s2 conv) -> LeakyReLU(0.2)
//   CVEncoder.forward             modules/networks.py:110-117   (torch.cat with image-prior features)
//   ConvBlock / ConvUpsampleAndConcatBlock  modules/networks_fast.py:17-40  (conv+ELU, nearest x2, cat)
//   regression heads              modules/networks_fast.py:102-132, modules/networks.py:60-63
//   upsample                      utils/generic_utils.py:95-104 (bilinear x2, align_corners=False)
//
// GEMM view: out[co][pixel] = sum_{tap,ci} W[co][tap][ci] * in[pixel+tap][ci]
//   v_mfma_f32_32x32x2_f32 with i = output channel (A = packed weights, straight from L2),
//   j = output pixel (B = input patch, staged per wave in LDS), lane = (pixel l&31, half l>>5).
// A workgroup (4 waves) owns ONE 32-channel x (4x8)-pixel output block; the four waves split K
// (8-channel input groups, round-robin) and reduce through LDS, so even the 15x20 level of the
// encoder yields >= 4 waves per MFMA block and the grid stays fine-grained enough to balance
// 1024 SIMDs.  Bias, residual add, LeakyReLU/ELU, channel concat of up to three sources and
// nearest x2 upsampling of a source are fused (never materialised).
#include <array>
#include <atomic>
#include <map>
#include <mutex>
#include <utility>
#include <vector>

#include "common.hpp"
#include "head_device.hpp"

namespace dt {

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct ConvArgs {
  const float* src[3];
  unsigned src_bytes[3];  // extent of each source in bytes (raw-buffer range check of the Winograd patch loads)
  int c[3];
  int up[3];
  int nsrc;
  const float* wp;
  const float* bias;
  const float* res;
  float* out;
  int n, h_out, w_out, c_out, h_in, w_in, act;
  int pad_replicate;  // 1: out-of-image taps read the clamped (edge) pixel instead of zero
  int xcd_remap;      // 1: XCD-contiguous block order (see xcd_contiguous_block)
  int cb_major;       // 1: channel block is the slowest block coordinate (weight-dominated layers)
  int groups;  // total 8-channel input groups over all sources
  int tiles_x, tiles_y, co_blocks;
  // K-split kernels only:
  int tr;             // 1: the image is tiled in the transposed frame (h/w fields hold the swapped extents, the weights
                      //    are packed with swapped taps): tall 8x4 instead of 4x8 output patches, same results
  int kparts;         // P > 1: P workgroups share one output block, each with 1/P of the K groups; the last to arrive sums
                      //    the partial blocks in part order (deterministic) and runs the epilogue
  int kplain;         // the first kplain blocks keep ONE workgroup each (launched first); only the remaining blocks are split:
                      //    with a few blocks more than CUs, the leftovers become many small workgroups that fill the second round
  float* part_buf;    // [blocks - kplain][P][4 waves][64 lanes] float4 partial blocks
  unsigned* part_cnt; // [blocks - kplain] arrival counters, zero between launches
#ifdef DT_CONV_TIMING
  unsigned long long* timing;
#endif
};
// (device pointers of the argument block, by position: what a launch program may patch -- common.hpp arg_pointers)
DT_ARG_POINTERS(ConvArgs, offsetof(ConvArgs, src) + 0 * sizeof(const float*), offsetof(ConvArgs, src) + 1 * sizeof(const float*),
                offsetof(ConvArgs, src) + 2 * sizeof(const float*), offsetof(ConvArgs, wp), offsetof(ConvArgs, bias),
                offsetof(ConvArgs, res), offsetof(ConvArgs, out), offsetof(ConvArgs, part_buf), offsetof(ConvArgs, part_cnt));

typedef float f32x4 __attribute__((ext_vector_type(4)));
// Hand-off of a partial output block between workgroups of ONE launch.  The per-CU L1 is never refreshed by other CUs'
// stores and the per-XCD L2s are not coherent with each other, so the payload travels write-through / cache-bypassing
// (sc0 sc1 on both sides) and is ordered against the arrival counter by draining vmcnt, not by cache-wide fences
// (an agent-scope release writes back the whole XCD L2: microseconds per workgroup).
__device__ __forceinline__ void store_f4_coherent(float4* p, float4 v) {
  const f32x4 x = {v.x, v.y, v.z, v.w};
  asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(x) : "memory");
}
__device__ __forceinline__ float4 load_f4_coherent(const float4* p) {
  f32x4 x;
  asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(x) : "v"(p) : "memory");
  return make_float4(x[0], x[1], x[2], x[3]);
}

__device__ __forceinline__ float apply_act(float v, int act) {
  if (act == DT_ACT_LRELU02) return v >= 0.f ? v : 0.2f * v;
  if (act == DT_ACT_ELU) return v > 0.f ? v : __expf(v) - 1.0f;  // ATen's elu: exp(x) - 1
  if (act == DT_ACT_RELU) return fmaxf(v, 0.f);
  return v;
}

constexpr int kPH = 4, kPW = 8;  // output patch of one MFMA pixel block

// Workgroups are dealt round-robin to the 8 XCDs (each with a private L2).  Map the hardware block index to
// a logical one so that every XCD works on one contiguous eighth of the block space: the co-blocks of a pixel
// tile and its neighbouring tiles (shared input halo) then hit the same L2 instead of eight different ones.
__device__ __forceinline__ long xcd_contiguous_block(int xcd_remap, unsigned b, unsigned nb) {
  if (!xcd_remap || (nb & 7u) != 0u) return (long)b;
  return (long)(b & 7u) * (nb >> 3) + (b >> 3);
}

// element offset of input pixel (iy, ix) of image n inside one NHWC source (nearest-upsampled when
// `up`), plus this lane's 4-channel half of the 8-channel group; -1 = zero padding
__device__ __forceinline__ int pixel_offset(bool inside, int n, int iy, int ix, int h_in, int w_in, int up, int cs,
                                            int lane, int tr = 0) {
  const int hs = up ? (h_in >> 1) : h_in, ws = up ? (w_in >> 1) : w_in;
  const int sy = up ? (iy >> 1) : iy, sx = up ? (ix >> 1) : ix;
  // tr: (iy, ix) are coordinates of the transposed frame, i.e. the real pixel is (row ix, column iy) of a ws x hs image
  const int pix = tr ? (n * ws + sx) * hs + sy : (n * hs + sy) * ws + sx;
  return inside ? pix * cs + (lane & 1) * 4 : -1;
}

// SPLIT = number of waves that share one 32-channel x 32-pixel output block by splitting K:
//   1  : every wave owns a whole block (4 blocks per 256-thread workgroup), no reduction;
//        used when the layer has enough blocks to fill the chip on its own
//   4  : one block per 256-thread workgroup, 4-way K split, LDS reduction
//   8  : one block per 512-thread workgroup, 8-way K split (the 30x40 level)
//   16 : one block per 1024-thread workgroup, 16-way K split (only the low-resolution 1x1 downsample convs:
//        for the 3x3 layers of the 15x20 level it measured 10 % slower than 8)
#ifndef DT_CONV_S2_SWZ
#define DT_CONV_S2_SWZ 1  // 0 = the pixel-major patch layout of the stride-2 bodies (A/B of the bank-conflict fix)
#endif
template <int KS, int ST, int SPLIT>
struct ConvMfmaCfg {
  static constexpr int NW = (SPLIT >= 8) ? SPLIT : 4;
  static constexpr int IH = (kPH - 1) * ST + KS, IW = (kPW - 1) * ST + KS;
  static constexpr int NPIX = IH * IW;
  // (padded row pitch 12 and, for stride 2, one plane per column parity: see the body)
  static constexpr int PATCH_FLOATS = (ST == 1) ? 2 * IH * 12 * 4 : (DT_CONV_S2_SWZ ? 2 * 2 * IH * 12 * 4 : NPIX * 8);
  static constexpr int TILE_FLOATS = (SPLIT == 1) ? PATCH_FLOATS : ((PATCH_FLOATS > 1024) ? PATCH_FLOATS : 1024);
  static constexpr int LDS_FLOATS = NW * TILE_FLOATS;
  static constexpr int THREADS = NW * 64;
};

// Workgroups per CU requested from the compiler (register budget = 512 / waves per SIMD).  1 = whatever the kernels need
// (128-160 VGPRs: one 512-thread workgroup per CU).  Experiment knob: -DDT_CONV_OCC=2 caps the 512-thread kernels at 128
// VGPRs so that two workgroups (e.g. of two HIP streams) share a CU -- measured 4 % SLOWER on bench.py in both the one- and
// the two-stream mode (623 vs 648 frames/s): the spills land in the K loop and co-resident workgroups move in lockstep.
#ifndef DT_CONV_OCC
#define DT_CONV_OCC 1
#endif
constexpr int conv_waves_per_eu(int threads) { return threads / 256 * DT_CONV_OCC; }

// -DDT_CONV_TIMING: wave 0 of every workgroup of the K-split kernels records s_memrealtime (100 MHz) at its phase
// boundaries into the buffer whose address the host reads from the environment (scripts/conv_phase_timing.py).
#ifndef DT_WINO_TIMING_WAVE
#define DT_WINO_TIMING_WAVE 0
#endif
#ifdef DT_CONV_TIMING
#define DT_STAMP(SLOT)                                                                              \
  do {                                                                                              \
    if (a.timing && threadIdx.x == 0) a.timing[(size_t)vblock * 24 + (SLOT)] = __builtin_amdgcn_s_memrealtime(); \
  } while (0)
#else
#define DT_STAMP(SLOT) do { } while (0)
#endif

// The body takes a VIRTUAL block index / grid size so that two convolutions can share one launch (conv_pair_kernel).
template <int KS, int ST, int SPLIT>
__device__ __forceinline__ void conv_mfma_body(const ConvArgs& a, float* __restrict__ lds, unsigned vblock, unsigned vgrid) {
  using Cfg = ConvMfmaCfg<KS, ST, SPLIT>;
  constexpr int NW = Cfg::NW;
  constexpr int IH = Cfg::IH, IW = Cfg::IW;
  constexpr int NPIX = Cfg::NPIX;
  constexpr int NLOAD = (NPIX + 31) / 32;  // float4 staging loads per lane and group
  constexpr int TILE_FLOATS = Cfg::TILE_FLOATS;
  constexpr int PAD = KS / 2;
  constexpr int TAPS = KS * KS;

  DT_STAMP(0);
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int half = lane >> 5, p = lane & 31;
  // Stride 1: the staged patch is laid out [channel half][row][column, pitch 12][4 floats] and the lanes that one
  // ds_read_b128 cycle services together ({0-3,12-15,20-27} / {4-11,16-19,28-31} of each half-wave) own pixel rows
  // {0,2} / {1,3} of the 4x8 block, so the 16 lanes of an access touch 16 different four-bank groups (12*row + col mod
  // 16).  The [pixel][8 channels] layout it replaces left SQ_LDS_BANK_CONFLICT at 63 % of this kernel's LDS cycles
  // (profiles/r2i_pmc_summary.json).
  // Stride 2: a lane reads column 2*px + kx, so the pixel-major layout put the 16 lanes of an access on 4 bank groups
  // (68*py + 4*px mod 16: SQ_LDS_BANK_CONFLICT = 50 % of the stride-2 kernels' LDS cycles, profiles/r3u_pmc_summary.json).
  // Layout [channel half][column parity][row][column / 2, pitch 12][4 floats] with the natural lane -> pixel map: the
  // bank group of a read is 24*py + px + const = 8*py + px (mod 16), and the lanes serviced together -- rows (0, px 0-3),
  // (1, 4-7), (2, 4-7), (3, 0-3) or the complement -- cover all 16 groups.  The staging side enumerates a patch row as
  // its even columns followed by its odd columns, so that the eight pixels x two halves of a store cycle are
  // consecutive within one parity plane.
  constexpr bool SWZ = (ST == 1);
  constexpr int PITCH = 12;
  const bool grp_a = (p < 4) || (p >= 12 && p < 16) || (p >= 20 && p < 28);
  const int gi = grp_a ? ((p < 4) ? p : ((p < 16) ? p - 8 : p - 12)) : ((p < 12) ? p - 4 : ((p < 20) ? p - 8 : p - 16));
  const int py = SWZ ? 2 * (gi >> 3) + (grp_a ? 0 : 1) : (p >> 3), px = SWZ ? (gi & 7) : (p & 7);
  constexpr bool SWZ2 = (ST == 2) && DT_CONV_S2_SWZ;
  auto lds_off = [&](int hf, int row, int col) {
    return SWZ ? ((hf * IH + row) * PITCH + col) * 4
               : (SWZ2 ? ((((hf * 2 + (col & 1)) * IH + row) * PITCH) + (col >> 1)) * 4 : (row * IW + col) * 8 + hf * 4);
  };
  // staged pixel idx -> (row, column) of the patch
  constexpr int NEVEN = (IW + 1) / 2;
  auto patch_row = [&](int idx) { return idx / IW; };
  auto patch_col = [&](int idx) {
    const int r = idx - (idx / IW) * IW;
    return SWZ2 ? ((r < NEVEN) ? 2 * r : 2 * (r - NEVEN) + 1) : r;
  };

  const long total_blocks = (long)a.n * a.tiles_y * a.tiles_x * a.co_blocks;
  // cross-workgroup K split: workgroups [0, kplain) own a whole block each, the others share the remaining blocks P ways;
  // the parts of a block are neighbours in the logical order (same XCD, shared input patch in L2)
  const bool split_wg = (SPLIT != 1) && a.kparts > 1 && vblock >= (unsigned)a.kplain;
  const int kparts = split_wg ? a.kparts : 1;
  long bid;
  if (SPLIT == 1) bid = (long)vblock * 4 + wave;
  else if (a.kparts <= 1) bid = xcd_contiguous_block(a.xcd_remap, vblock, vgrid);
  else if (!split_wg) bid = xcd_contiguous_block(a.xcd_remap, vblock, (unsigned)a.kplain);
  else bid = xcd_contiguous_block(a.xcd_remap, vblock - (unsigned)a.kplain, vgrid - (unsigned)a.kplain);
  const int part = split_wg ? (int)(bid % kparts) : 0;
  if (split_wg) bid = a.kplain + bid / kparts;
  const long blk = bid - a.kplain;  // index into the partial-block scratch
  const bool have_block = bid < total_blocks;
  if (!have_block) bid = total_blocks - 1;  // keep the wave alive (no barriers are skipped); it stores nothing
  int cb;
  if (SPLIT != 1 && a.cb_major) {
    // weight-dominated layers (few pixels, many channels): channel block slowest, so that with the
    // XCD-contiguous order each XCD's L2 holds the weights of only an eighth of the channel blocks
    const long px_tiles = (long)a.n * a.tiles_y * a.tiles_x;
    cb = (int)(bid / px_tiles);
    bid -= (long)cb * px_tiles;
  } else {
    cb = (int)(bid % a.co_blocks);
    bid /= a.co_blocks;
  }
  const int tx = (int)(bid % a.tiles_x);
  bid /= a.tiles_x;
  const int ty = (int)(bid % a.tiles_y);
  const int n = (int)(bid / a.tiles_y);

  const int oy = ty * kPH + py, ox = tx * kPW + px;
  const int iy0 = ty * kPH * ST - PAD, ix0 = tx * kPW * ST - PAD;
  const int tr = (SPLIT == 1) ? 0 : a.tr;
  float* tile = lds + wave * TILE_FLOATS;

  // one accumulator chain per wave is enough: dependent v_mfma_f32_32x32x2_f32 issue back to back
  // (scripts/mfma_chain_bench.hip: 92 % of peak from a single chain, one wave per SIMD)
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;

  const float4* wp4 = reinterpret_cast<const float4*>(a.wp);
  const int g_first = (SPLIT == 1) ? 0 : part * SPLIT + wave;
  const int g_step = (SPLIT == 1) ? 1 : SPLIT * kparts;

  // ---- per-lane, group-invariant addressing (hoisted out of the K loop) ----------------------------
  // element offset of each staged pixel inside each source (or -1: zero padding / outside the patch)
  int poff0[NLOAD], poff1[NLOAD], poff2[NLOAD];
#pragma unroll
  for (int it = 0; it < NLOAD; ++it) {
    const int idx = (lane >> 1) + it * 32;
    const int ly = patch_row(idx), lx = patch_col(idx);
    int iy = iy0 + ly, ix = ix0 + lx;
    bool inside = idx < NPIX && iy >= 0 && iy < a.h_in && ix >= 0 && ix < a.w_in;
    if (a.pad_replicate) {
      iy = min(max(iy, 0), a.h_in - 1);
      ix = min(max(ix, 0), a.w_in - 1);
      inside = idx < NPIX;
    }
    poff0[it] = pixel_offset(inside, n, iy, ix, a.h_in, a.w_in, a.up[0], a.c[0], lane, tr);
    poff1[it] = pixel_offset(inside && a.nsrc > 1, n, iy, ix, a.h_in, a.w_in, a.up[1], a.c[1], lane, tr);
    poff2[it] = pixel_offset(inside && a.nsrc > 2, n, iy, ix, a.h_in, a.w_in, a.up[2], a.c[2], lane, tr);
  }
  const int ng0 = a.c[0] >> 3, ng1 = a.c[1] >> 3;
  const float4* wbase = wp4 + ((size_t)cb * a.groups * TAPS * 2 + half) * 32 + p;

  float4 patch[NLOAD];

  // (macros, not lambdas: arrays captured or passed by reference end up in scratch memory)
  // issue the global loads of one 8-channel group's input patch
#define DT_PREFETCH_PATCH(G)                                                                         \
  do {                                                                                               \
    const int g_ = (G);                                                                              \
    const int sidx = (g_ < ng0) ? 0 : ((g_ < ng0 + ng1) ? 1 : 2);                                    \
    const int gl = (sidx == 0) ? g_ : ((sidx == 1) ? g_ - ng0 : g_ - ng0 - ng1);                     \
    const float* sp = ((sidx == 0) ? src0 : ((sidx == 1) ? src1 : src2)) + gl * 8;                   \
    _Pragma("unroll") for (int it = 0; it < NLOAD; ++it) {                                           \
      const int off = (sidx == 0) ? poff0[it] : ((sidx == 1) ? poff1[it] : poff2[it]);               \
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);                                                    \
      if (!(DT_ABL & 2) && off >= 0) v = *reinterpret_cast<const float4*>(sp + off);                 \
      patch[it] = v;                                                                                 \
    }                                                                                                \
  } while (0)
#ifndef DT_ABL
#define DT_ABL 0
#endif
#define DT_LOAD_WEIGHTS(WREG, G)                                                                     \
  do {                                                                                               \
    const float4* wg = wbase + (size_t)(G) * (TAPS * 64);                                            \
    _Pragma("unroll") for (int t = 0; t < TAPS; ++t) {                                               \
      if (DT_ABL & 1) WREG[t] = make_float4((float)(G), (float)t, 1.f, 2.f);                         \
      else WREG[t] = wg[(size_t)t * 64];                                                             \
    }                                                                                                \
  } while (0)
  // one K step: publish the prefetched patch, start the next group's loads, run this group's MFMAs
#define DT_K_STEP(WCUR, WNXT, G)                                                                     \
  do {                                                                                               \
    __builtin_amdgcn_wave_barrier();                                                                 \
    _Pragma("unroll") for (int it = 0; it < NLOAD; ++it) {                                           \
      const int idx = (lane >> 1) + it * 32;                                                         \
      if (idx < NPIX) *reinterpret_cast<float4*>(tile + lds_off(lane & 1, patch_row(idx), patch_col(idx))) = patch[it]; \
    }                                                                                                \
    if ((G) + g_step < a.groups) {                                                                   \
      DT_PREFETCH_PATCH((G) + g_step);                                                               \
      DT_LOAD_WEIGHTS(WNXT, (G) + g_step);                                                           \
    }                                                                                                \
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");                                           \
    __builtin_amdgcn_wave_barrier();                                                                 \
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");                                           \
    _Pragma("unroll") for (int t = 0; t < TAPS; ++t) {                                               \
      const int ky = t / KS, kx = t - ky * KS;                                                       \
      const float4 b4 = (DT_ABL & 4) ? make_float4(acc[0], acc[1], 1.f, (float)t) :                  \
          *reinterpret_cast<const float4*>(tile + lds_off(half, py * ST + ky, px * ST + kx));           \
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(WCUR[t].x, b4.x, acc, 0, 0, 0);                     \
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(WCUR[t].y, b4.y, acc, 0, 0, 0);                     \
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(WCUR[t].z, b4.z, acc, 0, 0, 0);                     \
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(WCUR[t].w, b4.w, acc, 0, 0, 0);                     \
    }                                                                                                \
  } while (0)

  const float* src0 = a.src[0];
  const float* src1 = a.src[1];
  const float* src2 = a.src[2];
  // K loop, unrolled by two so that the two weight register sets alternate without copies
  float4 wA[TAPS], wB[TAPS];
  DT_STAMP(1);
  if (g_first < a.groups) {
    DT_PREFETCH_PATCH(g_first);
    DT_LOAD_WEIGHTS(wA, g_first);
  }
  for (int g = g_first; g < a.groups; g += 2 * g_step) {
    DT_K_STEP(wA, wB, g);
#ifdef DT_CONV_TIMING
    if (g == g_first) {
      asm volatile("s_nop 0" ::"v"(acc[0]));  // the first step's MFMAs have produced their result
      DT_STAMP(2);
    }
#endif
    if (g + g_step < a.groups) DT_K_STEP(wB, wA, g + g_step);
  }
#ifdef DT_CONV_TIMING
  asm volatile("s_nop 0" ::"v"(acc[0]));
  DT_STAMP(3);
  if (a.timing && lane == 0 && wave < 8) {  // per wave: end of its K loop, and where it runs (HW_ID: SIMD, CU, ...)
    a.timing[(size_t)vblock * 24 + 8 + wave] = __builtin_amdgcn_s_memrealtime();
    a.timing[(size_t)vblock * 24 + 16 + wave] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));  // HW_REG_HW_ID, all 32 bits
  }
#endif
#undef DT_K_STEP
#undef DT_LOAD_WEIGHTS
#undef DT_PREFETCH_PATCH

  const bool in_image = have_block && oy < a.h_out && ox < a.w_out;
  const size_t pix_off = (tr ? ((size_t)n * a.w_out + ox) * a.h_out + oy : ((size_t)n * a.h_out + oy) * a.w_out + ox) * a.c_out;

  auto finish = [&](float4 o, int co) {
    const size_t off = pix_off + co;
    if (a.bias) {
      const float4 bv = *reinterpret_cast<const float4*>(a.bias + co);
      o.x += bv.x;
      o.y += bv.y;
      o.z += bv.z;
      o.w += bv.w;
    }
    if (a.res) {
      const float4 rv = *reinterpret_cast<const float4*>(a.res + off);
      o.x += rv.x;
      o.y += rv.y;
      o.z += rv.z;
      o.w += rv.w;
    }
    o.x = apply_act(o.x, a.act);
    o.y = apply_act(o.y, a.act);
    o.z = apply_act(o.z, a.act);
    o.w = apply_act(o.w, a.act);
    *reinterpret_cast<float4*>(a.out + off) = o;
  };

  if (SPLIT == 1) {
    // lane (p,h) holds channels (r&3) + 8*(r>>2) + 4h of its pixel: four float4 stores
    if (in_image) {
#pragma unroll
      for (int q = 0; q < 4; ++q)
        finish(make_float4(acc[q * 4 + 0], acc[q * 4 + 1], acc[q * 4 + 2], acc[q * 4 + 3]), cb * 32 + q * 8 + half * 4);
    }
  } else {
    // ---- cross-wave K reduction through LDS, epilogue by the first four waves -------------------
    // a wave parks its accumulators in ITS OWN patch region (its last patch reads precede these writes in program order,
    // and a wave's LDS operations complete in order), so the early finishers do this while the others still compute
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int r = 0; r < 16; ++r) tile[r * 64 + lane] = acc[r];
    __syncthreads();
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    if (wave < 4) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int r = wave * 4 + j;
        float sum = 0.f;
#pragma unroll
        for (int u = 0; u < NW; ++u) sum += lds[u * TILE_FLOATS + r * 64 + lane];
        v[j] = sum;
      }
      DT_STAMP(4);
      if (kparts == 1 && in_image) finish(make_float4(v[0], v[1], v[2], v[3]), cb * 32 + wave * 8 + half * 4);
    }
    if (kparts > 1 && have_block) {  // (workgroup-uniform)
      // ---- cross-workgroup K reduction: publish this part, count arrivals, the last one sums in part order ----
      float4* slot = reinterpret_cast<float4*>(a.part_buf) + ((size_t)blk * kparts * 4 + wave) * 64 + lane;
      if (wave < 4) store_f4_coherent(slot + (size_t)part * 256, make_float4(v[0], v[1], v[2], v[3]));
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();  // every partial store of this workgroup has been acknowledged; the reduction reads of lds are done
      unsigned* flag = reinterpret_cast<unsigned*>(lds);
      if (threadIdx.x == 0)
        *flag = __hip_atomic_fetch_add(a.part_cnt + blk, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __syncthreads();
      const unsigned arrived = *reinterpret_cast<volatile unsigned*>(flag);
      if (arrived == (unsigned)(kparts - 1)) {
        if (wave < 4) {
          float4 sum = make_float4(0.f, 0.f, 0.f, 0.f);
          for (int q = 0; q < kparts; ++q) {
            // own part from registers (bit-identical to what was stored): the sum does not depend on who arrives last
            const float4 pv = (q == part) ? make_float4(v[0], v[1], v[2], v[3]) : load_f4_coherent(slot + (size_t)q * 256);
            sum.x = (q == 0) ? pv.x : sum.x + pv.x;
            sum.y = (q == 0) ? pv.y : sum.y + pv.y;
            sum.z = (q == 0) ? pv.z : sum.z + pv.z;
            sum.w = (q == 0) ? pv.w : sum.w + pv.w;
          }
          if (in_image) finish(sum, cb * 32 + wave * 8 + half * 4);
        }
        if (threadIdx.x == 0) __hip_atomic_store(a.part_cnt + blk, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
#ifdef DT_CONV_TIMING
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the output stores have been acknowledged
    DT_STAMP(5);
#endif
  }
}

template <int KS, int ST, int SPLIT>
__global__ __launch_bounds__((ConvMfmaCfg<KS, ST, SPLIT>::THREADS), (conv_waves_per_eu(ConvMfmaCfg<KS, ST, SPLIT>::THREADS))) void conv_mfma_kernel(const ConvArgs a) {
  __shared__ __attribute__((aligned(16))) float lds[ConvMfmaCfg<KS, ST, SPLIT>::LDS_FLOATS];
  conv_mfma_body<KS, ST, SPLIT>(a, lds, blockIdx.x, gridDim.x);
}

// ---- many-block layers: four pixel tiles per workgroup share the weight fragments through LDS -------
// Ablation on the 240x320 128->64 layer (scripts/time_conv_layer.py, -DDT_ABL): of 139 us, 30 us were
// stalls on the per-wave weight loads (every wave re-read all 147 KB of its channel block's weights
// from L2) and 18 us on the patch loads.  Here the four waves of a workgroup own four pixel tiles of
// the SAME 32-channel block; each wave fetches a quarter of the group's weight fragments one K step
// ahead, publishes it to a double-buffered LDS slab, and all four read their A operands from there:
// weight traffic per MFMA drops 4x, one workgroup barrier per K step (2304 MFMA cycles).
template <int KS, int ST>
__global__ __launch_bounds__(256) void conv_mfma_wshare_kernel(const ConvArgs a) {
  constexpr int IH = (kPH - 1) * ST + KS, IW = (kPW - 1) * ST + KS;
  constexpr int NPIX = IH * IW;
  constexpr int NLOAD = (NPIX + 31) / 32;
  constexpr int PAD = KS / 2;
  constexpr int TAPS = KS * KS;
  constexpr int WSHARE = (TAPS + 3) / 4;  // weight fragments fetched per wave and K step
  __shared__ __attribute__((aligned(16))) float tiles[4 * NPIX * 8];
  __shared__ __attribute__((aligned(16))) float wlds[2 * TAPS * 256];

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int half = lane >> 5, p = lane & 31;
  const int py = p >> 3, px = p & 7;

  const long px_tiles = (long)a.n * a.tiles_y * a.tiles_x;
  const int cb = (int)(blockIdx.x % a.co_blocks);
  long pt = (long)(blockIdx.x / a.co_blocks) * 4 + wave;
  const bool have_tile = pt < px_tiles;
  if (!have_tile) pt = px_tiles - 1;
  const int tx = (int)(pt % a.tiles_x);
  const int ty = (int)((pt / a.tiles_x) % a.tiles_y);
  const int n = (int)(pt / ((long)a.tiles_x * a.tiles_y));

  const int oy = ty * kPH + py, ox = tx * kPW + px;
  const int iy0 = ty * kPH * ST - PAD, ix0 = tx * kPW * ST - PAD;
  float* tile = tiles + wave * NPIX * 8;

  int poff0[NLOAD], poff1[NLOAD], poff2[NLOAD];
#pragma unroll
  for (int it = 0; it < NLOAD; ++it) {
    const int idx = (lane >> 1) + it * 32;
    const int ly = idx / IW, lx = idx - ly * IW;
    int iy = iy0 + ly, ix = ix0 + lx;
    bool inside = idx < NPIX && iy >= 0 && iy < a.h_in && ix >= 0 && ix < a.w_in;
    if (a.pad_replicate) {
      iy = min(max(iy, 0), a.h_in - 1);
      ix = min(max(ix, 0), a.w_in - 1);
      inside = idx < NPIX;
    }
    poff0[it] = pixel_offset(inside, n, iy, ix, a.h_in, a.w_in, a.up[0], a.c[0], lane);
    poff1[it] = pixel_offset(inside && a.nsrc > 1, n, iy, ix, a.h_in, a.w_in, a.up[1], a.c[1], lane);
    poff2[it] = pixel_offset(inside && a.nsrc > 2, n, iy, ix, a.h_in, a.w_in, a.up[2], a.c[2], lane);
  }
  const int ng0 = a.c[0] >> 3, ng1 = a.c[1] >> 3;
  const float* src0 = a.src[0];
  const float* src1 = a.src[1];
  const float* src2 = a.src[2];
  // packed weights: [cb][g][tap][half][32][4] -> fragment (g, t) is 64 consecutive float4, one per lane
  const float4* wbase = reinterpret_cast<const float4*>(a.wp) + (size_t)cb * a.groups * TAPS * 64 + lane;

  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  float4 patch[NLOAD];
  float4 wq0 = make_float4(0.f, 0.f, 0.f, 0.f), wq1 = wq0, wq2 = wq0;  // named (an indexed array spills to scratch)

#define DT_WS_PREFETCH(G)                                                                            \
  do {                                                                                               \
    const int g_ = (G);                                                                              \
    const int sidx = (g_ < ng0) ? 0 : ((g_ < ng0 + ng1) ? 1 : 2);                                    \
    const int gl = (sidx == 0) ? g_ : ((sidx == 1) ? g_ - ng0 : g_ - ng0 - ng1);                     \
    const float* sp = ((sidx == 0) ? src0 : ((sidx == 1) ? src1 : src2)) + gl * 8;                   \
    _Pragma("unroll") for (int it = 0; it < NLOAD; ++it) {                                           \
      const int off = (sidx == 0) ? poff0[it] : ((sidx == 1) ? poff1[it] : poff2[it]);               \
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);                                                    \
      if (off >= 0) v = *reinterpret_cast<const float4*>(sp + off);                                  \
      patch[it] = v;                                                                                 \
    }                                                                                                \
    if (wave < TAPS) wq0 = wbase[((size_t)g_ * TAPS + wave) * 64];                                   \
    if (WSHARE > 1 && wave + 4 < TAPS) wq1 = wbase[((size_t)g_ * TAPS + wave + 4) * 64];             \
    if (WSHARE > 2 && wave + 8 < TAPS) wq2 = wbase[((size_t)g_ * TAPS + wave + 8) * 64];             \
  } while (0)

  DT_WS_PREFETCH(0);
  for (int g = 0; g < a.groups; ++g) {
    float* wbuf = wlds + (g & 1) * TAPS * 256;
#pragma unroll
    for (int it = 0; it < NLOAD; ++it) {
      const int idx = (lane >> 1) + it * 32;
      if (idx < NPIX) *reinterpret_cast<float4*>(tile + idx * 8 + (lane & 1) * 4) = patch[it];
    }
    if (wave < TAPS) *reinterpret_cast<float4*>(wbuf + wave * 256 + lane * 4) = wq0;
    if (WSHARE > 1 && wave + 4 < TAPS) *reinterpret_cast<float4*>(wbuf + (wave + 4) * 256 + lane * 4) = wq1;
    if (WSHARE > 2 && wave + 8 < TAPS) *reinterpret_cast<float4*>(wbuf + (wave + 8) * 256 + lane * 4) = wq2;
    if (g + 1 < a.groups) DT_WS_PREFETCH(g + 1);
    __syncthreads();
#pragma unroll
    for (int t = 0; t < TAPS; ++t) {
      const int ky = t / KS, kx = t - ky * KS;
      const float4 a4 = *reinterpret_cast<const float4*>(wbuf + t * 256 + lane * 4);
      const float4 b4 = *reinterpret_cast<const float4*>(tile + ((py * ST + ky) * IW + (px * ST + kx)) * 8 + half * 4);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.x, b4.x, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.y, b4.y, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.z, b4.z, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.w, b4.w, acc, 0, 0, 0);
    }
  }
#undef DT_WS_PREFETCH

  if (have_tile && oy < a.h_out && ox < a.w_out) {
    const size_t pix_off = (((size_t)n * a.h_out + oy) * a.w_out + ox) * a.c_out;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int co = cb * 32 + q * 8 + half * 4;
      const size_t off = pix_off + co;
      float4 o = make_float4(acc[q * 4 + 0], acc[q * 4 + 1], acc[q * 4 + 2], acc[q * 4 + 3]);
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

// ---- 1x1 convolution: no LDS, operands straight from global memory, 8 K-groups in flight -----------
// A 1x1 conv is a plain GEMM over the flattened pixel index: the lane that owns pixel p / k-half h
// loads its own B fragment (4 consecutive channels) directly, so nothing is staged and eight
// 8-channel groups (16 dwordx4 loads per lane) are issued before their 32 MFMAs -- with one tap per
// group there is no other way to cover the load latency.  One wave = one 32-channel x 32-pixel block.
constexpr int kC1x1Pitch = 36;                       // floats per pixel of the output stage (32 channels + pad: bank spread)
constexpr int kC1x1LdsFloats = 4 * 32 * kC1x1Pitch;   // four waves x 32 pixels
__device__ __forceinline__ void conv1x1_mfma_body(const ConvArgs& a, unsigned vblock, float* lds_stage) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int half = lane >> 5, p = lane & 31;
  const long npix = (long)a.n * a.h_out * a.w_out;
  const long pix_blocks = (npix + 31) / 32;
  long bid = (long)vblock * 4 + wave;
  const bool have_block = bid < pix_blocks * a.co_blocks;
  if (!have_block) bid = pix_blocks * a.co_blocks - 1;
  const int cb = (int)(bid % a.co_blocks);
  const long pb = bid / a.co_blocks;
  const long pix = pb * 32 + p;
  const bool live = have_block && pix < npix;
  const long pc = pix < npix ? pix : npix - 1;
  const int ox = (int)(pc % a.w_out);
  const int oy = (int)((pc / a.w_out) % a.h_out);
  const int n = (int)(pc / ((long)a.w_out * a.h_out));
  // element offset of this lane's pixel (+ its 4-channel half) in each source
  const int off0 = pixel_offset(true, n, oy, ox, a.h_in, a.w_in, a.up[0], a.c[0], half);
  const int off1 = pixel_offset(a.nsrc > 1, n, oy, ox, a.h_in, a.w_in, a.up[1], a.c[1], half);
  const int off2 = pixel_offset(a.nsrc > 2, n, oy, ox, a.h_in, a.w_in, a.up[2], a.c[2], half);
  const int ng0 = a.c[0] >> 3, ng1 = a.c[1] >> 3;
  const float* src0 = a.src[0];
  const float* src1 = a.src[1];
  const float* src2 = a.src[2];
  const float4* wbase = reinterpret_cast<const float4*>(a.wp) + ((size_t)cb * a.groups * 2 + half) * 32 + p;

  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  constexpr int U = 8;
  for (int g0 = 0; g0 < a.groups; g0 += U) {
    float4 bq[U], aq[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int g = min(g0 + u, a.groups - 1);
      const int sidx = (g < ng0) ? 0 : ((g < ng0 + ng1) ? 1 : 2);
      const int gl = (sidx == 0) ? g : ((sidx == 1) ? g - ng0 : g - ng0 - ng1);
      const float* sp = ((sidx == 0) ? src0 : ((sidx == 1) ? src1 : src2)) + gl * 8;
      const int off = (sidx == 0) ? off0 : ((sidx == 1) ? off1 : off2);
      bq[u] = *reinterpret_cast<const float4*>(sp + off);
      aq[u] = wbase[(size_t)g * 64];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (g0 + u < a.groups) {
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(aq[u].x, bq[u].x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(aq[u].y, bq[u].y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(aq[u].z, bq[u].z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(aq[u].w, bq[u].w, acc, 0, 0, 0);
      }
    }
  }
  // Epilogue.  The C layout gives lane (pixel p, half h) the channels q*8 + h*4 .. +3 of its pixel: stored from there a wave
  // instruction writes 64 pieces of 16 bytes, 512 B (= c_out floats) apart -- 9.8 M partial-line writes for the matching encoder's
  // 64 -> 128 layer at 8 images (round 5 trace: 68 us for 236 MB = 3.5 TB/s).  The block is transposed through a wave-private LDS
  // stage instead, so that eight consecutive lanes write the 128 contiguous bytes of one pixel's 32 channels: whole lines.
  float* st = lds_stage + wave * (32 * kC1x1Pitch);
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int co = cb * 32 + q * 8 + half * 4;
    float4 o = make_float4(acc[q * 4 + 0], acc[q * 4 + 1], acc[q * 4 + 2], acc[q * 4 + 3]);
    if (a.bias) {
      const float4 bv = *reinterpret_cast<const float4*>(a.bias + co);
      o.x += bv.x; o.y += bv.y; o.z += bv.z; o.w += bv.w;
    }
    if (a.res && live) {
      const float4 rv = *reinterpret_cast<const float4*>(a.res + (size_t)pix * a.c_out + co);
      o.x += rv.x; o.y += rv.y; o.z += rv.z; o.w += rv.w;
    }
    o.x = apply_act(o.x, a.act); o.y = apply_act(o.y, a.act); o.z = apply_act(o.z, a.act); o.w = apply_act(o.w, a.act);
    *reinterpret_cast<float4*>(st + p * kC1x1Pitch + q * 8 + half * 4) = o;
  }
  __builtin_amdgcn_wave_barrier();
  if (have_block) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int pp = j * 8 + (lane >> 3), ch = (lane & 7) * 4;
      const long gp = pb * 32 + pp;
      const float4 o = *reinterpret_cast<const float4*>(st + pp * kC1x1Pitch + ch);
      if (gp < npix) *reinterpret_cast<float4*>(a.out + (size_t)gp * a.c_out + cb * 32 + ch) = o;
    }
  }
}

__global__ __launch_bounds__(256) void conv1x1_mfma_kernel(const ConvArgs a) {
  __shared__ __attribute__((aligned(16))) float lds_stage[kC1x1LdsFloats];
  conv1x1_mfma_body(a, blockIdx.x, lds_stage);
}

// ---- Winograd F(2x2, 3x3) variant of the 3x3 stride-1 conv -------------------------------------------
// Y = A^T [ (G g G^T) .* (B^T d B) ] A per 4x4 input window / 2x2 output tile: 16 multiplies instead of 36,
// i.e. 2.25x fewer MFMAs.  Mapping: a workgroup owns 8x16 output pixels = 4x8 Winograd tiles (the MFMA N
// dimension) of one 32-channel block (M); wave a (of 4) owns row a of the 4x4 transform domain, i.e. the four
// positions xi = (a, 0..3), one accumulator each.  Per 8-channel group: the 10x18-pixel input patch is staged
// once in LDS for all four waves (double buffered), every lane reads the two patch rows its transform row
// combines (8 x ds_read_b128), forms its four B operands with 32 adds, and issues 16 MFMAs against the
// pre-transformed weights.  Epilogue: column inverse transform in registers, row inverse transform across
// the four waves through LDS, then wave w writes output sub-pixel (w>>1, w&1) of every tile.
#ifndef DT_WABL
#define DT_WABL 0
#endif
#ifndef DT_WINO1_WPE
#define DT_WINO1_WPE 3  // waves per SIMD requested for the 256-thread Winograd / pair kernels (4 = 128 registers: spills, A/B)
#endif
#ifndef DT_WINO_BUFLOAD
#define DT_WINO_BUFLOAD 1  // 0 = the round-3 exec-masked global loads (A/B)
#endif
constexpr int kWinoTH = 4, kWinoTW = 8;                      // Winograd tiles per workgroup (rows, cols)
constexpr int kWinoPH = 2 * kWinoTH + 2, kWinoPW = 2 * kWinoTW + 2;  // staged input patch 10 x 18
// LDS layout of a staged 8-channel patch: [channel half][column parity][row (10)][column / 2, pitch 10][4 floats].
// A lane reads, for a fixed (row offset r, column offset c), pixel (2*ty + r, 2*tx + c) of ITS channel half: with the
// plain [pixel][8 channels] layout the 16 lanes that one ds_read_b128 cycle services touch only 4 of the 16 four-bank
// groups (stride 64 B in tx, half the bytes of every pixel unused) -- SQ_LDS_BANK_CONFLICT was 64 % of the kernel's LDS
// cycles (profiles/r2i_pmc_summary.json).  Splitting halves and column parities makes tx advance by one 16-byte slot,
// a row pair by 20 slots (= 4 mod 16), and the lane -> tile map below puts tile rows {0,2} / {1,3} into the two lane
// groups of a b128 access: every access is conflict free.
constexpr int kWinoRowPitch = 10;  // columns per parity (9 used), padded so that two rows shift the bank group by 4
constexpr int kWinoPatchFloats = 2 * 2 * kWinoPH * kWinoRowPitch * 4;  // 1600
__device__ __forceinline__ int wino_lds_off(int half, int y, int x) {
  return ((((half * 2 + (x & 1)) * kWinoPH + y) * kWinoRowPitch) + (x >> 1)) * 4;
}

// KSPLIT > 1: KSPLIT groups of four waves take the 8-channel groups round-robin (each with its own patch
// buffers) and their partial results are summed in the row-inverse step; for layers with too few output
// blocks to fill the chip.
template <int KSPLIT>
__device__ __forceinline__ void conv_wino_body(const ConvArgs& a, float* __restrict__ lds_all, unsigned vblock, unsigned vgrid) {
  constexpr int NPIX = kWinoPH * kWinoPW;  // 180
  constexpr int NLOAD = (NPIX + 127) / 128;  // float4 staging loads per thread and group
  // lds_all: 8192 floats per K-split group: 2 patch buffers (2 x 1600 floats), later 4 waves x 2 x 1024 Z values
  const int tid = threadIdx.x & 255;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);           // transform row of this wave
  const int ks = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 8));  // K-split group
  float* lds = lds_all + ks * 8192;
  const int half = lane >> 5, t = lane & 31;
  // lane -> Winograd tile: ds_read_b128 services lanes {0-3,12-15,20-27} and {4-11,16-19,28-31} (of each half-wave)
  // together; the first group takes tile rows 0 and 2, the second rows 1 and 3 (see wino_lds_off)
  const bool grp_a = (t < 4) || (t >= 12 && t < 16) || (t >= 20 && t < 28);
  const int gi = grp_a ? ((t < 4) ? t : ((t < 16) ? t - 8 : t - 12)) : ((t < 12) ? t - 4 : ((t < 20) ? t - 8 : t - 16));
  const int ty = 2 * (gi >> 3) + (grp_a ? 0 : 1), tx = gi & 7;

  const int wt_x = (a.w_out + 2 * kWinoTW - 1) / (2 * kWinoTW), wt_y = (a.h_out + 2 * kWinoTH - 1) / (2 * kWinoTH);
  // cross-workgroup K split, as in conv_mfma_body: workgroups [0, kplain) own a whole block, the others 1/P of one
  const bool split_wg = a.kparts > 1 && vblock >= (unsigned)a.kplain;
  const int kparts = split_wg ? a.kparts : 1;
  long bid;
  if (a.kparts <= 1) bid = xcd_contiguous_block(a.xcd_remap, vblock, vgrid);
  else if (!split_wg) bid = xcd_contiguous_block(a.xcd_remap, vblock, (unsigned)a.kplain);
  else bid = xcd_contiguous_block(a.xcd_remap, vblock - (unsigned)a.kplain, vgrid - (unsigned)a.kplain);
  const int part = split_wg ? (int)(bid % kparts) : 0;
  if (split_wg) bid = a.kplain + bid / kparts;
  const long blk = bid - a.kplain;  // index into the partial-block scratch
  const int cb = (int)(bid % a.co_blocks);
  bid /= a.co_blocks;
  const int bx = (int)(bid % wt_x);
  bid /= wt_x;
  const int by = (int)(bid % wt_y);
  const int n = (int)(bid / wt_y);
  const int iy0 = by * 2 * kWinoTH - 1, ix0 = bx * 2 * kWinoTW - 1;

  int poff0[NLOAD], poff1[NLOAD], poff2[NLOAD];
#pragma unroll
  for (int it = 0; it < NLOAD; ++it) {
    const int idx = (tid >> 1) + it * 128;
    const int ly = idx / kWinoPW, lx = idx - ly * kWinoPW;
    int iy = iy0 + ly, ix = ix0 + lx;
    bool inside = idx < NPIX && iy >= 0 && iy < a.h_in && ix >= 0 && ix < a.w_in;
    if (a.pad_replicate) {
      iy = min(max(iy, 0), a.h_in - 1);
      ix = min(max(ix, 0), a.w_in - 1);
      inside = idx < NPIX;
    }
    poff0[it] = pixel_offset(inside, n, iy, ix, a.h_in, a.w_in, a.up[0], a.c[0], tid);
    poff1[it] = pixel_offset(inside && a.nsrc > 1, n, iy, ix, a.h_in, a.w_in, a.up[1], a.c[1], tid);
    poff2[it] = pixel_offset(inside && a.nsrc > 2, n, iy, ix, a.h_in, a.w_in, a.up[2], a.c[2], tid);
  }
  const int ng0 = a.c[0] >> 3, ng1 = a.c[1] >> 3;
  const float* src0 = a.src[0];
  const float* src1 = a.src[1];
  const float* src2 = a.src[2];
  // packed weights: [cb][g][xi = 4*row + col][half][32][4]; this wave reads xi = 4*wave .. 4*wave+3
  const float4* wbase = reinterpret_cast<const float4*>(a.wp) + ((size_t)cb * a.groups * 16 + wave * 4) * 64 + lane;

  f32x16 acc[4];
#pragma unroll
  for (int b = 0; b < 4; ++b)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[b][r] = 0.f;

  // the two patch rows transform row `wave` combines:  0: d0 - d2   1: d1 + d2   2: d2 - d1   3: d1 - d3
  const int r1 = (wave == 0) ? 0 : ((wave == 2) ? 2 : 1);
  const int r2 = (wave == 0) ? 2 : ((wave == 1) ? 2 : ((wave == 2) ? 1 : 3));
  const float sgn = (wave == 1) ? 1.0f : -1.0f;
  // LDS offsets of the eight window positions this lane reads per group: rows r1 / r2, columns 0..3
  // (column c of the window = column 2*tx + c of the patch: parity c & 1, slot tx + (c >> 1), i.e. a CONSTANT distance from
  //  column 0 -- the four reads of a row share one address register and differ in the instruction's immediate offset)
  const int wo1_0 = wino_lds_off(half, 2 * ty + r1, 2 * tx), wo2_0 = wino_lds_off(half, 2 * ty + r2, 2 * tx);
  constexpr int kWoff[4] = {0, kWinoPH * kWinoRowPitch * 4, 4, kWinoPH * kWinoRowPitch * 4 + 4};
#define wo1(c) (wo1_0 + kWoff[c])
#define wo2(c) (wo2_0 + kWoff[c])

  float4 patch[NLOAD];
  float4 w[4];
#pragma unroll
  for (int b = 0; b < 4; ++b) w[b] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int it = 0; it < NLOAD; ++it) patch[it] = make_float4(0.f, 0.f, 0.f, 0.f);
#if DT_WINO_BUFLOAD
  // Patch loads as RAW BUFFER loads (round 4): the hardware range check returns zeros for a byte offset beyond the source,
  // so zero padding (offset -1 -> 0xFFFFFFFC) needs no compare / exec mask / zero-initialised destination, and the address is
  // a 32-bit per-lane offset plus a scalar offset instead of 64-bit vector arithmetic.  On gfx950 every vector instruction
  // in this loop costs fp32-MFMA time (scripts/mfma_filler_bench.hip): ~15 fewer per K step.
  const __amdgpu_buffer_rsrc_t rs0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src0), 0, a.src_bytes[0], 0x00020000);
  const __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src1 ? src1 : src0), 0, src1 ? a.src_bytes[1] : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src2 ? src2 : src0), 0, src2 ? a.src_bytes[2] : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_empty = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src0), 0, 0, 0x00020000);
  // (byte offsets, once: -1 -> -4 = 0xFFFFFFFC stays out of range)
#pragma unroll
  for (int it = 0; it < NLOAD; ++it) {
    poff0[it] *= 4;
    poff1[it] *= 4;
    poff2[it] *= 4;
  }
  // W = the weight register set to fill.  A group index past the layer's groups (the padding iterations of an uneven K
  // split) loads zeros through the empty resource -- against the last group's weights, re-read harmlessly.
#define DTW_PREFETCH_BUF(G, W)                                                                       \
  do {                                                                                               \
    const int gq_ = (G);                                                                             \
    const bool live_ = gq_ < a.groups;                                                               \
    const int g_ = live_ ? gq_ : a.groups - 1;                                                       \
    const int sidx = (g_ < ng0) ? 0 : ((g_ < ng0 + ng1) ? 1 : 2);                                    \
    const int gl = (sidx == 0) ? g_ : ((sidx == 1) ? g_ - ng0 : g_ - ng0 - ng1);                     \
    const __amdgpu_buffer_rsrc_t rs = !live_ ? rs_empty : ((sidx == 0) ? rs0 : ((sidx == 1) ? rs1 : rs2)); \
    _Pragma("unroll") for (int it = 0; it < NLOAD; ++it) {                                           \
      const int off = (sidx == 0) ? poff0[it] : ((sidx == 1) ? poff1[it] : poff2[it]);               \
      typedef unsigned u32x4 __attribute__((ext_vector_type(4)));                                    \
      const u32x4 raw = __builtin_amdgcn_raw_buffer_load_b128(rs, off, gl * 32, 0);                  \
      patch[it] = make_float4(__uint_as_float(raw.x), __uint_as_float(raw.y), __uint_as_float(raw.z), __uint_as_float(raw.w)); \
    }                                                                                                \
    const float4* wg = wbase + (size_t)g_ * (16 * 64);                                               \
    _Pragma("unroll") for (int b = 0; b < 4; ++b)                                                    \
      W[b] = (DT_WABL & 1) ? make_float4((float)g_, (float)b, 1.f, 2.f) : wg[b * 64];                \
  } while (0)
#else
#define DTW_PREFETCH(G)                                                                              \
  do {                                                                                               \
    const int g_ = (G);                                                                              \
    const int sidx = (g_ < ng0) ? 0 : ((g_ < ng0 + ng1) ? 1 : 2);                                    \
    const int gl = (sidx == 0) ? g_ : ((sidx == 1) ? g_ - ng0 : g_ - ng0 - ng1);                     \
    const float* sp = ((sidx == 0) ? src0 : ((sidx == 1) ? src1 : src2)) + gl * 8;                   \
    _Pragma("unroll") for (int it = 0; it < NLOAD; ++it) {                                           \
      const int off = (sidx == 0) ? poff0[it] : ((sidx == 1) ? poff1[it] : poff2[it]);               \
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);                                                    \
      if (!(DT_WABL & 2) && off >= 0) v = *reinterpret_cast<const float4*>(sp + off);                \
      patch[it] = v;                                                                                 \
    }                                                                                                \
    const float4* wg = wbase + (size_t)g_ * (16 * 64);                                               \
    _Pragma("unroll") for (int b = 0; b < 4; ++b)                                                    \
      w[b] = (DT_WABL & 1) ? make_float4((float)g_, (float)b, 1.f, 2.f) : wg[b * 64];                \
  } while (0)
#endif

  // every K-split group runs the same number of iterations so that the workgroup barriers line up; a group
  // whose share is exhausted stages zeros against (re-read, harmless) weights
  const int g_stride = KSPLIT * kparts, g_base = part * KSPLIT;
  const int iters = (a.groups - g_base + g_stride - 1) / g_stride;
#ifdef DT_CONV_TIMING
  // phase accounting of ONE wave per workgroup (s_memrealtime, 10 ns ticks; reading a stamp drains lgkmcnt, i.e. it waits
  // for this wave's outstanding LDS operations -- which is what the phases are meant to include)
  unsigned long long tph[5] = {0, 0, 0, 0, 0}, tlast = __builtin_amdgcn_s_memrealtime(), tbegin = tlast;
#define DTW_STAMP(K)                                                        \
  do {                                                                      \
    const unsigned long long now_ = __builtin_amdgcn_s_memrealtime();       \
    tph[K] += now_ - tlast;                                                 \
    tlast = now_;                                                           \
  } while (0)
#else
#define DTW_STAMP(K) do { } while (0)
#endif
#if DT_WINO_BUFLOAD
  // One K step against the weight set WC while the next step's patch and weights (into WN) are in flight.  The loop below is
  // unrolled by two with the two weight sets swapping roles, so no register copy separates the steps.
#define DTW_STEP(I, WC, WN)                                                                                           \
  do {                                                                                                                \
    const int g = g_base + ks + (I) * g_stride;                                                                       \
    float* buf = lds + ((I) & 1) * kWinoPatchFloats;                                                                  \
    _Pragma("unroll") for (int it = 0; it < NLOAD; ++it) {                                                            \
      const int idx = (tid >> 1) + it * 128;                                                                          \
      if (idx < NPIX) *reinterpret_cast<float4*>(buf + wino_lds_off(tid & 1, idx / kWinoPW, idx % kWinoPW)) = patch[it]; \
    }                                                                                                                 \
    DTW_STAMP(0);                                                                                                     \
    if (!(DT_WABL & 8)) __syncthreads();                                                                              \
    DTW_STAMP(1);                                                                                                     \
    if ((I) + 1 < iters) DTW_PREFETCH_BUF(g + g_stride, WN);                                                          \
    DTW_STAMP(2);                                                                                                     \
    float4 tcol[4];                                                                                                   \
    _Pragma("unroll") for (int c = 0; c < 4; ++c) {                                                                   \
      const float4 u = *reinterpret_cast<const float4*>(buf + wo1(c));                                                \
      const float4 v = *reinterpret_cast<const float4*>(buf + wo2(c));                                                \
      tcol[c] = make_float4(u.x + sgn * v.x, u.y + sgn * v.y, u.z + sgn * v.z, u.w + sgn * v.w);                      \
    }                                                                                                                 \
    float4 V[4];                                                                                                      \
    V[0] = make_float4(tcol[0].x - tcol[2].x, tcol[0].y - tcol[2].y, tcol[0].z - tcol[2].z, tcol[0].w - tcol[2].w);   \
    V[1] = make_float4(tcol[1].x + tcol[2].x, tcol[1].y + tcol[2].y, tcol[1].z + tcol[2].z, tcol[1].w + tcol[2].w);   \
    V[2] = make_float4(tcol[2].x - tcol[1].x, tcol[2].y - tcol[1].y, tcol[2].z - tcol[1].z, tcol[2].w - tcol[1].w);   \
    V[3] = make_float4(tcol[1].x - tcol[3].x, tcol[1].y - tcol[3].y, tcol[1].z - tcol[3].z, tcol[1].w - tcol[3].w);   \
    DTW_STAMP(3);                                                                                                     \
    _Pragma("unroll") for (int b = 0; b < 4; ++b) {                                                                   \
      acc[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(WC[b].x, V[b].x, acc[b], 0, 0, 0);                                \
      acc[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(WC[b].y, V[b].y, acc[b], 0, 0, 0);                                \
      acc[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(WC[b].z, V[b].z, acc[b], 0, 0, 0);                                \
      acc[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(WC[b].w, V[b].w, acc[b], 0, 0, 0);                                \
    }                                                                                                                 \
    DTW_STAMP(4);                                                                                                     \
  } while (0)
  float4 w2[4];
#pragma unroll
  for (int b = 0; b < 4; ++b) w2[b] = make_float4(0.f, 0.f, 0.f, 0.f);
  if (iters > 0) DTW_PREFETCH_BUF(g_base + ks, w);
  for (int i = 0; i < iters; i += 2) {
    DTW_STEP(i, w, w2);
    if (i + 1 < iters) DTW_STEP(i + 1, w2, w);
  }
#undef DTW_STEP
#else
  if (g_base + ks < a.groups) DTW_PREFETCH(g_base + ks);
  for (int i = 0; i < iters; ++i) {
    const int g = g_base + ks + i * g_stride;
    const bool live_g = g < a.groups;
    float* buf = lds + (i & 1) * kWinoPatchFloats;
#pragma unroll
    for (int it = 0; it < NLOAD; ++it) {
      const int idx = (tid >> 1) + it * 128;
      if (idx < NPIX)
        *reinterpret_cast<float4*>(buf + wino_lds_off(tid & 1, idx / kWinoPW, idx % kWinoPW)) =
            live_g ? patch[it] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    float4 wc[4];
#pragma unroll
    for (int b = 0; b < 4; ++b) wc[b] = w[b];
#ifdef DT_CONV_TIMING
    asm volatile("s_nop 0" ::"v"(wc[3].w));  // the prefetched weights (and the patch before them) have arrived
#endif
    DTW_STAMP(0);  // wait for the prefetched global data + patch -> LDS
    if (!(DT_WABL & 8)) __syncthreads();  // patch visible; everyone is done with this buffer from two iterations ago
    DTW_STAMP(1);  // barrier
    if (g + g_stride < a.groups) DTW_PREFETCH(g + g_stride);
    DTW_STAMP(2);  // issue of the next step's global loads
    // B^T d B restricted to transform row `wave`
    float4 tcol[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const float4 u = (DT_WABL & 4) ? make_float4(acc[0][c], 1.f, 2.f, (float)g)
                                     : *reinterpret_cast<const float4*>(buf + wo1(c));
      const float4 v = (DT_WABL & 4) ? make_float4(acc[1][c], 3.f, 1.f, (float)c)
                                     : *reinterpret_cast<const float4*>(buf + wo2(c));
      tcol[c] = make_float4(u.x + sgn * v.x, u.y + sgn * v.y, u.z + sgn * v.z, u.w + sgn * v.w);
    }
    float4 V[4];
    V[0] = make_float4(tcol[0].x - tcol[2].x, tcol[0].y - tcol[2].y, tcol[0].z - tcol[2].z, tcol[0].w - tcol[2].w);
    V[1] = make_float4(tcol[1].x + tcol[2].x, tcol[1].y + tcol[2].y, tcol[1].z + tcol[2].z, tcol[1].w + tcol[2].w);
    V[2] = make_float4(tcol[2].x - tcol[1].x, tcol[2].y - tcol[1].y, tcol[2].z - tcol[1].z, tcol[2].w - tcol[1].w);
    V[3] = make_float4(tcol[1].x - tcol[3].x, tcol[1].y - tcol[3].y, tcol[1].z - tcol[3].z, tcol[1].w - tcol[3].w);
#ifdef DT_CONV_TIMING
    asm volatile("s_nop 0" ::"v"(V[0].x), "v"(V[1].y), "v"(V[2].z), "v"(V[3].w));
#endif
    DTW_STAMP(3);  // LDS window reads + input transform
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      acc[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(wc[b].x, V[b].x, acc[b], 0, 0, 0);
      acc[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(wc[b].y, V[b].y, acc[b], 0, 0, 0);
      acc[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(wc[b].z, V[b].z, acc[b], 0, 0, 0);
      acc[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(wc[b].w, V[b].w, acc[b], 0, 0, 0);
    }
    DTW_STAMP(4);  // issue of the 16 MFMAs (an MFMA issues when the pipe takes it)
  }
#endif
#ifdef DT_CONV_TIMING
  if (a.timing && (threadIdx.x & 63) == 0 && wave == DT_WINO_TIMING_WAVE && ks == 0) {
    unsigned long long* o = a.timing + (size_t)vblock * 8;
    o[0] = tbegin;
    o[1] = tph[0]; o[2] = tph[1]; o[3] = tph[2]; o[4] = tph[3]; o[5] = tph[4];
    o[6] = __builtin_amdgcn_s_memrealtime();
  }
#endif
#undef wo1
#undef wo2
#undef DTW_STAMP
#if DT_WINO_BUFLOAD
#undef DTW_PREFETCH_BUF
#else
#undef DTW_PREFETCH
#endif

  // ---- inverse transform: columns in registers, rows across the four waves through LDS ----------------
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
  const int oy = by * 2 * kWinoTH + 2 * ty + p, ox = bx * 2 * kWinoTW + 2 * tx + q;
  const bool in_image = oy < a.h_out && ox < a.w_out;
  const size_t pix_off = (((size_t)n * a.h_out + oy) * a.w_out + ox) * a.c_out;
  // the 4 channel quads of a tile are dealt to the KSPLIT groups
  constexpr int NQ = 4 / KSPLIT;
  float4 ov[NQ];
#pragma unroll
  for (int qi = 0; qi < NQ; ++qi) {
    const int qd = ks * NQ + qi;
    float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int k2 = 0; k2 < KSPLIT; ++k2) {
      const float* z = lds_all + k2 * 8192 + q * 1024 + qd * 256 + lane * 4;
      const float4 za = *reinterpret_cast<const float4*>(z + (p ? 1 : 0) * 2048);
      const float4 zbv = *reinterpret_cast<const float4*>(z + (p ? 2 : 1) * 2048);
      const float4 zc = *reinterpret_cast<const float4*>(z + (p ? 3 : 2) * 2048);
      if (p == 0) {
        o.x += za.x + zbv.x + zc.x; o.y += za.y + zbv.y + zc.y; o.z += za.z + zbv.z + zc.z; o.w += za.w + zbv.w + zc.w;
      } else {
        o.x += za.x - zbv.x - zc.x; o.y += za.y - zbv.y - zc.y; o.z += za.z - zbv.z - zc.z; o.w += za.w - zbv.w - zc.w;
      }
    }
    ov[qi] = o;
  }
  if (kparts > 1) {  // (workgroup-uniform)
    // publish this part's output block, count arrivals, the last workgroup sums the parts in part order (conv_mfma_body)
    float4* slot = reinterpret_cast<float4*>(a.part_buf) + (size_t)blk * kparts * 1024 + threadIdx.x;
#pragma unroll
    for (int qi = 0; qi < NQ; ++qi) store_f4_coherent(slot + ((size_t)part * NQ + qi) * (256 * KSPLIT), ov[qi]);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();  // all partial stores acknowledged; all reads of the Z values in lds_all are done
    unsigned* flag = reinterpret_cast<unsigned*>(lds_all);
    if (threadIdx.x == 0) *flag = __hip_atomic_fetch_add(a.part_cnt + blk, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (*reinterpret_cast<volatile unsigned*>(flag) != (unsigned)(kparts - 1)) return;
    if (threadIdx.x == 0) __hip_atomic_store(a.part_cnt + blk, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
    for (int qi = 0; qi < NQ; ++qi) {
      float4 sum = make_float4(0.f, 0.f, 0.f, 0.f);
      for (int q2 = 0; q2 < kparts; ++q2) {
        const float4 pv = (q2 == part) ? ov[qi] : load_f4_coherent(slot + ((size_t)q2 * NQ + qi) * (256 * KSPLIT));
        sum.x = (q2 == 0) ? pv.x : sum.x + pv.x;
        sum.y = (q2 == 0) ? pv.y : sum.y + pv.y;
        sum.z = (q2 == 0) ? pv.z : sum.z + pv.z;
        sum.w = (q2 == 0) ? pv.w : sum.w + pv.w;
      }
      ov[qi] = sum;
    }
  }
#pragma unroll
  for (int qi = 0; qi < NQ; ++qi) {
    const int qd = ks * NQ + qi;
    float4 o = ov[qi];
    if (in_image) {
      const int co = cb * 32 + qd * 8 + half * 4;
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

template <int KSPLIT>
// (KSPLIT = 1: three 256-thread workgroups per CU -- the register budget the chip-filling layers are tuned for -- is requested
// explicitly: left alone, the compiler spends registers on the unrolled K loop and drops to two)
__global__ __launch_bounds__(256 * KSPLIT, (KSPLIT == 1 ? DT_WINO1_WPE : conv_waves_per_eu(256 * KSPLIT))) void conv_wino_kernel(const ConvArgs a) {
  __shared__ __attribute__((aligned(16))) float lds_all[8192 * KSPLIT];
  conv_wino_body<KSPLIT>(a, lds_all, blockIdx.x, gridDim.x);
}

// ---- two independent convolutions of the SAME input in one launch ------------------------------------------------------
// BasicBlock (reference modules/layers.py:77-94) runs conv1 and the shortcut conv ("downsample": 1x1, or 3x3 stride 2)
// on the same input.  As two launches the second one costs a full kernel boundary (~5-7 us on the GPU for layers whose
// arithmetic is 1-3 us) although nothing depends on it until conv2.  Here the workgroups of both share one grid:
// blocks [0, nblocks_a) run convolution A, the rest convolution B, each with its own ConvArgs and virtual block index.
// Both bodies must use the same workgroup size.
template <class BodyA, class BodyB>
__global__ __launch_bounds__(BodyA::THREADS, (BodyA::THREADS == 256 ? DT_WINO1_WPE : conv_waves_per_eu(BodyA::THREADS))) void conv_pair_kernel(const ConvArgs a, const ConvArgs b, unsigned nblocks_a) {
  static_assert(BodyA::THREADS == BodyB::THREADS, "paired convolutions need equal workgroup sizes");
  constexpr int LDSF = BodyA::LDS_FLOATS > BodyB::LDS_FLOATS ? BodyA::LDS_FLOATS : BodyB::LDS_FLOATS;
  __shared__ __attribute__((aligned(16))) float lds[LDSF > 0 ? LDSF : 4];
  if (blockIdx.x < nblocks_a) BodyA::run(a, lds, blockIdx.x, nblocks_a);
  else BodyB::run(b, lds, blockIdx.x - nblocks_a, gridDim.x - nblocks_a);
}
// ---- a Winograd convolution with the coarse regression heads as extra workgroups of its grid (round 5) ----------------------
// SkipDecoderRegression (reference modules/networks_fast.py:134-141): the heads of scales 3, 2, 1 depend on decoder features
// that are final before the last block's 240x320 convolutions start, and nothing depends on them.  As a launch of their own
// they cost one workgroup's dependent MFMA chain (36 us for 788 small workgroups) at the end of the stream; here they are the
// LAST blocks of the conv launch and start as the conv's first round of workgroups drains (1200 blocks on 768 resident slots:
// the second round is 56 % full).  The launch is bound by slot time either way -- at the conv kernel's 144 registers a head
// workgroup occupies one of three slots per CU for its whole latency-bound chain, where the stand-alone head kernel (40
// registers) keeps all 788 resident at once -- so the merge returns 18 us of the 35 on one stream, not all of it; heads in FRONT
// of the conv blocks fill every slot of the first round and return 9.  Same bodies, hence bit-identical results to two launches.
__global__ __launch_bounds__(256, DT_WINO1_WPE) void conv_wino_heads_kernel(const ConvArgs a, const HeadMultiArgs m, unsigned nblocks_heads) {
  __shared__ __attribute__((aligned(16))) float lds_all[8192];
#ifndef DT_HEADS_FIRST
#define DT_HEADS_FIRST 0  // (measured, DESIGN 4.2 "Round 5": behind the conv blocks 0.900 ms conv stack, in front of them 0.909, two launches 0.919)
#endif
  const unsigned nconv = gridDim.x - nblocks_heads;
  if (DT_HEADS_FIRST) {
    if (blockIdx.x < nblocks_heads) head_multi_block(m, blockIdx.x, lds_all, lds_all + 4096);
    else conv_wino_body<1>(a, lds_all, blockIdx.x - nblocks_heads, nconv);
  } else {
    if (blockIdx.x >= nconv) head_multi_block(m, blockIdx.x - nconv, lds_all, lds_all + 4096);
    else conv_wino_body<1>(a, lds_all, blockIdx.x, nconv);
  }
}

template <int KS, int ST, int SPLIT>
struct MfmaBody {
  static constexpr int THREADS = ConvMfmaCfg<KS, ST, SPLIT>::THREADS;
  static constexpr int LDS_FLOATS = ConvMfmaCfg<KS, ST, SPLIT>::LDS_FLOATS;
  static __device__ __forceinline__ void run(const ConvArgs& a, float* lds, unsigned vb, unsigned vg) { conv_mfma_body<KS, ST, SPLIT>(a, lds, vb, vg); }
};
template <int KSPLIT>
struct WinoBody {
  static constexpr int THREADS = 256 * KSPLIT;
  static constexpr int LDS_FLOATS = 8192 * KSPLIT;
  static __device__ __forceinline__ void run(const ConvArgs& a, float* lds, unsigned vb, unsigned vg) { conv_wino_body<KSPLIT>(a, lds, vb, vg); }
};
struct OneByOneBody {
  static constexpr int THREADS = 256;
  static constexpr int LDS_FLOATS = kC1x1LdsFloats;
  static __device__ __forceinline__ void run(const ConvArgs& a, float* lds, unsigned vb, unsigned) { conv1x1_mfma_body(a, vb, lds); }
};

// OIHW 3x3 weights -> U = G g G^T per (co, ci), packed [co_block][group][xi][half][32][4]
__global__ void conv_wino_pack_kernel(const float* __restrict__ W, float* __restrict__ packed, int c_out, int c_in) {
  const int groups = c_in >> 3;
  const size_t total = (size_t)c_out * c_in * 16;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
    size_t r = idx;
    const int j = r & 3;
    r >>= 2;
    const int i = r & 31;
    r >>= 5;
    const int h = r & 1;
    r >>= 1;
    const int xi = r & 15;
    r >>= 4;
    const int g = r % groups;
    const int cb = (int)(r / groups);
    const int co = cb * 32 + i, ci = g * 8 + h * 4 + j;
    const float* k = W + ((size_t)co * c_in + ci) * 9;
    const int ra = xi >> 2, rb = xi & 3;
    // row ra of G applied to the kernel columns, then row rb of G
    float col[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float k0 = k[0 * 3 + c], k1 = k[1 * 3 + c], k2 = k[2 * 3 + c];
      col[c] = (ra == 0) ? k0 : ((ra == 1) ? 0.5f * (k0 + k1 + k2) : ((ra == 2) ? 0.5f * (k0 - k1 + k2) : k2));
    }
    packed[idx] = (rb == 0) ? col[0]
                            : ((rb == 1) ? 0.5f * (col[0] + col[1] + col[2])
                                         : ((rb == 2) ? 0.5f * (col[0] - col[1] + col[2]) : col[2]));
  }
}

// ---- weight packing: OIHW -> [co_block][group][tap][half][32][4] -----------------------------
__global__ void conv_pack_kernel(const float* __restrict__ W, float* __restrict__ packed, int c_out, int c_in, int ks) {
  const int taps = ks * ks;
  const int groups = c_in >> 3;
  const size_t total = (size_t)c_out * c_in * taps;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
    size_t r = idx;
    const int j = r & 3;
    r >>= 2;
    const int i = r & 31;
    r >>= 5;
    const int h = r & 1;
    r >>= 1;
    const int t = r % taps;
    r /= taps;
    const int g = r % groups;
    const int cb = (int)(r / groups);
    const int co = cb * 32 + i, ci = g * 8 + h * 4 + j;
    packed[idx] = W[((size_t)co * c_in + ci) * taps + t];
  }
}

// ---- direct conv (cross-check): one thread per output element --------------------------------
__global__ void conv_simple_kernel(const ConvArgs a, const float* __restrict__ W, int ks, int st) {
  const size_t total = (size_t)a.n * a.h_out * a.w_out * a.c_out;
  const int pad = ks / 2;
  int cin = 0;
  for (int s = 0; s < a.nsrc; ++s) cin += a.c[s];
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
    size_t r = idx;
    const int co = r % a.c_out;
    r /= a.c_out;
    const int ox = r % a.w_out;
    r /= a.w_out;
    const int oy = r % a.h_out;
    const int n = (int)(r / a.h_out);
    float acc = a.bias ? a.bias[co] : 0.f;
    for (int ky = 0; ky < ks; ++ky)
      for (int kx = 0; kx < ks; ++kx) {
        int iy = oy * st + ky - pad, ix = ox * st + kx - pad;
        if (a.pad_replicate) {
          iy = min(max(iy, 0), a.h_in - 1);
          ix = min(max(ix, 0), a.w_in - 1);
        } else if (iy < 0 || iy >= a.h_in || ix < 0 || ix >= a.w_in) {
          continue;
        }
        int cbase = 0;
        for (int s = 0; s < a.nsrc; ++s) {
          const int up = a.up[s];
          const int hs = up ? (a.h_in >> 1) : a.h_in, ws = up ? (a.w_in >> 1) : a.w_in;
          const int sy = up ? (iy >> 1) : iy, sx = up ? (ix >> 1) : ix;
          const float* sp = a.src[s] + (((size_t)n * hs + sy) * ws + sx) * a.c[s];
          for (int ci = 0; ci < a.c[s]; ++ci)
            acc += W[(((size_t)co * cin + cbase + ci) * ks + ky) * ks + kx] * sp[ci];
          cbase += a.c[s];
        }
      }
    if (a.res) acc += a.res[idx];
    a.out[idx] = apply_act(acc, a.act);
  }
}

// ---- 1x1 conv to one channel: one wave per 64 pixels, lanes own pixels -----------------------
__global__ __launch_bounds__(256) void conv1x1_head_kernel(const float* __restrict__ in, const float* __restrict__ w,
                                                          const float* __restrict__ bias, float* __restrict__ out,
                                                          float* __restrict__ out_exp, int64_t pixels, int c) {
  // 4 lanes cooperate on one pixel (float4 strided over channels), shuffle-reduce
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t pix = t >> 2;
  const int sub = (int)(t & 3);
  float acc = 0.f;
  if (pix < pixels) {
    const float* ip = in + pix * c;
    for (int ci = sub * 4; ci < c; ci += 16) {
      const float4 v = *reinterpret_cast<const float4*>(ip + ci);
      const float4 ww = *reinterpret_cast<const float4*>(w + ci);
      acc += v.x * ww.x + v.y * ww.y + v.z * ww.z + v.w * ww.w;
    }
  }
  acc += __shfl_xor(acc, 1, 64);
  acc += __shfl_xor(acc, 2, 64);
  if (pix < pixels && sub == 0) {
    const float v = acc + (bias ? bias[0] : 0.f);
    out[pix] = v;
    if (out_exp) out_exp[pix] = expf(v);
  }
}

// ---- bilinear x2 upsample (align_corners=False), NHWC, float4 over channels ------------------
__global__ void upsample2x_bilinear_kernel(const float* __restrict__ in, float* __restrict__ out, int n, int h, int w,
                                           int c) {
  const int c4 = c >> 2;
  const size_t total = (size_t)n * (2 * h) * (2 * w) * c4;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
    size_t r = idx;
    const int cc = r % c4;
    r /= c4;
    const int ox = r % (2 * w);
    r /= (2 * w);
    const int oy = r % (2 * h);
    const int b = (int)(r / (2 * h));
    const float sy = fmaxf(((float)oy + 0.5f) * 0.5f - 0.5f, 0.f), sx = fmaxf(((float)ox + 0.5f) * 0.5f - 0.5f, 0.f);
    const int y0 = (int)sy, x0 = (int)sx;
    const int y1 = min(y0 + 1, h - 1), x1 = min(x0 + 1, w - 1);
    const float ly = sy - (float)y0, lx = sx - (float)x0;
    const float hy = 1.f - ly, hx = 1.f - lx;
    const float4* base = reinterpret_cast<const float4*>(in) + (size_t)b * h * w * c4;
    const float4 v00 = base[((size_t)y0 * w + x0) * c4 + cc], v01 = base[((size_t)y0 * w + x1) * c4 + cc];
    const float4 v10 = base[((size_t)y1 * w + x0) * c4 + cc], v11 = base[((size_t)y1 * w + x1) * c4 + cc];
    float4 o;
    // same association as ATen's upsample_bilinear2d: hy*(hx*v00 + lx*v01) + ly*(hx*v10 + lx*v11)
    o.x = hy * (hx * v00.x + lx * v01.x) + ly * (hx * v10.x + lx * v11.x);
    o.y = hy * (hx * v00.y + lx * v01.y) + ly * (hx * v10.y + lx * v11.y);
    o.z = hy * (hx * v00.z + lx * v01.z) + ly * (hx * v10.z + lx * v11.z);
    o.w = hy * (hx * v00.w + lx * v01.w) + ly * (hx * v10.w + lx * v11.w);
    reinterpret_cast<float4*>(out)[idx] = o;
  }
}

static int fill_args(const dt_conv_desc* d, const float* in0, const float* in1, const float* in2, const float* bias,
                     const float* res, float* out, ConvArgs& a, const char* who, bool any_channels = false,
                     bool allow_tr = false) {
  DT_REQUIRE(d != nullptr, "%s: null descriptor", who);
  DT_REQUIRE(d->n > 0 && d->h_out > 0 && d->w_out > 0 && d->h_in > 0 && d->w_in > 0, "%s: bad extents", who);
  DT_REQUIRE(d->nsrc >= 1 && d->nsrc <= 3, "%s: nsrc=%d not in 1..3", who, d->nsrc);
  DT_REQUIRE(d->ksize == 1 || d->ksize == 3, "%s: ksize=%d (1 or 3)", who, d->ksize);
  DT_REQUIRE(d->stride == 1 || d->stride == 2, "%s: stride=%d (1 or 2)", who, d->stride);
  DT_REQUIRE(!(d->ksize == 1 && d->stride == 2), "%s: 1x1 stride-2 conv is not used by the reference", who);
  DT_REQUIRE(d->act >= 0 && d->act <= 3, "%s: act=%d", who, d->act);
  DT_REQUIRE(d->pad_mode == 0 || d->pad_mode == 1, "%s: pad_mode=%d (0 zeros, 1 replicate)", who, d->pad_mode);
  const int pad = d->ksize / 2;
  DT_REQUIRE(d->h_out == (d->h_in + 2 * pad - d->ksize) / d->stride + 1 &&
                 d->w_out == (d->w_in + 2 * pad - d->ksize) / d->stride + 1,
             "%s: output extent %dx%d inconsistent with input %dx%d k=%d s=%d", who, d->h_out, d->w_out, d->h_in,
             d->w_in, d->ksize, d->stride);
  const float* ins[3] = {in0, in1, in2};
  a.groups = 0;
  for (int s = 0; s < 3; ++s) {
    a.src[s] = nullptr;
    a.src_bytes[s] = 0;
    a.c[s] = 0;
    a.up[s] = 0;
  }
  for (int s = 0; s < d->nsrc; ++s) {
    DT_REQUIRE(ins[s] != nullptr, "%s: source %d is null", who, s);
    DT_REQUIRE(d->c[s] > 0 && (any_channels || d->c[s] % 8 == 0), "%s: source %d has %d channels (multiple of 8 required)", who,
               s, d->c[s]);
    DT_REQUIRE(!d->up[s] || (d->h_in % 2 == 0 && d->w_in % 2 == 0), "%s: upsampled source needs even input extent", who);
    a.src[s] = ins[s];
    a.c[s] = d->c[s];
    a.up[s] = d->up[s] ? 1 : 0;
    a.groups += d->c[s] >> 3;
    {  // bytes of the source as stored (a nearest-upsampled source is read at half resolution)
      const size_t px = (size_t)d->n * (d->up[s] ? d->h_in / 2 : d->h_in) * (d->up[s] ? d->w_in / 2 : d->w_in);
      const size_t bytes = px * (size_t)d->c[s] * sizeof(float);
      a.src_bytes[s] = bytes < 0xFFFFF000ull ? (unsigned)bytes : 0xFFFFF000u;  // (larger sources: range check effectively off)
    }
  }
  DT_REQUIRE(out != nullptr, "%s: null output", who);
  a.nsrc = d->nsrc;
  a.bias = bias;
  a.res = res;
  a.out = out;
  DT_REQUIRE(d->transposed == 0 || d->transposed == 1, "%s: transposed=%d (0 or 1)", who, d->transposed);
  DT_REQUIRE(!d->transposed || allow_tr, "%s: transposed tiling is only implemented by the K-split kernels of dt_conv2d_f32 / dt_conv2d_pair_f32", who);
#ifdef DT_CONV_TIMING
  {
    const char* e = getenv("DT_CONV_TIMING_PTR");
    a.timing = e ? reinterpret_cast<unsigned long long*>(strtoull(e, nullptr, 0)) : nullptr;
  }
#endif
  a.tr = d->transposed;
  a.kparts = 1;
  a.kplain = 0;
  a.part_buf = nullptr;
  a.part_cnt = nullptr;
  a.n = d->n;
  a.h_out = a.tr ? d->w_out : d->h_out;  // (the kernels work in the transposed frame)
  a.w_out = a.tr ? d->h_out : d->w_out;
  a.c_out = d->c_out;
  a.h_in = a.tr ? d->w_in : d->h_in;
  a.w_in = a.tr ? d->h_in : d->w_in;
  a.act = d->act;
  a.pad_replicate = d->pad_mode;
  static const int xcd_remap = [] { const char* e = getenv("DT_CONV_XCD_REMAP"); return e ? atoi(e) : 1; }();
  a.xcd_remap = xcd_remap;
  static const int cb_major_mode = [] { const char* e = getenv("DT_CONV_CB_MAJOR"); return e ? atoi(e) : 1; }();
  // weights (c_out x K x taps) outweigh the input (pixels x K) when pixels < c_out x taps
  a.cb_major = cb_major_mode && ((long)d->n * d->h_out * d->w_out < (long)d->c_out * d->ksize * d->ksize);
  a.tiles_x = (a.w_out + kPW - 1) / kPW;
  a.tiles_y = (a.h_out + kPH - 1) / kPH;
  a.co_blocks = d->c_out / 32;
  return 0;
}

// ---- cross-workgroup K split: policy and workspace ---------------------------------------------------------------------
static int device_cus() { return device_cu_count(); }

// Plan objective (dt_conv_set_plan_objective, round 5).  The launch plans below were tuned for the LATENCY of one launch on an
// otherwise idle chip: K split over 8 waves or over several workgroups so that a small layer still covers the CUs.  Those
// choices buy latency with resources -- a 512-thread K-split workgroup at 144 registers owns its CU (no second one fits), a
// cross-workgroup split adds partial-tile traffic and an atomic round trip -- which is the wrong trade when several
// independent frames are in flight and another stream's kernels could use what is left over (bench.py, 4 keyframes in flight:
// profiles/r5b_conv_env_probe.txt).  Bits of the objective mask:
//   1  Winograd layers never split K inside the workgroup (256-thread workgroups, three per CU)
//   2  direct 3x3 K-split kernels use 4 waves instead of 8
//   4  no K split across workgroups
//   8  no tail split (whole blocks first, leftovers in parts)
//   16 low-resolution 1x1 convolutions on the plain kernel (no in-workgroup K split)
// 0 = latency (default).  DT_CONV_OBJ presets it.
static std::atomic<int> g_conv_obj{[] { const char* e = getenv("DT_CONV_OBJ"); return e ? atoi(e) : 0; }()};
static inline int conv_obj() { return g_conv_obj.load(std::memory_order_relaxed); }
int conv_plan_objective_value() { return conv_obj(); }

// Number of workgroups P that share one output block of a K-split kernel.  Measured on the K-split kernels
// (scripts/conv_quantisation.py, profiles/r2p_conv_quantisation.txt): a launch lasts about
//     4 us + max over CUs of the sum over the CU's workgroups of (4.1 us + K steps per wave x waves per SIMD x step time),
// i.e. a workgroup's own latency chain (kernel arguments, first loads, reduction, epilogue: ~4 us) is NOT hidden by other
// workgroups of the same CU -- they all start together and move through their phases in lockstep -- and a second "round" of
// workgroups costs a full extra term (320 blocks of the 30x40 level: 27 us, 256 blocks: 16 us).  Splitting K over P
// workgroups therefore pays exactly when the P-fold workgroup count still fits one round (15x20 level: 120 blocks x 2 ->
// 14.8 instead of 20.6 us), and costs otherwise.  The model below ranks the candidates; workgroups are assumed to be dealt
// to the CUs round-robin in launch order.
struct KSeg {
  long blocks;  // output blocks of this convolution
  int groups, split, ksize;
  bool can_split;  // cross-workgroup split implemented for this body
};
static double launch_cost_us(const KSeg* seg, const int* P, int nseg, int cus) {
  // per-CU load of a round-robin deal: segment workgroups [o, o + n) put ceil/floor(n / cus) on every CU
  double best = 0.0;
  long o = 0;
  // loads of the (at most cus) distinct CUs; n is small (a few hundred), so the direct form is fine
  static thread_local std::vector<double> load;
  load.assign((size_t)cus, 0.0);
  for (int i = 0; i < nseg; ++i) {
    const long n = seg[i].blocks * P[i];
    const long steps = (seg[i].groups + (long)seg[i].split * P[i] - 1) / ((long)seg[i].split * P[i]);
    const double step_us = seg[i].ksize * seg[i].ksize * 4 * 64 / 2400.0;  // 4 MFMAs x 64 cycles per tap and group
    const double wg = 4.1 + (P[i] > 1 ? 0.7 : 0.0) + (double)steps * step_us * (seg[i].split / 4.0);
    const long full = n / cus, rem = n % cus;
    for (int c = 0; c < cus; ++c) {
      const long pos = (c - o % cus + cus) % cus;  // CU c is the pos-th to receive a workgroup of this segment
      load[(size_t)c] += (double)(full + (pos < rem ? 1 : 0)) * wg;
    }
    o += n;
  }
  for (int c = 0; c < cus; ++c) best = load[(size_t)c] > best ? load[(size_t)c] : best;
  return best;
}
static void plan_kparts(const KSeg* seg, int nseg, int* P) {
  static const int forced = [] { const char* e = getenv("DT_CONV_KPARTS"); return e ? atoi(e) : 0; }();
  for (int i = 0; i < nseg; ++i) P[i] = 1;
  if (conv_obj() & 4) return;
  if (forced == 1 || forced == 2 || forced == 4) {
    for (int i = 0; i < nseg; ++i)
      if (seg[i].can_split && seg[i].groups >= seg[i].split * forced) P[i] = forced;
    return;
  }
  // memo: the plan depends only on the launch shape and the CU count of the device it runs on
  typedef std::array<long, 9> Key;
  static std::mutex mtx;
  static std::map<Key, std::array<int, 2>> memo;
  Key key{};
  for (int i = 0; i < nseg && i < 2; ++i) {
    key[(size_t)i * 4 + 0] = seg[i].blocks;
    key[(size_t)i * 4 + 1] = seg[i].groups;
    key[(size_t)i * 4 + 2] = seg[i].split * 2 + (seg[i].can_split ? 1 : 0);
    key[(size_t)i * 4 + 3] = seg[i].ksize;
  }
  const int cus = device_cus();
  key[8] = cus;
  std::lock_guard<std::mutex> lock(mtx);
  auto it = memo.find(key);
  if (it == memo.end()) {
    std::array<int, 2> best = {1, 1};
    int cand[2] = {1, 1};
    double best_cost = launch_cost_us(seg, cand, nseg, cus);
    for (int pa = 1; pa <= 4; pa *= 2)
      for (int pb = 1; pb <= (nseg > 1 ? 4 : 1); pb *= 2) {
        cand[0] = pa;
        cand[1] = pb;
        bool ok = true;
        for (int i = 0; i < nseg; ++i)
          ok = ok && (cand[i] == 1 || (seg[i].can_split && seg[i].groups >= seg[i].split * cand[i]));
        if (!ok) continue;
        const double c = launch_cost_us(seg, cand, nseg, cus);
        if (c < best_cost - 0.75) {  // keep the plain launch unless the split clearly wins
          best_cost = c;
          best = {pa, pb};
        }
      }
    it = memo.emplace(key, best).first;
  }
  for (int i = 0; i < nseg && i < 2; ++i) P[i] = it->second[(size_t)i];
}

// Scratch of the cross-workgroup reduction: one per (device, stream), grown on demand, never shrunk or freed.  Launches on one stream
// are ordered, so consecutive layers reuse it; the counters are zero whenever no launch is in flight (the last workgroup to
// arrive at a block resets its counter).  Two host threads driving the SAME stream concurrently are not supported.
struct ConvScratch {
  float* parts = nullptr;
  unsigned* cnt = nullptr;
  size_t part_floats = 0, counters = 0;
};
static std::mutex g_scratch_mutex;
static std::map<std::pair<int, hipStream_t>, ConvScratch> g_scratch;
static std::vector<void*> g_retired;

static int conv_scratch(hipStream_t st, size_t part_floats, size_t counters, float** parts, unsigned** cnt) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) {
    (void)hipGetLastError();
    return fail("conv scratch: no current device");
  }
  std::lock_guard<std::mutex> lock(g_scratch_mutex);
  ConvScratch& ws = g_scratch[std::make_pair(dev, st)];
  if (ws.part_floats < part_floats || ws.counters < counters) {
    // growing: earlier launches on this stream -- or hipGraphs captured from it -- may still use the old buffers, so they
    // are retired, not freed (a few MB per growth step; the sizes below make growth rare)
    if (ws.parts || ws.cnt) {
      g_retired.push_back(ws.parts);
      g_retired.push_back(ws.cnt);
      ws = ConvScratch();
    }
    const size_t pf = part_floats > (size_t)(4u << 20) ? part_floats : (size_t)(4u << 20);  // >= 16 MB
    const size_t nc = counters > (size_t)16384 ? counters : (size_t)16384;
    if (hipMalloc(&ws.parts, pf * sizeof(float)) != hipSuccess || hipMalloc(&ws.cnt, nc * sizeof(unsigned)) != hipSuccess ||
        hipMemsetAsync(ws.cnt, 0, nc * sizeof(unsigned), st) != hipSuccess) {
      (void)hipGetLastError();
      (void)hipFree(ws.parts);
      (void)hipFree(ws.cnt);
      ws = ConvScratch();
      return fail("conv scratch: cannot allocate %zu + %zu bytes (if this stream is being captured into a hipGraph, run the "
                  "layer on it once before the capture: the scratch is allocated on first use)",
                  pf * sizeof(float), nc * sizeof(unsigned));
    }
    ws.part_floats = pf;
    ws.counters = nc;
  }
  *parts = ws.parts;
  *cnt = ws.cnt;
  return 0;
}

// attach scratch at the given offsets (a pair launch gives its two convolutions disjoint slices)
static int attach_scratch(hipStream_t st, ConvArgs* a, long blocks_a, ConvArgs* b, long blocks_b, size_t part_floats_a = 1024) {
  const long sa = (a && a->kparts > 1) ? blocks_a - a->kplain : 0, sb = (b && b->kparts > 1) ? blocks_b - b->kplain : 0;
  const size_t fa = (size_t)sa * (a ? a->kparts : 0) * part_floats_a, fb = (size_t)sb * (b ? b->kparts : 0) * 1024;
  if (fa + fb == 0) return 0;
  float* parts = nullptr;
  unsigned* cnt = nullptr;
  if (int rc = conv_scratch(st, fa + fb, (size_t)(sa + sb), &parts, &cnt)) return rc;
  if (fa) {
    a->part_buf = parts;
    a->part_cnt = cnt;
  }
  if (fb) {
    b->part_buf = parts + fa;
    b->part_cnt = cnt + sa;
  }
  return 0;
}

// the 16-way split only exists for stride 1 (a stride-2 patch per wave would need 78 KB of LDS)
template <int KS, int ST>
static void launch_split16(const ConvArgs& a, long blocks, hipStream_t st) {
  if constexpr (ST == 1)
    DT_LAUNCH((conv_mfma_kernel<KS, ST, 16>), dim3((unsigned)blocks), dim3(1024), 0, st, a);
}

}  // namespace dt

using namespace dt;

extern "C" {

int64_t dt_conv_pack_floats(int c_out, int c_in, int ksize) { return (int64_t)c_out * c_in * ksize * ksize; }

int dt_conv_pack_f32(const float* W, float* packed, int c_out, int c_in, int ksize, dt_stream_t s) {
  DT_REQUIRE(W && packed, "dt_conv_pack_f32: null pointer");
  DT_REQUIRE(c_out > 0 && c_out % 32 == 0, "dt_conv_pack_f32: c_out=%d must be a multiple of 32", c_out);
  DT_REQUIRE(c_in > 0 && c_in % 8 == 0, "dt_conv_pack_f32: c_in=%d must be a multiple of 8", c_in);
  DT_REQUIRE(ksize == 1 || ksize == 3, "dt_conv_pack_f32: ksize=%d", ksize);
  const size_t total = (size_t)c_out * c_in * ksize * ksize;
  const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
  DT_LAUNCH(conv_pack_kernel, dim3(blocks), dim3(256), 0, to_stream(s), W, packed, c_out, c_in, ksize);
  return check_launch("dt_conv_pack_f32");
}

int dt_conv2d_f32(const dt_conv_desc* d, const float* in0, const float* in1, const float* in2, const float* packed_w,
                  const float* bias, const float* residual, float* out, dt_stream_t s) {
  ConvArgs a;
  if (int rc = fill_args(d, in0, in1, in2, bias, residual, out, a, "dt_conv2d_f32", false, /*allow_tr=*/true)) return rc;
  DT_REQUIRE(packed_w != nullptr, "dt_conv2d_f32: null weights");
  DT_REQUIRE(d->c_out > 0 && d->c_out % 32 == 0, "dt_conv2d_f32: c_out=%d must be a multiple of 32", d->c_out);
  a.wp = packed_w;
  const long blocks = (long)a.n * a.tiles_y * a.tiles_x * a.co_blocks;
  DT_REQUIRE(blocks < 2147483647L, "dt_conv2d_f32: grid too large");
  hipStream_t st = to_stream(s);
  // K-split policy: enough output blocks to fill 1024 SIMDs several times over -> no split;
  // otherwise split K over 4 or 8 waves so that small layers still occupy the chip.
  const long k_steps = (long)a.groups * d->ksize * d->ksize;  // groups x taps per block
  int split = 4;
  if (blocks >= 4096 || k_steps <= 16) split = 1;  // (at 1200 blocks the 4-way K split measured faster)
  else if (blocks * 4 < 2048 && a.groups >= 16) split = 8;
  static const int force_split = [] { const char* e = getenv("DT_CONV_SPLIT"); return e ? atoi(e) : 0; }();
  if (split != 1 && (force_split == 4 || force_split == 8)) split = force_split;
  if (split == 8 && (conv_obj() & 2)) split = 4;
  long grid = blocks;
  if (d->ksize == 3 && split != 1) {
    const KSeg seg = {blocks, a.groups, split, 3, true};
    plan_kparts(&seg, 1, &a.kparts);
    // a few blocks more than CUs (the 30x40 level: 320 on 256): one whole block per CU first, and only the leftovers are
    // split, into about one small workgroup per CU -- a second round of 1/P-sized instead of full-sized workgroups
    static const int tail_mode = [] { const char* e = getenv("DT_CONV_TAIL_SPLIT"); return e ? atoi(e) : 1; }();
    const int cus = device_cus();
    if (tail_mode && !(conv_obj() & (4 | 8)) && a.kparts == 1 && blocks > cus && blocks < 2L * cus && cus % 8 == 0) {
      const long rest = blocks - cus;
      const int P = (rest * 4 <= cus + cus / 4) ? 4 : 2;  // rest x P ~ one workgroup per CU
      if (a.groups >= split * P) {
        a.kparts = P;
        a.kplain = cus;
      }
    }
    grid = a.kplain + (blocks - a.kplain) * a.kparts;
    if (int rc = attach_scratch(st, &a, blocks, nullptr, 0)) return rc;
  }
#define DT_LAUNCH_CONV(KS_, ST_)                                                                                   \
  do {                                                                                                             \
    DT_REQUIRE(!a.tr || split != 1, "dt_conv2d_f32: transposed tiling needs a K-split kernel (see dt_conv_transposed_tiling)"); \
    if (split == 1)                                                                                                \
      DT_LAUNCH((conv_mfma_wshare_kernel<KS_, ST_>),                                                      \
                         dim3((unsigned)((((long)a.n * a.tiles_y * a.tiles_x + 3) / 4) * a.co_blocks)), dim3(256), 0, st, a); \
    else if (split == 4)                                                                                           \
      DT_LAUNCH((conv_mfma_kernel<KS_, ST_, 4>), dim3((unsigned)grid), dim3(256), 0, st, a);               \
    else                                                                                                           \
      DT_LAUNCH((conv_mfma_kernel<KS_, ST_, 8>), dim3((unsigned)grid), dim3(512), 0, st, a);               \
  } while (0)
  if (d->ksize == 3 && d->stride == 1)
    DT_LAUNCH_CONV(3, 1);
  else if (d->ksize == 3 && d->stride == 2)
    DT_LAUNCH_CONV(3, 2);
  else {
    const long pix_blocks = ((long)a.n * a.h_out * a.w_out + 31) / 32;
    const long waves = pix_blocks * a.co_blocks;
    // few pixels, long K (the low-resolution downsample convs): one wave per block leaves most SIMDs idle
    // and runs hundreds of dependent MFMAs in a row -> K-split variant of the tiled kernel instead
    const bool plain_1x1 = (conv_obj() & 16) && !a.tr;
    if (!plain_1x1 && waves < 1024 && a.groups >= 32 && blocks * 8 < 2048)
      launch_split16<1, 1>(a, blocks, st);
    else if (!plain_1x1 && waves < 1024 && a.groups >= 16)
      DT_LAUNCH((conv_mfma_kernel<1, 1, 8>), dim3((unsigned)blocks), dim3(512), 0, st, a);
    else {
      DT_REQUIRE(!a.tr, "dt_conv2d_f32: transposed tiling needs a K-split kernel (see dt_conv_transposed_tiling)");
      DT_LAUNCH(conv1x1_mfma_kernel, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, st, a);
    }
  }
#undef DT_LAUNCH_CONV
  return check_launch("dt_conv2d_f32");
}

int dt_conv_transposed_tiling(const dt_conv_desc* d) {
  if (!d || d->nsrc < 1 || d->nsrc > 3 || d->c_out <= 0 || d->c_out % 32 != 0 || (d->ksize != 1 && d->ksize != 3)) return 0;
  int groups = 0;
  for (int s = 0; s < d->nsrc; ++s) {
    if (d->c[s] <= 0 || d->c[s] % 8 != 0) return 0;
    groups += d->c[s] >> 3;
  }
  auto px_tiles = [&](int h, int w) { return (long)((h + kPH - 1) / kPH) * ((w + kPW - 1) / kPW); };
  const long t_n = px_tiles(d->h_out, d->w_out), t_t = px_tiles(d->w_out, d->h_out);
  if (t_t >= t_n) return 0;
  // the transposed launch must land on a K-split kernel (same rules as dt_conv2d_f32 with the transposed block count)
  const long blocks = (long)d->n * t_t * (d->c_out / 32);
  if (d->ksize == 3) return (blocks < 4096 && (long)groups * 9 > 16) ? 1 : 0;
  const long waves = (((long)d->n * d->h_out * d->w_out + 31) / 32) * (d->c_out / 32);
  return (waves < 1024 && groups >= 16) ? 1 : 0;
}

int64_t dt_conv_wino_pack_floats(int c_out, int c_in) { return (int64_t)c_out * c_in * 16; }

int dt_conv_wino_pack_f32(const float* W, float* packed, int c_out, int c_in, dt_stream_t s) {
  DT_REQUIRE(W && packed, "dt_conv_wino_pack_f32: null pointer");
  DT_REQUIRE(c_out > 0 && c_out % 32 == 0, "dt_conv_wino_pack_f32: c_out=%d must be a multiple of 32", c_out);
  DT_REQUIRE(c_in > 0 && c_in % 8 == 0, "dt_conv_wino_pack_f32: c_in=%d must be a multiple of 8", c_in);
  const size_t total = (size_t)c_out * c_in * 16;
  const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
  DT_LAUNCH(conv_wino_pack_kernel, dim3(blocks), dim3(256), 0, to_stream(s), W, packed, c_out, c_in);
  return check_launch("dt_conv_wino_pack_f32");
}

int dt_conv2d_wino_f32(const dt_conv_desc* d, const float* in0, const float* in1, const float* in2, const float* packed_w,
                       const float* bias, const float* residual, float* out, dt_stream_t s) {
  ConvArgs a;
  if (int rc = fill_args(d, in0, in1, in2, bias, residual, out, a, "dt_conv2d_wino_f32")) return rc;
  DT_REQUIRE(packed_w != nullptr, "dt_conv2d_wino_f32: null weights");
  DT_REQUIRE(d->ksize == 3 && d->stride == 1, "dt_conv2d_wino_f32: only 3x3 stride-1 convolutions (k=%d s=%d)", d->ksize,
             d->stride);
  DT_REQUIRE(d->c_out > 0 && d->c_out % 32 == 0, "dt_conv2d_wino_f32: c_out=%d must be a multiple of 32", d->c_out);
  a.wp = packed_w;
  const long wt_x = (a.w_out + 2 * kWinoTW - 1) / (2 * kWinoTW), wt_y = (a.h_out + 2 * kWinoTH - 1) / (2 * kWinoTH);
  const long blocks = (long)a.n * wt_y * wt_x * a.co_blocks;
  DT_REQUIRE(blocks < 2147483647L, "dt_conv2d_wino_f32: grid too large");
  // few output blocks (the 60x80 level and below): split K over two groups of four waves
  static const int force_split = [] { const char* e = getenv("DT_WINO_KSPLIT"); const int v = e ? atoi(e) : 0; return (v == 1 || v == 2) ? v : 0; }();
  const int ksplit = force_split ? force_split : ((blocks < 256 && a.groups >= 8 && !(conv_obj() & 1)) ? 2 : 1);
  // cross-workgroup K split (see plan_kparts): all parts when the P-fold workgroup count still fits one round of CUs,
  // only the leftover blocks when the launch is slightly larger than the chip (300 blocks at 120x160)
  static const int wino_parts = [] { const char* e = getenv("DT_WINO_KPARTS"); return e ? atoi(e) : -1; }();
  const int cus = device_cus();
  if (wino_parts != 0 && ksplit <= 2 && !(conv_obj() & 4)) {
    if (blocks > cus && blocks < 2L * cus && cus % 8 == 0 && !(conv_obj() & 8)) {
      const long rest = blocks - cus;
      const int P = (rest * 4 <= cus + cus / 4) ? 4 : 2;
      if (a.groups >= ksplit * P * 2) {
        a.kparts = P;
        a.kplain = cus;
      }
    } else if (blocks < cus) {
      // few blocks (the 30x40 level: 96): the largest P that still fits one round and leaves every part >= 2 K groups
      for (int P = 4; P >= 2; P /= 2) {
        if ((wino_parts < 0 || P <= wino_parts) && blocks * P <= cus && a.groups >= ksplit * P * 2) {
          a.kparts = P;
          break;
        }
      }
    }
  }
  const long grid = a.kplain + (blocks - a.kplain) * a.kparts;
  if (int rc = attach_scratch(to_stream(s), &a, blocks, nullptr, 0, 4096)) return rc;
  // (a four-way in-workgroup split -- 1024 threads, 128 registers -- was an experiment of round 3: 18 spilled VGPRs and no
  //  faster anywhere; the instantiation is gone, DT_WINO_KSPLIT accepts 1 or 2)
  if (ksplit == 2)
    DT_LAUNCH(conv_wino_kernel<2>, dim3((unsigned)grid), dim3(512), 0, to_stream(s), a);
  else
    DT_LAUNCH(conv_wino_kernel<1>, dim3((unsigned)grid), dim3(256), 0, to_stream(s), a);
  return check_launch("dt_conv2d_wino_f32");
}

/* dt_conv2d_wino_f32 with n_heads fused regression heads (dt_head_mlp_multi_f32's tables) as the LAST workgroups (DT_HEADS_FIRST=0, the shipped order; 1 = in front) of the same
 * launch.  Only for Winograd launches of 256-thread workgroups without a K split across workgroups (the chip-filling layers);
 * anything else runs as the two launches it replaces.  Results are those of the two launches, bit for bit. */
int dt_conv2d_wino_heads_f32(const dt_conv_desc* d, const float* in0, const float* in1, const float* in2, const float* packed_w,
                             const float* bias, const float* residual, float* out, int n_heads, const float* const* h_in,
                             const float* const* h_wa, const float* const* h_wb, const float* const* h_tail, float* const* h_out,
                             float* const* h_out_exp, const int64_t* h_pixels, const int* h_cin, dt_stream_t s) {
  ConvArgs a;
  if (int rc = fill_args(d, in0, in1, in2, bias, residual, out, a, "dt_conv2d_wino_heads_f32")) return rc;
  DT_REQUIRE(packed_w != nullptr, "dt_conv2d_wino_heads_f32: null weights");
  DT_REQUIRE(d->ksize == 3 && d->stride == 1 && d->c_out > 0 && d->c_out % 32 == 0,
             "dt_conv2d_wino_heads_f32: a 3x3 stride-1 convolution with c_out %% 32 == 0 (k=%d s=%d c_out=%d)", d->ksize, d->stride, d->c_out);
  HeadMultiArgs m;
  unsigned head_blocks = 0;
  if (int rc = head_multi_fill(n_heads, h_in, h_wa, h_wb, h_tail, h_out, h_out_exp, h_pixels, h_cin, m, head_blocks,
                               "dt_conv2d_wino_heads_f32")) return rc;
  a.wp = packed_w;
  const long wt_x = (a.w_out + 2 * kWinoTW - 1) / (2 * kWinoTW), wt_y = (a.h_out + 2 * kWinoTH - 1) / (2 * kWinoTH);
  const long blocks = (long)a.n * wt_y * wt_x * a.co_blocks;
  const int cus = device_cus();
  static const int fuse_on = [] { const char* e = getenv("DT_HEADS_IN_CONV"); return e ? atoi(e) : 1; }();
  if (!fuse_on || blocks < 2L * cus || blocks + head_blocks >= 2147483647L) {
    // a launch that does not fill the chip (K split inside / across workgroups): the two launches it would replace
    if (int rc = dt_conv2d_wino_f32(d, in0, in1, in2, packed_w, bias, residual, out, s)) return rc;
    return dt_head_mlp_multi_f32(n_heads, h_in, h_wa, h_wb, h_tail, h_out, h_out_exp, h_pixels, h_cin, s);
  }
  DT_LAUNCH(conv_wino_heads_kernel, dim3((unsigned)(blocks + head_blocks)), dim3(256), 0, to_stream(s), a, m, head_blocks);
  return check_launch("dt_conv2d_wino_heads_f32");
}

// kernel the single-launch entry points would pick (kept in one place so that the pair launcher agrees with them)
enum ConvPick { PICK_WSHARE, PICK_SPLIT4, PICK_SPLIT8, PICK_1X1_PLAIN, PICK_1X1_SPLIT8, PICK_1X1_SPLIT16 };
static ConvPick pick_direct(const ConvArgs& a, const dt_conv_desc* d) {
  const long blocks = (long)a.n * a.tiles_y * a.tiles_x * a.co_blocks;
  if (d->ksize == 3) {
    const long k_steps = (long)a.groups * 9;
    if (blocks >= 4096 || k_steps <= 16) return PICK_WSHARE;
    if (blocks * 4 < 2048 && a.groups >= 16 && !(conv_obj() & 2)) return PICK_SPLIT8;
    return PICK_SPLIT4;
  }
  const long pix_blocks = ((long)a.n * a.h_out * a.w_out + 31) / 32;
  const long waves = pix_blocks * a.co_blocks;
  if ((conv_obj() & 16) && !a.tr) return PICK_1X1_PLAIN;
  if (waves < 1024 && a.groups >= 32 && blocks * 8 < 2048) return PICK_1X1_SPLIT16;
  if (waves < 1024 && a.groups >= 16) return PICK_1X1_SPLIT8;
  return PICK_1X1_PLAIN;
}

/* conv1 and the shortcut conv of a BasicBlock (modules/layers.py:77-94) in ONE launch: both read the same sources.
 * A: 3x3 (stride 1 or 2), packed for the Winograd kernel when a_wino != 0 (stride 1 only) else for the direct kernel;
 * B: 1x1 stride 1 (when A has stride 1) or 3x3 stride 2 (when A has stride 2), direct packing.  Combinations whose two
 * kernels do not share a workgroup size fall back to two launches inside this call (same products either way; the fp32
 * summation order may differ because the pair plans its K splits jointly). */
int dt_conv2d_pair_f32(const dt_conv_desc* da, const dt_conv_desc* db, const float* in0, const float* in1, const float* in2,
                       const float* packed_wa, int a_wino, const float* bias_a, float* out_a, const float* packed_wb,
                       const float* bias_b, float* out_b, dt_stream_t s) {
  ConvArgs a, b;
  if (int rc = fill_args(da, in0, in1, in2, bias_a, nullptr, out_a, a, "dt_conv2d_pair_f32(A)", false, /*allow_tr=*/!a_wino)) return rc;
  if (int rc = fill_args(db, in0, in1, in2, bias_b, nullptr, out_b, b, "dt_conv2d_pair_f32(B)", false, /*allow_tr=*/true)) return rc;
  DT_REQUIRE(packed_wa && packed_wb, "dt_conv2d_pair_f32: null weights");
  DT_REQUIRE(da->c_out % 32 == 0 && db->c_out % 32 == 0, "dt_conv2d_pair_f32: c_out must be a multiple of 32");
  DT_REQUIRE(da->ksize == 3 && da->stride == db->stride && da->h_out == db->h_out && da->w_out == db->w_out && da->n == db->n,
             "dt_conv2d_pair_f32: A must be 3x3 and both convolutions must produce the same extent");
  DT_REQUIRE((db->ksize == 1 && db->stride == 1) || (db->ksize == 3 && db->stride == 2),
             "dt_conv2d_pair_f32: B must be a 1x1 stride-1 or a 3x3 stride-2 convolution");
  DT_REQUIRE(!a_wino || da->stride == 1, "dt_conv2d_pair_f32: the Winograd kernel needs stride 1");
  a.wp = packed_wa;
  b.wp = packed_wb;
  hipStream_t st = to_stream(s);
  const long blocks_b = (long)b.n * b.tiles_y * b.tiles_x * b.co_blocks;
  const ConvPick pb = pick_direct(b, db);
  using M118 = MfmaBody<1, 1, 8>;
  using M318 = MfmaBody<3, 1, 8>;
  using M328 = MfmaBody<3, 2, 8>;
  using M324 = MfmaBody<3, 2, 4>;
#define DT_PAIR(BA, BB, NA, NB)                                                                                           \
  do {                                                                                                                    \
    DT_LAUNCH((conv_pair_kernel<BA, BB>), dim3((unsigned)((NA) + (NB))), dim3(BA::THREADS), 0, st, a, b,         \
                       (unsigned)(NA));                                                                                   \
    return check_launch("dt_conv2d_pair_f32");                                                                            \
  } while (0)
  // direct 3x3 K-split bodies: the two segments are planned jointly (plan_kparts with nseg = 2), so the split -- and with it
  // the fp32 summation order -- can differ from what a lone launch of either convolution would choose
#define DT_PAIR_KP(BA, BB, SPLIT_A, SPLIT_B)                                                                              \
  do {                                                                                                                    \
    const KSeg seg[2] = {{blocks_a, a.groups, SPLIT_A, 3, true}, {blocks_b, b.groups, SPLIT_B, db->ksize, true}};          \
    int P[2];                                                                                                             \
    plan_kparts(seg, 2, P);                                                                                               \
    a.kparts = P[0];                                                                                                      \
    b.kparts = P[1];                                                                                                      \
    if (int rc = attach_scratch(st, &a, blocks_a, &b, blocks_b)) return rc;                                               \
    DT_PAIR(BA, BB, blocks_a * P[0], blocks_b * P[1]);                                                                    \
  } while (0)
  if (a_wino) {
    const long wt_x = (a.w_out + 2 * kWinoTW - 1) / (2 * kWinoTW), wt_y = (a.h_out + 2 * kWinoTH - 1) / (2 * kWinoTH);
    const long blocks_a = (long)a.n * wt_y * wt_x * a.co_blocks;
    const int ksplit = (blocks_a < 256 && a.groups >= 8 && !(conv_obj() & 1)) ? 2 : 1;
    if (ksplit == 1 && pb == PICK_1X1_PLAIN) {
      const long pix_blocks = ((long)b.n * b.h_out * b.w_out + 31) / 32;
      DT_PAIR(WinoBody<1>, OneByOneBody, blocks_a, (pix_blocks * b.co_blocks + 3) / 4);
    }
    if (ksplit == 2 && (pb == PICK_1X1_SPLIT8 || pb == PICK_1X1_SPLIT16)) {
      // few Winograd blocks (the 30x40 level): split their K over workgroups as the single launch does; the light 1x1
      // workgroups fill in behind them
      static const int wino_parts = [] { const char* e = getenv("DT_WINO_KPARTS"); return e ? atoi(e) : -1; }();
      const int cus = device_cus();
      if (wino_parts != 0 && blocks_a < cus && !(conv_obj() & 4))
        for (int P = 4; P >= 2; P /= 2)
          if ((wino_parts < 0 || P <= wino_parts) && blocks_a * P <= cus && a.groups >= ksplit * P * 2) {
            a.kparts = P;
            break;
          }
      if (int rc = attach_scratch(st, &a, blocks_a, nullptr, 0, 4096)) return rc;
      DT_PAIR(WinoBody<2>, M118, blocks_a * a.kparts, blocks_b);
    }
  } else {
    const long blocks_a = (long)a.n * a.tiles_y * a.tiles_x * a.co_blocks;
    const ConvPick pa = pick_direct(a, da);
    if (da->stride == 1 && pa == PICK_SPLIT8 && (pb == PICK_1X1_SPLIT8 || pb == PICK_1X1_SPLIT16))
      DT_PAIR_KP(M318, M118, 8, 8);
    if (da->stride == 2 && pa == PICK_SPLIT8 && pb == PICK_SPLIT8) DT_PAIR_KP(M328, M328, 8, 8);
    if (da->stride == 2 && pa == PICK_SPLIT4 && pb == PICK_SPLIT4) DT_PAIR_KP(M324, M324, 4, 4);
  }
#undef DT_PAIR_KP
#undef DT_PAIR
  // no common workgroup shape: two launches
  int rc = a_wino ? dt_conv2d_wino_f32(da, in0, in1, in2, packed_wa, bias_a, nullptr, out_a, s)
                  : dt_conv2d_f32(da, in0, in1, in2, packed_wa, bias_a, nullptr, out_a, s);
  if (rc) return rc;
  return dt_conv2d_f32(db, in0, in1, in2, packed_wb, bias_b, nullptr, out_b, s);
}

int dt_conv_set_plan_objective(int mask) {
  g_conv_obj.store(mask < 0 ? 0 : mask, std::memory_order_relaxed);
  return conv_obj();
}

int dt_conv2d_simple_f32(const dt_conv_desc* d, const float* in0, const float* in1, const float* in2, const float* W,
                         const float* bias, const float* residual, float* out, dt_stream_t s) {
  ConvArgs a;
  if (int rc = fill_args(d, in0, in1, in2, bias, residual, out, a, "dt_conv2d_simple_f32", /*any_channels=*/true)) return rc;
  DT_REQUIRE(W != nullptr, "dt_conv2d_simple_f32: null weights");
  DT_REQUIRE(d->c_out > 0, "dt_conv2d_simple_f32: c_out=%d", d->c_out);
  const size_t total = (size_t)a.n * a.h_out * a.w_out * a.c_out;
  const int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
  DT_LAUNCH(conv_simple_kernel, dim3(blocks), dim3(256), 0, to_stream(s), a, W, d->ksize, d->stride);
  return check_launch("dt_conv2d_simple_f32");
}

int dt_conv1x1_head_f32(const float* in, const float* w, const float* bias, float* out, float* out_exp, int64_t pixels,
                        int c, dt_stream_t s) {
  DT_REQUIRE(in && w && out, "dt_conv1x1_head_f32: null pointer");
  DT_REQUIRE(pixels > 0 && c > 0 && c % 4 == 0, "dt_conv1x1_head_f32: bad extents pixels=%ld c=%d", (long)pixels, c);
  const int64_t threads = pixels * 4;
  DT_LAUNCH(conv1x1_head_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, to_stream(s), in, w, bias,
                     out, out_exp, pixels, c);
  return check_launch("dt_conv1x1_head_f32");
}

int dt_upsample2x_bilinear_f32(const float* in, float* out, int n, int h, int w, int c, dt_stream_t s) {
  DT_REQUIRE(in && out, "dt_upsample2x_bilinear_f32: null pointer");
  DT_REQUIRE(n > 0 && h > 0 && w > 0 && c > 0 && c % 4 == 0, "dt_upsample2x_bilinear_f32: bad extents");
  const size_t total = (size_t)n * 4 * h * w * (c / 4);
  const int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
  DT_LAUNCH(upsample2x_bilinear_kernel, dim3(blocks), dim3(256), 0, to_stream(s), in, out, n, h, w, c);
  return check_launch("dt_upsample2x_bilinear_f32");
}

}  // extern "C"
