// Element-wise / window kernels of the matching encoder (reference modules/networks.py:138-189:
// ResNet-18 stem + layer1, 1x1 conv, InstanceNorm, LeakyReLU, 3x3 replicate conv, InstanceNorm).
// The convolutions themselves run on the fp32-MFMA conv primitive (conv.hip); this file holds what
// sits between them.  All activations are NHWC fp32.  Every kernel here is HBM-bound: one read and
// one write of the activation, float4 along the channel axis.
#include <cstdlib>

#include "common.hpp"

namespace dt {

// ---- stem im2col: image NCHW [n,3,H,W] -> [n, Ho, Wo, 152] patches of the 7x7 stride-2 pad-3 conv ----
// Column order ci*49 + ky*7 + kx (= nn.Conv2d weight.reshape(c_out, -1)), columns 147..151 zero, so
// the stem becomes a 1x1 MFMA conv over 152 "channels" (19 groups of 8).
constexpr int kStemCols = 152;
__global__ __launch_bounds__(256) void stem_im2col_kernel(const float* __restrict__ img, float* __restrict__ cols, int n,
                                                         int H, int W, int Ho, int Wo) {
  // one thread per (pixel, 4-column quad): 38 quads per pixel
  const size_t total = (size_t)n * Ho * Wo * (kStemCols / 4);
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
    const int q = (int)(idx % (kStemCols / 4));
    size_t r = idx / (kStemCols / 4);
    const int ox = (int)(r % Wo);
    r /= Wo;
    const int oy = (int)(r % Ho);
    const int b = (int)(r / Ho);
    float v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int col = q * 4 + j;
      float val = 0.f;
      if (col < 147) {
        const int ci = col / 49, t = col - ci * 49;
        const int ky = t / 7, kx = t - ky * 7;
        const int iy = oy * 2 + ky - 3, ix = ox * 2 + kx - 3;
        if (iy >= 0 && iy < H && ix >= 0 && ix < W) val = img[(((size_t)b * 3 + ci) * H + iy) * W + ix];
      }
      v[j] = val;
    }
    *reinterpret_cast<float4*>(cols + idx * 4) = make_float4(v[0], v[1], v[2], v[3]);
  }
}

// ---- max pooling, NHWC, -inf padding (nn.MaxPool2d(k, stride, pad)) --------------------------------
__global__ __launch_bounds__(256) void maxpool_kernel(const float* __restrict__ in, float* __restrict__ out, int n, int h,
                                                     int w, int c, int ho, int wo, int k, int stride, int pad) {
  const int c4 = c >> 2;
  const size_t total = (size_t)n * ho * wo * c4;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
    const int cq = (int)(idx % c4);
    size_t r = idx / c4;
    const int ox = (int)(r % wo);
    r /= wo;
    const int oy = (int)(r % ho);
    const int b = (int)(r / ho);
    float4 m = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
    for (int ky = 0; ky < k; ++ky) {
      const int iy = oy * stride + ky - pad;
      if (iy < 0 || iy >= h) continue;
      for (int kx = 0; kx < k; ++kx) {
        const int ix = ox * stride + kx - pad;
        if (ix < 0 || ix >= w) continue;
        const float4 v = *reinterpret_cast<const float4*>(in + (((size_t)b * h + iy) * w + ix) * c + cq * 4);
        m.x = fmaxf(m.x, v.x);
        m.y = fmaxf(m.y, v.y);
        m.z = fmaxf(m.z, v.z);
        m.w = fmaxf(m.w, v.w);
      }
    }
    *reinterpret_cast<float4*>(out + idx * 4) = m;
  }
}

// ---- anti-aliasing blur + stride-2 subsample: depthwise 4x4 filter, reflect padding (1,2,1,2) ---------
__device__ __forceinline__ int reflect(int i, int n) {
  if (i < 0) i = -i;
  if (i >= n) i = 2 * (n - 1) - i;
  return i;
}
struct BlurFilt {
  float f[16];
};
DT_ARG_NO_POINTERS(BlurFilt);
__global__ __launch_bounds__(256) void blurpool_kernel(const float* __restrict__ in, float* __restrict__ out, int n, int h,
                                                      int w, int c, int ho, int wo, BlurFilt filt) {
  const int c4 = c >> 2;
  const size_t total = (size_t)n * ho * wo * c4;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
    const int cq = (int)(idx % c4);
    size_t r = idx / c4;
    const int ox = (int)(r % wo);
    r /= wo;
    const int oy = (int)(r % ho);
    const int b = (int)(r / ho);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int ky = 0; ky < 4; ++ky) {
      const int iy = reflect(oy * 2 + ky - 1, h);
#pragma unroll
      for (int kx = 0; kx < 4; ++kx) {
        const int ix = reflect(ox * 2 + kx - 1, w);
        const float4 v = *reinterpret_cast<const float4*>(in + (((size_t)b * h + iy) * w + ix) * c + cq * 4);
        const float g = filt.f[ky * 4 + kx];
        acc.x += v.x * g;
        acc.y += v.y * g;
        acc.z += v.z * g;
        acc.w += v.w * g;
      }
    }
    *reinterpret_cast<float4*>(out + idx * 4) = acc;
  }
}

// ---- MaxPool2d(2, stride 1) + BlurPool(4, stride 2) in one pass (the anti-aliased stem's `maxpool`) ------
// out(oy,ox) = sum_{ky,kx} filt[ky][kx] * M(reflect(2oy+ky-1), reflect(2ox+kx-1)),  M = 2x2 running max of
// the (h x w) input, extent (h-1) x (w-1).  One read of the input instead of read+write+read.
// one output pixel (4 channels) of MaxPool2d(2,1) + BlurPool(4, stride 2); base = image + channel quad
__device__ __forceinline__ float4 maxblur_one(const float* __restrict__ base, int oy, int ox, int hm, int wm, int w, int c,
                                              const BlurFilt& filt) {
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  const int y0 = oy * 2 - 1, x0 = ox * 2 - 1;
  if (y0 >= 0 && y0 + 3 < hm && x0 >= 0 && x0 + 3 < wm) {
    // interior: the 4x4 window of 2x2 maxima comes from a 5x5 input window, walked row by row
    float4 prev[4];
#pragma unroll
    for (int r = 0; r < 5; ++r) {
      const float* rp = base + ((size_t)(y0 + r) * w + x0) * c;
      float4 v[5];
#pragma unroll
      for (int q = 0; q < 5; ++q) v[q] = *reinterpret_cast<const float4*>(rp + (size_t)q * c);
      float4 hm4[4];
#pragma unroll
      for (int q = 0; q < 4; ++q)
        hm4[q] = make_float4(fmaxf(v[q].x, v[q + 1].x), fmaxf(v[q].y, v[q + 1].y), fmaxf(v[q].z, v[q + 1].z),
                             fmaxf(v[q].w, v[q + 1].w));
      if (r > 0) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float g = filt.f[(r - 1) * 4 + q];
          acc.x += fmaxf(prev[q].x, hm4[q].x) * g;
          acc.y += fmaxf(prev[q].y, hm4[q].y) * g;
          acc.z += fmaxf(prev[q].z, hm4[q].z) * g;
          acc.w += fmaxf(prev[q].w, hm4[q].w) * g;
        }
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) prev[q] = hm4[q];
    }
  } else {
#pragma unroll
    for (int ky = 0; ky < 4; ++ky) {
      const int my = reflect(y0 + ky, hm);
#pragma unroll
      for (int kx = 0; kx < 4; ++kx) {
        const int mx = reflect(x0 + kx, wm);
        const float4 a = *reinterpret_cast<const float4*>(base + ((size_t)my * w + mx) * c);
        const float4 bq = *reinterpret_cast<const float4*>(base + ((size_t)my * w + mx + 1) * c);
        const float4 cq4 = *reinterpret_cast<const float4*>(base + ((size_t)(my + 1) * w + mx) * c);
        const float4 d = *reinterpret_cast<const float4*>(base + ((size_t)(my + 1) * w + mx + 1) * c);
        const float g = filt.f[ky * 4 + kx];
        acc.x += fmaxf(fmaxf(a.x, bq.x), fmaxf(cq4.x, d.x)) * g;
        acc.y += fmaxf(fmaxf(a.y, bq.y), fmaxf(cq4.y, d.y)) * g;
        acc.z += fmaxf(fmaxf(a.z, bq.z), fmaxf(cq4.z, d.z)) * g;
        acc.w += fmaxf(fmaxf(a.w, bq.w), fmaxf(cq4.w, d.w)) * g;
      }
    }
  }
  return acc;
}

__global__ __launch_bounds__(256) void maxblur_kernel(const float* __restrict__ in, float* __restrict__ out, int n, int h,
                                                     int w, int c, int ho, int wo, BlurFilt filt) {
  const int c4 = c >> 2;
  const int hm = h - 1, wm = w - 1;
  const size_t total = (size_t)n * ho * wo * c4;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
    const int cq = (int)(idx % c4);
    size_t r = idx / c4;
    const int ox = (int)(r % wo);
    r /= wo;
    const int oy = (int)(r % ho);
    const int b = (int)(r / ho);
    const float* base = in + (size_t)b * h * w * c + cq * 4;
    *reinterpret_cast<float4*>(out + idx * 4) = maxblur_one(base, oy, ox, hm, wm, w, c, filt);
  }
}

// Round 5: the same outputs with a 2-D thread -> pixel mapping.  In the linear mapping a workgroup owns 256 / c4 consecutive
// outputs of ONE row, so the three input rows that vertically adjacent outputs share come from L2 every time (trace at 8 images:
// 71 us for 196 MB = 2.8 TB/s).  Here a workgroup owns a th x tw block of outputs (4 x 4 at 64 channels): its (2 th + 3) x (2 tw + 3)
// input window is read once into L1 and reused by all of them.  Same arithmetic per output (maxblur_one): bit-identical.
__global__ __launch_bounds__(256) void maxblur_tiled_kernel(const float* __restrict__ in, float* __restrict__ out, int n, int h,
                                                           int w, int c, int ho, int wo, int tw, int th, int tiles_x,
                                                           int tiles_y, BlurFilt filt) {
  const int c4 = c >> 2;
  const int hm = h - 1, wm = w - 1;
  const int cq = threadIdx.x % c4, pl = threadIdx.x / c4;  // channel quad, pixel of the tile
  unsigned blk = blockIdx.x;
  const int txb = (int)(blk % (unsigned)tiles_x);
  blk /= (unsigned)tiles_x;
  const int tyb = (int)(blk % (unsigned)tiles_y);
  const int b = (int)(blk / (unsigned)tiles_y);
  const int ox = txb * tw + pl % tw, oy = tyb * th + pl / tw;
  if (pl >= tw * th || ox >= wo || oy >= ho) return;
  const float* base = in + (size_t)b * h * w * c + cq * 4;
  float* o = out + ((((size_t)b * ho + oy) * wo + ox) * c4 + cq) * 4;
  *reinterpret_cast<float4*>(o) = maxblur_one(base, oy, ox, hm, wm, w, c, filt);
}

// ---- InstanceNorm2d (affine=False, biased variance, eps) ------------------------------------------
// Pass 1: per (image, pixel chunk) partial sum / sum of squares per channel in double; pass 2 reduces the
// partials per (image, channel), normalises, applies the optional LeakyReLU(0.2) and writes NHWC or NCHW.
// The input may carry more channels per pixel (c_stride) than are normalised (c): the 16-channel final
// conv is computed as a zero-padded 32-channel block.
constexpr int kNormChunk = 64;  // pixels per pass-1 workgroup
__global__ __launch_bounds__(256) void instnorm_partial_kernel(const float* __restrict__ in, double* __restrict__ part,
                                                              int hw, int c, int c_stride, int chunks) {
  // thread t: channel t % c, pixel phase t / c; requires c <= 256 and 256 % c == 0
  const int b = blockIdx.y, chunk = blockIdx.x;
  const int ch = threadIdx.x % c, phase = threadIdx.x / c, phases = 256 / c;
  const int p0 = chunk * kNormChunk, p1 = min(p0 + kNormChunk, hw);
  double s = 0.0, ss = 0.0;
  for (int p = p0 + phase; p < p1; p += phases) {
    const double v = (double)in[((size_t)b * hw + p) * c_stride + ch];
    s += v;
    ss += v * v;
  }
  __shared__ double red[2][256];
  red[0][threadIdx.x] = s;
  red[1][threadIdx.x] = ss;
  __syncthreads();
  if (phase == 0) {
    for (int q = 1; q < phases; ++q) {
      s += red[0][q * c + ch];
      ss += red[1][q * c + ch];
    }
    double* o = part + (((size_t)b * chunks + chunk) * c + ch) * 2;
    o[0] = s;
    o[1] = ss;
  }
}

__global__ __launch_bounds__(256) void instnorm_finalize_kernel(const double* __restrict__ part, float* __restrict__ stats,
                                                               int nimg, int hw, int c, int chunks, float eps) {
  // one wave per (image, channel): lanes stride over the chunk partials, fixed-order butterfly reduction
  const int i = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int lane = threadIdx.x & 63;
  if (i >= nimg * c) return;
  const int b = i / c, ch = i - b * c;
  double s = 0.0, ss = 0.0;
  for (int k = lane; k < chunks; k += 64) {
    const double* o = part + (((size_t)b * chunks + k) * c + ch) * 2;
    s += o[0];
    ss += o[1];
  }
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) {
    s += __shfl_xor(s, m, 64);
    ss += __shfl_xor(ss, m, 64);
  }
  if (lane == 0) {
    const double mean = s / hw;
    double var = ss / hw - mean * mean;
    if (var < 0.0) var = 0.0;
    stats[i * 2 + 0] = (float)mean;
    stats[i * 2 + 1] = (float)(1.0 / sqrt(var + (double)eps));
  }
}

__global__ __launch_bounds__(256) void instnorm_apply_kernel(const float* __restrict__ in, const float* __restrict__ stats,
                                                            float* __restrict__ out, int nimg, int hw, int c, int c_stride,
                                                            int act, int out_nchw) {
  if (!out_nchw) {
    // NHWC -> NHWC: one float4 (4 channels of one pixel) per thread
    const int c4 = c >> 2;
    const size_t total = (size_t)nimg * hw * c4;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
      const int cq = (int)(idx % c4);
      const size_t pix = idx / c4;
      const int b = (int)(pix / hw);
      const float4 x = *reinterpret_cast<const float4*>(in + pix * c_stride + cq * 4);
      const float4 s0 = *reinterpret_cast<const float4*>(stats + ((size_t)b * c + cq * 4) * 2);
      const float4 s1 = *reinterpret_cast<const float4*>(stats + ((size_t)b * c + cq * 4) * 2 + 4);
      float4 v = make_float4((x.x - s0.x) * s0.y, (x.y - s0.z) * s0.w, (x.z - s1.x) * s1.y, (x.w - s1.z) * s1.w);
      if (act == DT_ACT_LRELU02) {
        v.x = v.x >= 0.f ? v.x : 0.2f * v.x;
        v.y = v.y >= 0.f ? v.y : 0.2f * v.y;
        v.z = v.z >= 0.f ? v.z : 0.2f * v.z;
        v.w = v.w >= 0.f ? v.w : 0.2f * v.w;
      }
      *reinterpret_cast<float4*>(out + pix * c + cq * 4) = v;
    }
    return;
  }
  // NHWC -> NCHW: idx enumerates the OUTPUT (coalesced stores, strided L2-resident loads)
  const size_t total = (size_t)nimg * hw * c;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
    const int p = (int)(idx % hw);
    const size_t r = idx / hw;
    const int ch = (int)(r % c);
    const int b = (int)(r / c);
    const float mean = stats[((size_t)b * c + ch) * 2], inv = stats[((size_t)b * c + ch) * 2 + 1];
    float v = (in[((size_t)b * hw + p) * c_stride + ch] - mean) * inv;
    if (act == DT_ACT_LRELU02) v = v >= 0.f ? v : 0.2f * v;
    out[idx] = v;
  }
}

static int grid_for(size_t total) {
  const size_t b = (total + 255) / 256;
  return (int)(b < 16384 ? (b ? b : 1) : 16384);
}

}  // namespace dt

using namespace dt;

extern "C" {

int dt_stem_im2col_f32(const float* image_nchw, float* cols_nhwc, int n, int H, int W, dt_stream_t s) {
  DT_REQUIRE(image_nchw && cols_nhwc, "dt_stem_im2col_f32: null pointer");
  DT_REQUIRE(n > 0 && H > 0 && W > 0, "dt_stem_im2col_f32: bad extents n=%d H=%d W=%d", n, H, W);
  const int Ho = (H + 6 - 7) / 2 + 1, Wo = (W + 6 - 7) / 2 + 1;
  const size_t total = (size_t)n * Ho * Wo * (kStemCols / 4);
  DT_LAUNCH(stem_im2col_kernel, dim3(grid_for(total)), dim3(256), 0, to_stream(s), image_nchw, cols_nhwc, n, H, W,
                     Ho, Wo);
  return check_launch("dt_stem_im2col_f32");
}

int dt_maxpool_f32(const float* in, float* out, int n, int h, int w, int c, int ksize, int stride, int pad, dt_stream_t s) {
  DT_REQUIRE(in && out, "dt_maxpool_f32: null pointer");
  DT_REQUIRE(n > 0 && h > 0 && w > 0 && c > 0 && c % 4 == 0, "dt_maxpool_f32: bad extents");
  DT_REQUIRE(ksize >= 1 && stride >= 1 && pad >= 0 && 2 * pad <= ksize, "dt_maxpool_f32: k=%d stride=%d pad=%d", ksize,
             stride, pad);
  const int ho = (h + 2 * pad - ksize) / stride + 1, wo = (w + 2 * pad - ksize) / stride + 1;
  DT_REQUIRE(ho > 0 && wo > 0, "dt_maxpool_f32: empty output");
  const size_t total = (size_t)n * ho * wo * (c / 4);
  DT_LAUNCH(maxpool_kernel, dim3(grid_for(total)), dim3(256), 0, to_stream(s), in, out, n, h, w, c, ho, wo, ksize,
                     stride, pad);
  return check_launch("dt_maxpool_f32");
}

int dt_blurpool4_s2_f32(const float* in, float* out, const float* filt16_host, int n, int h, int w, int c, dt_stream_t s) {
  DT_REQUIRE(in && out && filt16_host, "dt_blurpool4_s2_f32: null pointer");
  DT_REQUIRE(n > 0 && h > 2 && w > 2 && c > 0 && c % 4 == 0, "dt_blurpool4_s2_f32: bad extents");
  const int ho = (h + 3 - 4) / 2 + 1, wo = (w + 3 - 4) / 2 + 1;
  BlurFilt f;
  for (int i = 0; i < 16; ++i) f.f[i] = filt16_host[i];
  const size_t total = (size_t)n * ho * wo * (c / 4);
  DT_LAUNCH(blurpool_kernel, dim3(grid_for(total)), dim3(256), 0, to_stream(s), in, out, n, h, w, c, ho, wo, f);
  return check_launch("dt_blurpool4_s2_f32");
}

int dt_maxblur_f32(const float* in, float* out, const float* filt16_host, int n, int h, int w, int c, dt_stream_t s) {
  DT_REQUIRE(in && out && filt16_host, "dt_maxblur_f32: null pointer");
  DT_REQUIRE(n > 0 && h > 3 && w > 3 && c > 0 && c % 4 == 0, "dt_maxblur_f32: bad extents");
  const int hm = h - 1, wm = w - 1;
  const int ho = (hm - 1) / 2 + 1, wo = (wm - 1) / 2 + 1;
  BlurFilt f;
  for (int i = 0; i < 16; ++i) f.f[i] = filt16_host[i];
  const size_t total = (size_t)n * ho * wo * (c / 4);
  const int c4 = c / 4;
  static const int tiled_on = [] { const char* e = getenv("DT_MAXBLUR_TILED"); return e ? atoi(e) : 1; }();
  if (tiled_on && c4 <= 64 && 256 % c4 == 0 && 256 / c4 >= 4) {
    const int P = 256 / c4;                       // output pixels per workgroup
    const int tw = P >= 16 ? 4 : 2, th = P / tw;  // 4 x 4 at 64 channels
    const int tiles_x = (wo + tw - 1) / tw, tiles_y = (ho + th - 1) / th;
    const long blocks = (long)n * tiles_x * tiles_y;
    if (blocks < 2147483647L) {
      DT_LAUNCH(maxblur_tiled_kernel, dim3((unsigned)blocks), dim3(256), 0, to_stream(s), in, out, n, h, w, c, ho, wo, tw, th,
                tiles_x, tiles_y, f);
      return check_launch("dt_maxblur_f32");
    }
  }
  DT_LAUNCH(maxblur_kernel, dim3(grid_for(total)), dim3(256), 0, to_stream(s), in, out, n, h, w, c, ho, wo, f);
  return check_launch("dt_maxblur_f32");
}

int64_t dt_instnorm_workspace_bytes(int n, int hw, int c) {
  const int64_t chunks = (hw + kNormChunk - 1) / kNormChunk;
  return (int64_t)n * chunks * c * 2 * sizeof(double) + (int64_t)n * c * 2 * sizeof(float);
}

int dt_instnorm_f32(const float* in, float* out, void* workspace, int n, int hw, int c, int c_stride, float eps, int act,
                    int out_nchw, dt_stream_t s) {
  DT_REQUIRE(in && out && workspace, "dt_instnorm_f32: null pointer");
  DT_REQUIRE(n > 0 && hw > 0 && c > 0 && c <= 256 && 256 % c == 0, "dt_instnorm_f32: c=%d must divide 256", c);
  DT_REQUIRE(c_stride >= c && c % 4 == 0 && c_stride % 4 == 0, "dt_instnorm_f32: c_stride=%d / c=%d (multiples of 4, stride >= c)",
             c_stride, c);
  DT_REQUIRE(act == DT_ACT_NONE || act == DT_ACT_LRELU02, "dt_instnorm_f32: act=%d", act);
  const int chunks = (hw + kNormChunk - 1) / kNormChunk;
  double* part = reinterpret_cast<double*>(workspace);
  float* stats = reinterpret_cast<float*>(part + (size_t)n * chunks * c * 2);
  hipStream_t st = to_stream(s);
  DT_LAUNCH(instnorm_partial_kernel, dim3(chunks, n), dim3(256), 0, st, in, part, hw, c, c_stride, chunks);
  DT_LAUNCH(instnorm_finalize_kernel, dim3((n * c + 3) / 4), dim3(256), 0, st, part, stats, n, hw, c, chunks, eps);
  const size_t total = out_nchw ? (size_t)n * hw * c : (size_t)n * hw * (c / 4);
  DT_LAUNCH(instnorm_apply_kernel, dim3(grid_for(total)), dim3(256), 0, st, in, stats, out, n, hw, c, c_stride, act,
                     out_nchw);
  return check_launch("dt_instnorm_f32");
}

}  // extern "C"

// =====================================================================================================
// Fused stem: 7x7 stride-2 pad-3 conv 3->64 (+ folded-BN bias, ReLU) straight from the NCHW image.
// The im2col route above moves 2 x 47 MB per 640x480 image through HBM; here a workgroup stages the
// 13x21x3 input patch of a 4x8 output tile in LDS (3.3 KB) and feeds fp32 MFMAs from it.
//
// K layout: (ci, ky, kx padded 7->8) = 168 -> 84 steps of v_mfma_f32_32x32x2_f32; the two K slots of a
// step are kx = 2j and 2j+1, so a lane's LDS address is base(pixel) + khalf + an immediate offset, and the
// padded kx = 7 slot multiplies whatever finite value sits in the next patch column by a zero weight.
// Each lane keeps its 84 weights (one output channel, one K parity) in registers for the whole kernel;
// workgroups are persistent over tiles.  4 waves = 2 pixel tiles x 2 blocks of 32 output channels.
namespace dt {

typedef float stem_f32x16 __attribute__((ext_vector_type(16)));
constexpr int kStemSteps = 84;
constexpr int kStemPH = 13, kStemPW = 22;                  // patch rows, padded row pitch (21 used)
constexpr int kStemPatch = 3 * kStemPH * kStemPW + 2;      // + slack for the zero-weight overread
constexpr int kStemStagePitch = 36;                        // floats per pixel of the output stage (32 channels + pad)

__global__ __launch_bounds__(256, 2) void stem_conv_kernel(const float* __restrict__ img, const float* __restrict__ wp,
                                                          const float* __restrict__ bias, float* __restrict__ out, int n,
                                                          int H, int W, int Ho, int Wo, int act) {
  __shared__ float patch[2][kStemPatch];
  __shared__ __attribute__((aligned(16))) float stage[4 * 32 * kStemStagePitch];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int cb = wave & 1, slot = wave >> 1;
  const int half = lane >> 5, p = lane & 31;
  const int py = p >> 3, px = p & 7;

  float w[kStemSteps];
#pragma unroll
  for (int s = 0; s < kStemSteps; ++s) w[s] = wp[((size_t)cb * kStemSteps + s) * 64 + lane];

  const int tiles_x = (Wo + 7) / 8, tiles_y = (Ho + 3) / 4;
  const long tiles = (long)n * tiles_y * tiles_x;
  const long pairs = (tiles + 1) / 2;
  const int tid2 = threadIdx.x & 127;  // thread index inside this tile's two waves
  float* my_patch = patch[slot];
  if (tid2 < 2) my_patch[3 * kStemPH * kStemPW + tid2] = 0.f;
  const int base_p = (2 * py) * kStemPW + 2 * px + half;
  constexpr int kLoads = (3 * kStemPH * 21 + 127) / 128;  // 7 global loads per thread and tile

  // Round 5: the patch of the NEXT tile is fetched into registers before this tile's MFMAs and written to LDS after them, so
  // that the global-load latency (7 scalar loads per thread and tile) hides behind the 84 MFMAs instead of standing between two
  // workgroup barriers (trace at 8 images: 157 us against 84 us of matrix time).
  float pre[kLoads];
  // per-thread constants of its kLoads patch elements (row, column, patch offset, image offset relative to the tile origin):
  // computed once -- in the loop they cost two integer divisions per element and tile, on the vector ALU that the MFMAs share
  int pr[kLoads], pcol[kLoads], pofs[kLoads];
  long gofs[kLoads];
#pragma unroll
  for (int it = 0; it < kLoads; ++it) {
    const int idx = tid2 + it * 128;
    const int ci = idx / (kStemPH * 21), rem = idx - ci * (kStemPH * 21);
    const int r = rem / 21, c = rem - r * 21;
    const bool ok = idx < 3 * kStemPH * 21;
    pr[it] = ok ? r : -100000;  // (a row that is never inside the image: the element loads nothing and is not stored)
    pcol[it] = c;
    pofs[it] = (ci * kStemPH + r) * kStemPW + c;
    gofs[it] = ((long)ci * H + r) * W + c;
  }
  auto fetch = [&](long pair_) {
    long t_ = pair_ * 2 + slot;
    if (t_ >= tiles) t_ = tiles - 1;
    const int tx_ = (int)(t_ % tiles_x);
    const int ty_ = (int)((t_ / tiles_x) % tiles_y);
    const int b_ = (int)(t_ / ((long)tiles_x * tiles_y));
    const int iy0_ = ty_ * 8 - 3, ix0_ = tx_ * 16 - 3;
    const float* origin = img + (size_t)b_ * 3 * H * W + (long)iy0_ * W + ix0_;  // (may point before the image: only dereferenced in bounds)
#pragma unroll
    for (int it = 0; it < kLoads; ++it) {
      const int iy = iy0_ + pr[it], ix = ix0_ + pcol[it];
      float v = 0.f;
      if (iy >= 0 && iy < H && ix >= 0 && ix < W) v = origin[gofs[it]];
      pre[it] = v;
    }
  };
  if ((long)blockIdx.x < pairs) fetch(blockIdx.x);

  for (long pair = blockIdx.x; pair < pairs; pair += gridDim.x) {
    long t = pair * 2 + slot;
    const bool have = t < tiles;
    if (!have) t = tiles - 1;
    const int tx = (int)(t % tiles_x);
    const int ty = (int)((t / tiles_x) % tiles_y);
    const int b = (int)(t / ((long)tiles_x * tiles_y));

    __syncthreads();  // previous tile's MFMAs are done reading the patch
#pragma unroll
    for (int it = 0; it < kLoads; ++it)
      if (pr[it] >= 0) my_patch[pofs[it]] = pre[it];
    if (tid2 < 3 * kStemPH) my_patch[tid2 * kStemPW + 21] = 0.f;  // pad column (zero weight, must be finite)
    __syncthreads();
    if (pair + (long)gridDim.x < pairs) fetch(pair + gridDim.x);  // in flight during the MFMAs below

    stem_f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int ci = 0; ci < 3; ++ci)
#pragma unroll
      for (int ky = 0; ky < 7; ++ky)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float bv = my_patch[base_p + (ci * kStemPH + ky) * kStemPW + 2 * j];
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w[(ci * 7 + ky) * 4 + j], bv, acc, 0, 0, 0);
        }

    // Epilogue through a wave-private LDS stage (round 5, as in conv1x1_mfma_body): from the C layout a lane would write four
    // 16-byte pieces 32 bytes apart; transposed, eight consecutive lanes write the 128 contiguous bytes of one pixel's 32 channels
    float* st = stage + wave * (32 * kStemStagePitch);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float4 bv = *reinterpret_cast<const float4*>(bias + cb * 32 + q * 8 + half * 4);
      float4 v = make_float4(acc[q * 4 + 0] + bv.x, acc[q * 4 + 1] + bv.y, acc[q * 4 + 2] + bv.z, acc[q * 4 + 3] + bv.w);
      if (act == DT_ACT_RELU) {
        v.x = fmaxf(v.x, 0.f);
        v.y = fmaxf(v.y, 0.f);
        v.z = fmaxf(v.z, 0.f);
        v.w = fmaxf(v.w, 0.f);
      }
      *reinterpret_cast<float4*>(st + p * kStemStagePitch + q * 8 + half * 4) = v;
    }
    __builtin_amdgcn_wave_barrier();
    if (have) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int pp = j * 8 + (lane >> 3), ch = (lane & 7) * 4;  // pixel (row j, column lane >> 3) of the 4 x 8 tile
        const int oy = ty * 4 + j, ox = tx * 8 + (lane >> 3);
        const float4 v = *reinterpret_cast<const float4*>(st + pp * kStemStagePitch + ch);
        if (oy < Ho && ox < Wo) *reinterpret_cast<float4*>(out + (((size_t)b * Ho + oy) * Wo + ox) * 64 + cb * 32 + ch) = v;
      }
    }
    __builtin_amdgcn_wave_barrier();  // (the stage is rewritten by this wave's next tile)
  }
}

__global__ void stem_pack_kernel(const float* __restrict__ W, float* __restrict__ packed) {
  // W: [64,3,7,7] -> packed[cb][step = (ci*7+ky)*4 + j][lane]: co = cb*32 + (lane&31), kx = 2j + (lane>>5)
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= 2 * kStemSteps * 64) return;
  const int lane = idx & 63, s = (idx >> 6) % kStemSteps, cb = idx / (64 * kStemSteps);
  const int j = s & 3, ky = (s >> 2) % 7, ci = s / 28;
  const int co = cb * 32 + (lane & 31), kx = 2 * j + (lane >> 5);
  packed[idx] = kx < 7 ? W[((co * 3 + ci) * 7 + ky) * 7 + kx] : 0.f;
}

}  // namespace dt

extern "C" {

int dt_stem_pack_floats(void) { return 2 * dt::kStemSteps * 64; }

int dt_stem_pack_f32(const float* W_64x3x7x7, float* packed, dt_stream_t s) {
  DT_REQUIRE(W_64x3x7x7 && packed, "dt_stem_pack_f32: null pointer");
  DT_LAUNCH(dt::stem_pack_kernel, dim3((2 * dt::kStemSteps * 64 + 255) / 256), dim3(256), 0, dt::to_stream(s),
                     W_64x3x7x7, packed);
  return dt::check_launch("dt_stem_pack_f32");
}

int dt_stem_conv_f32(const float* image_nchw, const float* packed_w, const float* bias64, float* out_nhwc, int n, int H,
                     int W, int act, dt_stream_t s) {
  DT_REQUIRE(image_nchw && packed_w && bias64 && out_nhwc, "dt_stem_conv_f32: null pointer");
  DT_REQUIRE(n > 0 && H > 0 && W > 0, "dt_stem_conv_f32: bad extents n=%d H=%d W=%d", n, H, W);
  DT_REQUIRE(act == DT_ACT_NONE || act == DT_ACT_RELU, "dt_stem_conv_f32: act=%d", act);
  const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
  const long tiles = (long)n * ((Ho + 3) / 4) * ((Wo + 7) / 8);
  const long pairs = (tiles + 1) / 2;
  const int cus = dt::device_cu_count();
  const long want = (long)cus * 3;  // 148 VGPRs -> three workgroups per CU
  DT_LAUNCH(dt::stem_conv_kernel, dim3((unsigned)(pairs < want ? pairs : want)), dim3(256), 0, dt::to_stream(s),
                     image_nchw, packed_w, bias64, out_nhwc, n, H, W, Ho, Wo, act);
  return dt::check_launch("dt_stem_conv_f32");
}

}  // extern "C"
