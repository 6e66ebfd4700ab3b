// Device side of the hint-mesh depth rasteriser (see raster.hip for the semantics it restates), shared by the
// stand-alone render entry points (raster.hip) and the fused marching-cubes -> depth kernel (mc.hip).
#pragma once
#include "common.hpp"

namespace dt {

static __global__ void raster_init_kernel(uint32_t* __restrict__ zb, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) zb[i] = 0xFFFFFFFFu;
}

// rasterise one triangle given in world coordinates
__device__ __forceinline__ void raster_triangle(const float (&X)[3], const float (&Y)[3], const float (&Z)[3],
                                                const float* __restrict__ cam_T_world, const float* __restrict__ K, int h,
                                                int w, uint32_t* __restrict__ zb) {
  const float fx = K[0], cx = K[2], fy = K[5], cy = K[6];
  float sx[3], sy[3], sz[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const float xc = cam_T_world[0] * X[i] + cam_T_world[1] * Y[i] + cam_T_world[2] * Z[i] + cam_T_world[3];
    const float yc = cam_T_world[4] * X[i] + cam_T_world[5] * Y[i] + cam_T_world[6] * Z[i] + cam_T_world[7];
    const float zc = cam_T_world[8] * X[i] + cam_T_world[9] * Y[i] + cam_T_world[10] * Z[i] + cam_T_world[11];
    if (!(zc > 1e-2f)) return;
    sx[i] = fx * xc / zc + cx;
    sy[i] = fy * yc / zc + cy;
    sz[i] = zc;
  }
  const float area = (sx[1] - sx[0]) * (sy[2] - sy[0]) - (sx[2] - sx[0]) * (sy[1] - sy[0]);
  if (fabsf(area) < 1e-12f) return;
  const float inv_area = 1.0f / area;
  const float minx = fminf(sx[0], fminf(sx[1], sx[2])), maxx = fmaxf(sx[0], fmaxf(sx[1], sx[2]));
  const float miny = fminf(sy[0], fminf(sy[1], sy[2])), maxy = fmaxf(sy[0], fmaxf(sy[1], sy[2]));
  const int x0 = max(0, (int)ceilf(minx - 0.5f)), x1 = min(w - 1, (int)floorf(maxx - 0.5f));
  const int y0 = max(0, (int)ceilf(miny - 0.5f)), y1 = min(h - 1, (int)floorf(maxy - 0.5f));
  const float iz0 = 1.0f / sz[0], iz1 = 1.0f / sz[1], iz2 = 1.0f / sz[2];
  for (int y = y0; y <= y1; ++y) {
    const float py = (float)y + 0.5f;
    for (int x = x0; x <= x1; ++x) {
      const float px = (float)x + 0.5f;
      // barycentrics from edge functions, normalised by the signed area (either winding)
      const float b0 = ((sx[1] - px) * (sy[2] - py) - (sx[2] - px) * (sy[1] - py)) * inv_area;
      const float b1 = ((sx[2] - px) * (sy[0] - py) - (sx[0] - px) * (sy[2] - py)) * inv_area;
      const float b2 = ((sx[0] - px) * (sy[1] - py) - (sx[1] - px) * (sy[0] - py)) * inv_area;
      // PyTorch3D 0.7.4 CheckPixelInsideFace (rasterize_meshes.cu): inside = all three barycentrics STRICTLY positive;
      // a sample exactly on an edge is covered by neither neighbour (tests/golden/make_raster_handcases.py, rule R3)
      if (!(b0 > 0.f && b1 > 0.f && b2 > 0.f)) continue;
      const float z = (b0 + b1 + b2) / (b0 * iz0 + b1 * iz1 + b2 * iz2);  // sum b'_i z_i of the corrected barycentrics
      if (z > 0.f) atomicMin(zb + (size_t)y * w + x, __float_as_uint(z));
    }
  }
}

static __global__ void raster_resolve_kernel(const uint32_t* __restrict__ zb, float* __restrict__ out, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    const uint32_t b = zb[i];
    out[i] = (b == 0xFFFFFFFFu) ? -1.0f : __uint_as_float(b);
  }
}

}  // namespace dt
