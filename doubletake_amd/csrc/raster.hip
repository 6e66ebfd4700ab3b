// Depth rasteriser for the hint mesh, gfx950.
//
// Replaces the PyTorch3D MeshRasterizer call of the reference's hint renderer
// (utils/rendering_utils.py:9-53: image_size=(h,w), blur_radius=0, faces_per_pixel=1, perspective
// cameras built by cameras_from_opencv_projection; only fragments.zbuf is used, background -1).
// Semantics restated: a pixel is covered by a triangle when its centre (x+0.5, y+0.5) lies STRICTLY inside
// the projected triangle (no back-face culling); zbuf is the perspective-correct depth
// 1 / sum(b_i / z_i) of the nearest such triangle.  Faces with a vertex closer than z = 1e-2 are
// dropped (stated deviation: PyTorch3D has no near clip for PerspectiveCameras and rasterises the
// wrapped-around projection of faces that cross the camera plane).
// PyTorch3D 0.7.4 is not installed here; its rules for this call (camera conversion, pixel grid, strict
// coverage, perspective-correct z, nearest face) are pinned by the hand-derived known-answer cases of
// tests/golden/make_raster_handcases.py, which this kernel and the independent numpy oracle
// (oracle/raster_ref.py) both have to reproduce.
//
// One thread per triangle, depth test with atomicMin on the IEEE bits (z > 0 => order preserving).
// Hint meshes are 0.04 m marching-cubes triangles seen from 0.5-3 m: a few pixels each, so the
// per-thread bounding-box loop is short and the kernel is bound by the vertex gather.
#include "common.hpp"

namespace dt {

__global__ void raster_init_kernel(uint32_t* __restrict__ zb, int64_t n) {
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

__global__ void raster_faces_kernel(const float* __restrict__ verts, const int64_t* __restrict__ faces, int64_t num_faces,
                                    const float* __restrict__ cam_T_world, const float* __restrict__ K, int h, int w,
                                    uint32_t* __restrict__ zb) {
  const int64_t f = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (f >= num_faces) return;
  float X[3], Y[3], Z[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int64_t vi = faces[f * 3 + i];
    X[i] = verts[vi * 3 + 0];
    Y[i] = verts[vi * 3 + 1];
    Z[i] = verts[vi * 3 + 2];
  }
  raster_triangle(X, Y, Z, cam_T_world, K, h, w, zb);
}

// triangle soup straight from dt_mc_generate: vertex 3f+i of face f, in (k,j,i) voxel-index coordinates;
// world = origin + (i,j,k) * voxel_size (what TSDF.to_mesh_pytorch3d(scale_to_world=True) computes after the
// axis flip of utils/pytorch3d_extras.py:105).  Skips the sort/unique vertex merge, which rendering ignores.
__global__ void raster_soup_kernel(const float* __restrict__ verts_kji, int64_t num_faces, float ox, float oy, float oz,
                                   float vs, const float* __restrict__ cam_T_world, const float* __restrict__ K, int h, int w,
                                   uint32_t* __restrict__ zb) {
  const int64_t f = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (f >= num_faces) return;
  float X[3], Y[3], Z[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const float* v = verts_kji + (f * 3 + i) * 3;
    X[i] = ox + v[2] * vs;
    Y[i] = oy + v[1] * vs;
    Z[i] = oz + v[0] * vs;
  }
  raster_triangle(X, Y, Z, cam_T_world, K, h, w, zb);
}

__global__ void raster_resolve_kernel(const uint32_t* __restrict__ zb, float* __restrict__ out, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    const uint32_t b = zb[i];
    out[i] = (b == 0xFFFFFFFFu) ? -1.0f : __uint_as_float(b);
  }
}

}  // namespace dt

using namespace dt;

extern "C" {

int dt_raster_depth_f32(const float* verts_v3, const int64_t* faces_f3, int64_t num_faces, const float* cam_T_world_44,
                        const float* K_44, int h, int w, uint32_t* workspace_hw, float* depth_hw, dt_stream_t s) {
  DT_REQUIRE(cam_T_world_44 && K_44 && workspace_hw && depth_hw, "dt_raster_depth_f32: null pointer");
  DT_REQUIRE(h > 0 && w > 0 && num_faces >= 0, "dt_raster_depth_f32: bad extents");
  DT_REQUIRE(num_faces == 0 || (verts_v3 && faces_f3), "dt_raster_depth_f32: null mesh");
  const int64_t n = (int64_t)h * w;
  hipStream_t st = to_stream(s);
  DT_LAUNCH(raster_init_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, workspace_hw, n);
  if (num_faces > 0)
    DT_LAUNCH(raster_faces_kernel, dim3((unsigned)((num_faces + 127) / 128)), dim3(128), 0, st, verts_v3, faces_f3,
                       num_faces, cam_T_world_44, K_44, h, w, workspace_hw);
  DT_LAUNCH(raster_resolve_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, workspace_hw, depth_hw, n);
  return check_launch("dt_raster_depth_f32");
}

int dt_raster_soup_depth_f32(const float* verts_kji_v3, int64_t num_faces, const float* origin3, float voxel_size,
                             const float* cam_T_world_44, const float* K_44, int h, int w, uint32_t* workspace_hw,
                             float* depth_hw, dt_stream_t s) {
  DT_REQUIRE(cam_T_world_44 && K_44 && workspace_hw && depth_hw && origin3, "dt_raster_soup_depth_f32: null pointer");
  DT_REQUIRE(h > 0 && w > 0 && num_faces >= 0 && voxel_size > 0.f, "dt_raster_soup_depth_f32: bad extents");
  DT_REQUIRE(num_faces == 0 || verts_kji_v3, "dt_raster_soup_depth_f32: null mesh");
  const int64_t n = (int64_t)h * w;
  hipStream_t st = to_stream(s);
  DT_LAUNCH(raster_init_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, workspace_hw, n);
  if (num_faces > 0)
    DT_LAUNCH(raster_soup_kernel, dim3((unsigned)((num_faces + 127) / 128)), dim3(128), 0, st, verts_kji_v3, num_faces,
                       origin3[0], origin3[1], origin3[2], voxel_size, cam_T_world_44, K_44, h, w, workspace_hw);
  DT_LAUNCH(raster_resolve_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, workspace_hw, depth_hw, n);
  return check_launch("dt_raster_soup_depth_f32");
}

}  // extern "C"
