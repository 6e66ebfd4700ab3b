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
#include "raster_device.hpp"

namespace dt {

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
