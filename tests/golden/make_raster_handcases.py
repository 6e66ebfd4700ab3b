#!/usr/bin/env python3
"""Hand-derived known-answer cases for the hint-mesh depth render.

The reference renders the hint depth with PyTorch3D 0.7.4 (environment.yml:18), a third-party package that is neither in
/root/reference nor installable here, through utils/rendering_utils.py:9-53:

    RasterizationSettings(image_size=(h, w), blur_radius=0.0, faces_per_pixel=1, bin_size=None)
    cameras_from_opencv_projection(R, tvec, K * (w, h), image_size)  ->  MeshRasterizer  ->  fragments.zbuf

This script does NOT run any rasteriser.  It applies, in exact rational arithmetic, the rules PyTorch3D 0.7.4 documents and
implements for that call, each cited to the file of the pinned release that states it, and writes the resulting expected
images to tests/golden/raster_handcases.json.  The simplest cases are additionally written out by hand below and asserted,
so that the derivation itself is checked.

Rules (pytorch3d 0.7.4):
 R1  camera: renderer/camera_conversions.py `_cameras_from_opencv_projection`: R_p3d = R^T with the x and y columns negated,
     T_p3d = tvec with x and y negated, focal = f / s, principal point = -(c - (W/2, H/2)) / s, s = min(W, H) / 2.
     With PerspectiveCameras (cameras.py) this gives  x_ndc = -(u - W/2) / s,  y_ndc = -(v - H/2) / s  where (u, v) is the
     OpenCV pixel coordinate fx X/Z + cx, fy Y/Z + cy of the point in the camera frame X_cam = R X_world + tvec.
 R2  pixel grid: renderer/mesh/rasterize_meshes.py + csrc/rasterize_meshes/rasterize_meshes.cu (`NonSquarePixToNdc`, and the
     `yi = H - 1 - y`, `xi = W - 1 - x` flips): output pixel (row y, column x) samples NDC x = r/2 - r (x + 1/2) / W with
     r = 2 W / min(W, H) (and the same for y with H).  Together with R1:  the sample point of pixel (y, x) is the OpenCV
     pixel coordinate (u, v) = (x + 1/2, y + 1/2) -- for square and non-square images alike.
 R3  coverage: csrc/rasterize_meshes/rasterize_meshes.cu `CheckPixelInsideFace`: barycentric coordinates from edge
     functions (csrc/utils/geometry_utils.cuh `BarycentricCoordsForward`); with blur_radius = 0 a face covers the pixel iff
     `inside = bary.x > 0 && bary.y > 0 && bary.z > 0` -- STRICT: a sample exactly on an edge or vertex is not covered.
     No back-face culling (cull_backfaces=False): either winding covers.  Faces with |area| <= 1e-8 (NDC) are skipped.
 R4  depth: MeshRasterizer.forward sets perspective_correct=True for perspective cameras and clip_barycentric_coords=False
     for blur_radius 0; z of a vertex is its camera-space depth (mesh/rasterizer.py `transform`: verts_ndc[..., 2] =
     verts_view[..., 2]).  `BarycentricPerspectiveCorrectionForward`: b'_i = b_i / z_i / sum_j(b_j / z_j), pz = sum b'_i z_i
     = 1 / sum_i (b_i / z_i).
 R5  visibility: faces_per_pixel=1 keeps the face with the smallest pz (> 0); a face whose three vertices are all behind
     the camera (zmax < 0) is skipped; pixels no face covers hold zbuf = -1.
 Not covered by these cases (stated deviation, DESIGN.md section 2, oracle table): faces that cross the camera plane.  PyTorch3D's
 z_clip_value is None for PerspectiveCameras (no znear), so it rasterises their wrapped-around projections; our renderer
 drops faces with a vertex nearer than 1 cm.
"""
import json
import os
from fractions import Fraction as Fr

HERE = os.path.dirname(os.path.abspath(__file__))


def mat_identity():
    return [[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 1, 0], [0, 0, 0, 1]]


def derive(case):
    """Expected zbuf [h][w] (Fractions; -1 background) by rules R1-R5."""
    h, w = case["h"], case["w"]
    K, T = case["K"], case["cam_T_world"]
    fx, fy, cx, cy = Fr(K[0][0]), Fr(K[1][1]), Fr(K[0][2]), Fr(K[1][2])
    cam = []
    for v in case["verts"]:
        p = [sum(Fr(T[r][c]) * Fr(v[c]) for c in range(3)) + Fr(T[r][3]) for r in range(3)]  # R1: X_cam = R X + t
        cam.append(p)
    out = [[Fr(-1)] * w for _ in range(h)]
    for f in case["faces"]:
        P = [cam[i] for i in f]
        if max(p[2] for p in P) < 0:  # R5
            continue
        assert all(p[2] > 0 for p in P), "cases with faces crossing the camera plane are out of scope (see docstring)"
        uv = [(fx * p[0] / p[2] + cx, fy * p[1] / p[2] + cy) for p in P]  # R1
        z = [p[2] for p in P]

        def edge(a, b, q):
            return (q[0] - a[0]) * (b[1] - a[1]) - (q[1] - a[1]) * (b[0] - a[0])

        area = edge(uv[0], uv[1], uv[2])
        if area == 0:
            continue
        for y in range(h):
            for x in range(w):
                q = (Fr(2 * x + 1, 2), Fr(2 * y + 1, 2))  # R2
                b = [edge(uv[1], uv[2], q) / area, edge(uv[2], uv[0], q) / area, edge(uv[0], uv[1], q) / area]
                if not (b[0] > 0 and b[1] > 0 and b[2] > 0):  # R3 (strict)
                    continue
                pz = 1 / sum(bi / zi for bi, zi in zip(b, z))  # R4
                if pz > 0 and (out[y][x] < 0 or pz < out[y][x]):  # R5
                    out[y][x] = pz
    return out


def cases():
    I = mat_identity()
    unit_K = [[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 1, 0], [0, 0, 0, 1]]  # fx = fy = 1, c = 0: (u, v) = (X/Z, Y/Z)
    half = Fr(1, 2)
    cs = []
    # A: edges and vertices through sample points (R2 + R3).  Triangle (0.5,0.5) (4.5,0.5) (0.5,4.5) at z = 1 in an 8x8
    # image: samples (x+0.5, y+0.5) strictly inside need x >= 1, y >= 1 and (x+0.5)+(y+0.5) < 5, i.e. x + y < 4:
    # (1,1) (2,1) (1,2).  Row 0 and column 0 lie ON the legs, (3,1) (2,2) (1,3) ON the hypotenuse: not covered.
    cs.append(dict(name="A_edges_through_samples", h=8, w=8, K=unit_K, cam_T_world=I,
                   verts=[[half, half, 1], [Fr(9, 2), half, 1], [half, Fr(9, 2), 1]], faces=[[0, 1, 2]],
                   by_hand={"covered_xy": [[1, 1], [2, 1], [1, 2]], "depth": 1}))
    # B: the same triangle moved by a quarter pixel, both windings: x >= 1, y >= 1, x + y + 1 < 5.5 -> x + y <= 4
    cs.append(dict(name="B_quarter_pixel_shift", h=8, w=8, K=unit_K, cam_T_world=I,
                   verts=[[Fr(3, 4), Fr(3, 4), 1], [Fr(19, 4), Fr(3, 4), 1], [Fr(3, 4), Fr(19, 4), 1]], faces=[[0, 2, 1]],
                   by_hand={"covered_xy": [[1, 1], [2, 1], [3, 1], [1, 2], [2, 2], [1, 3]], "depth": 1}))
    # C: non-square images, off-centre principal point, z = 2 (R1 + R2 for W != H).  u = 2 X / 2 + 1 = X + 1, v = Y + 0.5.
    K_c = [[2, 0, 1, 0], [0, 2, half, 0], [0, 0, 1, 0], [0, 0, 0, 1]]
    tri_c = [[Fr(1, 4), Fr(1, 4), 2], [Fr(21, 4), Fr(3, 4), 2], [Fr(5, 4), Fr(11, 4), 2]]
    cs.append(dict(name="C_landscape_8x4", h=4, w=8, K=K_c, cam_T_world=I, verts=tri_c, faces=[[0, 1, 2]]))
    cs.append(dict(name="C_portrait_4x8", h=8, w=4, K=K_c, cam_T_world=I,
                   verts=[[v[1], v[0], v[2]] for v in tri_c], faces=[[0, 1, 2]]))
    # D: perspective-correct depth (R4): vertex depths 1, 2, 4
    cs.append(dict(name="D_perspective_depth", h=8, w=8, K=[[4, 0, 1, 0], [0, 4, 1, 0], [0, 0, 1, 0], [0, 0, 0, 1]], cam_T_world=I,
                   verts=[[0, 0, 1], [Fr(7, 2), Fr(1, 2), 2], [1, 6, 4]], faces=[[0, 1, 2]]))
    # E: nearest face wins, background -1 (R5): a large far triangle, a small near one in front, one behind the camera
    cs.append(dict(name="E_nearest_wins", h=8, w=8, K=unit_K, cam_T_world=I,
                   verts=[[0, 0, 2], [16, 0, 2], [0, 16, 2], [3, 3, Fr(3, 2)], [9, 3, Fr(3, 2)], [3, 9, Fr(3, 2)],
                          [0, 0, -1], [8, 0, -1], [0, 8, -1]],
                   faces=[[0, 1, 2], [3, 4, 5], [6, 7, 8]]))
    # F: world -> camera pose (R1): camera rotated 90 degrees about its z axis and shifted; X_cam = R X_world + t
    T_f = [[0, -1, 0, 5], [1, 0, 0, Fr(1, 4)], [0, 0, 1, 1], [0, 0, 0, 1]]
    cs.append(dict(name="F_pose", h=6, w=6, K=unit_K, cam_T_world=T_f,
                   verts=[[0, Fr(9, 2), 0], [Fr(9, 2), Fr(9, 2), 1], [Fr(1, 4), Fr(1, 4), 0]], faces=[[0, 1, 2]]))
    return cs


def main():
    out = []
    for c in cases():
        exp = derive(c)
        if "by_hand" in c:  # the derivation against the answer written out by hand above
            cov = sorted([x, y] for y in range(c["h"]) for x in range(c["w"]) if exp[y][x] >= 0)
            assert cov == sorted(c["by_hand"]["covered_xy"]), (c["name"], cov)
            assert all(exp[y][x] == c["by_hand"]["depth"] for x, y in cov)
        n_cov = sum(1 for row in exp for v in row if v >= 0)
        assert n_cov > 0, c["name"]
        f = lambda m: [[float(v) for v in row] for row in m]
        out.append(dict(name=c["name"], h=c["h"], w=c["w"], K=f(c["K"]), cam_T_world=f(c["cam_T_world"]),
                        verts=f(c["verts"]), faces=c["faces"], expected=f(exp), covered=n_cov))
        print(f"{c['name']}: {n_cov} covered pixels")
    json.dump(out, open(os.path.join(HERE, "raster_handcases.json"), "w"), indent=0)


if __name__ == "__main__":
    main()
