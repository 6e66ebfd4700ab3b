"""Sanity of the (unpinned) depth-render oracle on analytic scenes.  CPU only."""
import numpy as np

from oracle import raster_ref


def test_fronto_parallel_quad_and_occlusion():
    K = np.array([[50.0, 0, 32, 0], [0, 50.0, 24, 0], [0, 0, 1, 0], [0, 0, 0, 1]])
    T = np.eye(4)
    # big quad at z=2 and a small one in front of it at z=1
    verts = np.array([[-5, -5, 2], [5, -5, 2], [5, 5, 2], [-5, 5, 2], [-0.2, -0.2, 1], [0.2, -0.2, 1], [0.2, 0.2, 1], [-0.2, 0.2, 1.0]])
    faces = np.array([[0, 1, 2], [0, 2, 3], [4, 5, 6], [6, 7, 4]])  # second pair with mixed winding
    d = raster_ref.render_depth(verts, faces, T, K, 48, 64)
    assert d.shape == (48, 64) and np.all(d > 0)
    assert np.isclose(d[0, 0], 2.0) and np.isclose(d[24, 32], 1.0)
    # the small quad covers |x|,|y| < 0.2 at z=1 -> 10 px around the principal point
    assert (d == 1.0).sum() == 20 * 20


def test_slanted_plane_is_perspective_correct_and_background():
    K = np.array([[60.0, 0, 32, 0], [0, 60.0, 24, 0], [0, 0, 1, 0], [0, 0, 0, 1]])
    # plane z = 2 + 0.5 x, as one large triangle pair covering the left part of the image only
    P = lambda x, y: [x, y, 2 + 0.5 * x]
    verts = np.array([P(-3, -3), P(0.2, -3), P(0.2, 3), P(-3, 3)])
    faces = np.array([[0, 1, 2], [0, 2, 3]])
    d = raster_ref.render_depth(verts, faces, np.eye(4), K, 48, 64)
    ys, xs = np.mgrid[0:48, 0:64]
    rx = (xs + 0.5 - 32) / 60.0
    want = 2.0 / (1 - 0.5 * rx)  # ray (rx, ry, 1) * t hits z = 2 + 0.5 x  ->  t = 2 / (1 - 0.5 rx)
    covered = d > 0
    assert covered.any() and (~covered).any()
    np.testing.assert_allclose(d[covered], want[covered], rtol=1e-6)
    assert np.all(d[~covered] == -1)
