"""Pins oracle/knn_oracle.c (CPU): grid search == brute force bit-for-bit == numpy brute force."""
import numpy as np

from oracle import raster as orc


def test_grid_equals_brute_and_numpy():
    rng = np.random.default_rng(0)
    for n in (5, 70, 3000):
        pts = rng.normal(size=(n, 3)).astype(np.float32)
        a, b = orc.dist2_knn3(pts), orc.dist2_knn3(pts, brute=True)
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
        d = pts[:, None, :] - pts[None, :, :]
        d2 = (d[..., 0] * d[..., 0] + d[..., 1] * d[..., 1]) + d[..., 2] * d[..., 2]
        np.fill_diagonal(d2, np.inf)
        s = np.sort(d2, axis=1)[:, :3].astype(np.float32)
        want = ((s[:, 0] + s[:, 1]) + s[:, 2]) / np.float32(3.0)
        assert np.array_equal(a.view(np.uint32), want.view(np.uint32))


def test_clustered_points_and_duplicates():
    rng = np.random.default_rng(1)
    pts = np.concatenate([rng.normal(size=(500, 3)) * 1e-3, rng.normal(size=(500, 3)) * 10 + 50]).astype(np.float32)
    pts[10] = pts[11]
    a, b = orc.dist2_knn3(pts), orc.dist2_knn3(pts, brute=True)
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
