"""GPU parity of distCUDA2 (csrc/knn.hip) against the C oracle: bit-exact (shared d2 arithmetic,
the 3 smallest values are order-independent)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("no HIP device")


@pytest.mark.parametrize("n,seed", [(1, 0), (2, 0), (3, 1), (4, 1), (257, 2), (8_192, 5), (8_193, 5), (10_000, 3), (50_000, 4)])
def test_distcuda2_bit_exact(n, seed):
    _need_gpu()
    import dreammesh4d_amd
    from oracle import raster as orc

    dreammesh4d_amd.install_compat()
    from simple_knn._C import distCUDA2   # the name the reference imports

    rng = np.random.default_rng(seed)
    pts = rng.normal(size=(n, 3)).astype(np.float32)
    if n > 100:
        pts[5] = pts[6]                      # a duplicate point: distance 0 counts, self does not
    got = distCUDA2(torch.tensor(pts, device="cuda:0")).cpu().numpy()
    want = orc.dist2_knn3(pts)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    if n > 100:
        assert got[5] >= 0 and np.isfinite(got).all()
        assert np.array_equal(orc.dist2_knn3(pts, brute=True).view(np.uint32), want.view(np.uint32))


def test_distcuda2_empty_and_cpu_rejected():
    _need_gpu()
    from dreammesh4d_amd.simple_knn._C import distCUDA2

    assert distCUDA2(torch.zeros(0, 3, device="cuda:0")).shape == (0,)
    with pytest.raises(RuntimeError):
        distCUDA2(torch.zeros(4, 3))


@pytest.mark.parametrize("kind", ["uniform", "clustered", "mesh"])
def test_distcuda2_one_million_points_box_search_bit_exact(kind):
    """BASELINE configs[4] size: 1 M points through the Morton-box search (csrc/knn.hip, dm4d_dist2_knn3_ws) -- bit-identical to
    the oracle's exact search; clustered / degenerate clouds (all boxes overlap, duplicates, a flat sheet) included; and the
    exhaustive kernel agrees on a 30k subset."""
    _need_gpu()
    import time

    from dreammesh4d_amd.simple_knn._C import distCUDA2
    from oracle import raster as orc

    rng = np.random.default_rng(7)
    n = 1_000_002
    if kind == "uniform":
        pts = rng.random((n, 3)).astype(np.float32) * 1.2 - 0.6
    elif kind == "clustered":
        c = rng.normal(size=(50, 3)) * 0.5
        pts = (c[rng.integers(0, 50, n)] + rng.normal(size=(n, 3)) * 0.01).astype(np.float32)
        pts[1000:1200] = pts[1000]                                   # 200 coincident points
    else:
        d = rng.normal(size=(n, 3)); d /= np.linalg.norm(d, axis=1, keepdims=True)
        pts = (0.6 * d).astype(np.float32)                           # a surface (what a mesh-bound cloud looks like)
        pts[:, 2] = np.where(np.arange(n) % 7 == 0, 0.0, pts[:, 2])  # + a flat sheet through it
    t = torch.tensor(pts, device="cuda:0")
    distCUDA2(t[:20000]); torch.cuda.synchronize()
    t0 = time.perf_counter()
    got = distCUDA2(t)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    want = orc.dist2_knn3(pts)
    got = got.cpu().numpy()
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    print(f"distCUDA2 {kind}: {n} points in {dt * 1e3:.1f} ms (box search)")
    sub = pts[:30_000]
    from dreammesh4d_amd import _lib
    L = _lib.lib()
    ts, out = torch.tensor(sub, device="cuda:0"), torch.empty(len(sub), device="cuda:0")
    _lib.check(L.dm4d_dist2_knn3(len(sub), ts.data_ptr(), out.data_ptr(), torch.cuda.current_stream().cuda_stream))     # exhaustive
    assert np.array_equal(out.cpu().numpy().view(np.uint32), distCUDA2(ts).cpu().numpy().view(np.uint32))
