"""GPU parity of distCUDA2 (csrc/knn.hip) against the C oracle: bit-exact (shared d2 arithmetic,
the 3 smallest values are order-independent)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("no HIP device")


@pytest.mark.parametrize("n,seed", [(1, 0), (2, 0), (3, 1), (4, 1), (257, 2), (10_000, 3), (50_000, 4)])
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
