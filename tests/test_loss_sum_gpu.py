"""loss_sum (csrc/imagehead.hip: dm4d_weighted_sum*, dm4d_partial_sums) against the torch expressions it replaces
(system/sugar_4dgen.py:296-330, sugar_static.py:246-340: `loss = lambda_a * a + lambda_b * b + ...` on 0-dim tensors)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_weighted_sum_is_the_torch_expression_bit_for_bit():
    from dreammesh4d_amd.loss_sum import weighted_sum

    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(3)
    ws = [5000.0, 500.0, 0.1, 1.0, 10.0, 0.37, 1e-3]
    vals = (torch.randn(7, generator=g) * torch.tensor([1e-3, 1e-2, 1e3, 1.0, 1e-4, 7.0, 1e5])).tolist()
    a = [torch.tensor(v, device=dev, requires_grad=True) for v in vals]
    b = [torch.tensor(v, device=dev, requires_grad=True) for v in vals]
    ref = 0.0
    for w, t in zip(ws, a):
        ref = ref + w * t
    # scalars, and the middle five as ONE vector term
    vec = torch.stack([t.detach() for t in b[1:6]]).requires_grad_(True)
    got = weighted_sum([(ws[0], b[0]), (tuple(ws[1:6]), vec), (ws[6], b[6])])
    assert got.dtype == torch.float32 and got.dim() == 0
    assert torch.equal(got.detach(), ref.detach())
    up = torch.tensor(0.731, device=dev)
    (ref * up).backward()
    (got * up).backward()
    assert torch.equal(b[0].grad, a[0].grad) and torch.equal(b[6].grad, a[6].grad)
    assert torch.equal(vec.grad, torch.stack([t.grad for t in a[1:6]]))


def test_weighted_sum_takes_the_torch_expression_for_anything_else():
    from dreammesh4d_amd.loss_sum import weighted_sum

    a, b = torch.tensor(2.0, requires_grad=True), torch.tensor(3.0, dtype=torch.float64)
    out = weighted_sum([(0.5, a), (2.0, b)])           # CPU / float64: not the kernel's domain
    assert float(out) == 7.0
    out.backward()
    assert float(a.grad) == 0.5
    assert weighted_sum([]) == 0.0


def test_partial_sums_and_limits():
    from dreammesh4d_amd import _lib
    from dreammesh4d_amd.loss_sum import partial_sums

    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(5)
    for n, k, m in ((2048, 2, 2), (1280, 8, 5), (1, 3, 8), (0, 4, 1)):
        p = torch.rand(n, k, generator=g).to(dev)
        mat = torch.rand(k, m, generator=g, dtype=torch.float64)
        got = partial_sums(p, mat.tolist())
        ref = (p.double().sum(0) @ mat.to(dev)).float()
        assert torch.allclose(got, ref, rtol=2e-6, atol=1e-7), (n, k, m)
    with pytest.raises(_lib.Dm4dError):
        partial_sums(torch.zeros(4, 9, device=dev), [[1.0]] * 9)
    with pytest.raises(ValueError):
        partial_sums(torch.zeros(4, 2, device=dev, dtype=torch.float64), [[1.0], [1.0]])


def test_upload_packed_is_one_copy_with_the_arrays_dtypes():
    import numpy as np

    from dreammesh4d_amd.dynamic_stage import upload_packed

    arrays = {"vm": np.arange(128, dtype=np.float32).reshape(8, 4, 4), "t": np.asarray([0.1, 0.7, 0.3], np.float32), "idx": np.asarray([3, 1, 0, 2, 5], np.int64),
              "pos": np.asarray([-1, 0, 1], np.int32), "empty": np.zeros((0,), np.int64)}
    out = upload_packed(arrays, torch.device("cuda:0"))
    base = {t.untyped_storage().data_ptr() for t in out.values()}
    assert len(base) == 1                                   # views of ONE device buffer
    for k, v in arrays.items():
        assert out[k].is_cuda and tuple(out[k].shape) == v.shape and str(out[k].dtype) == "torch." + str(v.dtype)
        assert np.array_equal(out[k].cpu().numpy(), v)
        assert out[k].data_ptr() % 8 == 0


def test_weighted_sum_beyond_the_kernel_domain_and_c_abi_limits():
    import ctypes as C

    from dreammesh4d_amd import _lib
    from dreammesh4d_amd.loss_sum import weighted_sum

    dev = torch.device("cuda:0")
    ts = [torch.tensor(float(i + 1), device=dev, requires_grad=True) for i in range(17)]      # 17 terms: more than one launch takes
    out = weighted_sum([(0.5, t) for t in ts])
    assert float(out) == 0.5 * sum(range(1, 18))
    out.backward()
    assert all(float(t.grad) == 0.5 for t in ts)
    L = _lib.lib()
    w = (C.c_float * 17)(*([1.0] * 17))
    p = (C.c_void_p * 17)(*[t.data_ptr() for t in ts])
    o = torch.zeros((), device=dev)
    assert L.dm4d_weighted_sum(17, p, w, o.data_ptr(), 0) != 0          # n > 16: refused, with a message
    assert b"16" in L.dm4d_last_error()
    assert L.dm4d_weighted_sum(0, p, w, o.data_ptr(), 0) != 0
    assert L.dm4d_weighted_sum_backward(3, None, w, o.data_ptr(), 0) != 0
    big = torch.rand(200_000, 8, device=dev)
    from dreammesh4d_amd.loss_sum import partial_sums
    eye = [[1.0 if i == j else 0.0 for j in range(8)] for i in range(8)]
    assert torch.allclose(partial_sums(big, eye), big.double().sum(0).float(), rtol=3e-6)
