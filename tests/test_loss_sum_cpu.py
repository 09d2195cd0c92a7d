"""Host logic of the loss arithmetic and of the packed batch upload (loss_sum.weighted_sum's torch branch, dynamic_stage.upload_packed's
host branch): what runs when the tensors are not on the HIP device."""
import numpy as np
import torch


def test_weighted_sum_torch_branch_matches_the_written_expression():
    from dreammesh4d_amd.loss_sum import weighted_sum

    a = torch.tensor(0.25, requires_grad=True)
    b = torch.tensor([1.0, 2.0, 4.0], requires_grad=True)
    c = torch.tensor(3.0, dtype=torch.float64)
    out = weighted_sum([(5000.0, a), ((0.5, 0.25, 0.125), b), (2.0, c)])
    assert float(out) == 5000.0 * 0.25 + 0.5 + 0.5 + 0.5 + 6.0
    out.backward()
    assert float(a.grad) == 5000.0 and b.grad.tolist() == [0.5, 0.25, 0.125]
    assert weighted_sum([]) == 0.0


def test_upload_packed_host_branch_keeps_dtypes_shapes_and_values():
    from dreammesh4d_amd.dynamic_stage import upload_packed

    arrays = {"vm": np.arange(32, dtype=np.float32).reshape(2, 4, 4), "idx": np.asarray([3, 1], np.int64), "pos": np.asarray([-1, 0, 1], np.int32),
              "empty": np.zeros((0,), np.int64)}
    out = upload_packed(arrays, "cpu")
    assert set(out) == set(arrays)
    for k, v in arrays.items():
        assert tuple(out[k].shape) == v.shape and str(out[k].dtype) == "torch." + str(v.dtype)
        assert np.array_equal(out[k].numpy(), v)
