"""csrc/attention.hip (dm4d_attention_f16): the UNet's self-attention on the matrix cores against softmax(q k^T / sqrt(d)) v in
float32 (extern/ldm_zero123/modules/attention.py:152-194), for the head dimensions and token counts of the Zero123 UNet at batch 8
(8 heads of 40 / 80 / 160 channels over 1024 / 256 / 64 tokens) and a few more."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("B,L,H,D", [(8, 1024, 8, 40), (8, 256, 8, 80), (8, 64, 8, 160), (2, 128, 3, 64), (1, 192, 2, 40), (3, 64, 1, 80)])
def test_attention_matches_float32_reference(B, L, H, D):
    if not torch.cuda.is_available():
        pytest.skip("no HIP device")
    from dreammesh4d_amd import conv_mfma

    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(L + D)
    qkv = torch.randn(B, L, 3, H, D, generator=g).to(dev).half()
    qkv[:, :, 0] *= 2.0                      # (peaked rows: the running maximum has work to do)
    out = conv_mfma.attention_qkv(qkv)
    q, k, v = (qkv[:, :, i].float().transpose(1, 2) for i in range(3))          # [B, H, L, D]
    ref = torch.softmax(q @ k.transpose(-1, -2) * D ** -0.5, dim=-1) @ v
    ref = ref.transpose(1, 2).reshape(B, L, H * D)
    assert out.shape == ref.shape and out.dtype == torch.float16
    err = float((out.float() - ref).abs().max())
    assert err <= 4e-3 * max(1.0, float(ref.abs().max())), err                 # float16 probabilities and output


def test_attention_rejects_what_it_does_not_take():
    if not torch.cuda.is_available():
        pytest.skip("no HIP device")
    from dreammesh4d_amd import conv_mfma

    dev = torch.device("cuda:0")
    with pytest.raises(ValueError):
        conv_mfma.attention_qkv(torch.zeros(1, 64, 3, 2, 48, device=dev, dtype=torch.float16))      # head dimension
    with pytest.raises(ValueError):
        conv_mfma.attention_qkv(torch.zeros(1, 48, 3, 2, 40, device=dev, dtype=torch.float16))      # L % 64
