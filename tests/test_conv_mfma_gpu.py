"""csrc/conv_mfma.hip: the 3x3 / stride 1 / padding 1 NHWC float16 convolution of the Zero123 UNet and VAE encoder on MFMA
(threestudio/models/guidance/stable_zero123_guidance.py:222-233 runs ldm's UNet, whose ResBlocks and Up/Downsample are
torch Conv2d).  Checked against torch's convolution of the same float16 operands, accumulated in float32 -- the bar is the
float16 rounding of the OUTPUT (2^-10 relative to the largest partial sum scale), for every tile configuration the plan
can choose: the direct kernel (UNet 8^2..32^2), the split-K implicit GEMM (4^2) and the plain implicit GEMM (VAE)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

SHAPES = [  # N, H, W, Cin, Cout       (a subset of the 22 shapes of one SDS step, tools/conv_shapes.py has all)
    (8, 32, 32, 320, 320), (8, 32, 32, 640, 320), (8, 16, 16, 640, 640), (8, 16, 16, 1280, 640), (8, 8, 8, 1280, 1280),
    (8, 8, 8, 2560, 1280), (8, 4, 4, 1280, 1280), (2, 128, 128, 128, 128), (2, 64, 64, 128, 256), (1, 32, 32, 512, 512),
    (3, 16, 16, 64, 96), (1, 5, 7, 32, 32),       # ragged: M not a multiple of any tile, W not a power of two
    # the direct kernel (W >= 64) off its comfortable path: one channel chunk (no prefetch of a next patch), fewer filters than
    # a tile, a row count that is no multiple of the tile's 8 rows, tiles that straddle two images, H != W
    (1, 64, 64, 32, 64), (1, 20, 64, 64, 32), (3, 12, 64, 96, 160), (2, 8, 128, 64, 128),
    # round 6: grids that fill the machine with 512 x 128 tiles (the plan's ping-pong kernel with the 512-pixel tile): the VAE encoder's 256^2 level,
    # and a ragged one (1000 image rows: no multiple of the tile's 16, tiles straddle images, a partial filter tile, H != W)
    (4, 256, 256, 128, 128), (5, 200, 128, 64, 160),
]


def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("needs a HIP device")


def _ref(x, w, bias, res):
    y = F.conv2d(x.float(), w.float(), None if bias is None else bias.float(), 1, 1)
    return y if res is None else y + res.float()


@pytest.mark.parametrize("shape", SHAPES)
@pytest.mark.parametrize("epilogue", ["plain", "bias+residual"])
def test_conv3x3_matches_float32_accumulated_reference(shape, epilogue):
    _need_gpu()
    from dreammesh4d_amd import conv_mfma

    N, H, W, Ci, Co = shape
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(Ci + Co + H)
    x = torch.randn(N, Ci, H, W, generator=g).to(dev).half().contiguous(memory_format=torch.channels_last)
    w = (torch.randn(Co, Ci, 3, 3, generator=g) * (9 * Ci) ** -0.5).to(dev).half()
    bias = res = None
    if epilogue != "plain":
        bias = torch.randn(Co, generator=g).to(dev).half()
        res = torch.randn(N, Co, H, W, generator=g).to(dev).half().contiguous(memory_format=torch.channels_last)
    assert conv_mfma.supported(x, w)
    y = conv_mfma.conv3x3(x, conv_mfma.pack_weight(w), bias, res)
    assert y.shape == (N, Co, H, W) and y.dtype == torch.float16 and y.is_contiguous(memory_format=torch.channels_last)
    ref = _ref(x, w, bias, res)
    err = (y.float() - ref).abs().max().item()
    assert err <= 2 ** -10 * ref.abs().max().item() + 1e-6, (err, ref.abs().max().item())
    assert torch.equal(y, conv_mfma.conv3x3(x, conv_mfma.pack_weight(w), bias, res))       # fixed reduction order


@pytest.mark.parametrize("cfg", [13, 14])
@pytest.mark.parametrize("shape", [(1, 20, 64, 64, 32), (3, 12, 64, 96, 160), (2, 8, 128, 64, 128), (1, 32, 32, 512, 512), (2, 40, 32, 32, 64)])
def test_conv3x3_ping_pong_tiles_forced(shape, cfg, monkeypatch):
    """The two ping-pong direct kernels (256- and 512-pixel tiles, csrc/conv_mfma.hip cfg 13 / 14) FORCED onto small and ragged shapes the plan
    would give to other kernels: one channel chunk, fewer filters than a tile, rows that are no multiple of the tile's, tiles that straddle
    images, most of a 512-pixel tile past the tensor's end."""
    _need_gpu()
    from dreammesh4d_amd import conv_mfma

    N, H, W, Ci, Co = shape
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(Ci + Co + H + cfg)
    x = torch.randn(N, Ci, H, W, generator=g).to(dev).half().contiguous(memory_format=torch.channels_last)
    w = (torch.randn(Co, Ci, 3, 3, generator=g) * (9 * Ci) ** -0.5).to(dev).half()
    bias = torch.randn(Co, generator=g).to(dev).half()
    res = torch.randn(N, Co, H, W, generator=g).to(dev).half().contiguous(memory_format=torch.channels_last)
    monkeypatch.setenv("DM4D_CONV_CFG", str(cfg))
    monkeypatch.setenv("DM4D_CONV_SPLITS", "1")
    y = conv_mfma.conv3x3(x, conv_mfma.pack_weight(w), bias, res)
    ref = _ref(x, w, bias, res)
    err = (y.float() - ref).abs().max().item()
    assert err <= 2 ** -10 * ref.abs().max().item() + 1e-6, (err, ref.abs().max().item())
    monkeypatch.setenv("DM4D_CONV_CFG", "7")          # the lock-step direct kernel: the same tile algebra and k order -> the same bits
    assert torch.equal(y, conv_mfma.conv3x3(x, conv_mfma.pack_weight(w), bias, res))


def test_frozen_conv_data_gradient_and_residual_gradient():
    _need_gpu()
    from dreammesh4d_amd import conv_mfma

    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(5)
    N, H, W, Ci, Co = 2, 24, 20, 64, 128
    x = torch.randn(N, Ci, H, W, generator=g).to(dev).half().contiguous(memory_format=torch.channels_last).requires_grad_(True)
    r = torch.randn(N, Co, H, W, generator=g).to(dev).half().contiguous(memory_format=torch.channels_last).requires_grad_(True)
    w = (torch.randn(Co, Ci, 3, 3, generator=g) * (9 * Ci) ** -0.5).to(dev).half()
    b = torch.randn(Co, generator=g).to(dev).half()
    gy = torch.randn(N, Co, H, W, generator=g).to(dev).half().contiguous(memory_format=torch.channels_last)
    y = conv_mfma.conv3x3_frozen(x, conv_mfma.pack_weight(w), conv_mfma.pack_weight_transposed(w), b, r)
    y.backward(gy)
    x32, r32 = x.detach().float().requires_grad_(True), r.detach().float().requires_grad_(True)
    (F.conv2d(x32, w.float(), b.float(), 1, 1) + r32).backward(gy.float())
    assert (x.grad.float() - x32.grad).abs().max() <= 2 ** -10 * x32.grad.abs().max() + 1e-6
    assert torch.equal(r.grad, gy)


def test_unsupported_operands_are_rejected_not_miscomputed():
    _need_gpu()
    from dreammesh4d_amd import conv_mfma

    dev = torch.device("cuda:0")
    w = torch.zeros(32, 32, 3, 3, device=dev, dtype=torch.float16)
    x = torch.zeros(1, 32, 8, 8, device=dev, dtype=torch.float16)
    assert not conv_mfma.supported(x, w)                                                     # NCHW-contiguous
    assert not conv_mfma.supported(x.float().contiguous(memory_format=torch.channels_last), w.float())
    assert not conv_mfma.supported(torch.zeros(1, 48, 8, 8, device=dev, dtype=torch.float16).contiguous(memory_format=torch.channels_last),
                                   torch.zeros(32, 48, 3, 3, device=dev, dtype=torch.float16))
    with pytest.raises((ValueError, RuntimeError)):
        conv_mfma.conv3x3(x, conv_mfma.pack_weight(w))


def test_first_convolution_data_gradient_on_the_small_cout_kernel():
    """128 -> 3 channels (the image gradient of the VAE encoder's conv_in): against torch's float32 autograd of the same operands."""
    _need_gpu()
    from dreammesh4d_amd import conv_mfma

    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(9)
    for (N, H, W) in ((2, 64, 48), (1, 17, 9)):
        x = torch.rand(N, 3, H, W, generator=g).to(dev).half().contiguous(memory_format=torch.channels_last).requires_grad_(True)
        w = (torch.randn(128, 3, 3, 3, generator=g) * 0.2).to(dev).half()
        b = torch.randn(128, generator=g).to(dev).half()
        gy = torch.randn(N, 128, H, W, generator=g).to(dev).half().contiguous(memory_format=torch.channels_last)
        assert conv_mfma.first_conv_supported(x, w)
        y = conv_mfma.conv3x3_first_frozen(x, w, b, conv_mfma.pack_weight_transposed(w))
        y.backward(gy)
        x32 = x.detach().float().requires_grad_(True)
        y32 = F.conv2d(x32, w.float(), b.float(), 1, 1)
        y32.backward(gy.float())
        assert (y.float() - y32).abs().max() <= 2 ** -9 * y32.abs().max()
        assert x.grad.shape == x.shape and x.grad.is_contiguous(memory_format=torch.channels_last)
        assert (x.grad.float() - x32.grad).abs().max() <= 2 ** -10 * x32.grad.abs().max() + 1e-6


@pytest.mark.parametrize("pad", [0, 1])
@pytest.mark.parametrize("shape", [(2, 32, 32, 64, 64), (1, 64, 48, 128, 96), (3, 16, 16, 320, 320), (2, 9, 14, 32, 32)])
def test_stride2_convolution_and_its_frozen_autograd_form(shape, pad):
    """Stride 2 with pad 1 (the UNet's Downsample) and with pad 0 + one zero behind each axis (the VAE encoder's: F.pad(x, (0, 1, 0, 1))
    + an unpadded convolution) against torch in float32; the data gradient of the autograd form against torch's."""
    _need_gpu()
    from dreammesh4d_amd import conv_mfma

    N, H, W, Ci, Co = shape
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(H * 7 + pad)
    x = torch.randn(N, Ci, H, W, generator=g).to(dev).half().contiguous(memory_format=torch.channels_last).requires_grad_(True)
    w = (torch.randn(Co, Ci, 3, 3, generator=g) * (9 * Ci) ** -0.5).to(dev).half()
    b = torch.randn(Co, generator=g).to(dev).half()
    ref_in = x.detach().float().requires_grad_(True)
    ref = F.conv2d(ref_in, w.float(), b.float(), 2, 1) if pad else F.conv2d(F.pad(ref_in, (0, 1, 0, 1)), w.float(), b.float(), 2, 0)
    y = conv_mfma.conv3x3_stride2_frozen(x, w, conv_mfma.pack_weight(w), b, pad)
    assert y.shape == ref.shape and y.is_contiguous(memory_format=torch.channels_last)
    assert (y.float() - ref).abs().max() <= 2 ** -10 * ref.abs().max() + 1e-6
    with torch.no_grad():
        assert torch.equal(y, conv_mfma.conv3x3(x.detach(), conv_mfma.pack_weight(w), b, None, stride=2, pad=pad))
    gy = torch.randn(ref.shape, generator=g).to(dev).half().contiguous(memory_format=torch.channels_last)
    y.backward(gy)
    ref.backward(gy.float())
    assert x.grad.shape == x.shape
    assert (x.grad.float() - ref_in.grad).abs().max() <= 2 ** -9 * ref_in.grad.abs().max() + 1e-6


@pytest.mark.parametrize("shape", [(2, 32, 32, 64, 64), (1, 64, 48, 128, 96), (3, 16, 16, 320, 320), (2, 10, 14, 32, 64), (4, 256, 256, 128, 128)])
def test_stride2_data_gradient_on_the_mfma_kernel(shape):
    """dm4d_conv3x3_s2_dgrad_nhwc_f16 (the VAE Downsample's data gradient as four stride-1 convolutions of dy, one per parity class
    of the input pixels) against torch's gradient of F.conv2d(F.pad(x, (0, 1, 0, 1)), w, stride 2) in float32; and through the
    autograd form, which must take this path when handed the packed filters."""
    _need_gpu()
    from dreammesh4d_amd import conv_mfma

    N, H, W, Ci, Co = shape
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(H * 5 + Ci)
    w = (torch.randn(Co, Ci, 3, 3, generator=g) * (9 * Ci) ** -0.5).to(dev).half()
    gy = torch.randn(N, Co, H // 2, W // 2, generator=g).to(dev).half().contiguous(memory_format=torch.channels_last)
    ref = torch.nn.grad.conv2d_input((N, Ci, H + 1, W + 1), w.float(), gy.float(), stride=2, padding=0)[:, :, :H, :W]
    w_cls = conv_mfma.pack_weight_s2_dgrad(w)
    assert [tuple(t.shape) for t in w_cls] == [(Ci, 2, 2, Co), (Ci, 2, 1, Co), (Ci, 1, 2, Co), (Ci, 1, 1, Co)]
    dx = conv_mfma.conv3x3_s2_dgrad(gy, w_cls, (N, Ci, H, W))
    assert dx.shape == (N, Ci, H, W) and dx.is_contiguous(memory_format=torch.channels_last)
    assert (dx.float() - ref).abs().max() <= 2 ** -9 * ref.abs().max() + 1e-6
    if N * H * W <= 1 << 16:
        x = torch.randn(N, Ci, H, W, generator=g).to(dev).half().contiguous(memory_format=torch.channels_last).requires_grad_(True)
        before = conv_mfma.FLOPS[0]
        y = conv_mfma.conv3x3_stride2_frozen(x, w, conv_mfma.pack_weight(w), None, 0, w_cls)
        y.backward(gy)
        assert conv_mfma.FLOPS[0] - before == 2 * (2 * N * (H // 2) * (W // 2) * Ci * Co * 9)      # forward AND data gradient were tallied: both ran here
        assert torch.equal(x.grad, dx)


LINEAR_SHAPES = [      # (M, K, N): the UNet's token counts x widths, tails, a split-K shape, the 64 x 64 tiles
    (8192, 320, 320), (8192, 320, 960), (8192, 1280, 320), (2048, 640, 1920), (512, 1280, 1280), (512, 5120, 1280), (128, 1280, 1280),
    (200, 64, 72), (1, 32, 8), (333, 96, 200), (130, 3840, 64),
]


@pytest.mark.parametrize("M,K,N", LINEAR_SHAPES)
@pytest.mark.parametrize("mode", ["plain", "bias_residual"])
def test_linear_matches_float32_reference(M, K, N, mode):
    """dm4d_linear_f16 against float32 matmul of the same float16 operands: the error is the float16 rounding of the result
    (+ of the residual sum) and the float32 accumulation order, nothing that grows with K."""
    _need_gpu()
    dev = torch.device("cuda:0")
    from dreammesh4d_amd import conv_mfma

    g = torch.Generator().manual_seed(M + 7 * K + N)
    x = torch.randn(M, K, generator=g).to(dev, torch.float16)
    w = (torch.randn(N, K, generator=g) / K ** 0.5).to(dev, torch.float16)
    b = torch.randn(N, generator=g).to(dev, torch.float16) if mode != "plain" else None
    r = torch.randn(M, N, generator=g).to(dev, torch.float16) if mode != "plain" else None
    y = conv_mfma.linear(x, w, b, r)
    ref = x.float() @ w.float().t()
    if b is not None:
        ref = (ref + b.float()).half().float() + r.float()
    assert y.shape == (M, N) and y.dtype == torch.float16
    err = float((y.float() - ref).abs().max())
    assert err <= 2e-3 * max(1.0, float(ref.abs().max())), err


@pytest.mark.parametrize("M,K,D", [(8192, 320, 1280), (2048, 640, 2560), (512, 1280, 5120), (100, 64, 64)])
def test_linear_geglu_matches_reference(M, K, D):
    """GEGLU in the epilogue (attention.py:48-56: value, gate = proj(x).chunk(2); value * gelu(gate)) with the rows interleaved by
    pack_geglu: the same as the float16 projection followed by the separate GEGLU."""
    _need_gpu()
    dev = torch.device("cuda:0")
    from dreammesh4d_amd import conv_mfma

    g = torch.Generator().manual_seed(M + K + D)
    x = torch.randn(M, K, generator=g).to(dev, torch.float16)
    w = (torch.randn(2 * D, K, generator=g) / K ** 0.5).to(dev, torch.float16)
    b = torch.randn(2 * D, generator=g).to(dev, torch.float16)
    wp, bp = conv_mfma.pack_geglu(w, b)
    y = conv_mfma.linear(x, wp, bp, act="geglu")
    proj = (x.float() @ w.float().t() + b.float()).half().float()
    val, gate = proj.chunk(2, dim=-1)
    ref = val * torch.nn.functional.gelu(gate)
    assert y.shape == (M, D)
    err = float((y.float() - ref).abs().max())
    assert err <= 4e-3 * max(1.0, float(ref.abs().max())), err


def test_linear_rejects_what_it_does_not_support():
    _need_gpu()
    dev = torch.device("cuda:0")
    from dreammesh4d_amd import conv_mfma

    x = torch.zeros(4, 48, device=dev, dtype=torch.float16)
    with pytest.raises(ValueError):
        conv_mfma.linear(x, torch.zeros(8, 48, device=dev, dtype=torch.float16))          # K % 32
    with pytest.raises(ValueError):
        conv_mfma.linear(torch.zeros(4, 64, device=dev, dtype=torch.float16), torch.zeros(12, 64, device=dev, dtype=torch.float16))   # N % 8
    with pytest.raises(ValueError):
        conv_mfma.linear(torch.zeros(4, 64, device=dev, dtype=torch.float16).t().contiguous().t(), torch.zeros(8, 64, device=dev, dtype=torch.float16))


@pytest.mark.parametrize("N,H,W,Ci,Co,grad", [(2, 32, 32, 8, 320, False), (2, 32, 32, 320, 4, False), (2, 16, 24, 512, 8, True), (1, 64, 64, 3, 128, False)])
def test_narrow_convolutions_on_zero_padded_channels(N, H, W, Ci, Co, grad):
    """The convolutions at the ends of the UNet (8 -> 320, 320 -> 4) and of the VAE encoder (3 -> 128, 512 -> 8) through
    zero123._conv3x3: the MFMA kernel on channels zero-padded to 32, sliced back -- against torch in float32; with a frozen filter
    and a gradient on the input, the data gradient too (the VAE's conv_out: the image is differentiated through it)."""
    _need_gpu()
    from dreammesh4d_amd import conv_mfma, zero123 as z

    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(Ci * 3 + Co)
    conv = torch.nn.Conv2d(Ci, Co, 3, padding=1).to(dev, torch.float16).requires_grad_(False)
    with torch.no_grad():
        conv.weight.copy_((torch.randn(Co, Ci, 3, 3, generator=g) * (9 * Ci) ** -0.5).to(dev))
        conv.bias.copy_(torch.randn(Co, generator=g).to(dev))
    x = torch.randn(N, Ci, H, W, generator=g).to(dev).half().contiguous(memory_format=torch.channels_last).requires_grad_(grad)
    before = conv_mfma.FLOPS[0]
    with torch.enable_grad() if grad else torch.no_grad():
        y = z._conv3x3(conv, x)
    assert conv_mfma.FLOPS[0] > before, "the convolution did not run on the MFMA kernel"
    x32 = x.detach().float().requires_grad_(grad)
    ref = F.conv2d(x32, conv.weight.float(), conv.bias.float(), 1, 1)
    assert y.shape == ref.shape
    assert (y.float() - ref).abs().max() <= 2 ** -9 * ref.abs().max() + 1e-6
    if grad:
        gy = torch.randn(ref.shape, generator=g).to(dev).half()
        y.backward(gy)
        ref.backward(gy.float())
        assert (x.grad.float() - x32.grad).abs().max() <= 2 ** -9 * x32.grad.abs().max() + 1e-6
