"""conv3x3_bias_relu (fp32 MFMA) vs torch conv2d on the CPU (fp32)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

CASES = [(2, 64, 64, 32, 128), (1, 64, 128, 8, 64), (2, 8, 64, 4, 32), (1, 128, 128, 64, 64), (1, 384, 64, 16, 128)]


@pytest.mark.parametrize("n,cin,cout,h,w", CASES)
@pytest.mark.parametrize("relu", [True, False])
def test_conv3x3_matches_torch(n, cin, cout, h, w, relu):
    from paddle3d_amd.ops import conv

    g = torch.Generator().manual_seed(cin * 7 + cout)
    x = torch.randn(n, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, 3, 3, generator=g) / (cin * 9) ** 0.5
    b = torch.randn(cout, generator=g)
    ref = F.conv2d(x, wt, b, padding=1)
    if relu:
        ref = torch.relu(ref)
    out = conv.conv3x3_bias_relu(x.cuda(), conv.pack_conv3x3_weight(wt.cuda()), b.cuda(), cout, relu).cpu()
    assert out.shape == ref.shape
    assert (out - ref).abs().max().item() < 2e-4, (out - ref).abs().max().item()


def test_asymmetric_identity():
    """A = I against an asymmetric input catches a transposed / mis-rowed accumulator layout."""
    from paddle3d_amd.ops import conv

    cin = cout = 64
    wt = torch.zeros(cout, cin, 3, 3)
    wt[torch.arange(64), torch.arange(64), 1, 1] = 1.0  # identity, centre tap
    x = torch.arange(1 * 64 * 4 * 128, dtype=torch.float32).reshape(1, 64, 4, 128) / 1000.0
    out = conv.conv3x3_bias_relu(x.cuda(), conv.pack_conv3x3_weight(wt.cuda()), None, cout, False).cpu()
    assert torch.equal(out, x)
    # shift taps: output = input shifted by one pixel in x (zero padding at the border)
    wt2 = torch.zeros(cout, cin, 3, 3)
    wt2[torch.arange(64), torch.arange(64), 1, 0] = 1.0
    out2 = conv.conv3x3_bias_relu(x.cuda(), conv.pack_conv3x3_weight(wt2.cuda()), None, cout, False).cpu()
    want = torch.zeros_like(x)
    want[..., 1:] = x[..., :-1]
    assert torch.equal(out2, want)


@pytest.mark.parametrize("n,cin,cout,h,w", [(2, 64, 64, 8, 128), (1, 64, 128, 16, 64), (1, 128, 256, 64, 128),
                                            (1, 8, 64, 4, 256)])
def test_conv3x3_stride2_matches_torch(n, cin, cout, h, w):
    """Stride-2 variant (the first convolution of every SECOND block, second_backbone.py:90-97)."""
    from paddle3d_amd.ops import conv

    g = torch.Generator().manual_seed(cin * 3 + cout + h)
    x = torch.randn(n, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, 3, 3, generator=g) / (cin * 9) ** 0.5
    b = torch.randn(cout, generator=g)
    ref = torch.relu(F.conv2d(x, wt, b, stride=2, padding=1))
    assert conv.supported(cin, cout, h, w, 2)
    out = conv.conv3x3_bias_relu(x.cuda(), conv.pack_conv3x3_weight(wt.cuda()), b.cuda(), cout, True, stride=2).cpu()
    assert out.shape == ref.shape
    assert (out - ref).abs().max().item() < 2e-4, (out - ref).abs().max().item()


@pytest.mark.parametrize("co,h,w", [(1, 16, 128), (2, 16, 128), (3, 16, 128), (4, 16, 128), (3, 45, 180), (2, 5, 12)])
def test_grouped_conv3x3_small_matches_torch(co, h, w):
    from paddle3d_amd.ops import conv

    groups, cg, n = 5, 64, 2
    g = torch.Generator().manual_seed(co)
    x = torch.randn(n, groups * cg, h, w, generator=g)
    wt = torch.randn(groups * co, cg, 3, 3, generator=g) / (cg * 9) ** 0.5
    b = torch.randn(groups * co, generator=g)
    ref = F.conv2d(x, wt, b, padding=1, groups=groups)
    assert conv.grouped_small_supported(cg, co, h, w)
    out = conv.grouped_conv3x3_small(x.cuda(), conv.pack_grouped_weight(wt.cuda(), groups), b.cuda(), groups).cpu()
    assert out.shape == ref.shape
    assert (out - ref).abs().max().item() < 2e-4, (out - ref).abs().max().item()


def test_unsupported_shapes_are_refused():
    from paddle3d_amd.ops import conv

    wp = torch.zeros(1, 1, 72, 64).cuda()
    with pytest.raises(RuntimeError):  # rows that are not float4-aligned
        conv.conv3x3_bias_relu(torch.randn(1, 8, 6, 46).cuda(), wp, None, 64, True)
    with pytest.raises(RuntimeError):  # odd input height at stride 2
        conv.conv3x3_bias_relu(torch.randn(1, 8, 7, 48).cuda(), wp, None, 64, True, stride=2)
    assert not conv.supported(8, 64, 7, 48, 2) and not conv.supported(4, 64, 8, 48)


@pytest.mark.parametrize("n,cin,cout,h,w,stride", [(1, 8, 64, 6, 48, 1), (2, 16, 64, 45, 180, 1), (1, 64, 128, 90, 180, 2),
                                                   (2, 128, 256, 180, 180, 2), (1, 8, 64, 10, 36, 2), (1, 8, 64, 5, 12, 1)])
def test_conv3x3_partial_tiles_match_torch(n, cin, cout, h, w, stride):
    """Maps that are not a multiple of the 4 x 32 pixel tile: partial border tiles with masked stores (the
    CenterPoint-Voxel stride-2 layer 180 x 180 -> 90 x 90 is one of them)."""
    from paddle3d_amd.ops import conv

    g = torch.Generator().manual_seed(cin + cout + h + w)
    x = torch.randn(n, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, 3, 3, generator=g) / (cin * 9) ** 0.5
    b = torch.randn(cout, generator=g)
    ref = torch.relu(F.conv2d(x, wt, b, stride=stride, padding=1))
    assert conv.supported(cin, cout, h, w, stride)
    out = conv.conv3x3_bias_relu(x.cuda(), conv.pack_conv3x3_weight(wt.cuda()), b.cuda(), cout, True, stride=stride).cpu()
    wo = ref.shape[3]
    assert out.shape == ref.shape[:3] + (conv.pitch4(wo),)  # rows zero-padded to a multiple of 4
    assert (out[..., :wo] - ref).abs().max().item() < 2e-4, (out[..., :wo] - ref).abs().max().item()
    assert not out[..., wo:].any()


def test_conv_chain_on_padded_rows_matches_torch():
    """A 90-wide stage (CenterPoint-Voxel block 2) as the model runs it: stride-2 conv into zero-padded rows of
    pitch 92, two Winograd convolutions and the transposed FPN convolution reading them through w_valid."""
    from paddle3d_amd.ops import conv

    g = torch.Generator().manual_seed(3)
    x = torch.randn(1, 16, 20, 180, generator=g)
    w0 = torch.randn(64, 16, 3, 3, generator=g) / 12
    w1 = torch.randn(64, 64, 3, 3, generator=g) / 24
    wt = torch.randn(64, 32, 2, 2, generator=g) / 8
    b = torch.randn(64, generator=g) * 0.1
    bt = torch.randn(32, generator=g) * 0.1
    ref = torch.relu(F.conv2d(x, w0, b, stride=2, padding=1))
    ref = torch.relu(F.conv2d(ref, w1, b, padding=1))
    ref = torch.relu(F.conv2d(ref, w1, b, padding=1))
    up = torch.relu(F.conv_transpose2d(ref, wt, bt, stride=2))
    y = conv.conv3x3_bias_relu(x.cuda(), conv.pack_conv3x3_weight(w0.cuda()), b.cuda(), 64, True, stride=2)
    assert y.shape == (1, 64, 10, 92)
    u = conv.pack_winograd43_weight(w1.cuda())
    y = conv.conv3x3_winograd43_bias_relu(y, u, b.cuda(), 64, True, w_valid=90)
    y = conv.conv3x3_winograd43_bias_relu(y, u, b.cuda(), 64, True, w_valid=90)
    assert (y.cpu()[..., :90] - ref).abs().max().item() < 5e-4 and not y[..., 90:].any()
    out = torch.full((1, 40, 20, 180), -7.0, device="cuda")
    conv.patch_conv_bias_relu(y, conv.pack_patch_weight(wt.cuda(), 2, True), bt.cuda(), 2, 32, out, 4, w_valid=90)
    got = out.cpu()
    assert (got[:, 4:36] - up).abs().max().item() < 5e-4
    assert (got[:, :4] == -7.0).all() and (got[:, 36:] == -7.0).all()


@pytest.mark.parametrize("n,cin,cout,h,w", [(2, 64, 64, 32, 128), (1, 8, 32, 8, 32), (1, 128, 96, 16, 64),
                                            (1, 384, 64, 24, 96), (3, 16, 32, 8, 32), (1, 24, 32, 8, 64), (2, 16, 64, 45, 180),
                                            (1, 8, 32, 5, 12), (1, 40, 32, 62, 124)])
@pytest.mark.parametrize("relu", [True, False])
def test_conv3x3_winograd_matches_torch(n, cin, cout, h, w, relu):
    """Winograd F(2x2,3x3) on the fp32 matrix cores vs torch conv2d on the CPU."""
    from paddle3d_amd.ops import conv

    g = torch.Generator().manual_seed(cin * 11 + cout + h)
    x = torch.randn(n, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, 3, 3, generator=g) / (cin * 9) ** 0.5
    b = torch.randn(cout, generator=g)
    ref = F.conv2d(x, wt, b, padding=1)
    if relu:
        ref = torch.relu(ref)
    assert conv.winograd_supported(cin, cout, h, w)
    out = conv.conv3x3_winograd_bias_relu(x.cuda(), conv.pack_winograd_weight(wt.cuda()), b.cuda(), cout, relu).cpu()
    assert out.shape == ref.shape
    assert (out - ref).abs().max().item() < 2e-4, (out - ref).abs().max().item()


def test_winograd_shift_taps():
    """Single-tap weights against an asymmetric input: catches transposed transforms / tile mis-addressing."""
    from paddle3d_amd.ops import conv

    cin = cout = 32
    x = torch.arange(1 * cin * 8 * 64, dtype=torch.float32).reshape(1, cin, 8, 64) / 512.0
    for ky in range(3):
        for kx in range(3):
            wt = torch.zeros(cout, cin, 3, 3)
            wt[torch.arange(cout), torch.arange(cin), ky, kx] = 1.0
            out = conv.conv3x3_winograd_bias_relu(x.cuda(), conv.pack_winograd_weight(wt.cuda()), None, cout, False).cpu()
            want = F.conv2d(x, wt, None, padding=1)
            assert (out - want).abs().max().item() < 1e-3, (ky, kx)


@pytest.mark.parametrize("mode", [0, 1, 2, 11, 12, 10])
def test_patch_conv_matches_torch(mode):
    """FPN patch convolutions (second_fpn.py:99-157) written at a channel offset of a wider tensor; modes 10 / 11 / 12
    are modes 0 / 1 / 2 on planes that are not a multiple of the 256-pixel tile (CenterPoint-Voxel: 180 x 180, 90 x 90)."""
    from paddle3d_amd.ops import conv

    g = torch.Generator().manual_seed(mode)
    n, off, ctot = 2, 64, 256
    odd = mode > 2
    mode = mode % 10
    if mode == 0:  # mode 10: a width that is not a multiple of the 2 x 128 output tile (CenterPoint-KITTI: 496 x 432)
        cin, cout, h, w = (16, 64, 12, 432) if odd else (16, 128, 8, 256)
        wt = torch.randn(cout, cin, 2, 2, generator=g) / (cin * 4) ** 0.5
        x = torch.randn(n, cin, h, w, generator=g)
        b = torch.randn(cout, generator=g)
        ref = torch.relu(F.conv2d(x, wt, b, stride=2))
        tr = False
    elif mode == 1:
        cin, cout, h, w = (32, 64, 45, 180) if odd else (32, 64, 16, 32)
        wt = torch.randn(cin, cout, 1, 1, generator=g) / cin ** 0.5
        x = torch.randn(n, cin, h, w, generator=g)
        b = torch.randn(cout, generator=g)
        ref = torch.relu(F.conv_transpose2d(x, wt, b, stride=1))
        tr = True
    else:
        cin, cout, h, w = (48, 128, 90, 90) if odd else (48, 128, 16, 16)
        wt = torch.randn(cin, cout, 2, 2, generator=g) / cin ** 0.5
        x = torch.randn(n, cin, h, w, generator=g)
        b = torch.randn(cout, generator=g)
        ref = torch.relu(F.conv_transpose2d(x, wt, b, stride=2))
        tr = True
    assert conv.patch_mode(wt, 1 if mode == 1 else 2, tr) == mode
    assert conv.patch_supported(mode, cin, cout, h, w)
    out = torch.full((n, ctot, ref.shape[2], ref.shape[3]), -7.0, device="cuda")
    conv.patch_conv_bias_relu(x.cuda(), conv.pack_patch_weight(wt.cuda(), mode, tr), b.cuda(), mode, cout, out, off)
    got = out.cpu()
    assert (got[:, off:off + cout] - ref).abs().max().item() < 2e-4
    assert (got[:, :off] == -7.0).all() and (got[:, off + cout:] == -7.0).all()


@pytest.mark.parametrize("mode,n,cin,cout,h,w,wv", [
    (0, 2, 64, 128, 32, 64, 64),      # CenterPoint's first level in small: two steps of dy x four 16-channel chunks
    (0, 3, 16, 256, 12, 64, 64),      # two row tiles, an output plane (6 x 32) below one pixel tile
    (0, 1, 32, 128, 20, 192, 192),    # planes that are no multiple of the 256-pixel tile (10 x 96 output pixels)
    (1, 2, 128, 128, 16, 48, 48),
    (1, 1, 32, 256, 45, 180, 180),    # ONE step per item: every step crosses an item boundary
    (1, 9, 64, 128, 7, 12, 12),       # nine tiny planes: 9 pixel tiles -> a partial last group of 8
    (2, 2, 256, 128, 16, 32, 32),     # CenterPoint's third level in small: (dy, 64-channel block) = 4 row tiles
    (2, 1, 64, 64, 45, 92, 90),       # row pitch 92, real width 90 (CenterPoint-Voxel)
    (2, 3, 32, 192, 9, 12, 12),
])
@pytest.mark.parametrize("relu", [True, False])
def test_patch_conv_bf16x3_is_fp32_arithmetic(mode, n, cin, cout, h, w, wv, relu):
    """SecondFPN's levels on the bf16 matrix cores with three pieces per operand (csrc/conv_patch_x3.hip): against the
    float64 layer its error is that of the fp32-MFMA kernel (<= 2x, and < 2e-6 of the magnitude -- a 16-bit operand format
    would leave 2e-4), values spread over e^+-3; channels outside [off, off + cout) are not touched."""
    from paddle3d_amd.ops import conv

    g = torch.Generator().manual_seed(mode * 1000 + cin + cout + h)
    off, ctot = 32, cout + 96
    x = torch.randn(n, cin, h, w, generator=g) * torch.exp(3 * (2 * torch.rand(n, cin, h, w, generator=g) - 1))
    if wv < w:
        x[..., wv:] = 0
    b = torch.randn(cout, generator=g)
    tr = mode == 2
    if mode == 0:
        wt = torch.randn(cout, cin, 2, 2, generator=g) / (cin * 4) ** 0.5
        ref = F.conv2d(x.double(), wt.double(), b.double(), stride=2)
    elif mode == 1:
        wt = torch.randn(cout, cin, 1, 1, generator=g) / cin ** 0.5
        ref = F.conv2d(x.double(), wt.double(), b.double())
    else:
        wt = torch.randn(cin, cout, 2, 2, generator=g) / cin ** 0.5
        ref = F.conv_transpose2d(x[..., :wv].double(), wt.double(), b.double(), stride=2)
    if relu:
        ref = torch.relu(ref)
    assert conv.patch_x3_supported(mode, cin, cout, h, w)
    wp = conv.pack_patch_weight_x3(wt.cuda(), mode, tr)
    steps = {0: 2 * (cin // 16), 1: cin // 32, 2: cin // 32}[mode]
    assert wp.dtype == torch.bfloat16 and tuple(wp.shape[1:]) == (steps, 16384)
    out = torch.full((n, ctot, ref.shape[2], ref.shape[3]), -7.0, device="cuda")
    conv.patch_conv_x3_bias_relu(x.cuda(), wp, b.cuda(), mode, cout, out, off, relu=relu, w_valid=wv)
    got = out.cpu()
    assert (got[:, :off] == -7.0).all() and (got[:, off + cout:] == -7.0).all()
    err = (got[:, off:off + cout].double() - ref).abs().max().item()
    out32 = torch.full_like(out, -7.0)
    conv.patch_conv_bias_relu(x.cuda(), conv.pack_patch_weight(wt.cuda(), mode, tr), b.cuda(), mode, cout, out32, off,
                              relu=relu, w_valid=wv)
    err32 = (out32.cpu()[:, off:off + cout].double() - ref).abs().max().item()
    mag = ref.abs().max().item()
    print(f"mode {mode}: bf16x3 {err:.3e}, fp32 kernel {err32:.3e}, magnitude {mag:.1f}")
    assert err <= 2 * err32 + 1e-7 * mag and err < 2e-6 * mag
    out2 = torch.full_like(out, -7.0)
    conv.patch_conv_x3_bias_relu(x.cuda(), wp, b.cuda(), mode, cout, out2, off, relu=relu, w_valid=wv)
    assert torch.equal(out, out2)  # run-to-run identical


@pytest.mark.parametrize("n,cin,cout,h,w", [
    (2, 64, 128, 32, 64),      # CenterPoint's first opener in small: 12 steps, one channel tile
    (1, 128, 256, 16, 128),    # the second opener: 24 steps, two channel tiles, two column tiles (left column from the map)
    (3, 16, 128, 12, 64),      # 6 output rows: two of a tile's eight waves own no row
    (1, 32, 384, 36, 192),     # 18 output rows (partial third row tile), three column tiles, three channel tiles
    (9, 16, 128, 2, 64),       # nine images of ONE output row: a partial last group of 8 tiles
    (2, 32, 128, 180, 180),    # CenterPoint-Voxel's opener in small: 90 output columns at pitch 92, partial column tile
    (1, 16, 256, 14, 90),      # an input of pitch 92 with 90 real columns -> 45 real output columns at pitch 48
])
@pytest.mark.parametrize("relu", [True, False])
def test_conv3x3_s2_bf16x3_is_fp32_arithmetic(n, cin, cout, h, w, relu):
    """The stride-2 block openers on the bf16 matrix cores with three pieces per operand (csrc/conv_s2_x3.hip): against
    the float64 convolution the error is that of the fp32 implicit GEMM (<= 2x) and < 2e-6 of the magnitude; values spread
    over e^+-3; padding rows / the padding column enter as exact zeros (a map of ones with the border taps isolated)."""
    from paddle3d_amd.ops import conv

    g = torch.Generator().manual_seed(cin + cout + h)
    x = torch.randn(n, cin, h, w, generator=g) * torch.exp(3 * (2 * torch.rand(n, cin, h, w, generator=g) - 1))
    wt = torch.randn(cout, cin, 3, 3, generator=g) / (cin * 9) ** 0.5
    b = torch.randn(cout, generator=g)
    ref = F.conv2d(x.double(), wt.double(), b.double(), stride=2, padding=1)
    if relu:
        ref = torch.relu(ref)
    assert conv.conv3x3_s2_x3_supported(cin, cout, h, w, n)
    wp = conv.pack_conv3x3_s2_x3_weight(wt.cuda())
    assert wp.dtype == torch.bfloat16 and tuple(wp.shape) == (cout // 128, 3 * (cin // 16), 24576)
    # rows live at a pitch that is a multiple of 4, the pad columns hold zeros (ops/conv.py: pitch4)
    xp = F.pad(x, (0, conv.pitch4(w) - w)).cuda()
    wo, wop = w // 2, conv.pitch4(w // 2)
    got = conv.conv3x3_s2_x3_bias_relu(xp, wp, b.cuda(), cout, relu=relu, w_valid=w)
    assert tuple(got.shape) == (n, cout, h // 2, wop) and (got[..., wo:] == 0).all()
    err = (got[..., :wo].cpu().double() - ref).abs().max().item()
    got32 = conv.conv3x3_bias_relu(xp, conv.pack_conv3x3_weight(wt.cuda()), b.cuda(), cout, relu=relu, stride=2, w_valid=w)
    err32 = (got32[..., :wo].cpu().double() - ref).abs().max().item()
    mag = ref.abs().max().item()
    print(f"bf16x3 {err:.3e}, fp32 kernel {err32:.3e}, magnitude {mag:.1f}")
    assert err <= 2 * err32 + 1e-7 * mag and err < 2e-6 * mag
    assert torch.equal(got, conv.conv3x3_s2_x3_bias_relu(xp, wp, b.cuda(), cout, relu=relu, w_valid=w))  # run-to-run identical
    # the taps one at a time on a map of ones: every output counts the taps that fall inside the image, exactly
    ones = torch.ones(1, cin, h, w)
    onesp = F.pad(ones, (0, conv.pitch4(w) - w)).cuda()
    for ky in range(3):
        for kx in range(3):
            w1 = torch.zeros(cout, cin, 3, 3)
            w1[:, :, ky, kx] = 1.0
            want = F.conv2d(ones, w1, None, stride=2, padding=1)
            got1 = conv.conv3x3_s2_x3_bias_relu(onesp, conv.pack_conv3x3_s2_x3_weight(w1.cuda()), None, cout, relu=False,
                                                w_valid=w).cpu()
            assert torch.equal(got1[..., :wo], want) and (got1[..., wo:] == 0).all(), (ky, kx)


def test_patch_conv_bf16x3_refuses_what_it_cannot_do():
    from paddle3d_amd._lib import Paddle3DAmdError
    from paddle3d_amd.ops import conv

    x = torch.zeros(1, 24, 8, 8, device="cuda")
    wp = torch.zeros(1, 1, 16384, dtype=torch.bfloat16, device="cuda")
    out = torch.zeros(1, 128, 8, 8, device="cuda")
    assert not conv.patch_x3_supported(1, 24, 128, 8, 8)
    with pytest.raises(Paddle3DAmdError):
        conv.patch_conv_x3_bias_relu(x, wp, None, 1, 128, out)
    with pytest.raises(Paddle3DAmdError):  # mode 3 stays on the fp32 kernel
        conv.patch_conv_x3_bias_relu(torch.zeros(1, 32, 8, 8, device="cuda"), wp, None, 3, 128, out)


@pytest.mark.parametrize("n,cin,cout,h,w", [(2, 64, 64, 32, 128), (1, 4, 32, 8, 64), (1, 128, 96, 16, 64),
                                            (1, 384, 64, 24, 96), (3, 16, 32, 8, 32), (1, 24, 32, 8, 64),
                                            (2, 16, 64, 45, 180), (1, 8, 32, 5, 12), (1, 40, 32, 62, 124)])
@pytest.mark.parametrize("relu", [True, False])
@pytest.mark.parametrize("tile", [32, 64])
def test_conv3x3_winograd43_matches_torch(n, cin, cout, h, w, relu, tile):
    """Winograd F(4x4,3x3) on the fp32 matrix cores vs torch conv2d on the CPU (looser: ~1e-5 from the transforms)."""
    from paddle3d_amd.ops import conv

    g = torch.Generator().manual_seed(cin * 13 + cout + h)
    x = torch.randn(n, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, 3, 3, generator=g) / (cin * 9) ** 0.5
    b = torch.randn(cout, generator=g)
    ref = F.conv2d(x, wt, b, padding=1)
    if relu:
        ref = torch.relu(ref)
    assert conv.winograd43_supported(cin, cout, h, w)
    if cout % tile:
        pytest.skip("cout not a multiple of the workgroup's channel tile")
    out = conv.conv3x3_winograd43_bias_relu(x.cuda(), conv.pack_winograd43_weight(wt.cuda(), tile), b.cuda(), cout,
                                            relu).cpu()
    assert out.shape == ref.shape
    assert (out - ref).abs().max().item() < 5e-4, (out - ref).abs().max().item()


def test_winograd43_shift_taps():
    """Single-tap weights against an asymmetric input: catches transposed transforms / tile mis-addressing."""
    from paddle3d_amd.ops import conv

    cin = cout = 32
    x = torch.arange(1 * cin * 8 * 64, dtype=torch.float32).reshape(1, cin, 8, 64) / 512.0
    for ky in range(3):
        for kx in range(3):
            wt = torch.zeros(cout, cin, 3, 3)
            wt[torch.arange(cout), torch.arange(cin), ky, kx] = 1.0
            out = conv.conv3x3_winograd43_bias_relu(x.cuda(), conv.pack_winograd43_weight(wt.cuda()), None, cout, False).cpu()
            want = F.conv2d(x, wt, None, padding=1)
            assert (out - want).abs().max().item() < 5e-3, (ky, kx, (out - want).abs().max().item())


@pytest.mark.parametrize("ny,nx,batch", [(512, 512, 2), (496, 432, 2), (64, 96, 3), (40, 24, 1)])
def test_scatter_fused_into_stride2_conv(ny, nx, batch):
    """PointPillarsScatter fused into the first backbone convolution (pd3_scatter_conv3x3_bias_relu): bit-identical
    to the canvas written out and convolved (every stride-2 tile shape: 128-, 64- and 32-pixel-wide output tiles,
    partial tiles, duplicate coordinates = the highest pillar row wins, padding rows with batch -1)."""
    import torch

    from paddle3d_amd.ops import conv
    from paddle3d_amd.ops import pointpillars_scatter as ps

    torch.manual_seed(ny + nx)
    cin, cout = 64, 64
    m = min(6000, ny * nx // 3) * batch
    coords = torch.stack([torch.randint(0, batch, (m,)), torch.zeros(m, dtype=torch.long),
                          torch.randint(0, ny, (m,)), torch.randint(0, nx, (m,))], 1).int()
    coords[::97, 0] = -1                     # padding rows of a fixed-shape voxelizer output
    coords[5] = coords[3]                    # a duplicate cell
    feats = torch.randn(m, cin)
    w = torch.randn(cout, cin, 3, 3) * 0.05
    b = torch.randn(cout)
    feats, coords, w, b = feats.cuda(), coords.cuda(), w.cuda(), b.cuda()
    wp = conv.pack_conv3x3_weight(w)
    canvas = ps.pointpillars_scatter(feats, coords, batch, ny, nx)
    want = conv.conv3x3_bias_relu(canvas, wp, b, cout, relu=True, stride=2)
    got = conv.scatter_conv3x3_bias_relu(ps.SparseCanvas(feats, coords, batch, ny, nx), wp, b, cout)
    assert got.shape == want.shape
    assert torch.equal(got, want)
    assert torch.equal(ps.SparseCanvas(feats, coords, batch, ny, nx).dense(), canvas)


@pytest.mark.parametrize("ny,nx,batch,occ", [(512, 512, 2, 0.11), (496, 432, 2, 0.11), (64, 96, 3, 0.3), (40, 24, 1, 0.02),
                                             (64, 64, 1, 0.0)])
@pytest.mark.parametrize("cout", [64, 128])
def test_scatter_conv_as_sparse_convolution(ny, nx, batch, occ, cout):
    """PointPillarsScatter + the strided first convolution as a SPARSE convolution over the occupied pillars (round 6,
    pd3_pillar_conv_rulebook -> tile order -> bf16x3 gather-GEMM -> pd3_rows_to_dense_fill) against the dense statement:
    an fp64 convolution of the scattered canvas.  The rulebook is exact (active pixels, their nine pillar rows, the
    inverse); the values carry the fp32 gather-GEMM's error -- within 2x the dense fp32 kernel's own error against fp64
    and far inside the 1e-3 contract; pixels without an occupied cell hold relu(bias) exactly.  Duplicate cells (highest
    row wins), padding rows, an empty canvas."""
    import torch
    import torch.nn.functional as F

    from paddle3d_amd._lib import lib
    from paddle3d_amd.ops import conv
    from paddle3d_amd.ops import pointpillars_scatter as ps
    from paddle3d_amd.ops._common import check, ptr, stream_ptr, workspace

    torch.manual_seed(ny * 7 + nx + cout)
    cin = 64
    m = max(8, int(ny * nx * occ)) * batch
    coords = torch.stack([torch.randint(0, batch, (m,)), torch.zeros(m, dtype=torch.long),
                          torch.randint(0, ny, (m,)), torch.randint(0, nx, (m,))], 1).int()
    coords[::97, 0] = -1
    coords[5] = coords[3]
    if occ == 0.0:
        coords[:, 0] = -1                    # nothing on the canvas at all
    feats = (torch.randn(m, cin) * torch.exp(torch.randn(m, cin))).cuda()
    w = (torch.randn(cout, cin, 3, 3) * 0.05).cuda()
    b = torch.randn(cout).cuda()
    coords = coords.cuda()
    sc = ps.SparseCanvas(feats, coords, batch, ny, nx)
    assert conv.scatter_conv_sparse_supported(cin, cout, ny, nx, 2)
    got, packed = conv.scatter_conv3x3_sparse(sc, w, b)
    again, _ = conv.scatter_conv3x3_sparse(sc, w, b, packed)
    assert torch.equal(got, again)           # run-to-run identical
    canvas = sc.dense()
    ref64 = F.relu(F.conv2d(canvas.double().cpu(), w.double().cpu(), b.double().cpu(), stride=2, padding=1))
    dense32 = conv.conv3x3_bias_relu(canvas, conv.pack_conv3x3_weight(w), b, cout, relu=True, stride=2)[..., : nx // 2]
    assert got.shape == ref64.shape == (batch, cout, ny // 2, nx // 2)
    mag = max(1.0, float(ref64.abs().max()))
    err = float((got.double().cpu() - ref64).abs().max())
    err_dense = float((dense32.double().cpu() - ref64).abs().max())
    assert err <= max(2.0 * err_dense, 2e-6 * mag) and err < 1e-3, (err, err_dense, mag)
    # pixels that see no occupied cell: relu(bias), bit for bit
    occd = (canvas.abs().sum(1, keepdim=True) > 0).float()
    active = F.max_pool2d(F.pad(occd, (1, 1, 1, 1)), 3, 2)[..., : ny // 2, : nx // 2] > 0
    fill = torch.relu(b).view(1, -1, 1, 1).expand_as(got)
    assert torch.equal(got[(~active).expand_as(got)], fill[(~active).expand_as(got)])
    # the rulebook itself against NumPy
    L = lib()
    ho, wo = ny // 2, nx // 2
    cap = batch * ho * wo
    nbr = torch.empty((cap, 9), dtype=torch.int32, device="cuda")
    out_cell = torch.empty((cap,), dtype=torch.int32, device="cuda")
    cell_row = torch.empty((batch, ho * wo), dtype=torch.int32, device="cuda")
    n_out = torch.empty((1,), dtype=torch.int32, device="cuda")
    ws = workspace(L.pd3_pillar_conv_rulebook_workspace(batch, ny, nx, 2), feats.device)
    order = torch.empty((int(L.pd3_sparse_tile_order_entries(cap)),), dtype=torch.int32, device="cuda")
    check(L.pd3_pillar_conv_rulebook(ptr(sc.inv), batch, ny, nx, 2, ptr(nbr), ptr(out_cell), ptr(cell_row), ptr(n_out), cap,
                                     ptr(order), ptr(ws), ws.numel(), stream_ptr(feats.device)), "pillar_conv_rulebook")
    inv = np.pad(sc.inv.cpu().numpy().reshape(batch, ny, nx), ((0, 0), (1, 1), (1, 1)), constant_values=-1)
    want_nbr = np.stack([inv[:, ky:ky + ny:2, kx:kx + nx:2] for ky in range(3) for kx in range(3)], -1).reshape(-1, 9)
    act = (want_nbr >= 0).any(1)
    n = int(n_out.item())
    assert n == int(act.sum()) == int(active.sum())
    np.testing.assert_array_equal(out_cell[:n].cpu().numpy(), np.nonzero(act)[0])          # raster order
    np.testing.assert_array_equal(nbr[:n].cpu().numpy(), want_nbr[act])
    want_cr = np.full(cap, -1, np.int32)
    want_cr[np.nonzero(act)[0]] = np.arange(n, dtype=np.int32)
    np.testing.assert_array_equal(cell_row.cpu().numpy().reshape(-1), want_cr)
    # the tile order: every window a permutation of its rows (then -1), grouped by tap mask in ascending order
    od = order.cpu().numpy()
    masks = ((want_nbr[act] >= 0) * (1 << np.arange(9))).sum(1)
    for w0 in range(0, len(od), 8192):   # (grouped per window of 8192 rows)
        nv = min(max(n - w0, 0), 8192)
        win = od[w0:w0 + 8192]
        assert (win[nv:] == -1).all()
        np.testing.assert_array_equal(np.sort(win[:nv]), np.arange(w0, w0 + nv))
        assert (np.diff(masks[win[:nv]]) >= 0).all()


# ---- mixed precision: the fp16 matrix-core form of the stride-1 layers (csrc/conv_f16.hip) -----------------------------
@pytest.mark.parametrize("out_f32", [False, True])
@pytest.mark.parametrize("cin,cout,h,w", [(64, 128, 32, 64), (32, 64, 64, 32), (128, 256, 16, 32), (16, 64, 32, 96),
                                          (48, 384, 48, 32), (32, 128, 45, 52), (16, 64, 50, 20), (64, 128, 180, 180)])
def test_conv3x3_f16_matches_fp32_math_on_fp16_operands(cin, cout, h, w, out_f32):
    """v_mfma_f32_32x32x16_f16 accumulates in fp32, so on operands that are already fp16 values the kernel must agree
    with an fp32 convolution of the same values to accumulation-order noise -- which pins the whole layout (A / B
    fragment order, tap order, channel tiles of 128 and 64, NHWC and NCHW epilogues, image borders) on asymmetric data."""
    import torch.nn.functional as F

    from paddle3d_amd.ops import conv

    g = torch.Generator(device="cuda").manual_seed(cin * 1000 + cout + h)
    x = torch.randn(2, cin, h, w, device="cuda", generator=g)
    wt = torch.randn(cout, cin, 3, 3, device="cuda", generator=g) / (cin * 9) ** 0.5
    b = torch.randn(cout, device="cuda", generator=g)
    xh = conv.to_f16_nhwc(x)
    assert torch.equal(xh, x.half().permute(0, 2, 3, 1).contiguous())  # round to nearest even, NHWC
    assert conv.f16_supported(cin, cout, h, w)
    ref = F.relu(F.conv2d(x.half().float().cpu(), wt.half().float().cpu(), b.cpu(), padding=1)).cuda()  # torch CPU fp32
    got = conv.conv3x3_f16_bias_relu(xh, conv.pack_conv3x3_f16_weight(wt), b, cout, relu=True, out_f32_nchw=out_f32)
    if out_f32:
        assert got.shape == ref.shape and got.dtype == torch.float32
        assert (got - ref).abs().max().item() < 2e-4 * max(1.0, ref.abs().max().item())
    else:
        assert got.shape == (2, h, w, cout) and got.dtype == torch.float16
        err = (got.float().permute(0, 3, 1, 2) - ref).abs().max().item()
        assert err < 2e-3 * max(1.0, ref.abs().max().item())  # + the fp16 rounding of the stored result
    # no ReLU, no bias
    got2 = conv.conv3x3_f16_bias_relu(xh, conv.pack_conv3x3_f16_weight(wt), None, cout, relu=False, out_f32_nchw=True)
    ref2 = F.conv2d(x.half().float().cpu(), wt.half().float().cpu(), None, padding=1).cuda()
    assert (got2 - ref2).abs().max().item() < 2e-4 * max(1.0, ref2.abs().max().item())


def test_conv3x3_f16_refuses_shapes_it_does_not_take():
    from paddle3d_amd._lib import Paddle3DAmdError
    from paddle3d_amd.ops import conv

    assert conv.f16_supported(64, 64, 180, 180)  # (round 5: border tiles are masked, any map size runs)
    assert not conv.f16_supported(8, 64, 32, 32) and not conv.f16_supported(64, 96, 32, 32)
    x = torch.zeros(1, 32, 32, 24, dtype=torch.float16, device="cuda")  # 24 input channels: not whole 16-channel chunks
    with pytest.raises(Paddle3DAmdError, match="status -3"):
        conv.conv3x3_f16_bias_relu(x, torch.zeros(1, 1, 9, 2, 64, 8, dtype=torch.float16, device="cuda"), None, 64)


@pytest.mark.parametrize("n,cin,cout,h,w", [(2, 64, 64, 32, 64), (1, 128, 128, 16, 128), (1, 8, 64, 24, 68), (1, 64, 192, 12, 64),
                                           (1, 384, 64, 8, 128), (1, 64, 128, 90, 92)])
@pytest.mark.parametrize("relu", [True, False])
def test_winograd43_pingpong_form(n, cin, cout, h, w, relu):
    """The ping-pong F(4x4, 3x3) kernel with U computed on the fly against torch's fp32 convolution (1e-3 contract; the
    packed form's measured error is ~1e-5) and against the packed-form kernel (same arithmetic up to the rounding of U):
    partial border tiles, a zero-padded 90-wide map, channel counts of one to 48 slots."""
    from paddle3d_amd.ops import conv

    g = torch.Generator().manual_seed(cin + cout + h)
    x = torch.randn(n, cin, h, w, generator=g)
    wv = 90 if w == 92 else w
    x[..., wv:] = 0
    wt = torch.randn(cout, cin, 3, 3, generator=g) / (cin * 9) ** 0.5
    b = torch.randn(cout, generator=g)
    ref = F.conv2d(x[..., :wv], wt, b, padding=1)
    if relu:
        ref = torch.relu(ref)
    assert conv.winograd43_pp_supported(cin, cout, h, w)
    ul = conv.pack_winograd43_lane_weight(wt.cuda())
    packed = conv.conv3x3_winograd43_bias_relu(x.cuda(), conv.pack_winograd43_weight(wt.cuda()), b.cuda(), cout, relu,
                                               w_valid=wv)
    got = conv.conv3x3_winograd43_pp_bias_relu(x.cuda(), ul, b.cuda(), cout, relu, w_valid=wv)
    assert got.shape == (n, cout, h, w)
    assert (got[..., :wv].cpu() - ref).abs().max().item() < 1e-3
    assert (got[..., wv:] == 0).all()
    assert torch.equal(got, packed)  # the same U, the same order of accumulation: the same bytes


@pytest.mark.parametrize("n,cin,cout,h,w,wv", [(2, 64, 192, 32, 128, 128), (1, 8, 64, 8, 64, 64), (3, 16, 128, 10, 36, 34),
                                               (1, 64, 576, 45, 180, 180), (2, 128, 64, 16, 64, 64)])
@pytest.mark.parametrize("relu", [True, False])
def test_conv3x3_winograd43_with_the_input_transformed_once(n, cin, cout, h, w, wv, relu):
    """The ping-pong Winograd kernel fed with V = B^T d B computed once by a pre-pass (the head's 36 channel blocks share
    their input): the same bytes as the kernel that transforms for itself -- the same w4_in sequence, the same MFMA order --
    on whole and partial tiles, padded rows (w_valid < pitch) and image borders."""
    from paddle3d_amd.ops import conv

    g = torch.Generator().manual_seed(cin + cout + h)
    x = torch.randn(n, cin, h, w, generator=g)
    x[..., wv:] = 0
    wt = torch.randn(cout, cin, 3, 3, generator=g) / (cin * 9) ** 0.5
    b = torch.randn(cout, generator=g)
    ul = conv.pack_winograd43_lane_weight(wt.cuda())
    want = conv.conv3x3_winograd43_pp_bias_relu(x.cuda(), ul, b.cuda(), cout, relu, w_valid=wv)
    v = conv.winograd43_input_transform(x.cuda(), w_valid=wv)
    tiles = n * -(-h // 8) * -(-w // 64)
    assert v.numel() == tiles * (cin // 8) * 2 * 8 * 16 * 36
    got = conv.conv3x3_winograd43_ppv_bias_relu(v, x.shape, ul, b.cuda(), cout, relu, w_valid=wv)
    assert torch.equal(got, want)
    ref = F.conv2d(x[..., :wv].double(), wt.double(), b.double(), padding=1)
    if relu:
        ref = torch.relu(ref)
    assert (got[..., :wv].cpu().double() - ref).abs().max().item() < 1e-3
    # a slice of the channel blocks (the head's two slices): blocks [1, 2) alone
    if cout >= 128:
        part = conv.conv3x3_winograd43_ppv_bias_relu(v, x.shape, ul[1:2], b.cuda()[64:128], 64, relu, w_valid=wv)
        assert torch.equal(part, want[:, 64:128])


@pytest.mark.parametrize("groups,co,h,w", [(6, 3, 32, 64), (4, 1, 20, 45), (36, 3, 16, 32), (2, 4, 9, 33), (3, 2, 128, 128)])
def test_grouped_conv3x3_small_f16_matches_fp32_math_on_fp16_operands(groups, co, h, w):
    """The AMP form of the final SeparateHead convolutions (fp16 NHWC in, fp32 NCHW out, v_dot2_f32_f16 with fp32
    accumulation) against torch's grouped fp32 convolution of the same fp16-rounded values: accumulation-order noise only.
    Partial tiles, every output width 1 .. 4, a slice of the groups written into a wider output."""
    from paddle3d_amd.ops import conv

    g = torch.Generator(device="cuda").manual_seed(groups * 100 + co + h)
    x = torch.randn(2, groups * 64, h, w, device="cuda", generator=g)
    wt = torch.randn(groups * co, 64, 3, 3, device="cuda", generator=g) / 24.0
    b = torch.randn(groups * co, device="cuda", generator=g)
    xh = conv.to_f16_nhwc(x)
    ref = F.conv2d(x.half().float().cpu(), wt.half().float().cpu(), b.cpu(), padding=1, groups=groups).cuda()
    wp = conv.pack_grouped_weight_f16(wt, groups)
    got = conv.grouped_conv3x3_small_f16(xh, wp, b, groups)
    assert got.shape == ref.shape and got.dtype == torch.float32
    per_ch = (got - ref).abs().amax(dim=(0, 2, 3)).cpu().numpy().round(4).tolist()
    assert (got - ref).abs().max().item() < 2e-4 * max(1.0, ref.abs().max().item()), per_ch
    # the group-major form of the input ([n, groups, h, w, 64]: persistent LDS-DMA kernel, round 6): the same bytes
    xg = xh.view(2, h, w, groups, 64).permute(0, 3, 1, 2, 4).contiguous()
    assert torch.equal(conv.grouped_conv3x3_small_f16(xg, wp, b, groups, group_major=True), got)
    if groups >= 4:  # the head's slices: groups [2, 4) of the input land at groups [3, 5) of a 6-group output
        out = torch.full((2, 6 * co, h, w), 7.0, device="cuda")
        xs = xh[..., 2 * 64:4 * 64].contiguous()
        conv.grouped_conv3x3_small_f16(xs, wp[2:4], b[2 * co:4 * co], 2, out=out, out_groups=6, out_group0=3)
        assert torch.equal(out[:, 3 * co:5 * co], got[:, 2 * co:4 * co])
        assert (out[:, :3 * co] == 7.0).all() and (out[:, 5 * co:] == 7.0).all()


@pytest.mark.parametrize("cin,cout,h,w", [(64, 128, 64, 64), (128, 256, 32, 96), (16, 128, 33, 45), (32, 384, 18, 70),
                                          (64, 128, 256, 256)])
def test_conv3x3_s2_f16_matches_fp32_math_on_fp16_operands(cin, cout, h, w):
    """The stride-2 convolution that opens a backbone block under AMP (fp16 NHWC in and out, de-interleaved patch columns)
    against torch's fp32 stride-2 convolution of the same fp16-rounded values; odd sizes, partial tiles, image borders."""
    from paddle3d_amd.ops import conv

    g = torch.Generator(device="cuda").manual_seed(cin + cout + h)
    x = torch.randn(2, cin, h, w, device="cuda", generator=g)
    wt = torch.randn(cout, cin, 3, 3, device="cuda", generator=g) / (cin * 9) ** 0.5
    b = torch.randn(cout, device="cuda", generator=g)
    assert conv.s2_f16_supported(cin, cout) and not conv.s2_f16_supported(cin, 64)
    ref = F.relu(F.conv2d(x.half().float().cpu(), wt.half().float().cpu(), b.cpu(), stride=2, padding=1)).cuda()
    got = conv.conv3x3_s2_f16_bias_relu(conv.to_f16_nhwc(x), conv.pack_conv3x3_f16_weight(wt, tile=128), b, cout)
    assert got.dtype == torch.float16 and got.shape == (2, ref.shape[2], ref.shape[3], cout)
    err = (got.float().permute(0, 3, 1, 2) - ref).abs().max().item()
    assert err < 2e-3 * max(1.0, ref.abs().max().item())
    got2 = conv.conv3x3_s2_f16_bias_relu(conv.to_f16_nhwc(x), conv.pack_conv3x3_f16_weight(wt, tile=128), None, cout,
                                         relu=False)
    ref2 = F.conv2d(x.half().float().cpu(), wt.half().float().cpu(), None, stride=2, padding=1).cuda()
    assert (got2.float().permute(0, 3, 1, 2) - ref2).abs().max().item() < 2e-3 * max(1.0, ref2.abs().max().item())


@pytest.mark.parametrize("ny,nx,batch,cout", [(512, 512, 2, 64), (496, 432, 2, 64), (64, 96, 3, 128), (40, 24, 1, 64)])
def test_scatter_fused_into_stride2_f16_conv(ny, nx, batch, cout):
    """PointPillarsScatter fused into the fp16 stride-2 convolution (pd3_scatter_conv3x3_s2_f16_bias_relu, channel tiles
    of 64 and 128): against torch's fp32 stride-2 convolution of the fp16-rounded canvas, and -- for the 128 tile -- the
    same bytes as the unfused fp16 kernel on the written-out canvas; duplicates, padding rows, partial tiles."""
    from paddle3d_amd.ops import conv
    from paddle3d_amd.ops import pointpillars_scatter as ps

    torch.manual_seed(ny + nx)
    cin = 64
    m = min(6000, ny * nx // 3) * batch
    coords = torch.stack([torch.randint(0, batch, (m,)), torch.zeros(m, dtype=torch.long),
                          torch.randint(0, ny, (m,)), torch.randint(0, nx, (m,))], 1).int()
    coords[::97, 0] = -1
    coords[5] = coords[3]
    feats, coords = torch.randn(m, cin).cuda(), coords.cuda()
    w, b = (torch.randn(cout, cin, 3, 3) * 0.05).cuda(), torch.randn(cout).cuda()
    assert conv.scatter_conv_s2_f16_supported(cin, cout, ny, nx)
    canvas = ps.pointpillars_scatter(feats, coords, batch, ny, nx)
    ref = F.relu(F.conv2d(canvas.half().float().cpu(), w.half().float().cpu(), b.cpu(), stride=2, padding=1)).cuda()
    for tile in ([64, 128] if cout % 128 == 0 else [64]):
        got = conv.scatter_conv3x3_s2_f16_bias_relu(ps.SparseCanvas(feats, coords, batch, ny, nx),
                                                    conv.pack_conv3x3_f16_weight(w, tile=tile), b, cout)
        assert got.dtype == torch.float16 and got.shape == (batch, ny // 2, nx // 2, cout)
        err = (got.float().permute(0, 3, 1, 2) - ref).abs().max().item()
        assert err < 2e-3 * max(1.0, ref.abs().max().item()), (tile, err)
        if tile == 128:
            dense = conv.conv3x3_s2_f16_bias_relu(conv.to_f16_nhwc(canvas), conv.pack_conv3x3_f16_weight(w, tile=128), b,
                                                  cout)
            assert torch.equal(got, dense)


@pytest.mark.parametrize("cin,cout,h,w", [(64, 64, 40, 64), (128, 128, 16, 50)])
def test_conv3x3_f16_dual_output_is_both_single_outputs(cin, cout, h, w):
    from paddle3d_amd.ops import conv

    g = torch.Generator(device="cuda").manual_seed(cin + h)
    xh = conv.to_f16_nhwc(torch.randn(2, cin, h, w, device="cuda", generator=g))
    wp = conv.pack_conv3x3_f16_weight(torch.randn(cout, cin, 3, 3, device="cuda", generator=g) / (cin * 9) ** 0.5)
    b = torch.randn(cout, device="cuda", generator=g)
    oh, of = conv.conv3x3_f16_bias_relu_dual(xh, wp, b, cout)
    assert torch.equal(oh, conv.conv3x3_f16_bias_relu(xh, wp, b, cout))
    assert torch.equal(of, conv.conv3x3_f16_bias_relu(xh, wp, b, cout, out_f32_nchw=True))


@pytest.mark.parametrize("cfg", ["pillars", "voxels"])
def test_second_fpn_f16_levels_match_fp32_levels(cfg):
    """SecondFPN under AMP: every level (kernel = stride convolution, 1 x 1 convolution, transposed convolution) is one
    fp16 gather-GEMM over a static pixel neighbour table writing its slice of the concatenated fp16 NHWC map; against
    the fp32 patch-GEMM levels on the same fp16-rounded inputs and weights."""
    from paddle3d_amd import centerpoint as cpm

    torch.manual_seed(4)
    if cfg == "pillars":  # second_fpn of CenterPoint-Pillars: strides (0.5, 1, 2), 128 channels each
        neck = cpm.SecondFPN((64, 128, 256), (128, 128, 128), (0.5, 1, 2), use_conv_for_no_stride=True).cuda().eval()
        shapes = [(64, 64, 96), (128, 32, 48), (256, 16, 24)]
    else:                 # CenterPoint-Voxel: strides (1, 2), 256 channels each, a 180-wide map
        neck = cpm.SecondFPN((128, 256), (256, 256), (1, 2), use_conv_for_no_stride=True).cuda().eval()
        shapes = [(128, 20, 180), (256, 10, 90)]
    with torch.no_grad():
        for m in neck.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.running_mean.normal_(0, 0.1)
                m.running_var.uniform_(0.5, 1.5)
                m.weight.uniform_(0.5, 1.5)
                m.bias.normal_(0, 0.1)
            if isinstance(m, (torch.nn.Conv2d, torch.nn.ConvTranspose2d)):
                m.weight.copy_(m.weight.half().float())
    xs = [torch.randn(2, c, h, w, device="cuda").half().float() for c, h, w in shapes]
    want = neck(xs)
    assert neck.amp_ok(None)
    got = neck([x.half().permute(0, 2, 3, 1).contiguous() for x in xs])
    assert got.dtype == torch.float16 and got.shape == (2, want.shape[2], want.shape[3], want.shape[1])
    err = (got.float().permute(0, 3, 1, 2) - want).abs().max().item()
    assert err < 3e-3 * max(1.0, want.abs().max().item()), err  # (the folded weights are rounded to fp16 once more)
