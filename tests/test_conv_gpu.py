"""conv3x3_bias_relu (fp32 MFMA) vs torch conv2d on the CPU (fp32)."""
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

    x = torch.randn(1, 8, 6, 48).cuda()
    wp = torch.zeros(1, 1, 72, 64).cuda()
    with pytest.raises(RuntimeError):
        conv.conv3x3_bias_relu(x, wp, None, 64, True)


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


@pytest.mark.parametrize("mode", [0, 1, 2])
def test_patch_conv_matches_torch(mode):
    """FPN patch convolutions (second_fpn.py:99-157) written at a channel offset of a wider tensor."""
    from paddle3d_amd.ops import conv

    g = torch.Generator().manual_seed(mode)
    n, off, ctot = 2, 64, 256
    if mode == 0:
        cin, cout, h, w = 16, 128, 8, 256
        wt = torch.randn(cout, cin, 2, 2, generator=g) / (cin * 4) ** 0.5
        x = torch.randn(n, cin, h, w, generator=g)
        b = torch.randn(cout, generator=g)
        ref = torch.relu(F.conv2d(x, wt, b, stride=2))
        tr = False
    elif mode == 1:
        cin, cout, h, w = 32, 64, 16, 32
        wt = torch.randn(cin, cout, 1, 1, generator=g) / cin ** 0.5
        x = torch.randn(n, cin, h, w, generator=g)
        b = torch.randn(cout, generator=g)
        ref = torch.relu(F.conv_transpose2d(x, wt, b, stride=1))
        tr = True
    else:
        cin, cout, h, w = 48, 128, 16, 16
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
