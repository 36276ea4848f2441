"""The dense BEV graph's convolutions on the library's fp32-MFMA kernels (SecondBackbone / SecondFPN / CenterHead
with BatchNorm folded): conv3x3_winograd_bias_relu (stride 1), conv3x3_bias_relu (stride 1 / 2, implicit GEMM),
patch_conv_bias_relu (kernel = stride FPN levels, written at a channel offset), grouped_conv3x3_small (the final
SeparateHead convolutions), plus the host-side weight packers each kernel expects."""
from __future__ import annotations

import torch

from ._common import check, lib, ptr, require_gpu, stream_ptr

__all__ = ["pack_conv3x3_weight", "conv3x3_bias_relu", "supported", "pitch4", "pack_grouped_weight", "grouped_conv3x3_small",
           "grouped_small_supported", "winograd_supported", "pack_winograd_weight", "conv3x3_winograd_bias_relu",
           "patch_mode", "patch_supported", "pack_patch_weight", "patch_conv_bias_relu",
           "patch_x3_supported", "pack_patch_weight_x3", "patch_conv_x3_bias_relu", "split_bf16x3",
           "conv3x3_s2_x3_supported", "pack_conv3x3_s2_x3_weight", "conv3x3_s2_x3_bias_relu",
           "winograd43_input_transform", "conv3x3_winograd43_ppv_bias_relu",
           "winograd43_supported", "winograd43_tile", "pack_winograd43_weight", "conv3x3_winograd43_bias_relu",
           "WINOGRAD43_PP_MIN_CIN", "winograd43_pp_supported", "pack_winograd43_lane_weight", "conv3x3_winograd43_pp_bias_relu",
           "f16_supported", "f16_tile", "pack_conv3x3_f16_weight", "conv3x3_f16_bias_relu", "to_f16_nhwc",
           "pack_grouped_weight_f16", "grouped_conv3x3_small_f16", "conv3x3_f16_bias_relu_dual",
           "s2_f16_supported", "conv3x3_s2_f16_bias_relu", "scatter_conv_s2_f16_supported",
           "scatter_conv3x3_s2_f16_bias_relu", "scatter_conv_sparse_supported", "scatter_conv3x3_sparse"]


def pitch4(w: int) -> int:
    """Row pitch of a map of width w: widths that are not a multiple of 4 are padded with zero columns."""
    return (int(w) + 3) // 4 * 4


def supported(cin: int, cout: int, h: int, w: int, stride: int = 1) -> bool:
    """Shapes pd3_conv3x3_bias_relu takes ([h, w] = input size, w the VALID width: rows of other widths live
    zero-padded to a multiple of 4, see pitch4): partial border tiles are masked, so any map."""
    return stride in (1, 2) and cin % 8 == 0 and cout % 64 == 0 and h % stride == 0 and w % stride == 0


def pack_conv3x3_weight(weight: torch.Tensor) -> torch.Tensor:
    """[Cout, Cin, 3, 3] -> [Cout/64][Cin/8][4 channel pairs][9 taps][2 channels of the pair][64]
    (the order the kernel stages into LDS: one MFMA K step = one tap of one channel pair)."""
    cout, cin = weight.shape[:2]
    assert weight.shape[2:] == (3, 3) and cout % 64 == 0 and cin % 8 == 0
    w = weight.reshape(cout // 64, 64, cin // 8, 4, 2, 9).permute(0, 2, 3, 5, 4, 1)
    return w.reshape(cout // 64, cin // 8, 72, 64).contiguous()


def conv3x3_bias_relu(x: torch.Tensor, w_packed: torch.Tensor, bias, cout: int, relu: bool = True,
                      stride: int = 1, out: torch.Tensor | None = None, w_valid: int | None = None) -> torch.Tensor:
    """x [n, cin, h, pitch]: pitch % 4 == 0; w_valid (default pitch) = real width, columns beyond hold zeros.
    Returns [n, cout, h / stride, pitch4(w_valid / stride)] with zeros in its own padding columns."""
    xx = require_gpu(x, "conv3x3_bias_relu")
    n, cin, h, w = xx.shape
    wv = w if w_valid is None else int(w_valid)
    if out is None:
        out = torch.empty((n, cout, h // stride, pitch4(wv // stride)), dtype=torch.float32, device=xx.device)
    check(lib().pd3_conv3x3_bias_relu(ptr(xx), ptr(w_packed), ptr(bias), n, cin, cout, h, w, wv, int(stride),
                                      int(bool(relu)), ptr(out), int(out.shape[3]), stream_ptr(xx.device)),
          "conv3x3_bias_relu")
    return out


def scatter_conv_supported(cin: int, cout: int, ny: int, nx: int, stride: int) -> bool:
    return stride == 2 and cin % 8 == 0 and cout % 64 == 0 and ny % 2 == 0 and nx % 2 == 0


def scatter_conv3x3_bias_relu(canvas, w_packed: torch.Tensor, bias, cout: int, relu: bool = True) -> torch.Tensor:
    """conv3x3 / pad 1 / stride 2 + bias + ReLU over a SparseCanvas (ops.pointpillars_scatter) without writing the
    canvas: -> [B, cout, ny/2, pitch4(nx/2)] (columns >= nx/2 zero).  Bit-identical to scatter + conv3x3_bias_relu."""
    f, inv = canvas.features, canvas.inv
    n, cin, ny, nx = canvas.shape
    wo = pitch4(nx // 2)
    out = torch.empty((n, cout, ny // 2, wo), dtype=torch.float32, device=f.device)
    check(lib().pd3_scatter_conv3x3_bias_relu(ptr(f), ptr(inv), ptr(w_packed), ptr(bias), n, cin, cout, ny, nx, 2,
                                              int(bool(relu)), ptr(out), wo, stream_ptr(f.device)),
          "scatter_conv3x3_bias_relu")
    return out


def scatter_conv_sparse_supported(cin: int, cout: int, ny: int, nx: int, stride: int) -> bool:
    """Shapes scatter_conv3x3_sparse serves: the bf16x3 gather-GEMM's (cin % 16 == 0, cout 64 or 128), stride 2, an
    output plane that is a whole number of float4."""
    ho, wo = (ny - 1) // stride + 1, (nx - 1) // stride + 1
    return stride == 2 and cin % 16 == 0 and cout in (64, 128) and wo % 4 == 0 and (ho * wo) % 4 == 0


def scatter_conv3x3_sparse(canvas, weight: torch.Tensor, bias: torch.Tensor, packed=None, relu: bool = True):
    """PointPillarsScatter + conv3x3 / pad 1 / stride 2 + bias + ReLU as a SPARSE convolution over the occupied pillars
    (pd3_pillar_conv_rulebook -> tile order -> the bf16x3 gather-GEMM = fp32 arithmetic on the bf16 matrix cores ->
    pd3_rows_to_dense_fill): a nuScenes canvas is 11 % occupied, 8.7 x fewer products than the dense kernel multiplies.
    canvas: ops.pointpillars_scatter.SparseCanvas; weight [cout, cin, 3, 3] (BatchNorm folded), bias [cout].  Returns
    ([B, cout, ny / 2, nx / 2] fp32 NCHW, packed weight for reuse).  No host sync: every array lives at the worst-case
    capacity (all output pixels), the row count stays on the device.  Against scatter + the dense kernel the values
    differ by the summation order (1e-6 relative; the dense kernel multiplies the empty taps' zeros)."""
    from . import sparse_conv3d as sp
    from ._common import workspace

    f, inv = canvas.features, canvas.inv
    n, cin, ny, nx = canvas.shape
    cout = int(weight.shape[0])
    ho, wo = (ny - 1) // 2 + 1, (nx - 1) // 2 + 1
    cap = n * ho * wo
    dev = f.device
    L = lib()
    nbr = torch.empty((cap, 9), dtype=torch.int32, device=dev)
    out_cell = torch.empty((cap,), dtype=torch.int32, device=dev)
    cell_row = torch.empty((n, ho * wo), dtype=torch.int32, device=dev)
    n_out = torch.empty((1,), dtype=torch.int32, device=dev)
    ws = workspace(L.pd3_pillar_conv_rulebook_workspace(n, ny, nx, 2), dev)
    order = torch.empty((int(L.pd3_sparse_tile_order_entries(cap)),), dtype=torch.int32, device=dev)
    check(L.pd3_pillar_conv_rulebook(ptr(inv), n, ny, nx, 2, ptr(nbr), ptr(out_cell), ptr(cell_row), ptr(n_out), cap,
                                     ptr(order), ptr(ws), ws.numel(), stream_ptr(dev)), "pillar_conv_rulebook")
    idx = sp.SparseIndices(None, nbr, cap, (1, ho, wo), 9, order, n_out)
    if packed is None:  # [kd = 1, kh, kw, cin, cout]: offset ky * 3 + kx, as the rulebook numbers the taps
        packed = sp.pack_weight_bf16x3(weight.permute(2, 3, 1, 0).reshape(1, 3, 3, cin, cout).contiguous())
    rows = sp.features_bf16x3(f, idx, packed, cin, cout, bias, None, None, None, relu)
    fill = torch.relu(bias) if relu else bias
    out = torch.empty((n, cout, ho, wo), dtype=torch.float32, device=dev)
    check(L.pd3_rows_to_dense_fill(ptr(rows), ptr(cell_row), ptr(fill.contiguous()), n, cout, ho, wo, ptr(out),
                                   stream_ptr(dev)), "rows_to_dense_fill")
    return out, packed


def winograd_supported(cin: int, cout: int, h: int, w: int) -> bool:
    return cin % 8 == 0 and cout % 32 == 0 and w % 4 == 0


def pack_winograd_weight(weight: torch.Tensor) -> torch.Tensor:
    """[Cout, Cin, 3, 3] -> U = G g G^T packed [Cout/32][Cin/8][2 blocks][8 ci][16 co][16 components]."""
    cout, cin = weight.shape[:2]
    assert weight.shape[2:] == (3, 3) and cout % 32 == 0 and cin % 8 == 0
    g = torch.tensor([[1.0, 0.0, 0.0], [0.5, 0.5, 0.5], [0.5, -0.5, 0.5], [0.0, 0.0, 1.0]], dtype=torch.float64,
                     device=weight.device)
    u = torch.einsum("ij,ocjk,lk->ocil", g, weight.double(), g).float()  # [cout, cin, 4, 4]
    u = u.reshape(cout // 32, 2, 16, cin // 8, 8, 16).permute(0, 3, 1, 4, 2, 5)
    return u.contiguous()


def conv3x3_winograd_bias_relu(x: torch.Tensor, u_packed: torch.Tensor, bias, cout: int, relu: bool = True,
                               out: torch.Tensor | None = None) -> torch.Tensor:
    xx = require_gpu(x, "conv3x3_winograd_bias_relu")
    n, cin, h, w = xx.shape
    if out is None:
        out = torch.empty((n, cout, h, w), dtype=torch.float32, device=xx.device)
    check(lib().pd3_conv3x3_winograd_bias_relu(ptr(xx), ptr(u_packed), ptr(bias), n, cin, cout, h, w,
                                               int(bool(relu)), ptr(out), stream_ptr(xx.device)),
          "conv3x3_winograd_bias_relu")
    return out


def winograd43_supported(cin: int, cout: int, h: int, w: int) -> bool:
    """w = the valid width; the tensor's rows are pitch4(w) long (zero padded)."""
    return cin % 4 == 0 and cout % 32 == 0


def winograd43_tile(cout: int) -> int:
    """Output channels per workgroup the kernel is run with: 64 where the layer allows, else 32."""
    return 64 if cout % 64 == 0 else 32


def pack_winograd43_weight(weight: torch.Tensor, tile: int | None = None) -> torch.Tensor:
    """[Cout, Cin, 3, 3] -> U = G g G^T (6x6, F(4x4,3x3)) packed [Cout/T][Cin/4][T/16 blocks][4 ci][16 co][36],
    T = channels per workgroup (winograd43_tile(Cout) by default)."""
    cout, cin = weight.shape[:2]
    t = winograd43_tile(cout) if tile is None else tile
    assert weight.shape[2:] == (3, 3) and cout % t == 0 and cin % 4 == 0 and t in (32, 64)
    g = torch.tensor([[1 / 4, 0, 0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6], [1 / 24, 1 / 12, 1 / 6],
                      [1 / 24, -1 / 12, 1 / 6], [0, 0, 1]], dtype=torch.float64, device=weight.device)
    u = torch.einsum("ij,ocjk,lk->ocil", g, weight.double(), g).float()  # [cout, cin, 6, 6]
    u = u.reshape(cout // t, t // 16, 16, cin // 4, 4, 36).permute(0, 3, 1, 4, 2, 5)
    return u.contiguous()


def conv3x3_winograd43_bias_relu(x: torch.Tensor, u_packed: torch.Tensor, bias, cout: int, relu: bool = True,
                                 out: torch.Tensor | None = None, w_valid: int | None = None) -> torch.Tensor:
    """x [n, cin, h, pitch] (pitch % 4 == 0, columns >= w_valid zero) -> [n, cout, h, pitch], same convention."""
    xx = require_gpu(x, "conv3x3_winograd43_bias_relu")
    n, cin, h, w = xx.shape
    tile = 16 * u_packed.shape[2]  # the packing records the workgroup shape
    if out is None:
        out = torch.empty((n, cout, h, w), dtype=torch.float32, device=xx.device)
    check(lib().pd3_conv3x3_winograd43_bias_relu(ptr(xx), ptr(u_packed), ptr(bias), n, cin, cout, h, w,
                                                 w if w_valid is None else int(w_valid), int(bool(relu)), ptr(out),
                                                 tile, stream_ptr(xx.device)),
          "conv3x3_winograd43_bias_relu")
    return out


def patch_mode(weight: torch.Tensor, stride: int, transpose: bool):
    """Which pd3_patch_conv_bias_relu mode computes this FPN layer (None if it is not a patch convolution)."""
    k = weight.shape[2]
    if weight.shape[2] != weight.shape[3] or k != stride:
        return None
    if k == 1:
        return 1
    if k == 2:
        return 2 if transpose else 0
    if k == 4 and transpose:
        return 3
    return None


def patch_supported(mode: int, cin: int, cout: int, h: int, w: int) -> bool:
    """[h, w] = input map, w its row pitch (pitch4 of the valid width for modes 2, 3)."""
    if mode == 0:
        return h % 4 == 0 and w % 4 == 0 and (cin * 4) % 16 == 0 and cout % 64 == 0
    if (h * w) % 4 or cin % 16:
        return False
    return {1: cout >= 1, 2: (cout * 4) % 64 == 0, 3: (cout * 16) % 64 == 0}.get(mode, False)


def pack_patch_weight(weight: torch.Tensor, mode: int, transpose: bool) -> torch.Tensor:
    """GEMM A matrix [M, K] of the layer, packed [M/64][K/16][16][64].
    Conv2D weights are [cout, cin, k, k], Conv2DTranspose weights [cin, cout, k, k].  Mode 1 pads M with zero rows to a
    multiple of 64 (the kernel never stores them)."""
    if mode == 0:
        a = weight.reshape(weight.shape[0], -1)                       # [co][ci*4 + py*2 + px]
    elif mode == 1:
        a = weight[:, :, 0, 0].t() if transpose else weight[:, :, 0, 0]  # [co][ci]
        if a.shape[0] % 64:
            a = torch.cat([a, a.new_zeros(64 - a.shape[0] % 64, a.shape[1])], 0)
    else:
        a = weight.permute(1, 2, 3, 0).reshape(-1, weight.shape[0])   # [co*k*k + dy*k + dx][ci], k = 2 or 4
    m, k = a.shape
    assert m % 64 == 0 and k % 16 == 0
    return a.reshape(m // 64, 64, k // 16, 16).permute(0, 2, 3, 1).contiguous()


def patch_conv_bias_relu(x: torch.Tensor, w_packed: torch.Tensor, bias, mode: int, cout: int, out: torch.Tensor,
                         channel_offset: int = 0, relu: bool = True, w_valid: int | None = None) -> torch.Tensor:
    """Writes relu(conv(x) + bias) into out[:, channel_offset:channel_offset + cout].  w_valid (modes 2, 3 only): real
    width of a zero-padded x."""
    xx = require_gpu(x, "patch_conv_bias_relu")
    n, cin, h, w = xx.shape
    check(lib().pd3_patch_conv_bias_relu(ptr(xx), ptr(w_packed), ptr(bias), int(mode), n, cin, cout, h, w,
                                         w if w_valid is None else int(w_valid), int(bool(relu)), ptr(out),
                                         out.shape[1], int(channel_offset), stream_ptr(xx.device)),
          "patch_conv_bias_relu")
    return out


# SecondFPN's levels in fp32 arithmetic on the bf16 matrix cores (three pieces per operand, six products; round 6).
# False: the fp32-MFMA kernel everywhere.
PATCH_BF16X3 = True


def patch_x3_supported(mode: int, cin: int, cout: int, h: int, w: int) -> bool:
    """pd3_patch_conv_x3_bias_relu's shapes ([h, w] = input map, w its row pitch)."""
    if mode == 0:
        return cin % 16 == 0 and cout % 128 == 0 and h % 2 == 0 and w % 64 == 0
    if mode == 1:
        return cin % 32 == 0 and cout % 128 == 0 and (h * w) % 4 == 0
    if mode == 2:
        return cin % 32 == 0 and cout % 64 == 0 and (h * w) % 4 == 0
    return False


def split_bf16x3(a: torch.Tensor) -> torch.Tensor:
    """[...] fp32 -> [3, ...] bf16 pieces with hi + mid + lo == a exactly (round to nearest even at every cut)."""
    a = a.float()
    hi = a.bfloat16()
    r1 = a - hi.float()
    mid = r1.bfloat16()
    lo = (r1 - mid.float()).bfloat16()
    return torch.stack([hi, mid, lo])


def pack_patch_weight_x3(weight: torch.Tensor, mode: int, transpose: bool) -> torch.Tensor:
    """The layer's GEMM A matrix as bf16 pieces in the bf16x3 kernel's order: [row tile][step][16384] bf16 = per step the
    LDS image the kernel fetches as it is, [piece 3][row 128][40] (32 values of K + 8 of padding) + 1024 of padding
    (include/paddle3d_amd.h: pd3_patch_conv_x3_bias_relu).  Conv2D weights are [cout, cin, k, k], Conv2DTranspose
    weights [cin, cout, k, k]."""
    w = weight.detach().float()
    if mode == 0:
        cout, cin = int(w.shape[0]), int(w.shape[1])
        a = w.reshape(cout // 128, 128, cin // 16, 16, 2, 2).permute(0, 4, 2, 1, 3, 5)  # [mt][dy][c][row][ci][dx]
        a = a.reshape(cout // 128, 2 * (cin // 16), 128, 32)
    elif mode == 1:
        m = w[:, :, 0, 0].t() if transpose else w[:, :, 0, 0]                            # [co][ci]
        cout, cin = int(m.shape[0]), int(m.shape[1])
        a = m.reshape(cout // 128, 128, cin // 32, 32).permute(0, 2, 1, 3)
    elif mode == 2:
        cin, cout = int(w.shape[0]), int(w.shape[1])
        a = w.reshape(cin // 32, 32, cout // 64, 64, 2, 2).permute(4, 2, 0, 5, 3, 1)     # [dy][cb][st][dx][col][k]
        a = a.reshape(2 * (cout // 64), cin // 32, 128, 32)
    else:
        raise ValueError(f"pack_patch_weight_x3: mode {mode}")
    pc = split_bf16x3(a.contiguous()).permute(1, 2, 0, 3, 4)                            # [mt][step][piece][128][32]
    pc = torch.nn.functional.pad(pc, (0, 8)).reshape(pc.shape[0], pc.shape[1], 3 * 128 * 40)
    return torch.nn.functional.pad(pc, (0, 16384 - 3 * 128 * 40)).contiguous()


def patch_conv_x3_bias_relu(x: torch.Tensor, w_packed: torch.Tensor, bias, mode: int, cout: int, out: torch.Tensor,
                            channel_offset: int = 0, relu: bool = True, w_valid: int | None = None) -> torch.Tensor:
    """patch_conv_bias_relu on the bf16x3 kernel (w_packed from pack_patch_weight_x3)."""
    xx = require_gpu(x, "patch_conv_x3_bias_relu")
    n, cin, h, w = xx.shape
    check(lib().pd3_patch_conv_x3_bias_relu(ptr(xx), ptr(w_packed), ptr(bias), int(mode), n, cin, cout, h, w,
                                            w if w_valid is None else int(w_valid), int(bool(relu)), ptr(out),
                                            out.shape[1], int(channel_offset), stream_ptr(xx.device)),
          "patch_conv_x3_bias_relu")
    return out


# The stride-2 block openers in fp32 arithmetic on the bf16 matrix cores (round 6).  False: the fp32 implicit GEMM.
S2_BF16X3 = True


def conv3x3_s2_x3_supported(cin: int, cout: int, h: int, w: int, batch: int = 1) -> bool:
    """pd3_conv3x3_s2_x3_bias_relu's shapes ([h, w] = input map, w its REAL width: rows live at pitch4(w))."""
    return (cin % 16 == 0 and cout % 128 == 0 and cout <= 1024 and h % 2 == 0 and w % 2 == 0
            and batch * max(cin, cout) * h * pitch4(w) * 4 < 0x7ffffff0)


def pack_conv3x3_s2_x3_weight(weight: torch.Tensor) -> torch.Tensor:
    """[cout, cin, 3, 3] -> bf16 [cout/128][(cin/16) * 3][24576]: per step (16-channel chunk, ky) the kernel's LDS image
    [piece 3][row 128][56] with k = kx * 16 + channel (include/paddle3d_amd.h: pd3_conv3x3_s2_x3_bias_relu)."""
    w = weight.detach().float()
    cout, cin = int(w.shape[0]), int(w.shape[1])
    a = w.reshape(cout // 128, 128, cin // 16, 16, 3, 3).permute(0, 2, 4, 1, 5, 3)      # [mt][c][ky][row][kx][ci]
    a = a.reshape(cout // 128, (cin // 16) * 3, 128, 48)
    pc = split_bf16x3(a.contiguous()).permute(1, 2, 0, 3, 4)                            # [mt][step][piece][128][48]
    pc = torch.nn.functional.pad(pc, (0, 8)).reshape(pc.shape[0], pc.shape[1], 3 * 128 * 56)
    return torch.nn.functional.pad(pc, (0, 24576 - 3 * 128 * 56)).contiguous()


def conv3x3_s2_x3_bias_relu(x: torch.Tensor, w_packed: torch.Tensor, bias, cout: int, relu: bool = True,
                            w_valid: int | None = None) -> torch.Tensor:
    """x [n, cin, h, pitch] (pitch % 4 == 0, w_valid real columns, the rest zero) -> [n, cout, h / 2, pitch4(w_valid / 2)]
    with zeros in its own padding columns (conv3x3_bias_relu's convention)."""
    xx = require_gpu(x, "conv3x3_s2_x3_bias_relu")
    n, cin, h, w = xx.shape
    wv = w if w_valid is None else int(w_valid)
    out = torch.empty((n, cout, h // 2, pitch4(wv // 2)), dtype=torch.float32, device=xx.device)
    check(lib().pd3_conv3x3_s2_x3_bias_relu(ptr(xx), ptr(w_packed), ptr(bias), n, cin, cout, h, w, wv, int(bool(relu)),
                                            ptr(out), int(out.shape[3]), stream_ptr(xx.device)),
          "conv3x3_s2_x3_bias_relu")
    return out


# A Winograd layer with at least this many 64-channel output blocks over one input runs with its input transform computed
# once (csrc/conv_winograd43_ppv.hip).  Measured at 16 frames (tools/prof/prof_head_ppv.py): the pass writes 2.25 x the
# input (151 MB in 43-52 us for 64 x 128^2 or 256 x 64^2 planes) and the convolution loses 19-29 % of its time: 2 blocks
# (128 -> 128: 244 -> 114 + 194 us) do not pay for it; 4 (256 -> 256) do alone, with the input cold (248 -> 49 + 176), but
# not inside the model's step, where the layer's input is still in the caches (213 -> 52 + 173: profiles/
# r06_b16_launches.txt); the head's 18 per slice do four times over (1140 -> 910 per slice, one pass for both).  0 = never.
WINOGRAD43_PPV_MIN_BLOCKS = 8


def winograd43_input_transform(x: torch.Tensor, w_valid: int | None = None, out: torch.Tensor | None = None) -> torch.Tensor:
    """V = B^T d B of every (image, 4 x 4 tile, input channel) of x [n, cin, h, pitch], in the ping-pong kernel's LDS order."""
    xx = require_gpu(x, "winograd43_input_transform")
    n, cin, h, w = xx.shape
    floats = int(lib().pd3_winograd43_input_transform_floats(n, cin, h, w))
    if floats == 0:
        raise RuntimeError("winograd43_input_transform: cin must be a multiple of 8")
    if out is None or out.numel() < floats:
        out = torch.empty((floats,), dtype=torch.float32, device=xx.device)
    check(lib().pd3_winograd43_input_transform(ptr(xx), n, cin, h, w, w if w_valid is None else int(w_valid), ptr(out),
                                               stream_ptr(xx.device)), "winograd43_input_transform")
    return out


def conv3x3_winograd43_ppv_bias_relu(v_pre: torch.Tensor, shape, u_lane: torch.Tensor, bias, cout: int, relu: bool = True,
                                     out: torch.Tensor | None = None, w_valid: int | None = None) -> torch.Tensor:
    """conv3x3_winograd43_pp_bias_relu on the pre-transformed input v_pre = winograd43_input_transform(x), shape = x.shape."""
    n, cin, h, w = (int(v) for v in shape)
    ul = require_gpu(u_lane, "conv3x3_winograd43_ppv_bias_relu")
    if ul.numel() != cout * cin * 36:
        raise RuntimeError("conv3x3_winograd43_ppv_bias_relu: u_lane does not belong to a [cout, cin, 3, 3] weight")
    if out is None:
        out = torch.empty((n, cout, h, w), dtype=torch.float32, device=v_pre.device)
    check(lib().pd3_conv3x3_winograd43_ppv_bias_relu(ptr(v_pre), ptr(ul), ptr(bias), n, cin, cout, h, w,
                                                     w if w_valid is None else int(w_valid), int(bool(relu)), ptr(out),
                                                     stream_ptr(v_pre.device)), "conv3x3_winograd43_ppv_bias_relu")
    return out


def grouped_small_supported(cin_per_group: int, cout_per_group: int, h: int, w: int) -> bool:
    return 1 <= cout_per_group <= 4 and cin_per_group % 4 == 0 and w % 4 == 0


def pack_grouped_weight(weight: torch.Tensor, groups: int) -> torch.Tensor:
    """[groups*co, cg, 3, 3] -> [groups][cg][co][9]."""
    gco, cg = weight.shape[:2]
    co = gco // groups
    return weight.reshape(groups, co, cg, 9).permute(0, 2, 1, 3).contiguous()


def grouped_conv3x3_small(x: torch.Tensor, w_grouped: torch.Tensor, bias, groups: int,
                          out: torch.Tensor | None = None, out_groups: int | None = None,
                          out_group0: int = 0) -> torch.Tensor:
    """Grouped 3x3 convolution + bias with 1..4 output channels per group (the final SeparateHead convolutions).
    out_groups / out_group0: x, w_grouped and bias describe a slice of `groups` consecutive groups whose outputs land
    at groups [out_group0, out_group0 + groups) of `out` [n, out_groups * co, h, w]."""
    xx = require_gpu(x, "grouped_conv3x3_small")
    n, c, h, w = xx.shape
    cg, co = w_grouped.shape[1], w_grouped.shape[2]
    assert c == groups * cg
    total = groups if out_groups is None else int(out_groups)
    if out is None:
        out = torch.empty((n, total * co, h, w), dtype=torch.float32, device=xx.device)
    assert out.is_contiguous() and tuple(out.shape) == (n, total * co, h, w)
    check(lib().pd3_grouped_conv3x3_small_slice(ptr(xx), ptr(w_grouped), ptr(bias), n, groups, cg, co, h, w, ptr(out),
                                                total, int(out_group0), stream_ptr(xx.device)),
          "grouped_conv3x3_small")
    return out


# ---- mixed precision (AMP): the stride-1 3x3 layers on the fp16 matrix cores (csrc/conv_f16.hip) -----------------------
def f16_tile(cout: int) -> int:
    """Output channels per workgroup of pd3_conv3x3_f16_bias_relu: 128 where the layer allows, else 64."""
    return 128 if cout % 128 == 0 else 64


def f16_supported(cin: int, cout: int, h: int, w: int) -> bool:
    t = f16_tile(cout)
    return cin % 16 == 0 and cout % t == 0 and h > 0 and w > 0  # (border tiles of any map size are masked)


def pack_conv3x3_f16_weight(weight: torch.Tensor, tile: int | None = None) -> torch.Tensor:
    """[Cout, Cin, 3, 3] fp32 -> fp16 [Cout/T][Cin/16][9 taps (dy*3+dx)][2 halves][T co][8 ci]: a 16-channel chunk of a
    workgroup's weights is one contiguous piece in exactly its LDS order; a lane's A operand (8 consecutive ci of one co)
    is 16 aligned bytes and the 32 lanes of a channel block read 32 consecutive slots (no LDS bank conflict)."""
    cout, cin = weight.shape[:2]
    t = f16_tile(cout) if tile is None else tile
    assert weight.shape[2:] == (3, 3) and cout % t == 0 and cin % 16 == 0 and t in (64, 128)
    w = weight.to(torch.float16).reshape(cout // t, t, cin // 16, 2, 8, 9).permute(0, 2, 5, 3, 1, 4)
    return w.contiguous()


def to_f16_nhwc(x: torch.Tensor) -> torch.Tensor:
    """[n, c, h, w] fp32 NCHW -> [n, h, w, c] fp16 NHWC on the device (pd3_f32_nchw_to_f16_nhwc)."""
    xx = require_gpu(x, "f32_nchw_to_f16_nhwc")
    n, c, h, w = xx.shape
    out = torch.empty((n, h, w, c), dtype=torch.float16, device=xx.device)
    check(lib().pd3_f32_nchw_to_f16_nhwc(ptr(xx), n, c, h, w, ptr(out), stream_ptr(xx.device)), "f32_nchw_to_f16_nhwc")
    return out


def conv3x3_f16_bias_relu(x: torch.Tensor, w_packed: torch.Tensor, bias, cout: int, relu: bool = True,
                          out_f32_nchw: bool = False, out: torch.Tensor | None = None,
                          group_major: bool = False) -> torch.Tensor:
    """x [n, h, w, cin] fp16 NHWC -> [n, h, w, cout] fp16 NHWC, or (out_f32_nchw) [n, cout, h, w] fp32 NCHW, or
    (group_major) [n, cout / 64, h, w, 64] fp16: the head's branches one after the other."""
    if x.dtype != torch.float16 or not x.is_cuda or not x.is_contiguous():
        raise RuntimeError("conv3x3_f16_bias_relu: x must be a contiguous fp16 NHWC tensor on the GPU")
    n, h, w, cin = x.shape
    tile = int(w_packed.shape[4])
    if out is None:
        out = (torch.empty((n, cout, h, w), dtype=torch.float32, device=x.device) if out_f32_nchw
               else torch.empty((n, cout // 64, h, w, 64) if group_major else (n, h, w, cout), dtype=torch.float16,
                                device=x.device))
    assert not (group_major and out_f32_nchw)
    check(lib().pd3_conv3x3_f16_bias_relu(ptr(x), ptr(w_packed), ptr(bias), n, cin, cout, h, w, int(bool(relu)),
                                          ptr(out), 1 if out_f32_nchw else (3 if group_major else 0), tile,
                                          stream_ptr(x.device)),
          "conv3x3_f16_bias_relu")
    return out


def conv3x3_f16_bias_relu_dual(x: torch.Tensor, w_packed: torch.Tensor, bias, cout: int, relu: bool = True):
    """conv3x3_f16_bias_relu leaving BOTH forms of the result: ([n, h, w, cout] fp16 NHWC, [n, cout, h, w] fp32 NCHW)."""
    if x.dtype != torch.float16 or not x.is_cuda or not x.is_contiguous():
        raise RuntimeError("conv3x3_f16_bias_relu: x must be a contiguous fp16 NHWC tensor on the GPU")
    n, h, w, cin = x.shape
    tile = int(w_packed.shape[4])
    oh = torch.empty((n, h, w, cout), dtype=torch.float16, device=x.device)
    of = torch.empty((n, cout, h, w), dtype=torch.float32, device=x.device)
    check(lib().pd3_conv3x3_f16_bias_relu_dual(ptr(x), ptr(w_packed), ptr(bias), n, cin, cout, h, w, int(bool(relu)),
                                               ptr(oh), ptr(of), tile, stream_ptr(x.device)),
          "conv3x3_f16_bias_relu_dual")
    return oh, of


def s2_f16_supported(cin: int, cout: int) -> bool:
    return cin % 16 == 0 and cout % 128 == 0


def conv3x3_s2_f16_bias_relu(x: torch.Tensor, w_packed: torch.Tensor, bias, cout: int, relu: bool = True) -> torch.Tensor:
    """Stride-2 3x3 / pad 1 convolution on fp16 NHWC: x [n, h, w, cin] -> [n, (h - 1) // 2 + 1, (w - 1) // 2 + 1, cout]
    fp16 NHWC; w_packed = pack_conv3x3_f16_weight(weight, tile=128)."""
    if x.dtype != torch.float16 or not x.is_cuda or not x.is_contiguous():
        raise RuntimeError("conv3x3_s2_f16_bias_relu: x must be a contiguous fp16 NHWC tensor on the GPU")
    n, h, w, cin = x.shape
    assert int(w_packed.shape[4]) == 128
    out = torch.empty((n, (h - 1) // 2 + 1, (w - 1) // 2 + 1, cout), dtype=torch.float16, device=x.device)
    check(lib().pd3_conv3x3_s2_f16_bias_relu(ptr(x), ptr(w_packed), ptr(bias), n, cin, cout, h, w, int(bool(relu)),
                                             ptr(out), stream_ptr(x.device)), "conv3x3_s2_f16_bias_relu")
    return out


def scatter_conv_s2_f16_supported(cin: int, cout: int, ny: int, nx: int) -> bool:
    return cin % 16 == 0 and cout % 64 == 0 and ny % 2 == 0 and nx % 2 == 0


def scatter_conv3x3_s2_f16_bias_relu(canvas, w_packed: torch.Tensor, bias, cout: int, relu: bool = True) -> torch.Tensor:
    """scatter_conv3x3_bias_relu on the fp16 matrix cores: a SparseCanvas (pillar features converted to fp16 once, the
    inverse map) -> [B, ny / 2, nx / 2, cout] fp16 NHWC; w_packed = pack_conv3x3_f16_weight(weight, tile=64 | 128)."""
    f, inv = canvas.features, canvas.inv
    fh = f if f.dtype == torch.float16 else f.half()
    n, cin, ny, nx = canvas.shape
    tile = int(w_packed.shape[4])
    out = torch.empty((n, ny // 2, nx // 2, cout), dtype=torch.float16, device=f.device)
    check(lib().pd3_scatter_conv3x3_s2_f16_bias_relu(ptr(fh), ptr(inv), ptr(w_packed), ptr(bias), n, cin, cout, ny, nx,
                                                     int(bool(relu)), ptr(out), tile, stream_ptr(f.device)),
          "scatter_conv3x3_s2_f16_bias_relu")
    return out


def pack_grouped_weight_f16(weight: torch.Tensor, groups: int) -> torch.Tensor:
    """[groups * co, 64, 3, 3] fp32 -> fp16 [groups][9 taps (dy*3+dx)][co][64]: the operand order of
    grouped_conv3x3_small_f16."""
    gco, cg = weight.shape[:2]
    co = gco // groups
    assert weight.shape[2:] == (3, 3) and cg == 64 and gco == groups * co and 1 <= co <= 4
    return weight.to(torch.float16).reshape(groups, co, cg, 9).permute(0, 3, 1, 2).contiguous()


def grouped_conv3x3_small_f16(x_h: torch.Tensor, w_f16: torch.Tensor, bias, groups: int,
                              out: torch.Tensor | None = None, out_groups: int | None = None,
                              out_group0: int = 0, group_major: bool = False) -> torch.Tensor:
    """grouped_conv3x3_small on the first stage's fp16 output: x_h [n, h, w, groups * 64] fp16 NHWC -- or (group_major)
    [n, groups, h, w, 64], what conv3x3_f16_bias_relu(..., group_major=True) leaves -- -> fp32 NCHW maps
    [n, out_groups * co, h, w] (this slice's groups at [out_group0, out_group0 + groups))."""
    if x_h.dtype != torch.float16 or not x_h.is_cuda or not x_h.is_contiguous():
        raise RuntimeError("grouped_conv3x3_small_f16: x must be a contiguous fp16 tensor on the GPU")
    if group_major:
        n, gg, h, w, c64 = x_h.shape
        c = gg * c64
        assert c64 == 64
    else:
        n, h, w, c = x_h.shape
    co = int(w_f16.shape[2])
    assert c == groups * 64 and tuple(w_f16.shape) == (groups, 9, co, 64)
    total = groups if out_groups is None else int(out_groups)
    if out is None:
        out = torch.empty((n, total * co, h, w), dtype=torch.float32, device=x_h.device)
    assert out.is_contiguous() and tuple(out.shape) == (n, total * co, h, w)
    fn = lib().pd3_grouped_conv3x3_small_f16_gm if group_major else lib().pd3_grouped_conv3x3_small_f16
    check(fn(ptr(x_h), ptr(w_f16.contiguous()), ptr(bias), n, groups, 64, co, h, w, ptr(out), total, int(out_group0),
             stream_ptr(x_h.device)), "grouped_conv3x3_small_f16")
    return out


# ---- F(4x4, 3x3) as the ping-pong kernel fed by LDS alone (round 4) ------------------------------------------------------
WINOGRAD43_PP_MIN_CIN = 64  # from here on the ping-pong form beats the packed one (measured per layer shape)


def winograd43_pp_supported(cin: int, cout: int, h: int, w: int) -> bool:
    return cin % 8 == 0 and cout % 64 == 0 and w % 4 == 0


def pack_winograd43_lane_weight(weight: torch.Tensor) -> torch.Tensor:
    """[Cout, Cin, 3, 3] -> U = G g G^T (the values of pack_winograd43_weight) in the lane order of the ping-pong kernel:
    [Cout/64][Cin/8][2 trips][4 blocks][9][64 lanes][4], lane = 16 (ci % 4) + (co % 16)."""
    cout, cin = weight.shape[:2]
    assert weight.shape[2:] == (3, 3) and cout % 64 == 0 and cin % 8 == 0
    g = torch.tensor([[1 / 4, 0, 0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6], [1 / 24, 1 / 12, 1 / 6],
                      [1 / 24, -1 / 12, 1 / 6], [0, 0, 1]], dtype=torch.float64, device=weight.device)
    u = torch.einsum("ij,ocjk,lk->ocil", g, weight.double(), g).float()  # [cout, cin, 6, 6]
    #          ct          cb 16co  slot      trip ci4 q  j
    u = u.reshape(cout // 64, 4, 16, cin // 8, 2, 4, 9, 4).permute(0, 3, 4, 1, 6, 5, 2, 7)
    return u.contiguous()


def conv3x3_winograd43_pp_bias_relu(x: torch.Tensor, u_lane: torch.Tensor, bias, cout: int, relu: bool = True,
                                    out: torch.Tensor | None = None, w_valid: int | None = None) -> torch.Tensor:
    """x [n, cin, h, pitch] fp32, u_lane from pack_winograd43_lane_weight -> [n, cout, h, pitch]."""
    xx = require_gpu(x, "conv3x3_winograd43_pp_bias_relu")
    ul = require_gpu(u_lane, "conv3x3_winograd43_pp_bias_relu")
    n, cin, h, w = xx.shape
    if ul.numel() != cout * cin * 36:
        raise RuntimeError("conv3x3_winograd43_pp_bias_relu: u_lane does not belong to a [cout, cin, 3, 3] weight")
    if out is None:
        out = torch.empty((n, cout, h, w), dtype=torch.float32, device=xx.device)
    check(lib().pd3_conv3x3_winograd43_pp_bias_relu(ptr(xx), ptr(ul), ptr(bias), n, cin, cout, h, w,
                                                    w if w_valid is None else int(w_valid), int(bool(relu)),
                                                    ptr(out), stream_ptr(xx.device)),
          "conv3x3_winograd43_pp_bias_relu")
    return out
