"""CenterPoint / PointPillars inference graph over the HIP ops -- the host-side mirror of the reference's
forward path, with the reference's layer names, constructor arguments and parameter names so that the
forward graph (and a converted ``.pdparams`` state dict) drops in unchanged.

Reference call stack: CenterPoint.test_forward (paddle3d/models/detection/centerpoint/
centerpoint.py:155-166) -> extract_feat (:126-138) -> HardVoxelizer (models/voxelizers/voxelize.py:27-82)
-> PillarFeatureNet (models/voxel_encoders/pillar_encoder.py:64-210) -> PointPillarsScatter
(models/middle_encoders/pillar_scatter.py:34-105) -> SecondBackbone (models/backbones/second_backbone.py:
72-120) -> SecondFPN (models/necks/second_fpn.py:99-157) -> CenterHead.forward (center_head.py:212-220)
-> CenterHead.predict_by_custom_op (:294-339).

What differs from the reference, by design:
  * voxelize / PFN / scatter / postprocess are single fused HIP ops taking the WHOLE batch (the reference
    loops over samples in Python and syncs on num_voxels per sample, voxelize.py:43);
  * padded rows beyond num_voxels are never sliced off on the host: fixed-shape [B, V, ...] tensors flow
    through PFN and scatter, which ignore rows >= num_voxels through the batch column -1;
  * the dense 2-D convolutions run on the library's own fp32-MFMA kernels (Winograd F(4x4,3x3) for the
    stride-1 3x3 layers, an implicit-GEMM kernel for the stride-2 ones, a patch-GEMM kernel for the FPN levels
    writing into the concatenated map, a grouped kernel for the final head convolutions), with BatchNorm
    folded; the 36 first-stage head convolutions that read the same shared feature map are issued as ONE
    convolution.  There is no other backend: a shape no kernel takes raises Paddle3DAmdError (status -3), and
    the layers refuse to run in training mode (the torch statement of the same layers that the tests compare
    against lives in oracle/pyoracle.py).
torch is plumbing here (device memory, streams, parameter containers); the work is in libpaddle3d_amd.so.
"""
from __future__ import annotations

import math

import numpy as np
import torch
import torch.nn as nn

from ._lib import Paddle3DAmdError
from .ops import centerpoint_postprocess as _cp
from .ops import conv as _conv
from .ops import pointpillars_scatter as _ps
from .ops import sparse_conv3d as _sp3
from .ops import voxel_encoder as _ve
from .ops import voxelize as _vox

__all__ = ["HardVoxelizer", "PillarFeatureNet", "HardVFE", "VoxelMean", "PointPillarsScatter", "SecondBackbone",
           "SecondFPN", "CenterHead", "CenterPoint", "centerpoint_pillars_nuscenes", "centerpoint_pillars_kitti",
           "centerpoint_voxels_nuscenes", "centerpoint_voxels_kitti", "load_paddle_state_dict"]


def _grid(voxel_size, point_cloud_range):
    pr = np.array(point_cloud_range, dtype=np.float32)
    vs = np.array(voxel_size, dtype=np.float32)
    return np.round((pr[3:] - pr[:3]) / vs).astype(np.int64)  # pillar_scatter.py:46-51


class HardVoxelizer(nn.Module):
    """models/voxelizers/voxelize.py:27-82.  forward(points [B, N, D]) -> (voxels [B,V,P,D],
    coors [B,V,4] int32 (batch, z, y, x; batch = -1 on padding rows), num_points [B,V], num_voxels [B])."""

    def __init__(self, voxel_size, point_cloud_range, max_num_points_in_voxel, max_num_voxels, path: int = 0):
        super().__init__()
        self.path = int(path)  # pd3_hard_voxelize_path selector (0 = the library's choice)
        self.voxel_size = list(map(float, voxel_size))
        self.point_cloud_range = list(map(float, point_cloud_range))
        self.max_num_points_in_voxel = int(max_num_points_in_voxel)
        if isinstance(max_num_voxels, (tuple, list)):
            self.max_num_voxels = list(max_num_voxels)
        else:
            self.max_num_voxels = [max_num_voxels, max_num_voxels]

    def forward(self, points: torch.Tensor, num_points: torch.Tensor | None = None):
        v = self.max_num_voxels[0] if self.training else self.max_num_voxels[1]
        # the batch column (voxelize.py:51-57 builds it through a float cast + F.pad) is written by the op
        # itself; -1 marks padding rows
        voxels, _, npv, nv, coors = _vox.hard_voxelize_batch(points, self.voxel_size, self.point_cloud_range,
                                                             self.max_num_points_in_voxel, v, num_points,
                                                             with_batch_coors=True, path=self.path)
        return voxels, coors, npv, nv

    def index(self, points: torch.Tensor, num_points: torch.Tensor | None = None):
        """The same voxels as forward() as an INDEX of the points instead of padded copies of them: (vox_span
        [B, V, 2], point_list, coors [B, V, 4], num_points [B, V], num_voxels [B]), or None where the voxelizer's
        wave forms do not serve the grid (ops.voxelize.hard_voxelize_index_batch)."""
        if self.path != 0:
            return None  # a forced path is a measurement / test form of the full operator
        v = self.max_num_voxels[0] if self.training else self.max_num_voxels[1]
        res = _vox.hard_voxelize_index_batch(points, self.voxel_size, self.point_cloud_range,
                                             self.max_num_points_in_voxel, v, num_points)
        if res is None:
            return None
        span, plist, _, npv, nv, coors = res
        return span, plist, coors, npv, nv


class PFNLayer(nn.Module):
    """pillar_encoder.py:64-105; holds parameters only -- the arithmetic runs inside the fused HIP op."""

    def __init__(self, in_channels, out_channels, last_layer=False):
        super().__init__()
        self.last_vfe = last_layer
        self.units = out_channels if last_layer else out_channels // 2
        self.linear = nn.Linear(in_channels, self.units, bias=False)
        self.norm = nn.BatchNorm1d(self.units, eps=1e-3, momentum=0.01)
        bound = 1 / math.sqrt(in_channels)
        nn.init.uniform_(self.linear.weight, -bound, bound)


class PillarFeatureNet(nn.Module):
    """pillar_encoder.py:108-210 (legacy=False, with_distance=False).  The folded parameters are rebuilt on
    every call in training mode and whenever the module is moved / reloaded (see _InferenceCache below)."""

    def __init__(self, in_channels=4, feat_channels=(64,), with_distance=False, max_num_points_in_voxel=20,
                 voxel_size=(0.2, 0.2, 4), point_cloud_range=(0, -40, -3, 70.4, 40, 1), legacy=False):
        super().__init__()
        if with_distance or legacy:
            raise NotImplementedError("with_distance / legacy PFN variants are not on the hot path")
        if len(feat_channels) not in (1, 2):
            raise NotImplementedError("the fused PFN op covers one or two PFN layers")
        self.in_channels = in_channels
        chans = [in_channels + 5] + list(feat_channels)
        self.pfn_layers = nn.ModuleList(
            PFNLayer(chans[i], chans[i + 1], last_layer=(i == len(chans) - 2)) for i in range(len(chans) - 1))
        # layer i+1 consumes [y_i | max(y_i)]: 2 * units_i inputs
        for i in range(1, len(self.pfn_layers)):
            assert self.pfn_layers[i].linear.in_features == 2 * self.pfn_layers[i - 1].units
        self.vx, self.vy = float(voxel_size[0]), float(voxel_size[1])
        self.x_offset = self.vx / 2 + point_cloud_range[0]
        self.y_offset = self.vy / 2 + point_cloud_range[1]
        self.max_num_points_in_voxel = max_num_points_in_voxel
        self._folded = None

    def train(self, mode: bool = True):
        self._folded = None
        return super().train(mode)

    def _fold(self):
        out = []
        for l in self.pfn_layers:
            scale, shift = _ve.fold_batchnorm(l.norm.weight, l.norm.bias, l.norm.running_mean,
                                              l.norm.running_var, l.norm.eps)
            out += [l.linear.weight.t().contiguous(), scale, shift]  # torch [out,in] -> Paddle [in,out]
        return [t.detach() for t in out]

    def forward(self, features, num_points_per_voxel, coors):
        """features [M,P,D], num_points [M] int32, coors [M,4] int32 -> [M, C]."""
        sig = _param_signature(self)
        if self.training or self._folded is None or self._folded[0] != sig:
            self._folded = (sig, self._fold())
        return _ve.pillar_feature_net(features, num_points_per_voxel, coors, self.vx, self.vy, self.x_offset,
                                      self.y_offset, *self._folded[1])

    def forward_indexed(self, points, vox_span, point_list, coors):
        """forward() reading the pillars' points through the voxelizer's index (HardVoxelizer.index) from the point
        cloud [B, N, D] itself; None for shapes the indexed kernel does not serve.  Same bytes as forward()."""
        if len(self.pfn_layers) != 2:
            return None
        sig = _param_signature(self)
        if self.training or self._folded is None or self._folded[0] != sig:
            self._folded = (sig, self._fold())
        return _ve.pillar_feature_net_indexed(points, vox_span, point_list, coors, self.max_num_points_in_voxel,
                                              self.vx, self.vy, self.x_offset, self.y_offset, *self._folded[1])


class HardVFE(nn.Module):
    """voxel_encoder.py:142-283 with with_cluster_center = with_voxel_center = True, two VFE layers (the
    BEVFusion LiDAR stream, configs/bevfusion/bevf_pp_2x8_1x_nusc.yaml:93-101); parameters only, arithmetic
    in the fused HIP op."""

    def __init__(self, in_channels=4, feat_channels=(64, 64), with_distance=False, with_cluster_center=True,
                 with_voxel_center=True, voxel_size=(0.2, 0.2, 4), point_cloud_range=(0, -40, -3, 70.4, 40, 1)):
        super().__init__()
        if with_distance or not (with_cluster_center and with_voxel_center) or len(feat_channels) != 2:
            raise NotImplementedError("HardVFE: only the BEVFusion configuration is on the hot path")
        self.voxel_size, self.point_cloud_range = tuple(voxel_size), tuple(point_cloud_range)
        chans = [in_channels + 6] + list(feat_channels)

        class _VFE(nn.Module):
            def __init__(self, cin, cout):
                super().__init__()
                self.linear = nn.Linear(cin, cout, bias=False)
                self.norm = nn.BatchNorm1d(cout, eps=1e-3, momentum=0.01)

        self.vfe_layers = nn.ModuleList([_VFE(chans[0], chans[1]), _VFE(2 * chans[1], chans[2])])

    def forward(self, features, num_points, coors):
        args = []
        for l in self.vfe_layers:
            scale, shift = _ve.fold_batchnorm(l.norm.weight, l.norm.bias, l.norm.running_mean, l.norm.running_var,
                                              l.norm.eps)
            args += [l.linear.weight.t().contiguous().detach(), scale.detach(), shift.detach()]
        return _ve.hard_vfe(features, num_points, coors, self.voxel_size, self.point_cloud_range, *args)


class VoxelMean(nn.Module):
    """voxel_encoder.py:44-57: mean of the points of a voxel."""

    def __init__(self, in_channels=4):
        super().__init__()
        self.in_channels = in_channels

    def forward(self, features, num_points_per_voxel, coors=None):
        assert self.in_channels == features.shape[-1]
        return _ve.voxel_mean(features, num_points_per_voxel)


class PointPillarsScatter(nn.Module):
    """pillar_scatter.py:34-105."""

    def __init__(self, in_channels, voxel_size, point_cloud_range):
        super().__init__()
        self.in_channels = in_channels
        g = _grid(voxel_size, point_cloud_range)
        self.nx, self.ny = int(g[0]), int(g[1])

    def forward(self, voxel_features, coords, batch_size):
        """The reference's contract: the dense [B, C, ny, nx] pseudo image, always."""
        return _ps.pointpillars_scatter(voxel_features, coords, batch_size, self.ny, self.nx)

    def sparse(self, voxel_features, coords, batch_size):
        """The same scatter as a SparseCanvas (pillar features + inverse map) for a consumer that gathers from the
        pillar features itself (SecondBackbone's first convolution): no canvas is written.  The models call this
        when they fuse (`model.scatter`); the module's own forward keeps returning a tensor for everyone else."""
        return _ps.SparseCanvas(voxel_features, coords, batch_size, self.ny, self.nx)


def _param_signature(module):
    """(storage, version, device) of every parameter and buffer below `module`: changes when any of them is
    reloaded, moved or written in place THROUGH THE TENSOR ITSELF (`load_state_dict` on this module or a child,
    `.to()`, `with torch.no_grad(): p.mul_(2)`, an optimizer step).  It does NOT see a write through `p.data`
    (`p.data.copy_(...)`, `p.data.mul_(2)`): `.data` is a detached alias with a version counter of its own, and
    reading the contents instead would cost a device -> host round trip on every forward.  Code that writes through
    `.data` (or through a raw pointer) calls `invalidate_derived(model)` afterwards."""
    return tuple((t.data_ptr(), t._version, str(t.device)) for t in list(module.parameters()) + list(module.buffers()))


def invalidate_derived(module):
    """Drop every weight derived for inference (folded BatchNorm, packed kernel layouts) below `module`; the next
    forward rebuilds them from the current parameters.  Needed only after a write the signature cannot see
    (`param.data.*`, see _param_signature); `load_state_dict`, `.to()`, `train()` and in-place tensor ops are
    detected without it."""
    for m in module.modules():
        if hasattr(m, "_drop_cache"):
            m._drop_cache()
        if hasattr(m, "_folded"):
            m._folded = None
        if getattr(m, "_pd3_folded", None) is not None:
            object.__setattr__(m, "_pd3_folded", None)
    return module


class _InferenceCache:
    """Mixin of the parameter-holding layers: weights derived for inference (BatchNorm folded, packed in the
    kernels' layouts) live in ``self._cache`` together with the signature of the tensors they were derived from
    and are rebuilt whenever that signature differs (a reload of this module or of a child, ``.to()``, an in-place
    write) -- checked on every forward, a few microseconds; ``train()`` drops them outright."""

    _cache = None
    _cache_sig = None

    def _drop_cache(self):
        self.__dict__["_cache"] = None
        self.__dict__["_cache_sig"] = None

    def _cache_valid(self):
        return self._cache is not None and self._cache_sig == _param_signature(self)

    def _store_cache(self, value):
        self.__dict__["_cache"] = value
        self.__dict__["_cache_sig"] = _param_signature(self)
        return value

    def train(self, mode: bool = True):
        self._drop_cache()
        return super().train(mode)

    def invalidate(self):
        """Public form of _drop_cache for this module and everything below it (see invalidate_derived)."""
        invalidate_derived(self)
        return self

    def _require_eval(self):
        if self.training:
            raise RuntimeError(f"{type(self).__name__}: paddle3d_amd implements the inference path only "
                               "(BatchNorm folded into the HIP kernels); call .eval() first")


def _conv_bn_relu(cin, cout, k, stride=1, padding=0, transpose=False, eps=1e-3, momentum=0.01, bias=False):
    conv = (nn.ConvTranspose2d if transpose else nn.Conv2d)(cin, cout, k, stride=stride, padding=padding, bias=bias)
    return [conv, nn.BatchNorm2d(cout, eps=eps, momentum=momentum), nn.ReLU()]


def _fold_conv_bn(conv: nn.Conv2d, bn: nn.BatchNorm2d):
    scale = bn.weight / torch.sqrt(bn.running_var + bn.eps)
    if isinstance(conv, nn.ConvTranspose2d):
        w = conv.weight * scale.reshape(1, -1, 1, 1)
    else:
        w = conv.weight * scale.reshape(-1, 1, 1, 1)
    b = (conv.bias if conv.bias is not None else torch.zeros_like(bn.running_mean))
    return w.detach().contiguous(), ((b - bn.running_mean) * scale + bn.bias).detach().contiguous()


class _Conv3x3:
    """One folded 3x3 / pad 1 convolution + bias + ReLU on the library's kernels: stride 1 by Winograd F(4x4,3x3),
    otherwise (and for stride 2) the implicit-GEMM kernel.  A shape no kernel takes raises (PD3_EUNSUPPORTED);
    nothing falls back to another backend.  Packed weights are kept next to the folded ones."""

    def __init__(self, w, b, stride):
        self.w, self.b, self.stride = w, b, int(stride)
        self.cout, self.cin = int(w.shape[0]), int(w.shape[1])
        self.packed = {}

    def __call__(self, x, w_valid=None):
        """x [n, cin, h, pitch]; w_valid = its real width (default pitch; a width that is not a multiple of 4 lives in
        zero-padded rows, ops/conv.py:pitch4).  Returns (y, real width of y)."""
        h, wv = int(x.shape[2]), int(x.shape[3] if w_valid is None else w_valid)
        if (self.stride == 1 and self.cin >= _conv.WINOGRAD43_PP_MIN_CIN
                and _conv.winograd43_pp_supported(self.cin, self.cout, h, int(x.shape[3]))):
            # the ping-pong form: the same bytes as the packed form, 2-9 % faster on this model's layers
            if "w43pp" not in self.packed:
                self.packed["w43pp"] = _conv.pack_winograd43_lane_weight(self.w)
            if 0 < _conv.WINOGRAD43_PPV_MIN_BLOCKS <= self.cout // 64:
                # enough channel blocks over this input to compute its Winograd transform once for all of them (round 6)
                vpre = _conv.winograd43_input_transform(x, w_valid=wv)
                return _conv.conv3x3_winograd43_ppv_bias_relu(vpre, x.shape, self.packed["w43pp"], self.b, self.cout,
                                                              relu=True, w_valid=wv), wv
            return _conv.conv3x3_winograd43_pp_bias_relu(x, self.packed["w43pp"], self.b, self.cout, relu=True,
                                                         w_valid=wv), wv
        if self.stride == 1 and _conv.winograd43_supported(self.cin, self.cout, h, wv):
            if "w43" not in self.packed:
                self.packed["w43"] = _conv.pack_winograd43_weight(self.w)
            return _conv.conv3x3_winograd43_bias_relu(x, self.packed["w43"], self.b, self.cout, relu=True,
                                                      w_valid=wv), wv
        if (self.stride == 2 and _conv.S2_BF16X3 and int(x.shape[3]) % 4 == 0
                and _conv.conv3x3_s2_x3_supported(self.cin, self.cout, h, wv, int(x.shape[0]))):
            # fp32 arithmetic on the bf16 matrix cores (three pieces per operand, csrc/conv_s2_x3.hip)
            if "s2x3" not in self.packed:
                self.packed["s2x3"] = _conv.pack_conv3x3_s2_x3_weight(self.w)
            return _conv.conv3x3_s2_x3_bias_relu(x, self.packed["s2x3"], self.b, self.cout, relu=True,
                                                 w_valid=wv), wv // 2
        if _conv.supported(self.cin, self.cout, h, wv, self.stride):
            if "direct" not in self.packed:
                self.packed["direct"] = _conv.pack_conv3x3_weight(self.w)
            return _conv.conv3x3_bias_relu(x, self.packed["direct"], self.b, self.cout, relu=True, stride=self.stride,
                                           w_valid=wv), wv // self.stride
        raise Paddle3DAmdError(f"conv3x3: unsupported configuration (cin {self.cin}, cout {self.cout}, stride "
                               f"{self.stride}, input {h}x{wv}) (status -3)")

    def f16_ok(self, h, w):
        return self.stride == 1 and _conv.f16_supported(self.cin, self.cout, int(h), int(w))

    def f16_s2_ok(self):
        return self.stride == 2 and _conv.s2_f16_supported(self.cin, self.cout)

    def f16_s2(self, x_h):
        """The stride-2 layer in mixed precision: x_h [n, h, w, cin] fp16 NHWC -> fp16 NHWC at half the size."""
        if "f16s2" not in self.packed:
            self.packed["f16s2"] = _conv.pack_conv3x3_f16_weight(self.w, tile=128)
        return _conv.conv3x3_s2_f16_bias_relu(x_h, self.packed["f16s2"], self.b, self.cout, relu=True)

    def f16_dual(self, x_h):
        """f16() leaving both forms of the result: (fp16 NHWC, fp32 NCHW)."""
        if "f16" not in self.packed:
            self.packed["f16"] = _conv.pack_conv3x3_f16_weight(self.w)
        return _conv.conv3x3_f16_bias_relu_dual(x_h, self.packed["f16"], self.b, self.cout, relu=True)

    def f16(self, x_h, out_f32_nchw=False, out=None, tiles=None, group_major=False):
        """The same layer in mixed precision (AMP): x_h [n, h, w, cin] fp16 NHWC -> fp16 NHWC, or fp32 NCHW for the
        fp32 kernels behind a chain.  tiles = (first, last) channel tile of the packed weight (the head's slices)."""
        if "f16" not in self.packed:
            self.packed["f16"] = _conv.pack_conv3x3_f16_weight(self.w)
        wp, b, cout = self.packed["f16"], self.b, self.cout
        if tiles is not None:
            t = int(wp.shape[4])  # [cout / T][cin / 16][9][2][T][8]
            wp, b, cout = wp[tiles[0]:tiles[1]], self.b[tiles[0] * t:tiles[1] * t], (tiles[1] - tiles[0]) * t
        return _conv.conv3x3_f16_bias_relu(x_h, wp, b, cout, relu=True, out_f32_nchw=out_f32_nchw, out=out,
                                           group_major=group_major)


def _valid_w(t) -> int:
    """Real width of a feature map (its rows may be zero-padded to a multiple of 4)."""
    return int(getattr(t, "_pd3_valid_w", t.shape[3]))


def _f16_nhwc_to_f32_nchw(xh):
    """fp16 NHWC [n, h, w, c] -> the fp32 kernels' form: fp32 NCHW with rows zero-padded to a multiple of 4 (pitch4) and
    the real width tagged -- a 90-wide fp16 stage must not reach an fp32 kernel as rows of 90."""
    x = xh.permute(0, 3, 1, 2).float()
    w = int(x.shape[3])
    pad = _conv.pitch4(w) - w
    if pad:
        x = torch.nn.functional.pad(x, (0, pad))
    return _tag_valid_w(x.contiguous(), w)


def _tag_valid_w(t, wv):
    if t.dtype == torch.float16:  # fp16 NHWC stage outputs: rows are exactly as wide as the map (dim 3 = channels)
        return t
    if wv != t.shape[3]:
        t._pd3_valid_w = int(wv)
    return t


def _fold_conv3x3(conv, bn):
    if (isinstance(conv, nn.ConvTranspose2d) or tuple(conv.kernel_size) != (3, 3) or tuple(conv.padding) != (1, 1)
            or tuple(conv.stride) not in ((1, 1), (2, 2))):
        raise Paddle3DAmdError(f"unsupported configuration: {conv} is not a 3x3 / pad 1 / stride 1|2 convolution")
    w, b = _fold_conv_bn(conv, bn)
    return _Conv3x3(w, b, conv.stride[0])


class SecondBackbone(_InferenceCache, nn.Module):
    """second_backbone.py:72-120 (parameter names blocks.<i>.<j>.* as in the reference).  The modules hold the
    parameters; forward runs every convolution (BatchNorm folded, ReLU fused) on the library's kernels."""

    def __init__(self, in_channels=128, out_channels=(128, 128, 256), layer_nums=(3, 5, 5),
                 downsample_strides=(2, 2, 2)):
        super().__init__()
        in_filters = [in_channels, *out_channels[:-1]]
        blocks = []
        for i, n in enumerate(layer_nums):
            block = _conv_bn_relu(in_filters[i], out_channels[i], 3, stride=downsample_strides[i], padding=1)
            for _ in range(n):
                block += _conv_bn_relu(out_channels[i], out_channels[i], 3, padding=1)
            blocks.append(nn.Sequential(*block))
        self.blocks = nn.ModuleList(blocks)
        # the scatter-fused first layer as a sparse convolution over the occupied pillars (round 6); False = the dense
        # implicit-GEMM kernel of round 3 (bit-identical to scatter + convolution; the sparse form differs from it by the
        # summation order, ~1e-6 relative)
        self.sparse_first = True
        self.amp = False  # True: the stride-1 layers run on the fp16 matrix cores (the reference's amp_cfg O2 configs)
        self.amp_out_f16 = False  # under AMP the stage outputs leave as fp16 NHWC (set by CenterPoint.set_amp when the
        #                           neck reads that form: no fp32 NCHW copy of a stage is written at all)

    def _plan(self):
        if not self._cache_valid():
            plan = []
            for blk in self.blocks:
                mods = list(blk)
                plan.append([_fold_conv3x3(mods[i], mods[i + 1]) for i in range(0, len(mods), 3)])
            self._store_cache(plan)
        return self._cache

    def forward(self, x):
        """-> the block outputs.  A stage whose width is not a multiple of 4 (CenterPoint-Voxel: 90) comes back in
        zero-padded rows, tagged with its real width (_valid_w); SecondFPN reads the tag."""
        self._require_eval()
        plan = self._plan()
        first = None
        first_h = None  # mixed precision: the scatter-fused first layer's result as fp16 NHWC
        if isinstance(x, _ps.SparseCanvas):
            # PointPillarsScatter fused into the first convolution where the kernel takes it (stride 2), else written out
            c0 = plan[0][0]
            if (self.amp and c0.stride == 2 and x.shape[1] == c0.cin and len(plan[0]) > 1
                    and _conv.scatter_conv_s2_f16_supported(c0.cin, c0.cout, x.ny, x.nx)
                    and plan[0][1].f16_ok(x.ny // 2, x.nx // 2)):
                # (round 5) on the fp16 matrix cores as well: the only fp32 layer of the AMP graph and its conversion go
                key = "f16s2_64" if c0.cout % 128 else "f16s2"
                if key not in c0.packed:
                    c0.packed[key] = _conv.pack_conv3x3_f16_weight(c0.w, tile=64 if c0.cout % 128 else 128)
                first_h = _conv.scatter_conv3x3_s2_f16_bias_relu(x, c0.packed[key], c0.b, c0.cout)
                first = (None, x.nx // 2)
            elif (self.sparse_first and x.shape[1] == c0.cin
                  and _conv.scatter_conv_sparse_supported(c0.cin, c0.cout, x.ny, x.nx, c0.stride)):
                # (round 6) the first layer as a SPARSE convolution over the occupied pillars: a nuScenes canvas is 11 %
                # occupied and the dense kernel multiplies 8.7 x more products than exist (csrc/pillar_conv.hip)
                y, c0.packed["x3sparse"] = _conv.scatter_conv3x3_sparse(x, c0.w, c0.b, c0.packed.get("x3sparse"))
                first = (y, x.nx // 2)
            elif _conv.scatter_conv_supported(c0.cin, c0.cout, x.ny, x.nx, c0.stride) and x.shape[1] == c0.cin:
                if "direct" not in c0.packed:
                    c0.packed["direct"] = _conv.pack_conv3x3_weight(c0.w)
                first = (_conv.scatter_conv3x3_bias_relu(x, c0.packed["direct"], c0.b, c0.cout), x.nx // 2)
            else:
                x = x.dense()
        if first is None and x.shape[3] % 4:
            raise Paddle3DAmdError(f"SecondBackbone: unsupported configuration (input width {x.shape[3]} is not a "
                                   "multiple of 4) (status -3)")
        outs, wv = [], (0 if first is not None else int(x.shape[3]))
        carry = None  # mixed precision: the previous block's result as fp16 NHWC, for this block's stride-2 convolution
        for bi, layers in enumerate(plan):
            li = 0
            while li < len(layers):
                conv = layers[li]
                xh = None
                if first is not None and bi == 0 and li == 0:
                    x, wv = first
                    li += 1
                    if first_h is None:
                        continue
                    xh = first_h  # fp16 NHWC already: the run below starts without a conversion
                    conv = layers[li]
                if self.amp and li == 0 and carry is not None and conv.f16_s2_ok() and wv == carry.shape[2]:
                    # (round 5) the block opens on the fp16 matrix cores too: no fp32 stride-2 kernel, no conversion
                    xh = conv.f16_s2(carry)
                    li, wv = 1, wv // 2
                    x = None  # (the fp32 NCHW form of this layer's result is never needed)
                carry = None
                # mixed precision (set_amp): a run of stride-1 layers the fp16 kernel takes travels as fp16 NHWC --
                # one conversion in front (none behind an fp16 stride-2 layer); the last layer of the run writes fp32
                # NCHW for the kernels behind it, and fp16 NHWC as well where the next block can open on it
                if xh is None and x.dtype == torch.float16:  # (a stage left as fp16 NHWC whose successor stays fp32)
                    x = _f16_nhwc_to_f32_nchw(x)
                h_now = int(xh.shape[1]) if xh is not None else int(x.shape[2])
                w_now = int(xh.shape[2]) if xh is not None else int(x.shape[3])
                run = li
                while (self.amp and run < len(layers) and wv == w_now and layers[run].f16_ok(h_now, w_now)):
                    run += 1
                if run > li and w_now % 4 and not (run == len(layers) and self.amp_out_f16):
                    run = li  # the run would end in fp32 NCHW rows of a width the fp32 kernels do not take: stay fp32
                if run > li:
                    if xh is None:
                        xh = _conv.to_f16_nhwc(x)
                    nxt = plan[bi + 1][0] if bi + 1 < len(plan) else None
                    opens = run == len(layers) and nxt is not None and nxt.f16_s2_ok()
                    stage_h = run == len(layers) and self.amp_out_f16  # the neck reads fp16 NHWC: no fp32 form at all
                    for k in range(li, run):
                        if k == run - 1 and stage_h:
                            xh = layers[k].f16(xh)
                            x, carry = xh, (xh if opens else None)
                        elif k == run - 1 and opens:
                            carry, x = layers[k].f16_dual(xh)
                        else:
                            xh = layers[k].f16(xh, out_f32_nchw=(k == run - 1))
                            x = xh
                    li = run
                    continue
                if xh is not None:  # an fp16 stride-2 layer with no fp16 run behind it: back to fp32 NCHW
                    x = _f16_nhwc_to_f32_nchw(xh)
                    continue
                x, wv = conv(x, wv)
                li += 1
            outs.append(_tag_valid_w(x, wv))
        return tuple(outs)


class SecondFPN(_InferenceCache, nn.Module):
    """second_fpn.py:99-157 (use_spatial_attn_before_concat unsupported: unused on the path).  Every level is a
    kernel = stride convolution / transposed convolution: one patch GEMM each, written straight into its channel
    slice of the concatenated map (no concat pass)."""

    def __init__(self, in_channels=(128, 128, 256), out_channels=(256, 256, 256), upsample_strides=(1, 2, 4),
                 use_conv_for_no_stride=False):
        super().__init__()
        deblocks = []
        for i, oc in enumerate(out_channels):
            stride = upsample_strides[i]
            if stride > 1 or (stride == 1 and not use_conv_for_no_stride):
                layer = _conv_bn_relu(in_channels[i], oc, int(stride), stride=int(stride), transpose=True)
            else:
                s = round(1 / stride)
                layer = _conv_bn_relu(in_channels[i], oc, s, stride=s)
            deblocks.append(nn.Sequential(*layer))
        self.deblocks = nn.ModuleList(deblocks)

    def _plan(self):
        if not self._cache_valid():
            plan, off = [], 0
            for blk in self.deblocks:
                conv, bn = blk[0], blk[1]
                tr = isinstance(conv, nn.ConvTranspose2d)
                w, b = _fold_conv_bn(conv, bn)
                mode = _conv.patch_mode(w, conv.stride[0], tr)
                if mode is None or tuple(conv.padding) != (0, 0) or conv.stride[0] != conv.stride[1]:
                    raise Paddle3DAmdError(f"unsupported configuration: FPN level {conv} is not a kernel = stride "
                                           "(conv 1 / 2, transposed conv 1 / 2 / 4) layer (status -3)")
                cout = int(w.shape[1] if tr else w.shape[0])
                cin = int(w.shape[0] if tr else w.shape[1])
                plan.append(dict(mode=mode, w=_conv.pack_patch_weight(w, mode, tr), wraw=w, tr=tr, b=b, cin=cin, cout=cout, off=off,
                                 scale={0: 0.5, 1: 1, 2: 2, 3: 4}[mode]))
                off += cout
            self._store_cache((plan, off))
        return self._cache

    # ---- mixed precision (CenterPoint.set_amp): every level as one fp16 gather-GEMM over the pixels ------------------
    def _amp_level(self, i, p, n, h, w, dev, tag=None):
        """Static pieces of level i on an [n, h, w, cin] fp16 NHWC input: the neighbour table over the output pixels
        (a kernel = stride convolution reads s x s input pixels, a transposed one exactly one, at the tap its position
        selects), its tile order, and the weight as [taps][cin][cout] in the fp16 kernel's operand order."""
        key = (i, n, h, w, str(dev))  # (per device: a model moved with .to() must not reuse the old device's tables)
        cache = self.__dict__.setdefault("_amp_tables", {})
        if len(cache) > 64:  # (input shapes seen so far: bounded, the tables are rebuilt in a millisecond)
            cache.clear()
        if key not in cache:
            conv = self.deblocks[i][0]
            s = int(conv.stride[0])
            tr = isinstance(conv, nn.ConvTranspose2d)
            ho, wo = (h * s, w * s) if tr else (h // s, w // s)
            b = torch.arange(n, device=dev).view(n, 1, 1)
            y = torch.arange(ho, device=dev).view(1, ho, 1)
            x = torch.arange(wo, device=dev).view(1, 1, wo)
            if tr:  # out (y, x) <- in (y // s, x // s) through tap (y % s, x % s)
                src = ((b * h + y // s) * w + x // s).reshape(-1)
                tap = ((y % s) * s + x % s).expand(n, ho, wo).reshape(-1)
                nbr = torch.full((n * ho * wo, s * s), -1, dtype=torch.int32, device=dev)
                nbr[torch.arange(n * ho * wo, device=dev), tap] = src.int()
            else:   # out (y, x) <- in (s y + dy, s x + dx), tap dy s + dx
                cols = [((b * h + y * s + dy) * w + x * s + dx).reshape(-1) for dy in range(s) for dx in range(s)]
                nbr = torch.stack(cols, 1).int().contiguous()
            order = _sp3.tile_order(nbr) if tr and s > 1 else None
            cache[key] = (nbr, order, ho, wo)
        wkey = ("w", i, str(dev))
        if tag is None:
            tag = _param_signature(self)
        if wkey not in cache or cache[wkey][0] != tag:
            conv, bn = self.deblocks[i][0], self.deblocks[i][1]
            tr = isinstance(conv, nn.ConvTranspose2d)
            wf, bf = _fold_conv_bn(conv, bn)
            # [taps = ky * s + kx][cin][cout]
            wk = wf.permute(2, 3, 0, 1) if tr else wf.permute(2, 3, 1, 0)
            wk = wk.reshape(-1, wk.shape[2], wk.shape[3]).contiguous()
            parts = []
            step = 128 if wk.shape[2] % 128 == 0 else (64 if wk.shape[2] % 64 == 0 else 32)
            for c0 in range(0, wk.shape[2], step):
                parts.append((c0, step, _sp3.pack_weight_f16(wk[:, :, c0:c0 + step].contiguous()), bf[c0:c0 + step].contiguous()))
            cache[wkey] = (tag, parts)
        return cache[key], cache[wkey][1]

    def amp_ok(self, xs):
        plan, _ = self._plan()
        return all(p["cin"] % 16 == 0 and p["cout"] % 32 == 0 and p["scale"] in (0.5, 1, 2, 4) for p in plan)

    def forward_f16(self, xs):
        """xs: the backbone's stage outputs as fp16 NHWC [n, h, w, c].  Returns the concatenated map as fp16 NHWC."""
        self._require_eval()
        plan, ctot = self._plan()
        n = int(xs[0].shape[0])
        out = None
        tag = _param_signature(self)  # once per forward, not once per level
        for i, (p, x) in enumerate(zip(plan, xs)):
            h, w = int(x.shape[1]), int(x.shape[2])
            (nbr, order, ho, wo), parts = self._amp_level(i, p, n, h, w, x.device, tag)
            if out is None:
                out = torch.empty((n, ho, wo, ctot), dtype=torch.float16, device=x.device)
            elif tuple(out.shape[1:3]) != (ho, wo):
                raise Paddle3DAmdError("SecondFPN: the levels do not meet at one resolution")
            for c0, step, wp, bias in parts:
                _sp3.gather_gemm_f16(x.view(n * h * w, p["cin"]), nbr, wp, p["cin"], step, out.view(n * ho * wo, ctot),
                                     p["off"] + c0, bias=bias, relu=True, order=order)
        return out

    def forward(self, xs):
        self._require_eval()
        plan, ctot = self._plan()
        if all(x.dtype == torch.float16 for x in xs):
            return self.forward_f16(xs)
        xs = [_f16_nhwc_to_f32_nchw(x) if x.dtype == torch.float16 else x for x in xs]
        sizes = {(int(x.shape[2] * p["scale"]), int(_valid_w(x) * p["scale"])) for p, x in zip(plan, xs)}
        if len(sizes) != 1 or len(plan) != len(xs):
            raise Paddle3DAmdError(f"SecondFPN: the levels do not meet at one resolution ({sorted(sizes)})")
        hw = sizes.pop()
        out = torch.empty((xs[0].shape[0], ctot, hw[0], hw[1]), dtype=torch.float32, device=xs[0].device)
        for p, x in zip(plan, xs):
            if not _conv.patch_supported(p["mode"], p["cin"], p["cout"], int(x.shape[2]), int(x.shape[3])):
                raise Paddle3DAmdError(f"patch_conv: unsupported configuration (mode {p['mode']}, cin {p['cin']}, "
                                       f"cout {p['cout']}, input {tuple(x.shape[2:])}) (status -3)")
            if p["mode"] < 2 and _valid_w(x) != x.shape[3]:
                raise Paddle3DAmdError(f"patch_conv: unsupported configuration (mode {p['mode']} on a map of width "
                                       f"{_valid_w(x)}, not a multiple of 4) (status -3)")
            if _conv.PATCH_BF16X3 and _conv.patch_x3_supported(p["mode"], p["cin"], p["cout"], int(x.shape[2]),
                                                               int(x.shape[3])):
                # fp32 arithmetic on the bf16 matrix cores (three pieces per operand, csrc/conv_patch_x3.hip)
                if "wx3" not in p:
                    p["wx3"] = _conv.pack_patch_weight_x3(p["wraw"], p["mode"], p["tr"])
                _conv.patch_conv_x3_bias_relu(x, p["wx3"], p["b"], p["mode"], p["cout"], out, p["off"], relu=True,
                                              w_valid=_valid_w(x))
                continue
            _conv.patch_conv_bias_relu(x, p["w"], p["b"], p["mode"], p["cout"], out, p["off"], relu=True,
                                       w_valid=_valid_w(x))
        return out


class ConvModule(nn.Module):
    """center_head.py:43-78: conv(bias) -> BN(eps 1e-5) -> ReLU.  Parameters only (CenterHead runs them fused)."""

    def __init__(self, cin, cout, k, padding=0):
        super().__init__()
        self.conv = nn.Conv2d(cin, cout, k, padding=padding, bias=True)
        self.bn = nn.BatchNorm2d(cout, eps=1e-5, momentum=0.1)
        self.activate = nn.ReLU()


class SeparateHead(nn.Module):
    """center_head.py:81-153: per head `num_conv-1` ConvModules then a biased conv to `classes` maps.
    Parameters only (CenterHead runs all heads of all tasks as two fused convolutions)."""

    def __init__(self, in_channels, heads, head_conv=64, final_kernel=1, init_bias=-2.19):
        super().__init__()
        self.heads = heads
        for head, (classes, num_conv) in heads.items():
            layers, c_in = [], in_channels
            for _ in range(num_conv - 1):
                layers.append(ConvModule(c_in, head_conv, final_kernel, padding=final_kernel // 2))
                c_in = head_conv
            layers.append(nn.Conv2d(head_conv, classes, final_kernel, padding=final_kernel // 2, bias=True))
            setattr(self, head, nn.Sequential(*layers))
        with torch.no_grad():
            getattr(self, "hm")[-1].bias.fill_(init_bias)


class CenterHead(_InferenceCache, nn.Module):
    """center_head.py:156-220 forward + :294-339 predict_by_custom_op (inference only): BatchNorm folded, the 36
    first-stage 3x3 convolutions on the shared map run as ONE Winograd convolution, the 36 final convolutions as
    one grouped launch."""

    def __init__(self, in_channels, tasks, common_heads, init_bias=-2.19, share_conv_channel=64, num_hm_conv=2,
                 **_unused):
        super().__init__()
        self.num_classes = [len(t["class_names"]) for t in tasks]
        self.class_names = [t["class_names"] for t in tasks]
        self.with_velocity = "vel" in common_heads
        self.box_n_dim = 9 if self.with_velocity else 7
        self.shared_conv = ConvModule(in_channels, share_conv_channel, 3, padding=1)
        # branches per slice of the head (0 = automatic: see forward)
        self.head_chunk = 0
        self.head_chunk_bytes = 1 << 30
        self.amp = False  # True: shared + first-stage convolutions on the fp16 matrix cores (CenterPoint.set_amp)
        self.tasks = nn.ModuleList()
        for ncls in self.num_classes:
            heads = dict(common_heads)
            heads.update(hm=(ncls, num_hm_conv))
            self.tasks.append(SeparateHead(share_conv_channel, heads, final_kernel=3, init_bias=init_bias))

    def _plan(self):
        if not self._cache_valid():
            w0, b0 = _fold_conv_bn(self.shared_conv.conv, self.shared_conv.bn)
            ws, bs, plan, finals = [], [], [], []
            for t, task in enumerate(self.tasks):
                for head in task.heads:
                    seq = getattr(task, head)
                    if len(seq) != 2 or tuple(seq[1].kernel_size) != (3, 3):
                        raise Paddle3DAmdError("unsupported configuration: the fused CenterHead expects num_conv == 2 "
                                               "and 3x3 final convolutions for every head (status -3)")
                    w, b = _fold_conv_bn(seq[0].conv, seq[0].bn)
                    ws.append(w)
                    bs.append(b)
                    plan.append((t, head))
                    finals.append((seq[1].weight.detach(), seq[1].bias.detach()))
            cmax = max(f[0].shape[0] for f in finals)
            hc = ws[0].shape[0]
            # grouped second stage: group g maps its own hc channels to cmax (zero padded) outputs
            wf = torch.zeros(len(finals) * cmax, hc, 3, 3, device=w0.device, dtype=w0.dtype)
            bf = torch.zeros(len(finals) * cmax, device=w0.device, dtype=w0.dtype)
            for g, (w, b) in enumerate(finals):
                wf[g * cmax:g * cmax + w.shape[0]] = w
                bf[g * cmax:g * cmax + w.shape[0]] = b
            self._store_cache(dict(
                shared=_Conv3x3(w0, b0, 1), first=_Conv3x3(torch.cat(ws, 0).contiguous(), torch.cat(bs, 0).contiguous(), 1),
                pf=_conv.pack_grouped_weight(wf, len(finals)), bf=bf, hc=int(hc), plan=plan, cmax=int(cmax), wf=wf,
                groups=len(finals), ncls=[int(f[0].shape[0]) for f in finals]))
        return self._cache

    def forward(self, x, want_shared=True):
        """x [B, C, H, W] -> (per task dict of head maps, shared feature map), center_head.py:212-220.
        want_shared=False (the model's own test_forward, which drops the shared map): under AMP the second item is None
        instead of the shared map converted back to fp32 NCHW (an 18 us pass per 16 frames nobody reads)."""
        self._require_eval()
        f = self._plan()
        nhwc = x.dtype == torch.float16  # the neck's map as fp16 NHWC (the whole dense graph under AMP)
        hh, ww = (int(x.shape[1]), int(x.shape[2])) if nhwc else (int(x.shape[2]), int(x.shape[3]))
        if ww % 4:
            raise Paddle3DAmdError(f"CenterHead: unsupported configuration (map width {ww} is not a multiple "
                                   "of 4) (status -3)")
        amp = self.amp and f["shared"].f16_ok(hh, ww) and f["first"].f16_ok(hh, ww) and f["hc"] == 64
        if nhwc and not amp:
            x, nhwc = x.permute(0, 3, 1, 2).float().contiguous(), False
        if amp:
            # mixed precision: shared convolution and the 36 first-stage convolutions on the fp16 matrix cores (fp16
            # NHWC between them), the grouped final convolutions read the first stage's fp16 NHWC output
            x = f["shared"].f16(x if nhwc else _conv.to_f16_nhwc(x))
            n, h, w, _ = (int(v) for v in x.shape)
        else:
            x, _ = f["shared"](x)
            n, _, h, w = (int(v) for v in x.shape)
        if not _conv.grouped_small_supported(f["hc"], f["cmax"], h, w):
            raise Paddle3DAmdError(f"grouped_conv3x3_small: unsupported configuration ({f['hc']} -> {f['cmax']} "
                                   f"channels per group, map {(h, w)}) (status -3)")
        first, groups = f["first"], f["groups"]
        # The branches in slices: a slice's first-stage map goes through ONE reused buffer and is read back by the
        # slice's final convolutions right away.  Measured at 16 frames of 128 x 128 (tools/prof/prof_head_chunk.py;
        # bit-identical results): all 36 branches at once (2.4 GB map) 3.60 ms, two slices of 18 3.43 ms, slices of
        # 1 / 2 / 4 branches (67 MB each, inside the last-level cache) 4.60 / 3.93 / 3.55 ms -- the short launches
        # lose more than the cache gives, so a map above head_chunk_bytes is cut in two and no further.
        full = n * f["hc"] * h * w * 4 * groups
        k = self.head_chunk if self.head_chunk else ((groups + 1) // 2 if full > self.head_chunk_bytes else groups)
        if amp:
            # the first stage leaves fp16 NHWC (round 5; fp32 NCHW before: 2.4 GB written and fetched back per 16 frames),
            # the grouped final convolutions read a group's 64 channels of a pixel as one 128-byte line
            if "pf16" not in f:
                f["pf16"] = _conv.pack_grouped_weight_f16(f["wf"], groups)
            full = full // 2
            k = self.head_chunk if self.head_chunk else ((groups + 1) // 2 if full > self.head_chunk_bytes else groups)
            k = k + (k & 1) if k < groups else groups  # slices of whole 128-channel tiles (two branches each)
            z = torch.empty((n, groups * f["cmax"], h, w), dtype=torch.float32, device=x.device)
            buf = torch.empty((n * min(k, groups) * 64 * h * w,), dtype=torch.float16, device=x.device)
            per = _conv.f16_tile(first.cout) // 64
            for c0 in range(0, groups, k):
                c1 = min(c0 + k, groups)
                # the slice's first-stage map group-major ([n, branch, h, w, 64], round 6): the final convolution of a
                # branch then fetches contiguous patch rows (NHWC: 0.48 ms per 16 frames for the 36 branches, this: see
                # tools/prof/prof_grouped_f16.py)
                y = buf[: n * (c1 - c0) * 64 * h * w].view(n, c1 - c0, h, w, 64)
                first.f16(x, out_f32_nchw=False, out=y, tiles=(c0 // per, c1 // per), group_major=True)
                _conv.grouped_conv3x3_small_f16(y, f["pf16"][c0:c1], f["bf"][c0 * f["cmax"]:c1 * f["cmax"]], c1 - c0,
                                                out=z, out_groups=groups, out_group0=c0, group_major=True)
            rets = [dict() for _ in self.tasks]
            for g, (t, head) in enumerate(f["plan"]):
                rets[t][head] = z[:, g * f["cmax"]:g * f["cmax"] + f["ncls"][g]]
            # the shared map in the reference's layout and dtype in both modes (center_head.py:212-220 returns
            # `ret_dicts, x` with x fp32 [n, 64, h, w]); one 16 MB elementwise pass per 16 frames
            return rets, (x.permute(0, 3, 1, 2).float() if want_shared else None)
        chunked = (k < groups and f["hc"] == 64 and first.stride == 1 and _conv.winograd43_supported(first.cin, first.cout, h, w)
                   and w % 4 == 0)
        if chunked:
            pp = first.cin >= _conv.WINOGRAD43_PP_MIN_CIN and _conv.winograd43_pp_supported(first.cin, first.cout, h, w)
            if pp:  # [cout / 64][...]: channel tiles are slices of the first dimension in both packings
                if "w43pp" not in first.packed:
                    first.packed["w43pp"] = _conv.pack_winograd43_lane_weight(first.w)
                u = first.packed["w43pp"]
            else:
                if "w43" not in first.packed:
                    first.packed["w43"] = _conv.pack_winograd43_weight(first.w)
                u = first.packed["w43"]
                chunked = int(u.shape[2]) * 16 == 64
        if chunked:
            wino = _conv.conv3x3_winograd43_pp_bias_relu if pp else _conv.conv3x3_winograd43_bias_relu
            z = torch.empty((n, groups * f["cmax"], h, w), dtype=torch.float32, device=x.device)
            buf = torch.empty((n * k * 64 * h * w,), dtype=torch.float32, device=x.device)
            # round 6: all branches read the same 64-channel map, so its Winograd input transform is computed once for the
            # 36 channel blocks instead of by every one of them (csrc/conv_winograd43_ppv.hip; the same bytes out)
            vpre = (_conv.winograd43_input_transform(x)
                    if pp and 0 < _conv.WINOGRAD43_PPV_MIN_BLOCKS <= groups else None)
            for c0 in range(0, groups, k):
                c1 = min(c0 + k, groups)
                y = buf[: n * (c1 - c0) * 64 * h * w].view(n, (c1 - c0) * 64, h, w)
                if vpre is not None:
                    _conv.conv3x3_winograd43_ppv_bias_relu(vpre, x.shape, u[c0:c1], first.b[c0 * 64:c1 * 64],
                                                           (c1 - c0) * 64, relu=True, out=y)
                else:
                    wino(x, u[c0:c1], first.b[c0 * 64:c1 * 64], (c1 - c0) * 64, relu=True, out=y)
                _conv.grouped_conv3x3_small(y, f["pf"][c0:c1], f["bf"][c0 * f["cmax"]:c1 * f["cmax"]], c1 - c0, out=z,
                                            out_groups=groups, out_group0=c0)
        else:
            y, _ = first(x)
            z = _conv.grouped_conv3x3_small(y, f["pf"], f["bf"], groups)
        rets = [dict() for _ in self.tasks]
        for g, (t, head) in enumerate(f["plan"]):
            # channel slices of z stay views: the postprocess op takes them with their common batch stride
            rets[t][head] = z[:, g * f["cmax"]:g * f["cmax"] + f["ncls"][g]]
        return rets, x

    @torch.no_grad()
    def predict_by_custom_op(self, preds_dicts, test_cfg, device_only=False, records=0):
        """center_head.py:294-339.  Returns per frame dict(box3d_lidar, label_preds, scores); device_only=True the
        padded device tensors + counts without a host sync (records = max_per_img: plus the hand-off record)."""
        hm, reg, height, dim, vel, rot = [], [], [], [], [], []
        for preds in preds_dicts:
            hm.append(preds["hm"])
            reg.append(preds["reg"])
            height.append(preds["height"])
            dim.append(preds["dim"])
            vel.append(preds["vel"] if self.with_velocity else preds["reg"])
            rot.append(preds["rot"])
        # the reference builds a len(tasks)**2 list of running class offsets (:303-309); entry t is task t's
        offsets = np.concatenate([[0], np.cumsum(self.num_classes)[:-1]]).astype(int).tolist()
        num_classes = offsets * len(preds_dicts)
        out = _cp.centerpoint_postprocess_device(
            hm, reg, height, dim, vel, rot, test_cfg["voxel_size"], test_cfg["point_cloud_range"],
            test_cfg["post_center_limit_range"], num_classes, test_cfg["down_ratio"], test_cfg["score_threshold"],
            test_cfg["nms"]["nms_iou_threshold"], test_cfg["nms"]["nms_pre_max_size"],
            test_cfg["nms"]["nms_post_max_size"], self.with_velocity, allow_batch=True,
            records=records if device_only else 0)
        if device_only:
            return out
        b, s, l, n = out
        counts = n.cpu().tolist()
        return [dict(box3d_lidar=b[i, :k], label_preds=l[i, :k], scores=s[i, :k]) for i, k in enumerate(counts)]


class CenterPoint(nn.Module):
    """centerpoint.py:45-166, inference path.  `test_forward(points)` takes a [B, N, D] tensor (or a list of
    [N_i, D] tensors, padded internally) and returns the per-frame detections."""

    def __init__(self, voxelizer, voxel_encoder, middle_encoder, backbone, neck, bbox_head, test_cfg,
                 box_with_velocity=True):
        super().__init__()
        self.voxelizer = voxelizer
        self.voxel_encoder = voxel_encoder
        self.middle_encoder = middle_encoder
        self.backbone = backbone
        self.neck = neck
        self.bbox_head = bbox_head
        self.test_cfg = test_cfg
        self.box_with_velocity = box_with_velocity
        # the scatter is fused into the backbone's first convolution where it can be (a property of this model's
        # forward, not of the scatter module: `model.middle_encoder(...)` still returns the dense pseudo image)
        self.fuse_scatter = isinstance(middle_encoder, PointPillarsScatter) and isinstance(backbone, SecondBackbone)
        # the row writer is fused away where the voxel encoder can read the points through the voxelizer's index
        # (`model.voxelizer(...)` still returns the padded tensor: the operator's contract is unchanged)
        self.fuse_rows = isinstance(voxel_encoder, PillarFeatureNet) and isinstance(middle_encoder, PointPillarsScatter)

    def scatter(self, feats, coors, batch_size):
        """The middle encoder as this model's forward runs it (a SparseCanvas when the scatter is fused)."""
        if self.fuse_scatter:
            return self.middle_encoder.sparse(feats, coors, batch_size)
        return self.middle_encoder(feats, coors, batch_size)

    def invalidate(self):
        """Rebuild every folded / packed weight on the next forward (after a write through `param.data`)."""
        invalidate_derived(self)
        return self

    def set_amp(self, enabled: bool = True):
        """Mixed precision, the reference's `amp_cfg: level O2` configurations
        (configs/centerpoint/centerpoint_pillars_02voxel_nuscenes_10sweep_ampO2_ultra.yml:5-9): the WHOLE dense graph
        in fp16 NHWC on the fp16 matrix cores with fp32 accumulation (csrc/conv_f16.hip) -- the scatter-fused stride-2
        first layer, the stride-1 and stride-2 layers of the backbone, the FPN levels (fp16 gather-GEMMs into the
        concatenated map), the head's shared / first-stage convolutions and its grouped final convolutions (which write
        the fp32 maps the post-processing reads); a sparse middle encoder from 16 -> 32 channels on.  The front half (PFN
        / VoxelMean), the post-processing and every bias / BatchNorm fold stay fp32.  Off by default; a layer the fp16
        kernels do not take (cin % 16, cout % 64) falls back to its fp32 kernel with one conversion on either side."""
        for m in (self.backbone, self.bbox_head, self.middle_encoder):
            if hasattr(m, "amp"):
                m.amp = bool(enabled)
        if hasattr(self.backbone, "amp_out_f16"):
            # the whole dense graph in fp16 NHWC: backbone stages -> FPN levels (fp16 gather-GEMMs) -> head, no fp32 copy
            # and no layout conversion between them
            self.backbone.amp_out_f16 = bool(enabled) and hasattr(self.neck, "amp_ok") and self.neck.amp_ok(None)
        return self

    def _pack(self, points):
        if isinstance(points, torch.Tensor):
            return points, None
        n = max(p.shape[0] for p in points)
        d = points[0].shape[1]
        out = torch.zeros((len(points), n, d), dtype=torch.float32, device=points[0].device)
        for i, p in enumerate(points):
            out[i, : p.shape[0]] = p
        lens = torch.tensor([p.shape[0] for p in points], dtype=torch.int32, device=points[0].device)
        return out, lens

    def extract_pillars(self, points, num_points=None, dense=True):
        """voxelize -> voxel encoder -> middle encoder: the LiDAR front half (dense BEV features; dense=False leaves a
        PointPillarsScatter result as the SparseCanvas the backbone consumes without writing the pseudo image)."""
        if self.fuse_rows and isinstance(points, torch.Tensor) and points.dim() == 3:
            # voxelizer -> PFN through an index of the points: the padded [B, V, P, D] tensor (78 % zeros, written by
            # the row writer and read back by the PFN) never exists.  Same bytes as the pair below.
            idx = self.voxelizer.index(points, num_points)
            if idx is not None:
                span, plist, coors, npv, nv = idx
                b, v = int(coors.shape[0]), int(coors.shape[1])
                coors = coors.view(b * v, 4)
                feats = self.voxel_encoder.forward_indexed(points, span, plist, coors)
                if feats is not None:
                    return self.middle_encoder(feats, coors, b) if dense else self.scatter(feats, coors, b)
        voxels, coors, npv, nv = self.voxelizer(points, num_points)
        b, v, p, d = voxels.shape
        voxels, coors, npv = voxels.view(b * v, p, d), coors.view(b * v, 4), npv.view(b * v)
        if not isinstance(self.middle_encoder, PointPillarsScatter) and \
                not getattr(self.middle_encoder, "accepts_padding_rows", False):
            # a middle encoder that wants the occupied voxels only (one host sync, like the reference's
            # voxels[0:num_voxels] slice, voxelize.py:43); the sparse encoders of this package skip padding
            # rows (batch index -1) themselves
            keep = coors[:, 0] >= 0
            voxels, coors, npv = voxels[keep], coors[keep].contiguous(), npv[keep]
        feats = self.voxel_encoder(voxels, npv, coors)
        if dense:
            return self.middle_encoder(feats, coors, b)
        return self.scatter(feats, coors, b)

    def dense_forward(self, x):
        """SecondBackbone -> SecondFPN (centerpoint.py:133-137)."""
        return self.neck(self.backbone(x))

    @torch.no_grad()
    def test_forward(self, points, device_only=False):
        """points -> detections.  device_only=False reads the detections to the host; a sparse middle encoder may then
        plan from remembered capacities (no host sync inside the graph) because the overflow word is checked HERE and
        the frame recomputed with exact sizes when a set outgrew its capacity.  device_only=True never synchronises,
        so the encoder plans with its one host sync (exact, like the reference) unless the caller set
        `middle_encoder.remember_capacities = True` and reads `middle_encoder.take_overflow()` itself."""
        import contextlib

        pts, lens = self._pack(points)
        checked = getattr(self.middle_encoder, "overflow_checked", None)
        take = getattr(self.middle_encoder, "take_overflow", None)
        for _attempt in range(2):
            with (checked() if (checked is not None and not device_only) else contextlib.nullcontext()):
                x = self.extract_pillars(pts, lens, dense=False)
            x = self.dense_forward(x)
            preds, _ = self.bbox_head(x, want_shared=False)
            out = self.bbox_head.predict_by_custom_op(preds, self.test_cfg, device_only=device_only)
            if device_only or take is None or not take():
                break
        return out

    forward = test_forward


NUSC_TASKS = [dict(num_class=1, class_names=["car"]), dict(num_class=2, class_names=["truck", "construction_vehicle"]),
              dict(num_class=2, class_names=["bus", "trailer"]), dict(num_class=1, class_names=["barrier"]),
              dict(num_class=2, class_names=["motorcycle", "bicycle"]),
              dict(num_class=2, class_names=["pedestrian", "traffic_cone"])]


def centerpoint_pillars_nuscenes(max_num_voxels=(30000, 60000)) -> CenterPoint:
    """configs/centerpoint/centerpoint_pillars_02voxel_nuscenes_10sweep.yml:110-179, random init."""
    pcr, vs = [-51.2, -51.2, -5.0, 51.2, 51.2, 3.0], [0.2, 0.2, 8]
    test_cfg = dict(post_center_limit_range=[-61.2, -61.2, -10.0, 61.2, 61.2, 10.0], max_per_img=500,
                    nms=dict(nms_pre_max_size=1000, nms_post_max_size=83, nms_iou_threshold=0.2),
                    score_threshold=0.1, point_cloud_range=[-51.2, -51.2], down_ratio=4, voxel_size=[0.2, 0.2])
    return CenterPoint(
        voxelizer=HardVoxelizer(vs, pcr, 20, list(max_num_voxels)),
        voxel_encoder=PillarFeatureNet(5, (64, 64), False, 20, vs, pcr, legacy=False),
        middle_encoder=PointPillarsScatter(64, vs, pcr),
        backbone=SecondBackbone(64, (64, 128, 256), (3, 5, 5), (2, 2, 2)),
        neck=SecondFPN((64, 128, 256), (128, 128, 128), (0.5, 1, 2), use_conv_for_no_stride=True),
        bbox_head=CenterHead(384, NUSC_TASKS, dict(reg=(2, 2), height=(1, 2), dim=(3, 2), rot=(2, 2), vel=(2, 2))),
        test_cfg=test_cfg, box_with_velocity=True)


KITTI_TASKS = [dict(num_class=1, class_names=["Car"]), dict(num_class=2, class_names=["Cyclist", "Pedestrian"])]


def centerpoint_pillars_kitti(max_num_voxels=(12000, 40000)) -> CenterPoint:
    """configs/centerpoint/centerpoint_pillars_016voxel_kitti.yml:108-163, random init: 0.16 m pillars on the KITTI
    range (432 x 496), 100 points per pillar, a stride-1 first backbone block (head map 248 x 216, down_ratio 2), two
    tasks, boxes without velocity."""
    pcr, vs = [0.0, -39.68, -3.0, 69.12, 39.68, 1.0], [0.16, 0.16, 4]
    test_cfg = dict(post_center_limit_range=[-10.0, -50.0, -10.0, 80.0, 50.0, 10.0], max_per_img=500,
                    nms=dict(nms_pre_max_size=1000, nms_post_max_size=83, nms_iou_threshold=0.1),
                    score_threshold=0.1, point_cloud_range=pcr[:2], down_ratio=2, voxel_size=[0.16, 0.16])
    return CenterPoint(
        voxelizer=HardVoxelizer(vs, pcr, 100, list(max_num_voxels)),
        voxel_encoder=PillarFeatureNet(4, (64, 64), False, 100, vs, pcr, legacy=False),
        middle_encoder=PointPillarsScatter(64, vs, pcr),
        backbone=SecondBackbone(64, (64, 128, 256), (3, 5, 5), (1, 2, 2)),
        neck=SecondFPN((64, 128, 256), (128, 128, 128), (0.5, 1, 2), use_conv_for_no_stride=True),
        bbox_head=CenterHead(384, KITTI_TASKS, dict(reg=(2, 2), height=(1, 2), dim=(3, 2), rot=(2, 2))),
        test_cfg=test_cfg, box_with_velocity=False)


def centerpoint_voxels_nuscenes(max_num_voxels=(120000, 160000), point_cloud_range=None) -> CenterPoint:
    """configs/centerpoint/centerpoint_voxels_0075voxel_nuscenes_10sweep.yml:111-173, random init.
    point_cloud_range overrides the config's [-54, -54, -5, 54, 54, 3] (the tests run a quarter-range copy whose
    dense statement fits a CPU)."""
    from .sparse import SparseResNet3D

    pcr = [-54.0, -54.0, -5.0, 54.0, 54.0, 3.0] if point_cloud_range is None else list(map(float, point_cloud_range))
    vs = [0.075, 0.075, 0.2]
    test_cfg = dict(post_center_limit_range=[-61.2, -61.2, -10.0, 61.2, 61.2, 10.0], max_per_img=500,
                    nms=dict(nms_pre_max_size=1000, nms_post_max_size=83, nms_iou_threshold=0.2),
                    score_threshold=0.1, point_cloud_range=pcr[:2], down_ratio=8, voxel_size=[0.075, 0.075])
    return CenterPoint(
        voxelizer=HardVoxelizer(vs, pcr, 10, list(max_num_voxels)),
        voxel_encoder=VoxelMean(5),
        middle_encoder=SparseResNet3D(5, vs, pcr),
        backbone=SecondBackbone(256, (128, 256), (5, 5), (1, 2)),
        neck=SecondFPN((128, 256), (256, 256), (1, 2), use_conv_for_no_stride=True),
        bbox_head=CenterHead(512, NUSC_TASKS, dict(reg=(2, 2), height=(1, 2), dim=(3, 2), rot=(2, 2), vel=(2, 2))),
        test_cfg=test_cfg, box_with_velocity=True)


def centerpoint_voxels_kitti(max_num_voxels=(12000, 40000)) -> CenterPoint:
    """configs/centerpoint/centerpoint_voxels_008voxel_kitti.yml:108-160, random init: 0.08 m x 0.08 m x 0.1 m voxels
    on the KITTI range (864 x 992 x 40), 100 points per voxel, SparseResNet3D on 4 input channels, 124 x 108 head maps
    (down_ratio 8), two tasks, boxes without velocity."""
    from .sparse import SparseResNet3D

    pcr, vs = [0.0, -39.68, -3.0, 69.12, 39.68, 1.0], [0.08, 0.08, 0.1]
    test_cfg = dict(post_center_limit_range=[-10.0, -50.0, -10.0, 80.0, 50.0, 10.0], max_per_img=500,
                    nms=dict(nms_pre_max_size=1000, nms_post_max_size=83, nms_iou_threshold=0.1),
                    score_threshold=0.1, point_cloud_range=pcr[:2], down_ratio=8, voxel_size=[0.08, 0.08])
    return CenterPoint(
        voxelizer=HardVoxelizer(vs, pcr, 100, list(max_num_voxels)),
        voxel_encoder=VoxelMean(4),
        middle_encoder=SparseResNet3D(4, vs, pcr),
        backbone=SecondBackbone(256, (128, 256), (5, 5), (1, 2)),
        neck=SecondFPN((128, 256), (256, 256), (1, 2), use_conv_for_no_stride=True),
        bbox_head=CenterHead(512, KITTI_TASKS, dict(reg=(2, 2), height=(1, 2), dim=(3, 2), rot=(2, 2))),
        test_cfg=test_cfg, box_with_velocity=False)


from .checkpoint import load_paddle_state_dict  # noqa: E402,F401  (re-exported: the models' loader)
