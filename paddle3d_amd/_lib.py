"""ctypes loader for libpaddle3d_amd.so (the C ABI declared in include/paddle3d_amd.h).

There is NO fallback: if the shared object is missing, was built for another ABI, or lacks a symbol,
importing an op raises.  A GPU box must never silently run a PyTorch/CPU substitute.
"""
from __future__ import annotations

import ctypes as C
import os
from functools import lru_cache

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libpaddle3d_amd.so")

c_f32p = C.c_void_p
c_i32p = C.c_void_p

# symbol -> (restype, argtypes); mirrors include/paddle3d_amd.h one to one
_SIGNATURES = {
    "pd3_version": (C.c_int, []),
    "pd3_target_arch": (C.c_char_p, []),
    "pd3_hard_voxelize_workspace": (C.c_size_t, [C.c_int, C.c_int64, C.c_int, C.c_void_p, C.c_void_p,
                                                 C.c_int, C.c_int]),
    "pd3_hard_voxelize": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int64, C.c_int, C.c_void_p,
                                    C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                    C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "pd3_hard_voxelize_f64": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int64, C.c_int, C.c_void_p,
                                    C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                    C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "pd3_hard_voxelize_path": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int64, C.c_int, C.c_void_p,
                                         C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                         C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_int]),
    "pd3_hard_voxelize_index_list_entries": (C.c_int64, [C.c_int, C.c_int64]),
    "pd3_hard_voxelize_index": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int64, C.c_int, C.c_void_p,
                                          C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                          C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "pd3_pillar_feature_net_indexed": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p,
                                                 C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float,
                                                 C.c_float, C.c_float, C.c_float, C.c_float, C.c_void_p, C.c_void_p,
                                                 C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                                 C.c_void_p, C.c_void_p]),
    "pd3_pointpillars_scatter_workspace": (C.c_size_t, [C.c_int, C.c_int, C.c_int]),
    "pd3_pointpillars_scatter": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_int,
                                           C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "pd3_pillar_feature_net": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int,
                                         C.c_int, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float,
                                         C.c_float, C.c_void_p,
                                         C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p,
                                         C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "pd3_pillar_feature_net_path": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int,
                                              C.c_int, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float,
                                              C.c_float, C.c_void_p,
                                              C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p,
                                              C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p]),
    "pd3_voxel_mean": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_void_p,
                                 C.c_void_p]),
    "pd3_nms_workspace": (C.c_size_t, [C.c_int]),
    "pd3_nms_bev": (C.c_int, [C.c_void_p, C.c_int, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p,
                              C.c_size_t, C.c_void_p]),
    "pd3_nms_normal": (C.c_int, [C.c_void_p, C.c_int, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p,
                                 C.c_size_t, C.c_void_p]),
    "pd3_boxes_iou_bev": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "pd3_boxes_overlap_bev": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p,
                                        C.c_void_p]),
    "pd3_libm_eval": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "pd3_centerpoint_postprocess_workspace": (C.c_size_t, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "pd3_centerpoint_postprocess": (C.c_int, [C.c_void_p] * 6 + [C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int,
                                              C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                              C.c_float, C.c_float, C.c_int, C.c_int, C.c_int,
                                              C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                              C.c_void_p, C.c_size_t, C.c_void_p]),
    "pd3_centerpoint_postprocess_strided": (C.c_int, [C.c_void_p] * 6 + [C.c_int64, C.c_int, C.c_int, C.c_void_p,
                                                      C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                                      C.c_void_p, C.c_int, C.c_float, C.c_float, C.c_int, C.c_int,
                                                      C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                                      C.c_void_p, C.c_size_t, C.c_void_p, C.c_int]),
    "pd3_centerpoint_postprocess_records": (C.c_int, [C.c_void_p] * 6 + [C.c_int64, C.c_int, C.c_int, C.c_void_p,
                                                      C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                                      C.c_void_p, C.c_int, C.c_float, C.c_float, C.c_int, C.c_int,
                                                      C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                                      C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p]),
    "pd3_bev_pool_v2": (C.c_int, [C.c_void_p] * 7 + [C.c_int, C.c_int, C.c_int64, C.c_void_p,
                                                     C.c_void_p]),
    "pd3_bev_pool_v2_bkwd": (C.c_int, [C.c_void_p] * 8 + [C.c_int, C.c_int64, C.c_int, C.c_int64, C.c_int64,
                                                          C.c_void_p, C.c_void_p, C.c_void_p]),
    "pd3_frustum_to_lidar": (C.c_int, [C.c_void_p, C.c_int64, C.c_int, C.c_int] + [C.c_void_p] * 7),
    "pd3_voxel_pooling_prepare_workspace": (C.c_size_t, [C.c_int64]),
    "pd3_voxel_pooling_prepare": (C.c_int, [C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                            C.c_void_p, C.c_int] + [C.c_void_p] * 7 + [C.c_size_t, C.c_void_p]),
    "pd3_sparse_conv3d_workspace": (C.c_size_t, [C.c_int, C.c_void_p, C.c_int, C.c_int]),
    "pd3_sparse_conv3d_indices": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                            C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                            C.c_void_p, C.c_size_t, C.c_void_p]),
    "pd3_sparse_conv3d_features": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                             C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                             C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "pd3_sparse_conv3d_features_ordered": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                                     C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                                     C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "pd3_sparse_pack_weight_bf16x3": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "pd3_sparse_conv3d_features_bf16x3": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                                    C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                                    C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "pd3_sparse_pack_weight_f16": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "pd3_sparse_conv3d_features_f16": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                                 C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                                 C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "pd3_gather_gemm_f16": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                      C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p,
                                      C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "pd3_sparse_tile_order_entries": (C.c_int64, [C.c_int]),
    "pd3_sparse_tile_order": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "pd3_sparse_plan_workspace": (C.c_size_t, [C.c_int]),
    "pd3_sparse_conv_outputs_workspace": (C.c_size_t, [C.c_int] + [C.c_void_p] * 4),
    "pd3_sparse_sort_coords": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                         C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "pd3_sparse_conv_outputs": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                          C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p,
                                          C.c_size_t, C.c_void_p]),
    "pd3_sparse_rulebook_workspace": (C.c_size_t, [C.c_int, C.c_void_p]),
    "pd3_sparse_rulebook": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                      C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p,
                                      C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "pd3_sparse_to_dense": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                      C.c_void_p, C.c_void_p, C.c_void_p]),
    "pd3_selfcheck_lds_atomic_order": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                                 C.c_void_p]),
    "pd3_conv3x3_f16_bias_relu": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                            C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "pd3_f32_nchw_to_f16_nhwc": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "pd3_conv3x3_f16_bias_relu_dual": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                                 C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "pd3_conv3x3_s2_f16_bias_relu": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                               C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "pd3_scatter_conv3x3_s2_f16_bias_relu": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                                       C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int,
                                                       C.c_void_p]),
    "pd3_grouped_conv3x3_small_f16": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                                C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "pd3_grouped_conv3x3_small_f16_gm": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                                   C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "pd3_stable_argsort_workspace": (C.c_size_t, [C.c_int64, C.c_uint32]),
    "pd3_stable_argsort": (C.c_int, [C.c_void_p, C.c_int64, C.c_int, C.c_uint32, C.c_void_p, C.c_void_p, C.c_size_t,
                                     C.c_void_p]),
    "pd3_conv3x3_winograd43_pp_bias_relu": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                                       C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "pd3_conv3x3_winograd43_pp_trace": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                                  C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "pd3_merge_sweeps_workspace": (C.c_size_t, [C.c_int64]),
    "pd3_merge_sweeps": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                   C.c_void_p, C.c_int, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t,
                                   C.c_void_p]),
    "pd3_dynamic_voxelize": (C.c_int, [C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "pd3_conv3x3_bias_relu": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                        C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p]),
    "pd3_conv3x3_winograd_bias_relu": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                                 C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "pd3_conv3x3_winograd43_bias_relu": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                                   C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int,
                                                   C.c_void_p]),
    "pd3_patch_conv_bias_relu": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                           C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int,
                                           C.c_void_p]),
    "pd3_patch_conv_x3_bias_relu": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                              C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int,
                                              C.c_void_p]),
    "pd3_conv3x3_s2_x3_bias_relu": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                              C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p]),
    "pd3_winograd43_input_transform_floats": (C.c_size_t, [C.c_int] * 4),
    "pd3_winograd43_input_transform": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                                 C.c_void_p]),
    "pd3_conv3x3_winograd43_ppv_bias_relu": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                                       C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "pd3_ssd_postprocess_workspace": (C.c_size_t, [C.c_int] * 7),
    "pd3_ssd_postprocess": (C.c_int, [C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                      C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64,
                                      C.c_int, C.c_int, C.c_float, C.c_float, C.c_void_p, C.c_float, C.c_int, C.c_int,
                                      C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t,
                                      C.c_void_p, C.c_int]),
    "pd3_pointpillars_inverse_map": (C.c_int, [C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                               C.c_void_p]),
    "pd3_scatter_conv3x3_bias_relu": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                                C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int,
                                                C.c_void_p]),
    "pd3_pillar_conv_rulebook_workspace": (C.c_size_t, [C.c_int] * 4),
    "pd3_pillar_conv_rulebook": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                           C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_size_t,
                                           C.c_void_p]),
    "pd3_rows_to_dense_fill": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                         C.c_void_p, C.c_void_p]),
    "pd3_grouped_conv3x3_small": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                            C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "pd3_grouped_conv3x3_small_slice": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                                  C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int,
                                                  C.c_void_p]),
}

SYMBOLS = tuple(_SIGNATURES)


class Paddle3DAmdError(RuntimeError):
    pass


@lru_cache(maxsize=None)
def lib() -> C.CDLL:
    if not os.path.exists(LIB_PATH):
        raise Paddle3DAmdError(
            f"{LIB_PATH} not found: build it with `python -m paddle3d_amd.build` "
            "(there is no CPU / PyTorch fallback for the HIP ops)")
    handle = C.CDLL(LIB_PATH)
    for name, (res, args) in _SIGNATURES.items():
        try:
            fn = getattr(handle, name)
        except AttributeError as e:  # pragma: no cover
            raise Paddle3DAmdError(f"{LIB_PATH} does not export {name}") from e
        fn.restype = res
        fn.argtypes = args
    return handle


def check(status: int, what: str) -> None:
    if status == 0:
        return
    if status < 0:
        msg = {-1: "invalid argument", -2: "workspace too small", -3: "unsupported configuration"}.get(
            status, "error")
        raise Paddle3DAmdError(f"{what}: {msg} (status {status})")
    raise Paddle3DAmdError(f"{what}: HIP error {status}")
