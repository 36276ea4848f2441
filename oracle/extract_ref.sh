#!/bin/sh
# Extract the Paddle-free arithmetic line ranges of the reference into $2 (git-ignored build dir).
# usage: extract_ref.sh /root/reference oracle/_ref
# The ranges are the ones SURVEY.md section 8(c) verified to compile standalone.
set -e
REF="$1"; OUT="$2"
mkdir -p "$OUT"
OPS="$REF/paddle3d/ops"
# hard_voxelize_cpu_kernel<T,T_int>                      voxelize_op.cc:19-82
sed -n '19,82p'   "$OPS/voxel/voxelize_op.cc"                          > "$OUT/gen_voxelize_cpu.inc"
# min/max/EPS/Point/cross/.../box_overlap/iou_bev         iou3d_cpu.cpp:31-239
sed -n '31,239p'  "$OPS/iou3d_nms/iou3d_cpu.cpp"                       > "$OUT/gen_iou3d_cpu.inc"
# host greedy sweep of nms_gpu                            iou3d_nms.cpp:119-137
sed -n '119,137p' "$OPS/iou3d_nms/iou3d_nms.cpp"                       > "$OUT/gen_nms_sweep.inc"
# iou_normal (axis aligned)                               iou3d_nms_kernel.cu:365-378
sed -n '365,378p' "$OPS/iou3d_nms/iou3d_nms_kernel.cu"                 > "$OUT/gen_iou_normal.inc"
# CenterPoint decode_kernel (barrier-free __global__)     postprocess.cu:32-80
sed -n '32,80p'   "$OPS/centerpoint_postprocess/postprocess.cu"        > "$OUT/gen_decode_kernel.inc"
# bev_pool_v2_kernel (barrier-free __global__)            bev_pool_cuda.cu:18-44
sed -n '18,44p'   "$OPS/bev_pool_v2/bev_pool_cuda.cu"                  > "$OUT/gen_bev_pool_kernel.inc"
# bev_pool_grad_kernel (barrier-free __global__)          bev_pool_cuda_bkwd.cu:44-94
sed -n '44,94p'   "$OPS/bev_pool_v2_backward/bev_pool_cuda_bkwd.cu"    > "$OUT/gen_bev_pool_grad_kernel.inc"
