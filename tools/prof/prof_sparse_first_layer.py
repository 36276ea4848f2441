"""Feasibility (not a test): the scatter-fused first backbone convolution (64 -> 64, stride 2, 512^2 -> 256^2, canvas 11 %
occupied) as a SPARSE convolution over the occupied cells with the library's own sparse ops, stage by stage, against the
dense fused kernel.  usage: prof_sparse_first_layer.py [batch]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from paddle3d_amd import centerpoint as cpm, synth  # noqa: E402
from paddle3d_amd.ops import conv as C, sparse_conv3d as sp  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
torch.manual_seed(0)
model = cpm.centerpoint_pillars_nuscenes(max_num_voxels=(30000, 30000)).cuda().eval()
synth.trained_like_batchnorm(model, 3)
pts = torch.from_numpy(np.stack([synth.nuscenes_sweep(100 + i) for i in range(B)])).cuda()


def timed(fn, it=10):
    for _ in range(3):
        out = fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it):
        out = fn()
    e1.record()
    torch.cuda.synchronize()
    return out, e0.elapsed_time(e1) / it * 1e3


with torch.no_grad():
    canvas = model.extract_pillars(pts, dense=False)   # SparseCanvas: features [B*V, 64], coords [B*V, 4], inv
    conv0 = model.backbone._plan()[0][0]
    ref, t_ref = timed(lambda: conv0.__class__.__call__ and C.scatter_conv3x3_bias_relu(
        canvas, conv0.packed.setdefault("direct", C.pack_conv3x3_weight(conv0.w)), conv0.b, conv0.cout))
    print(f"dense fused kernel: {t_ref:.0f} us, output {tuple(ref.shape)}")
    coords, feats = canvas.coords, canvas.features
    spec = sp.ConvSpec((1, 3, 3), (1, 2, 2), (0, 1, 1), False, None, True)
    pl, t_plan = timed(lambda: sp.plan(coords, B, (1, 512, 512), [spec]))
    caps = [int(coords.shape[0]), B * 256 * 256]
    plc, t_planc = timed(lambda: sp.plan(coords, B, (1, 512, 512), [spec], caps=caps))
    idx = pl.indices[0]
    print(f"plan with the host sync: {t_plan:.0f} us; without (worst-case capacities): {t_planc:.0f} us; "
          f"rows in {pl.n_in}, rows out {idx.n_out} = {idx.n_out / (B * 65536):.3f} of the cells, pairs per row "
          f"{float((idx.nbr >= 0).sum()) / idx.n_out:.2f}")
    w = conv0.w.permute(2, 3, 1, 0).reshape(1, 3, 3, conv0.cin, conv0.cout).contiguous()  # [kd, kh, kw, cin, cout]
    wp = sp.pack_weight_bf16x3(w)
    fin, t_g = timed(lambda: feats.index_select(0, pl.order))
    rows, t_f = timed(lambda: sp.features_bf16x3(fin, idx, wp, conv0.cin, conv0.cout, conv0.b, None, None, None, True))
    idxc = plc.indices[0]
    finc = feats.index_select(0, plc.order)
    rowsc, t_fc = timed(lambda: sp.features_bf16x3(finc, idxc, wp, conv0.cin, conv0.cout, conv0.b, None, None, None, True))
    dense, t_d = timed(lambda: sp.to_dense(rows, idx.out_coords, B, (1, 256, 256)))
    print(f"gather of the input rows {t_g:.0f} us, features (exact sizes) {t_f:.0f} us, (at capacity) {t_fc:.0f} us, "
          f"to_dense {t_d:.0f} us")
    # agreement where the output is active; elsewhere the dense kernel holds relu(bias)
    act = torch.zeros(B, 256, 256, dtype=torch.bool, device="cuda")
    oc = idx.out_coords[: idx.n_out].long()
    act[oc[:, 0], oc[:, 2], oc[:, 3]] = True
    diff = (dense - ref)[act.unsqueeze(1).expand_as(ref)].abs().max()
    fill = (ref - torch.relu(conv0.b).view(1, -1, 1, 1))[(~act).unsqueeze(1).expand_as(ref)].abs().max()
    print(f"active cells: max |sparse - dense| {float(diff):.2e} (max |ref| {float(ref.abs().max()):.2f}); empty cells: "
          f"max |dense - relu(bias)| {float(fill):.2e}")
    # the specialised path (round 6): rulebook from the inverse map, no sort, no input gather, dense writer with fill
    (out, packed), _ = timed(lambda: C.scatter_conv3x3_sparse(canvas, conv0.w, conv0.b), it=2)
    (out, _p), t_all = timed(lambda: C.scatter_conv3x3_sparse(canvas, conv0.w, conv0.b, packed))
    print(f"scatter_conv3x3_sparse: {t_all:.0f} us (dense fused kernel {t_ref:.0f}); max |sparse - dense| over ALL cells "
          f"{float((out - ref).abs().max()):.2e}")
    # how much does the tile order's reach matter?  The same rulebook with the rows grouped by mask over windows of
    # 2048 (the device order), 8192, 65536 and over ALL rows (host-side stable sort: an experiment, not a path)
    L = C.lib()
    from paddle3d_amd.ops._common import check, ptr, stream_ptr, workspace
    n, cin, ny, nx = canvas.shape
    ho, wo = ny // 2, nx // 2
    cap = n * ho * wo
    nbr = torch.empty((cap, 9), dtype=torch.int32, device="cuda")
    out_cell = torch.empty((cap,), dtype=torch.int32, device="cuda")
    cell_row = torch.empty((n, ho * wo), dtype=torch.int32, device="cuda")
    n_out = torch.empty((1,), dtype=torch.int32, device="cuda")
    ws = workspace(L.pd3_pillar_conv_rulebook_workspace(n, ny, nx, 2), canvas.features.device)
    order = torch.empty((int(L.pd3_sparse_tile_order_entries(cap)),), dtype=torch.int32, device="cuda")
    check(L.pd3_pillar_conv_rulebook(ptr(canvas.inv), n, ny, nx, 2, ptr(nbr), ptr(out_cell), ptr(cell_row), ptr(n_out), cap,
                                     ptr(order), ptr(ws), ws.numel(), stream_ptr(canvas.features.device)), "rulebook")
    no = int(n_out.item())
    mask = ((nbr[:no] >= 0).to(torch.int64) << torch.arange(9, device="cuda")).sum(1)
    for reach in (0, 2048, 8192, 65536, 1 << 30):
        if reach == 0:
            od, name = order, "device order (2048)"
        else:
            key = (torch.arange(no, device="cuda") // reach) * 512 + mask
            perm = torch.sort(key, stable=True)[1].to(torch.int32)
            od = torch.full_like(order, -1)
            od[:no] = perm
            name = f"mask groups over {reach if reach < (1 << 30) else 'all'} rows"
        idx2 = sp.SparseIndices(None, nbr, cap, (1, ho, wo), 9, od, n_out)
        r2, t2 = timed(lambda: sp.features_bf16x3(canvas.features, idx2, packed, cin, conv0.cout, conv0.b, None, None, None, True))
        # executed steps: per 256-row tile the taps any row has, x 2 chunks
        om = mask[od[:no].long()]
        pad = (-no) % 256
        om = torch.cat([om, om.new_zeros(pad)]).reshape(-1, 256)
        tile_or = om[:, 0].clone()
        for j in range(1, 256):
            tile_or |= om[:, j]
        taps = sum(((tile_or >> k) & 1).sum().item() for k in range(9))
        print(f"{name}: features {t2:.0f} us; taps executed per tile {taps / om.shape[0]:.2f} (pairs per row "
              f"{float((nbr[:no] >= 0).sum()) / no:.2f})")
