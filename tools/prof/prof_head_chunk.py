import sys, torch, numpy as np
sys.path.insert(0, '/root/repo')
from paddle3d_amd import centerpoint as cpm
torch.manual_seed(0)
model = cpm.centerpoint_pillars_nuscenes(max_num_voxels=(30000, 30000)).cuda().eval()
head = model.bbox_head
x = torch.randn(16, 384, 128, 128, device='cuda')
ref = None
for k in (36, 1, 2, 3, 4, 6, 9, 12, 18):
    head.head_chunk = k
    with torch.no_grad():
        for _ in range(2): rets, _ = head(x)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): rets, _ = head(x)
        e1.record(); torch.cuda.synchronize()
    cat = torch.cat([rets[t][h] for t in range(len(rets)) for h in sorted(rets[t])], 1)
    if ref is None: ref = cat
    print(k, "heads per slice: %.3f ms per 16 frames; equal to unsliced: %s" % (e0.elapsed_time(e1) / 10, torch.equal(cat, ref)))
