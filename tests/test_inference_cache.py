"""The derived-weight cache of the parameter-holding layers (paddle3d_amd/centerpoint.py: _InferenceCache,
_param_signature, invalidate_derived): which writes it detects by itself, and the explicit invalidation for the one
kind it cannot see (`param.data.*`)."""
import pytest
import torch

from paddle3d_amd import centerpoint as cpm


def test_signature_sees_tensor_writes_but_not_data_writes():
    m = cpm.SecondBackbone(8, (8,), (1,), (2,)).eval()
    w = next(m.parameters())
    s0 = cpm._param_signature(m)
    with torch.no_grad():
        w.mul_(2.0)
    s1 = cpm._param_signature(m)
    assert s1 != s0                      # in-place op on the parameter: version bump
    w.data.mul_(2.0)
    assert cpm._param_signature(m) == s1  # `.data` has a version counter of its own: invisible (documented)
    m.load_state_dict({k: v.clone() for k, v in m.state_dict().items()})
    assert cpm._param_signature(m) != s1  # copy_ into the parameters


def test_invalidate_drops_every_derived_weight():
    model = cpm.centerpoint_pillars_nuscenes(max_num_voxels=(100, 100)).eval()
    holders = [m for m in model.modules() if hasattr(m, "_drop_cache")]
    assert len(holders) >= 3
    for m in holders:
        m._store_cache(("sentinel",))
    model.voxel_encoder._folded = ("sig", ("sentinel",))
    assert all(m._cache_valid() for m in holders)
    assert model.invalidate() is model
    assert all(m._cache is None for m in holders) and model.voxel_encoder._folded is None
    holders[0]._store_cache(("again",))
    holders[0].invalidate()
    assert holders[0]._cache is None


@pytest.mark.gpu
def test_data_write_needs_invalidate_on_device():
    """forward, then `weight.data.mul_(2)`, then forward: the stale folded weights are still used (the documented
    limit of the signature); `invalidate()` makes the next forward see the write; an in-place op on the parameter
    itself needs nothing."""
    torch.manual_seed(0)
    m = cpm.SecondBackbone(64, (64,), (1,), (2,)).cuda().eval()
    x = torch.randn(1, 64, 64, 64, device="cuda")
    conv = m.blocks[0][0]
    with torch.no_grad():
        y0 = m(x)[0].clone()
        conv.weight.data.mul_(2.0)
        y_stale = m(x)[0].clone()
        m.invalidate()
        y1 = m(x)[0].clone()
        conv.weight.mul_(0.5)          # back to the original weights, through the tensor: detected
        y2 = m(x)[0].clone()
    assert torch.equal(y_stale, y0)
    assert not torch.allclose(y1, y0) and (y1 - y0).abs().max() > 1e-3
    assert torch.allclose(y2, y0, atol=1e-5)
