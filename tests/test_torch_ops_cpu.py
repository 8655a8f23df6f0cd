"""CPU: the `azhip::*` dispatcher registration (activezero_amd/torch_ops.py): schemas exist, shape inference
runs on the meta device (what FakeTensor / torch.compile tracing uses) with no GPU, CPU tensors are refused."""
import pytest
import torch

import activezero_amd.torch_ops  # noqa: F401  (registers the operators)


def test_schemas_registered():
    for name in ("warp_scatter", "cost_volume", "softargmin", "softargmin_fwd", "softargmin_bwd", "warp_gather",
                 "warp_gather_bwd", "local_contrast_norm", "cost_volume_bwd", "conv3d", "conv3d_input_grad",
                 "conv3d_weight_grad", "bn3d", "patch_reproj", "patch_reproj_bwd"):
        assert hasattr(torch.ops.azhip, name), name
    assert "Tensor logits" in str(torch.ops.azhip.softargmin.default._schema)


def test_meta_shape_inference_without_a_gpu():
    m = lambda *s, dtype=torch.float32: torch.empty(*s, dtype=dtype, device="meta")
    assert torch.ops.azhip.softargmin(m(2, 48, 34, 60)).shape == (2, 1, 136, 240)
    assert torch.ops.azhip.softargmin(m(2, 1, 48, 34, 60)).shape == (2, 1, 136, 240)
    assert torch.ops.azhip.cost_volume(m(2, 32, 34, 60), m(2, 32, 34, 60), 48).shape == (2, 64, 48, 34, 60)
    assert torch.ops.azhip.warp_gather(m(2, 3, 16, 24), m(2, 1, 16, 24)).shape == (2, 3, 16, 24)
    assert torch.ops.azhip.warp_scatter(m(2, 1, 16, 24), m(2, 1, 16, 24, dtype=torch.int32), 1).shape == (2, 1, 16, 24)
    n, s = torch.ops.azhip.local_contrast_norm(m(2, 1, 16, 24), 9, 1e-5)
    assert n.shape == s.shape == (2, 1, 16, 24)
    # K4/K5 on a channels-last volume: stride 1, stride 2, transposed stride 2 (psmnet_3.py:15-58)
    assert torch.ops.azhip.conv3d(m(2, 12, 34, 60, 32), m(32, 32, 3, 3, 3), 0).shape == (2, 12, 34, 60, 32)
    assert torch.ops.azhip.conv3d(m(2, 12, 34, 60, 32), m(64, 32, 3, 3, 3), 1).shape == (2, 6, 17, 30, 64)
    assert torch.ops.azhip.conv3d(m(2, 6, 17, 30, 64), m(64, 32, 3, 3, 3), 2).shape == (2, 12, 34, 60, 32)
    assert torch.ops.azhip.bn3d(m(2, 6, 17, 30, 64), m(64), m(64), None, True).shape == (2, 6, 17, 30, 64)
    assert torch.ops.azhip.patch_reproj(m(2, 1, 32, 48), m(2, 1, 32, 48), m(2, 1, 32, 48), None, 11).shape == ()


def test_cpu_tensors_are_refused():
    with pytest.raises(RuntimeError):
        torch.ops.azhip.softargmin(torch.zeros(1, 4, 3, 3))
    with pytest.raises(RuntimeError):
        torch.ops.azhip.warp_gather(torch.zeros(1, 1, 4, 4), torch.zeros(1, 1, 4, 4))
