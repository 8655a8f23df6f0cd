"""CPU: the `azhip::*` dispatcher registration (activezero_amd/torch_ops.py): schemas exist, shape inference
runs on the meta device (what FakeTensor / torch.compile tracing uses) with no GPU, CPU tensors are refused."""
import pytest
import torch

import activezero_amd.torch_ops  # noqa: F401  (registers the operators)


def test_schemas_registered():
    for name in ("warp_scatter", "cost_volume", "softargmin", "softargmin_fwd", "softargmin_bwd", "warp_gather",
                 "warp_gather_bwd", "local_contrast_norm", "cost_volume_bwd"):
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


def test_cpu_tensors_are_refused():
    with pytest.raises(RuntimeError):
        torch.ops.azhip.softargmin(torch.zeros(1, 4, 3, 3))
    with pytest.raises(RuntimeError):
        torch.ops.azhip.warp_gather(torch.zeros(1, 1, 4, 4), torch.zeros(1, 1, 4, 4))
