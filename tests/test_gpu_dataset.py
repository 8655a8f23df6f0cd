"""GPU: the data side of the path: IR-pattern extraction on the device against the oracle's
restatement of datasets/dataset_utils.py:33-46 (cv2 INTER_AREA restated: parity unpinned, see the oracle's
header), and the synthetic MessytableDataset-shaped loader driving the train.py-shaped step."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from activezero_amd.datasets.dataset_utils_gpu import get_smoothed_ir_pattern2  # noqa: E402
from activezero_amd.datasets.messytable_synthetic import SyntheticMessytableDataset  # noqa: E402
from oracle import ir_pattern_oracle as io  # noqa: E402

DEV = "cuda:0"


@pytest.mark.parametrize("hw,ks", [((540, 960), 11), ((256, 512), 11), ((90, 131), 7), ((44, 66), 11)])
def test_ir_pattern_vs_oracle(hw, ks):
    rng = np.random.default_rng(hw[0])
    base = rng.random(hw)
    dots = (rng.random(hw) < 0.08) * 0.4
    ir, plain = np.clip(base * 0.5 + dots, 0, 1), base * 0.5
    want, margin = io.get_smoothed_ir_pattern2(ir, plain, ks, 0.005, return_margin=True)
    got = get_smoothed_ir_pattern2(torch.tensor(ir, dtype=torch.float32, device=DEV),
                                   torch.tensor(plain, dtype=torch.float32, device=DEV), ks, 0.005).cpu().numpy()
    # the binary decision may only differ where the fp64 margin is within fp32 rounding of the threshold
    differs = got != want
    assert np.all(np.abs(margin[differs] - 0.005) < 2e-6), np.abs(margin[differs] - 0.005).max()
    assert differs.mean() < 1e-4
    assert 0.01 < got.mean() < 0.5


def test_ir_pattern_batched_equals_per_image():
    g = torch.Generator(device=DEV).manual_seed(0)
    a = torch.rand(3, 128, 160, device=DEV, generator=g)
    b = torch.rand(3, 128, 160, device=DEV, generator=g)
    whole = get_smoothed_ir_pattern2(a, b)
    for i in range(3):
        assert torch.equal(whole[i], get_smoothed_ir_pattern2(a[i], b[i]))


def test_synthetic_item_dictionary_matches_reference_layout():
    """keys / shapes / dtypes of datasets/messytable.py:184-306 (sim) and :308-404 (real, training)"""
    ds = SyntheticMessytableDataset(length=3, height=256, width=512, onReal=True, device=DEV)
    item = ds[1]
    h, w = 256, 512
    want = {"img_sim_L": (3, h, w), "img_sim_R": (3, h, w), "img_sim_L_reproj": (1, h, w), "img_sim_R_reproj": (1, h, w),
            "img_disp_L": (1, 2 * h, 2 * w), "img_depth_L": (1, 2 * h, 2 * w), "img_disp_R": (1, 2 * h, 2 * w),
            "img_depth_R": (1, 2 * h, 2 * w), "focal_length": (1, 1, 1), "baseline": (1, 1, 1),
            "img_real_L": (3, h, w), "img_real_R": (3, h, w), "img_real_L_reproj": (1, h, w), "img_real_R_reproj": (1, h, w)}
    for k, shape in want.items():
        assert tuple(item[k].shape) == shape and item[k].dtype == torch.float32, k
    assert isinstance(item["prefix"], str)
    assert set(item["img_sim_L_reproj"].unique().tolist()) <= {0.0, 1.0}
    # determinism per index, variation across indices
    again = ds[1]
    assert all(torch.equal(item[k], again[k]) for k in want)
    assert not torch.equal(item["img_sim_L"], ds[2]["img_sim_L"])
    # the default collate of a DataLoader stacks it like the reference's items
    batch = next(iter(torch.utils.data.DataLoader(ds, batch_size=2, num_workers=0)))
    assert tuple(batch["img_disp_L"].shape) == (2, 1, 2 * h, 2 * w) and tuple(batch["focal_length"].shape) == (2, 1, 1, 1)


def test_train_rehearsal_runs_and_learns():
    """two iterations of the train.py-shaped loop on the synthetic loader: finite losses, parameters move"""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import train_rehearsal as tr
    from activezero_amd.nets.psmnet.psmnet_3 import PSMNet

    torch.manual_seed(1)
    ds = SyntheticMessytableDataset(length=2, height=256, width=512, device=DEV)
    loader = torch.utils.data.DataLoader(ds, batch_size=1, num_workers=0)
    model = PSMNet(tr.MAX_DISP).to(DEV)
    opt = torch.optim.Adam(model.parameters(), lr=2e-4)
    w0 = model.classif3[2].weight.detach().clone()
    for sample in loader:
        s, r = tr.train_sample(sample, model, opt)
        assert np.isfinite(s) and np.isfinite(r)
    assert float((model.classif3[2].weight - w0).abs().max()) > 0
