"""Synthetic stand-in for the reference's MessytableDataset (datasets/messytable.py:184-306 sim item,
:308-404 real item) -- SURVEY.md 8f-4.  The private MessyTable data cannot travel, so this dataset RENDERS
items of the same dictionary shape on the device: same keys, shapes, dtypes and value conventions, so that
the loop of train.py:220-432 (see tools/train_rehearsal.py, its restatement for the keys this path consumes)
runs end to end without the dataset:

  img_sim_L / img_sim_R            [3,H,W]   ImageNet-normalised grey images (dataset_utils.py:76-81)
  img_sim_L_reproj / _R_reproj     [1,H,W]   binary IR pattern, extracted ON THE GPU from the (IR, no-IR)
                                             image pair with get_smoothed_ir_pattern2 (dataset_utils.py:33-46)
  img_disp_L / img_disp_R          [1,2H,2W] disparity at the 2x resolution of the depth maps (messytable.py:252-261)
  img_depth_L / img_depth_R        [1,2H,2W] metres;  focal_length, baseline [1,1,1];  prefix (str)
  (onReal) img_real_L / img_real_R [3,H,W],  img_real_L_reproj / img_real_R_reproj [1,H,W]

Geometry: a smooth random depth field -> disparity = focal * baseline / depth; the right view is the left one
moved by the (integer-rounded) disparity with the scatter warp K1, so disparity, images and patterns are
mutually consistent and the losses have signal."""
import torch
import torch.nn.functional as F

from activezero_amd.datasets.dataset_utils_gpu import get_smoothed_ir_pattern2
from activezero_amd.utils.warp_ops import apply_disparity_cu

_MEAN = (0.485, 0.456, 0.406)
_STD = (0.229, 0.224, 0.225)


class SyntheticMessytableDataset(torch.utils.data.Dataset):
    def __init__(self, length=64, height=256, width=512, onReal=True, device="cuda:0", seed=0, max_disp=192):
        self.length, self.h, self.w, self.onReal = int(length), int(height), int(width), bool(onReal)
        self.device, self.seed, self.max_disp = torch.device(device), int(seed), int(max_disp)
        self.focal_length, self.baseline = 446.31, 0.055  # the order of the MessyTable rig (metres, half-res pixels)

    def __len__(self):
        return self.length

    def _gen(self, idx, salt):
        return torch.Generator(device=self.device).manual_seed(self.seed * 1000003 + idx * 7 + salt)

    def _smooth(self, g, h, w, cells):
        low = torch.rand(1, 1, cells, 2 * cells, device=self.device, generator=g)
        return F.interpolate(low, size=(h, w), mode="bicubic", align_corners=False).clamp(0, 1)[0, 0]

    def _views(self, g):
        """(left, right) grey images with IR dots, their no-IR versions, left/right disparity at 2x resolution"""
        h, w = self.h, self.w
        depth2 = 0.45 + 1.1 * self._smooth(g, 2 * h, 2 * w, 6)             # metres, 2x resolution
        disp2 = self.focal_length * self.baseline / depth2                 # half-res pixels (messytable.py:205-213)
        disp = F.avg_pool2d(disp2[None, None], 2)[0, 0]
        tex = 0.25 + 0.5 * self._smooth(g, h, w, 24)
        dots = (torch.rand(h, w, device=self.device, generator=g) < 0.06).float()
        left_no_ir, left = tex, (tex + 0.35 * dots).clamp(0, 1)
        shift = (-disp.round()).int()[None, None].contiguous()             # left -> right: x - d
        warp = lambda im: apply_disparity_cu(im[None, None].contiguous(), shift)[0, 0]
        right, right_no_ir = warp(left), warp(left_no_ir)
        disp_r2 = F.interpolate(warp(disp)[None, None], scale_factor=2, mode="nearest")[0, 0]
        depth_r2 = torch.where(disp_r2 > 0, self.focal_length * self.baseline / disp_r2.clamp_min(1e-6),
                               torch.zeros_like(disp_r2))
        return left, right, left_no_ir, right_no_ir, disp2, depth2, disp_r2, depth_r2

    def _normalise(self, grey):
        rgb = grey[None].expand(3, -1, -1)
        mean = torch.tensor(_MEAN, device=self.device).view(3, 1, 1)
        std = torch.tensor(_STD, device=self.device).view(3, 1, 1)
        return ((rgb - mean) / std).contiguous()

    def __getitem__(self, idx):
        g = self._gen(idx, 1)
        left, right, left0, right0, disp2, depth2, disp_r2, depth_r2 = self._views(g)
        pat = get_smoothed_ir_pattern2(torch.stack([left, right]), torch.stack([left0, right0]))
        item = {
            "img_sim_L": self._normalise(left), "img_sim_R": self._normalise(right),
            "img_sim_L_reproj": pat[0:1].contiguous(), "img_sim_R_reproj": pat[1:2].contiguous(),
            "img_disp_L": disp2[None].contiguous(), "img_depth_L": depth2[None].contiguous(),
            "img_disp_R": disp_r2[None].contiguous(), "img_depth_R": depth_r2[None].contiguous(),
            "prefix": f"synthetic-{idx:05d}",
            "focal_length": torch.full((1, 1, 1), self.focal_length, device=self.device),
            "baseline": torch.full((1, 1, 1), self.baseline, device=self.device),
        }
        if self.onReal:
            g = self._gen(idx, 2)
            rl, rr, rl0, rr0, *_ = self._views(g)
            noise = lambda im: (im + 0.02 * torch.randn(im.shape, device=self.device, generator=g)).clamp(0, 1)
            rl, rr = noise(rl), noise(rr)
            rpat = get_smoothed_ir_pattern2(torch.stack([rl, rr]), torch.stack([rl0, rr0]))
            item.update({"img_real_L": self._normalise(rl), "img_real_R": self._normalise(rr),
                         "img_real_L_reproj": rpat[0:1].contiguous(), "img_real_R_reproj": rpat[1:2].contiguous()})
        return item
