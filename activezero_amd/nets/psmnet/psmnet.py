"""PSMNet with 6-channel input (image + adapter-transformed image); drop-in for the
reference module nets/psmnet/psmnet.py, which differs from psmnet_3.py only in
the forward signature (psmnet.py:144-148) and the first 2-D conv."""
from activezero_amd.nets.psmnet.psmnet_submodule import *  # noqa: F401,F403
import torch

from activezero_amd.nets.psmnet import psmnet_3 as _p3
from activezero_amd.nets.psmnet import psmnet_submodule as _sub6

hourglass = _p3.hourglass


class PSMNet(_p3.PSMNet):
    _feature_module = _sub6

    def forward(self, img_L, img_R, img_L_transformed, img_R_transformed):
        cat = lambda a, b: self._nhwc(torch.cat((a, b), 1))  # psmnet_submodule.py:167 (image + adapter output)
        a = self._pass_arith(img_L)
        return self._from_features(*self.feature_extraction.forward_pair(cat(img_L, img_L_transformed),
                                                                          cat(img_R, img_R_transformed), a), a)
