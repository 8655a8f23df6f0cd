"""6-channel-input twin of psmnet_submodule_3 (image + adapter output are
concatenated in front of the first conv); mirrors the reference module
nets/psmnet/psmnet_submodule.py, which differs from the 3-channel file only at
lines 99 and 167-174."""
from activezero_amd.nets.psmnet.psmnet_submodule_3 import *  # noqa: F401,F403
from activezero_amd.nets.psmnet import psmnet_submodule_3 as _base

__all__ = list(_base.__all__)


class FeatureExtraction(_base.FeatureExtraction):
    IN_CHANNELS = 6

    def forward(self, x, x_transformed):
        """([B,3,H,W], [B,3,H,W]) -> [B,32,H/4,W/4]"""
        return self._trunk(torch.cat((x, x_transformed), 1).contiguous(memory_format=torch.channels_last))
