"""activezero_amd -- MI355X-native hot path of ActiveZero (PSMNet cost volume,
3-D aggregation, soft-argmin, warp / reprojection ops).

Host side is Python and mirrors the reference's module surface
(`nets.psmnet.psmnet[_3]`, `utils.warp_ops`, `utils.reprojection`); the
computation lives in hand-written gfx950 kernels behind include/azhip.h.
"""
__version__ = "0.1.0"
