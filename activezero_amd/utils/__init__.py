# Drop-in shim: when /root/repo/activezero_amd precedes the reference checkout on
# sys.path, `utils.warp_ops`, `utils.reprojection` and `utils.cascade_metrics` resolve here
# (plus the new `utils.disp_losses`) while the other modules (utils.losses, utils.util,
# utils.reduce, ...) keep resolving from the reference tree.
from pkgutil import extend_path

__path__ = extend_path(__path__, __name__)
