# Drop-in shim: when /root/repo/activezero_amd precedes the reference checkout on
# sys.path, `nets.psmnet.*` resolves here while the reference's other sub-packages
# (nets.adapter, nets.dispnet, nets.raft, ...) keep resolving from its own tree.
from pkgutil import extend_path

__path__ = extend_path(__path__, __name__)
