# Drop-in shim (as nets/ and utils/): with activezero_amd/ ahead of the reference checkout on sys.path,
# `datasets.dataset_utils_gpu` and `datasets.messytable_synthetic` resolve here while `datasets.messytable`,
# `datasets.dataset_utils` keep resolving from the reference tree.
from pkgutil import extend_path

__path__ = extend_path(__path__, __name__)
