"""Procedural, name-keyed weights shared by tools/make_goldens.py (which loads
them into the imported reference) and the tests (which load them into the
oracle and the product module).  Nothing but (name, shape) decides a tensor, so
no state-dict has to be committed and key compatibility is tested for free."""
import zlib

import numpy as np
import torch


def procedural_tensor(name, shape, dtype=torch.float32):
    rng = np.random.default_rng(zlib.crc32(name.encode()))
    leaf = name.rsplit(".", 1)[-1]
    if leaf == "num_batches_tracked":
        return torch.zeros(shape, dtype=torch.long)
    if len(shape) >= 3:  # conv / deconv kernels
        fan = int(np.prod(shape[2:])) * shape[0]
        arr = rng.normal(0.0, np.sqrt(2.0 / fan), size=shape)
        if len(shape) == 5 and shape[0] == 1:
            # 32->1 classifier: the reference's init rule (fan = 27) makes the logits so
            # large that softmax is one-hot and the disparity saturates to integers; a
            # trained net is far softer.  Damp it so the goldens probe the soft regime.
            arr = arr * 0.2
    elif leaf == "running_var":
        arr = rng.uniform(0.5, 1.5, size=shape)
    elif leaf == "running_mean":
        arr = rng.normal(0.0, 0.1, size=shape)
    elif leaf == "weight":  # BN gamma
        arr = rng.uniform(0.5, 1.5, size=shape)
    else:  # BN beta
        arr = rng.normal(0.0, 0.1, size=shape)
    return torch.from_numpy(np.asarray(arr, dtype=np.float32)).to(dtype)


def procedural_state_dict(module, prefix=""):
    return {k: procedural_tensor(prefix + k, tuple(v.shape), v.dtype)
            for k, v in module.state_dict().items()}


def load_procedural(module, prefix=""):
    module.load_state_dict(procedural_state_dict(module, prefix), strict=True)
    return module


def seeded(shape, seed, lo=-1.0, hi=1.0):
    rng = np.random.default_rng(seed)
    return torch.from_numpy(rng.uniform(lo, hi, size=shape).astype(np.float32))


def load_bn_buffers(module, golden, prefix="buf::"):
    """Overwrite BatchNorm running statistics with the calibrated ones a golden file
    carries (keys `buf::<state-dict name>`); everything else stays procedural."""
    sd = module.state_dict()
    n = 0
    for key in golden.files:
        if key.startswith(prefix):
            name = key[len(prefix):]
            sd[name].copy_(torch.from_numpy(golden[key]))
            n += 1
    assert n > 0
    return module
