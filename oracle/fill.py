"""Deterministic parameter fill keyed by state-dict names (TEST INFRASTRUCTURE): lets the
reference model (in oracle/gen_golden.py) and this package's mirror (in tests/) hold identical
weights without shipping them."""
import zlib

import numpy as np
import torch

from . import synth


def fill_state(model, seed=0):
    sd = model.state_dict()
    with torch.no_grad():
        for name in sorted(sd):
            t = sd[name]
            if not torch.is_floating_point(t):
                continue
            s = (zlib.crc32(name.encode()) ^ (seed * 2654435761)) & 0x7FFFFFFF
            shape = tuple(t.shape)
            if name.endswith("running_var"):
                v = synth.uniform(s, shape, 0.5, 1.5)
            elif name.endswith("running_mean"):
                v = synth.uniform(s, shape, -0.1, 0.1)
            elif t.dim() >= 2:
                fan_in = max(int(np.prod(shape[1:])), 1) if "W_branch" not in name else shape[1]
                b = 1.0 / np.sqrt(fan_in)
                v = synth.uniform(s, shape, -b, b)
            elif name.endswith("weight"):            # norm scales
                v = synth.uniform(s, shape, 0.5, 1.5)
            else:                                    # biases
                v = synth.uniform(s, shape, -0.05, 0.05)
            t.copy_(torch.from_numpy(v))
    return model
