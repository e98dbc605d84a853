"""Deterministic, name-keyed parameter fill shared by the golden generator (applied to the reference modules),
the oracle and the HIP modules: every tensor depends only on (name, shape), so no weights need to be stored.

The reference zero-initialises many layers (openaimodel3d.py:179,269-270,383-384,555; attention.py:288-290,360-362),
which would make a freshly built network output exactly 0 — here every tensor is random."""
import zlib

import torch


def fill_tensor(name, shape, gain=1.0):
    g = torch.Generator(device="cpu").manual_seed(zlib.crc32(name.encode()) & 0x7FFFFFFF)
    t = torch.randn(tuple(shape), generator=g, dtype=torch.float32)
    if len(shape) <= 1:
        if name.endswith("weight"):      # norm gain
            return 1.0 + 0.1 * t
        return 0.05 * t                  # bias / norm shift / scalars
    fan_in = 1
    for s in shape[1:]:
        fan_in *= s
    return t * (gain / fan_in ** 0.5)


def seeded_state_dict(shapes, gain=1.0):
    """shapes: mapping name -> shape (e.g. {k: v.shape for k, v in module.state_dict().items()})."""
    return {k: fill_tensor(k, tuple(s), gain) for k, s in shapes.items()}


def fill_module_(module, gain=1.0):
    sd = module.state_dict()
    new = seeded_state_dict({k: v.shape for k, v in sd.items()}, gain)
    module.load_state_dict(new, strict=True)
    return module
