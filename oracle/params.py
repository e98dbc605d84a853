"""Deterministic, name-keyed parameter fill shared by the golden generator (applied to the reference modules),
the oracle and the HIP modules: every tensor depends only on (name, shape), so no weights need to be stored.

The reference zero-initialises many layers (openaimodel3d.py:179,269-270,383-384,555; attention.py:288-290,360-362),
which would make a freshly built network output exactly 0 — here every tensor is random."""
import zlib

import torch


BIG = 1 << 21     # tensors above 2 Mi elements (only the shipped 1.44 B config has them) are tiled from a 1 Mi-sample draw


def _randn(shape, g):
    n = 1
    for s in shape:
        n *= s
    if n <= BIG:
        return torch.randn(tuple(shape), generator=g, dtype=torch.float32)
    # the scalar CPU generator needs minutes for 1.44e9 samples: draw 2^20 normals and lay them end to end, each copy with its own
    # sign (still a pure function of (name, shape); the per-tensor statistics are those of the draw)
    base = torch.randn(1 << 20, generator=g, dtype=torch.float32)
    reps = (n + base.numel() - 1) // base.numel()
    signs = (torch.randint(0, 2, (reps, 1), generator=g).float() * 2 - 1)
    return (base[None, :] * signs).reshape(-1)[:n].reshape(tuple(shape))


def fill_tensor(name, shape, gain=1.0):
    g = torch.Generator(device="cpu").manual_seed(zlib.crc32(name.encode()) & 0x7FFFFFFF)
    t = _randn(shape, g)
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
