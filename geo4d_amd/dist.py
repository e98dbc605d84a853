"""Multi-GPU layer: one process per GPU (``torch.distributed``, backend "nccl" = RCCL over xGMI on ROCm).

The hot path shards at WINDOW granularity only: a 16-frame window is an independent unit (own noise,
own conditioning, test_geo4d.py:431-443), while inside a window every temporal layer couples all frames. Weights are
replicated (2.9 GB bf16), windows are dealt round-robin, and the ONLY collective is one all-gather of the decoded maps
``[W_local, 11, 16, H, W]`` fp32 (115 MB per window at 320x512) so that the alignment stage sees the whole clip.
xGMI is point-to-point (7 links x ~153 GB/s per GPU): a single large all-gather per clip keeps every link busy once
instead of many small ones; per-rank window counts may differ by one, so chunks are padded to the maximum count.
"""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Initialise the default process group from RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* (torchrun contract)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    elif torch.cuda.is_available():
        torch.cuda.set_device(local)
    return rank, world, local


def shard_windows(num_windows, rank, world):
    """Window w -> rank w % world. Every rank computes the same integer table (no communication)."""
    return list(range(rank, num_windows, world))


def window_owner_table(num_windows, world):
    counts = [len(range(r, num_windows, world)) for r in range(world)]
    return counts, max(counts) if counts else 0


def all_gather_windows(local, num_windows, rank=None, world=None, group=None):
    """local: [n_local, ...] decoded maps of this rank's windows (in increasing window order).
    Returns [num_windows, ...] in global window order on every rank."""
    if world is None:
        world = dist.get_world_size(group) if dist.is_initialized() else 1
    if rank is None:
        rank = dist.get_rank(group) if dist.is_initialized() else 0
    if world == 1:
        assert local.shape[0] == num_windows
        return local
    counts, cmax = window_owner_table(num_windows, world)
    assert local.shape[0] == counts[rank], (local.shape, counts, rank)
    pad = local
    if counts[rank] < cmax:
        pad = torch.cat([local, local.new_zeros((cmax - counts[rank],) + tuple(local.shape[1:]))], 0)
    pad = pad.contiguous()
    gathered = torch.empty((world * cmax,) + tuple(local.shape[1:]), device=local.device, dtype=local.dtype)
    dist.all_gather_into_tensor(gathered, pad, group=group) if hasattr(dist, "all_gather_into_tensor") and local.is_cuda else \
        dist.all_gather(list(gathered.chunk(world, 0)), pad, group=group)
    gathered = gathered.reshape((world, cmax) + tuple(local.shape[1:]))
    out = torch.empty((num_windows,) + tuple(local.shape[1:]), device=local.device, dtype=local.dtype)
    for r in range(world):
        ids = list(range(r, num_windows, world))
        if ids:
            out[torch.tensor(ids, device=local.device)] = gathered[r, :len(ids)]
    return out
