"""Multi-GPU layer: one process per GPU (``torch.distributed``, backend "nccl" = RCCL over xGMI on ROCm).

The DENOISING shards at WINDOW granularity only: a 16-frame window is an independent unit (own noise,
own conditioning, test_geo4d.py:431-443), while inside a window every temporal layer couples all frames. Weights are
replicated (2.9 GB bf16), windows are dealt round-robin, and the collective on that path is one all-gather of the decoded
maps ``[W_local, 11, 16, H, W]`` fp32 (115 MB per window at 320x512) so that the alignment stage sees the whole clip; it
can be issued asynchronously (RCCL's own stream) while the next window denoises. The VAE DECODE has no cross-frame
dependency (ddpm3d.py:810-819), so a window's 4 x T frame-modalities can also be FRAME-SHARDED: the owner broadcasts its
2.6 MB latent, rank r decodes frames ``frame_shard(T, r, world)`` of all four modalities and one all-gather along the frame
axis reassembles ``[B, 11, T, H, W]`` (pipeline.decode_modalities_sharded) — the mode for single-window latency and for
ragged last rounds (14 windows on 8 GPUs).
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
    backend = backend or os.environ.get("GEO4D_DIST_BACKEND") or None
    if os.environ.get("GEO4D_SINGLE_DEVICE") == "1":      # test rig: several ranks share GPU 0 (gloo only; RCCL refuses duplicate devices)
        local = 0
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if torch.cuda.is_available():
            torch.cuda.set_device(local)
        # a collective that a peer never joins (a rank-specific failure) must end in an error, not a hang: GEO4D_DIST_TIMEOUT_S (default 600 s)
        import datetime
        dist.init_process_group(backend=backend, rank=rank, world_size=world,
                                timeout=datetime.timedelta(seconds=float(os.environ.get("GEO4D_DIST_TIMEOUT_S", "600"))))
    elif torch.cuda.is_available():
        torch.cuda.set_device(local)
    return rank, world, local


def shard_windows(num_windows, rank, world):
    """Window w -> rank w % world. Every rank computes the same integer table (no communication)."""
    return list(range(rank, num_windows, world))


def window_owner_table(num_windows, world):
    counts = [len(range(r, num_windows, world)) for r in range(world)]
    return counts, max(counts) if counts else 0


class _Done:
    def wait(self):
        return True


def _all_gather(out, inp, group=None, async_op=False):
    if inp.is_cuda and dist.get_backend(group) == "nccl":
        return dist.all_gather_into_tensor(out, inp, group=group, async_op=async_op)    # one RCCL all-gather over xGMI
    if inp.is_cuda:        # gloo with device tensors (single-GPU test rig of the N > 1 path): stage through the host
        host = torch.empty(out.shape, dtype=out.dtype)
        dist.all_gather(list(host.chunk(dist.get_world_size(group), 0)), inp.cpu(), group=group)
        out.copy_(host)
        return _Done() if async_op else None
    return dist.all_gather(list(out.chunk(dist.get_world_size(group), 0)), inp, group=group, async_op=async_op)   # gloo (CPU tests)


class PendingGather:
    """An all-gather in flight (RCCL runs it on its own stream: the caller keeps enqueueing the next window's denoise).
    ``wait()`` orders the current stream after the collective and returns the windows in global order."""

    def __init__(self, work, gathered, num_windows, world, cmax):
        self.work, self.gathered, self.num_windows, self.world, self.cmax = work, gathered, num_windows, world, cmax
        self._out = None

    def wait(self):
        if self._out is None:
            if self.work is not None:
                self.work.wait()
            g = self.gathered.reshape((self.world, self.cmax) + tuple(self.gathered.shape[1:]))
            # window w = j * world + r sits at gathered[r, j]: one strided device copy puts the clip in window order
            self._out = g.transpose(0, 1).reshape((self.world * self.cmax,) + tuple(g.shape[2:]))[: self.num_windows]
        return self._out


def all_gather_windows(local, num_windows, rank=None, world=None, group=None, async_op=False, force=False):
    """local: [n_local, ...] decoded maps of this rank's windows (in increasing window order).
    Returns [num_windows, ...] in global window order on every rank (or, with ``async_op``, a PendingGather)."""
    if world is None:
        world = dist.get_world_size(group) if dist.is_initialized() else 1
    if rank is None:
        rank = dist.get_rank(group) if dist.is_initialized() else 0
    if world == 1 and not (force and dist.is_initialized()):    # `force`: run the collective even on one rank (RCCL smoke test)
        assert local.shape[0] == num_windows
        return PendingGather(None, local, num_windows, 1, num_windows) if async_op else local
    counts, cmax = window_owner_table(num_windows, world)
    assert local.shape[0] == counts[rank], (local.shape, counts, rank)
    pad = local
    if counts[rank] < cmax:
        pad = torch.cat([local, local.new_zeros((cmax - counts[rank],) + tuple(local.shape[1:]))], 0)
    pad = pad.contiguous()
    gathered = torch.empty((world * cmax,) + tuple(local.shape[1:]), device=local.device, dtype=local.dtype)
    work = _all_gather(gathered, pad, group=group, async_op=async_op)
    pending = PendingGather(work if async_op else None, gathered, num_windows, world, cmax)
    return pending if async_op else pending.wait()


def frame_shard(T, rank, world):
    """Frames [lo, hi) of a T-frame window decoded by `rank` when the VAE decode is frame-sharded (ddpm3d.py:810-819: the
    reference's per-frame loop has no cross-frame dependency). Contiguous, sizes differ by at most one."""
    base, extra = divmod(T, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


class PendingFrames:
    """A frame-axis all-gather in flight; ``wait()`` returns [..., T, ...] with the frame axis back at `dim`."""

    def __init__(self, work, gathered, sizes, tmax, dim):
        self.work, self.gathered, self.sizes, self.tmax, self.dim = work, gathered, sizes, tmax, dim
        self._out = None

    def wait(self):
        if self._out is None:
            if self.work is not None:
                self.work.wait()
            if all(hi - lo == self.tmax for lo, hi in self.sizes):
                frames = self.gathered
            else:
                frames = torch.cat([self.gathered[r * self.tmax: r * self.tmax + (hi - lo)] for r, (lo, hi) in enumerate(self.sizes)], 0)
            self._out = frames.movedim(0, self.dim)
        return self._out


def all_gather_frames(local, T, rank=None, world=None, group=None, dim=2, async_op=False, force=False):
    """local: this rank's decoded frames [..., t_local, ...] (frame axis `dim`, the slice frame_shard gives) -> all T frames
    on every rank. Ragged slices are padded to the largest one so a single all-gather serves every rank. With ``async_op`` a
    PendingFrames is returned (the collective runs on RCCL's stream while the caller enqueues the next window)."""
    if world is None:
        world = dist.get_world_size(group) if dist.is_initialized() else 1
    if rank is None:
        rank = dist.get_rank(group) if dist.is_initialized() else 0
    if world == 1 and not (force and dist.is_initialized()):
        return PendingFrames(None, local.movedim(dim, 0), [(0, T)], T, dim) if async_op else local
    sizes = [frame_shard(T, r, world) for r in range(world)]
    tmax = max(hi - lo for lo, hi in sizes)
    x = local.movedim(dim, 0)                                   # frames first: a rank's chunk is contiguous
    if x.shape[0] < tmax:
        x = torch.cat([x, x.new_zeros((tmax - x.shape[0],) + tuple(x.shape[1:]))], 0)
    x = x.contiguous()
    gathered = torch.empty((world * tmax,) + tuple(x.shape[1:]), device=x.device, dtype=x.dtype)
    work = _all_gather(gathered, x, group=group, async_op=async_op)
    pending = PendingFrames(work if async_op else None, gathered, sizes, tmax, dim)
    return pending if async_op else pending.wait()


def broadcast_from(t, src, group=None):
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.broadcast(t, src=src, group=group)
    return t
