"""world_size-2 gloo test of the window sharding + all-gather reassembly (the N>1 path of bench.py / pipeline)."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, num_windows, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from geo4d_amd import dist as gd
    r, w, _ = gd.init_from_env(backend="gloo")
    mine = gd.shard_windows(num_windows, r, w)
    local = torch.stack([torch.full((3, 4), float(i)) for i in mine]) if mine else torch.zeros((0, 3, 4))
    full = gd.all_gather_windows(local, num_windows)
    ok = all(bool((full[i] == float(i)).all()) for i in range(num_windows)) and full.shape == (num_windows, 3, 4)
    q.put((rank, mine, ok))
    dist.barrier()
    dist.destroy_process_group()


def test_window_sharding_and_allgather_world2():
    ctx = mp.get_context("spawn")
    for num_windows in (5, 14):          # odd count: rank 1 holds one window fewer (padded chunk)
        q = ctx.Queue()
        port = _free_port()
        procs = [ctx.Process(target=_worker, args=(r, 2, port, num_windows, q)) for r in range(2)]
        [p.start() for p in procs]
        res = sorted(q.get(timeout=120) for _ in range(2))
        [p.join(timeout=60) for p in procs]
        assert res[0][1] == list(range(0, num_windows, 2)) and res[1][1] == list(range(1, num_windows, 2))
        assert res[0][2] and res[1][2]


def _stub_synth(model, prompts, videos, noise_shape, n_samples=1, x_T=None, cond=None, **kw):
    """Stands in for the HIP synthesis on CPU: a deterministic function of the window's frames, its noise and the CPU RNG
    state run_clip seeds per window."""
    B, _, T, h, w = noise_shape
    v = videos.mean(dim=(1, 3, 4)).reshape(B, 1, 1, T, 1, 1)
    out = torch.zeros((B, 1, 11, T, 8 * h, 8 * w)) + v + x_T.mean() + torch.randn(1) + cond["c_crossattn"][0].mean()
    return out


class _StubModel:
    class model:
        conditioning_key = "hybrid"

        class diffusion_model:
            out_channels = 16


def _clip_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from geo4d_amd import dist as gd
    from geo4d_amd.pipeline import run_clip
    gd.init_from_env(backend="gloo")
    video = torch.arange(1 * 3 * 22 * 16 * 16, dtype=torch.float32).reshape(1, 3, 22, 16, 16) / 1e4
    slices, maps = run_clip(_StubModel, video, torch.ones((1, 333, 8)), ddim_steps=2, synthesize=_stub_synth)
    q.put((rank, [(s.start, s.stop) for s in slices], maps.numpy()))   # by value: a torch tensor would travel as a shared-memory fd
                                                                      # served by this process, which may have exited before the parent reads it
    dist.barrier()
    dist.destroy_process_group()


def test_run_clip_world2_equals_world1():
    """run_clip shards windows over ranks and all-gathers; per-window seeding makes the result independent of world size."""
    from geo4d_amd.pipeline import run_clip
    video = torch.arange(1 * 3 * 22 * 16 * 16, dtype=torch.float32).reshape(1, 3, 22, 16, 16) / 1e4
    slices1, maps1 = run_clip(_StubModel, video, torch.ones((1, 333, 8)), ddim_steps=2, synthesize=_stub_synth)
    assert [(s.start, s.stop) for s in slices1] == [(0, 16), (4, 20), (6, 22)] and maps1.shape == (3, 11, 16, 16, 16)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_clip_worker, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in procs]
    res = sorted((q.get(timeout=180) for _ in range(2)), key=lambda t: t[0])
    [p.join(timeout=60) for p in procs]
    for rank, sl, maps in res:
        assert sl == [(0, 16), (4, 20), (6, 22)]
        assert torch.equal(torch.from_numpy(maps), maps1), f"rank {rank}: sharded clip differs from the single-process clip"


def test_shard_tables():
    from geo4d_amd.dist import shard_windows, window_owner_table
    assert [len(shard_windows(30, r, 8)) for r in range(8)] == [4, 4, 4, 4, 4, 4, 3, 3]
    assert window_owner_table(14, 8) == ([2, 2, 2, 2, 2, 2, 1, 1], 2)
    assert sorted(sum((shard_windows(14, r, 8) for r in range(8)), [])) == list(range(14))
