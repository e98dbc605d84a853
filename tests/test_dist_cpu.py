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
    pending = gd.all_gather_windows(local, num_windows, async_op=True)       # the overlapped form used by bench.py
    ok = ok and torch.equal(pending.wait(), full)
    # frame-sharded decode reassembly: ragged frame slices (T = 5 over 2 ranks: 3 + 2)
    T = 5
    lo, hi = gd.frame_shard(T, r, w)
    part = torch.arange(lo, hi, dtype=torch.float32).reshape(1, 1, hi - lo, 1).expand(2, 11, hi - lo, 3).contiguous()
    frames = gd.all_gather_frames(part, T, dim=2)
    ok = ok and frames.shape == (2, 11, T, 3) and bool((frames[0, 0, :, 0] == torch.arange(T, dtype=torch.float32)).all())
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


def _stub_decoder(model, samples, pointmap_vae=None):
    """Frame-independent stand-in for the 4-modality decode: [B,16,t,h,w] -> [B,11,t,8h,8w]."""
    b, _, t, h, w = samples.shape
    return samples.mean(dim=(1, 3, 4)).reshape(b, 1, t, 1, 1).expand(b, 11, t, 8 * h, 8 * w).contiguous()


def _stub_synth(model, prompts, videos, noise_shape, n_samples=1, x_T=None, cond=None, decode=True, **kw):
    """Stands in for the HIP synthesis on CPU: a deterministic function of the window's frames, its noise and the CPU RNG
    state run_clip seeds per window. decode=False returns the 'latent' like image_guided_synthesis does."""
    B, _, T, h, w = noise_shape
    v = videos.mean(dim=(1, 3, 4)).reshape(B, 1, T, 1, 1)
    lat = torch.zeros((B, 16, T, h, w)) + v + x_T.mean() + torch.randn(1) + cond["c_crossattn"][0].mean()
    return (_stub_decoder(model, lat) if decode else lat)[:, None]


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
    _, maps_sh = run_clip(_StubModel, video, torch.ones((1, 333, 8)), ddim_steps=2, synthesize=_stub_synth, decode="sharded",
                          decoder=_stub_decoder)   # rounds of 2 windows, latents broadcast, frames decoded 8 + 8, gathered
    assert torch.equal(maps, maps_sh), "frame-sharded decode mode differs from the local-decode mode"
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
    from geo4d_amd.dist import frame_shard, shard_windows, window_owner_table
    assert [frame_shard(16, r, 8) for r in range(8)] == [(2 * r, 2 * r + 2) for r in range(8)]
    assert [frame_shard(5, r, 2) for r in range(2)] == [(0, 3), (3, 5)]
    assert [frame_shard(3, r, 4) for r in range(4)] == [(0, 1), (1, 2), (2, 3), (3, 3)]
    assert [len(shard_windows(30, r, 8)) for r in range(8)] == [4, 4, 4, 4, 4, 4, 3, 3]
    assert window_owner_table(14, 8) == ([2, 2, 2, 2, 2, 2, 1, 1], 2)
    assert sorted(sum((shard_windows(14, r, 8) for r in range(8)), [])) == list(range(14))


def test_bench_self_launches_n_ranks():
    """`python bench.py --gpus 2` with no WORLD_SIZE in the environment (the driver's command form) must bring up 2 ranks by itself;
    --launch-check runs the rendezvous + barrier / max-over-ranks protocol without kernels, so it works on the gloo CPU rig."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--launch-check"], env=env, capture_output=True,
                       text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout                                   # ONE JSON line, from rank 0
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["max_rank_seen"] == 1 and rec["config"]["parallelism"] == "window-dp2"
    # a launcher whose rank count disagrees with --gpus is refused with a message, not an assert
    env2 = dict(env, WORLD_SIZE="3", RANK="0", LOCAL_RANK="0")
    r2 = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--launch-check"], env=env2, capture_output=True,
                        text=True, timeout=120)
    assert r2.returncode != 0 and "--gpus 2 but WORLD_SIZE=3" in (r2.stdout + r2.stderr)
