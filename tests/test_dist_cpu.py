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


def _stub_synth_rows(model, prompts, videos, noise_shape, n_samples=1, x_T=None, cond=None, decode=True, **kw):
    """Like _stub_synth but sample-wise (what the real sampler is): row b depends on window b's frames, noise, context and video latent only."""
    B, _, T, h, w = noise_shape
    if "c_concat" not in cond:           # image_guided_synthesis encodes the window itself when the caller did not (pipeline.py: get_latent_z)
        cond = dict(cond, c_concat=[model.encode_first_stage(videos)])
    assert videos.shape[0] == B and x_T.shape[0] == B and cond["c_crossattn"][0].shape[0] == B and cond["c_concat"][0].shape[0] == B
    v = videos.mean(dim=(1, 3, 4)).reshape(B, 1, T, 1, 1)
    per = (x_T.mean(dim=(1, 2, 3, 4)) + cond["c_crossattn"][0].mean(dim=(1, 2)) + cond["c_concat"][0].mean(dim=(1, 2, 3, 4))).reshape(B, 1, 1, 1, 1)
    lat = torch.zeros((B, 16, T, h, w)) + v + per
    return (_stub_decoder(model, lat) if decode else lat)[:, None]


class _StubModelEnc(_StubModel):
    @staticmethod
    def encode_first_stage(videos):      # "posterior sample": the window's frames + noise from the CPU RNG run_clip seeds per window
        b, _, t, H, W = videos.shape
        return videos.mean(dim=1, keepdim=True).expand(b, 4, t, H, W)[..., ::8, ::8] + torch.randn((b, 4, t, H // 8, W // 8))


def _batch_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from geo4d_amd import dist as gd
    from geo4d_amd.pipeline import run_clip
    gd.init_from_env(backend="gloo")
    video = torch.arange(1 * 3 * 38 * 16 * 16, dtype=torch.float32).reshape(1, 3, 38, 16, 16) / 1e4
    outs = [run_clip(_StubModelEnc, video, torch.ones((1, 333, 8)), ddim_steps=2, synthesize=_stub_synth_rows, decode=d, decoder=_stub_decoder,
                     window_batch=wb)[1] for d in ("local", "sharded") for wb in (2, 3)]
    q.put((rank, [o.numpy() for o in outs]))
    dist.barrier()
    dist.destroy_process_group()


def test_run_clip_window_batch_is_invisible_in_the_result():
    """Round 6: run_clip(window_batch = k) denoises k of a rank's windows as one batch. Noise, VAE-encode sampling and conditioning stay per
    window, so with a sample-wise sampler the result is IDENTICAL to window_batch = 1 - on one rank (7 windows: groups 2 + 2 + 2 + 1 and
    3 + 3 + 1) and on two ranks in both decode modes (sharded: rounds of world x window_batch windows, rank r owning the r-th run)."""
    from geo4d_amd.pipeline import run_clip
    video = torch.arange(1 * 3 * 38 * 16 * 16, dtype=torch.float32).reshape(1, 3, 38, 16, 16) / 1e4
    run = lambda wb: run_clip(_StubModelEnc, video, torch.ones((1, 333, 8)), ddim_steps=2, synthesize=_stub_synth_rows, window_batch=wb)
    slices, ref = run(1)
    assert len(slices) == 7 and ref.shape == (7, 11, 16, 16, 16) and len({float(ref[i].mean()) for i in range(7)}) == 7
    for wb in (2, 3, 8):
        assert torch.equal(run(wb)[1], ref), wb
    assert torch.equal(run_clip(_StubModelEnc, video, torch.ones((1, 333, 8)), ddim_steps=2, ddim_eta=1.0, synthesize=_stub_synth_rows, window_batch=2)[1], ref)   # eta > 0: one window at a time
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_batch_worker, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in procs]
    res = [q.get(timeout=180) for _ in range(2)]
    [p.join(timeout=60) for p in procs]
    for rank, outs in res:
        for o in outs:
            assert torch.equal(torch.from_numpy(o), ref), f"rank {rank}"


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


# ---- alignment sharded over ranks (geo4d_amd/align_dist.py; north_star: "point-map alignment shard over the 8 GPUs") ------------------
def _align_problem(n=11, S=4, H=6, W=8, seed=3):
    """A small multi-window alignment problem in the oracle's format: windows of S consecutive images, stride 1 (overlapping)."""
    g = torch.Generator().manual_seed(seed)
    groups = [list(range(s, s + S)) for s in range(0, n - S + 1)]
    G, HW = len(groups), H * W
    data = dict(pred=torch.randn((G * S, HW, 3), generator=g) + torch.tensor([0.0, 0.0, 3.0]), conf=torch.rand((G * S, HW), generator=g) * 12,
                e_all=torch.tensor([i for grp in groups for i in grp]), H=H, W=W,
                invdepth=torch.rand((G * S, HW), generator=g), traj=torch.eye(4).repeat(G * S, 1, 1) + 0.01 * torch.randn((G * S, 4, 4), generator=g))
    P = dict(im_depthmaps=0.3 * torch.randn((n, HW), generator=g) + 1.0,
             im_poses=torch.cat([0.1 * torch.randn((n, 3), generator=g), torch.ones(n, 1), 0.2 * torch.randn((n, 3), generator=g)], 1),
             im_focals=torch.full((1, 1), 20.0 * 2.2), pw_poses=torch.cat([0.1 * torch.randn((G, 3), generator=g), torch.ones(G, 1), 0.1 * torch.randn((G, 4), generator=g)], 1),
             s_depth=torch.ones(G, 1) + 0.1 * torch.randn((G, 1), generator=g), t_depth=0.05 * torch.randn((G, 1), generator=g),
             traj_align_poses=torch.cat([0.05 * torch.randn((G, 3), generator=g), torch.ones(G, 1), 0.05 * torch.randn((G, 4), generator=g)], 1))
    state = dict(invalid_depth_groups=[1], valid_traj_groups=[0, 2, G - 1])
    return groups, data, P, state


def _adam_run(P, objective, niter=8, lr=0.02):
    """torch.optim.Adam arithmetic (betas 0.9 / 0.9) on a dict of tensors with externally supplied gradients."""
    m = {k: torch.zeros_like(v) for k, v in P.items()}
    v2 = {k: torch.zeros_like(v) for k, v in P.items()}
    losses = []
    for it in range(1, niter + 1):
        loss, grads = objective(P)
        losses.append(float(loss))
        for k in P:
            m[k].mul_(0.9).add_(grads[k], alpha=0.1)
            v2[k].mul_(0.9).addcmul_(grads[k], grads[k], value=0.1)
            P[k] = P[k] - lr / (1 - 0.9 ** it) * m[k] / ((v2[k] / (1 - 0.9 ** it)).sqrt() + 1e-8)
    return losses


def _oracle_objective(data, state, local_groups=None, pose_terms=True):
    from oracle import align as oalign

    def f(P):
        Q = {k: v.detach().clone().requires_grad_(True) for k, v in P.items()}
        loss = oalign.alignment_loss(Q, data, temporal_smoothing_weight=0.015, translation_weight=1.0, state=state, local_groups=local_groups,
                                     pose_terms=pose_terms)
        loss.backward()
        return loss.detach(), {k: (q.grad if q.grad is not None else torch.zeros_like(q)) for k, q in Q.items()}
    return f


def _align_worker(rank, world, port, q, exchange="halo", n=11):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from geo4d_amd import dist as gd
    from geo4d_amd.align_dist import AlignShard
    gd.init_from_env(backend="gloo")
    groups, data, P, state = _align_problem(n=n)
    shard = AlignShard(groups, P["im_depthmaps"].shape[0], exchange=exchange)
    local = _oracle_objective(data, state, local_groups=shard.local_groups, pose_terms=shard.primary)
    first = {}

    def objective(Pc):
        loss, grads = shard.reduce(*local(Pc))
        if not first:
            first.update(loss=float(loss), grads={k: v.clone() for k, v in grads.items()})
        return loss, grads
    losses = _adam_run(P, objective)
    P["im_depthmaps"] = shard.gather_depthmaps(P["im_depthmaps"])
    tab = shard.merge_rows(torch.arange(len(groups) * 3, dtype=torch.float32).reshape(-1, 3) + 1, shard.local_groups)
    # plain numpy through the queue (torch's shared-memory tensor hand-off dies with the worker)
    first["grads"] = {k: v.numpy().copy() for k, v in first["grads"].items()}
    q.put((rank, shard.local_groups, shard.shared, first, losses, {k: v.detach().numpy().copy() for k, v in P.items()}, tab.numpy().copy()))
    dist.barrier()
    dist.destroy_process_group()


import pytest


@pytest.mark.parametrize("world,exchange,n", [(2, "halo", 11), (2, "allreduce", 11), (3, "halo", 7)])
def test_alignment_sharded_over_two_ranks_equals_single_rank(world, exchange, n):
    """Window blocks on 2 (3) ranks; per iteration one small all-reduce of [loss | small gradients] and the depth gradients of the shared
    images by the neighbour halo exchange (round 6) or, as in rounds 3-5, inside that all-reduce: the first loss / gradients and 8 Adam
    iterations equal the un-sharded objective to fp32 round-off; depth maps of images that only one rank touches are final on that
    rank alone and assembled by gather_depthmaps. world 3 with n = 7 images (4 windows of 4 at stride 1, blocks of 2 / 1 / 1): images
    2 and 3 are touched by ALL THREE ranks - the partial rows are added in ascending rank order on each of them."""
    from geo4d_amd.align_dist import AlignShard, partition_windows
    assert [partition_windows(30, r, 8) for r in range(8)][0] == [0, 1, 2, 3] and sum(len(partition_windows(30, r, 8)) for r in range(8)) == 30
    s8 = AlignShard([list(range(4 * w, 4 * w + 16)) for w in range(29)], 128, rank=3, world=8)
    assert len(s8.shared) == 7 * 12 and s8.local_groups == partition_windows(29, 3, 8)      # 12 images per block boundary
    # halo: rank 3 of 8 talks to its two neighbours only, 12 images each; per iteration it sends 24 rows instead of the 84 of the all-reduce
    assert s8.peers == [2, 4] and [len(s8.peer_images[q]) for q in s8.peers] == [12, 12] and all(len(t) <= 2 for t in s8.touch)
    HW = 320 * 512
    assert s8.bytes_per_iteration(HW, 1000) == 4 * 1001 + 4 * HW * 24
    assert AlignShard(s8.groups, 128, rank=3, world=8, exchange="allreduce").bytes_per_iteration(HW, 1000) == 4 * (1001 + 84 * HW)
    if world == 3:
        s3 = [AlignShard([list(range(s, s + 4)) for s in range(4)], 7, rank=r, world=3) for r in range(3)]
        assert s3[0].touch[2] == (0, 1) and s3[0].touch[3] == (0, 1, 2) and s3[1].peers == [0, 2] and (0, 1, 2) in s3[2].sets
    groups, data, P, state = _align_problem(n=n)
    ref_obj = _oracle_objective(data, state)
    ref_loss, ref_grads = ref_obj(P)
    ref_P = {k: v.clone() for k, v in P.items()}
    ref_losses = _adam_run(ref_P, ref_obj)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_align_worker, args=(r, world, port, q, exchange, n)) for r in range(world)]
    [p.start() for p in procs]
    res = sorted((q.get(timeout=300) for _ in range(world)), key=lambda r: r[0])
    [p.join(timeout=60) for p in procs]
    res = [(r, lg, sh, dict(loss=f["loss"], grads={k: torch.from_numpy(v) for k, v in f["grads"].items()}), ls,
            {k: torch.from_numpy(v) for k, v in Pr.items()}, torch.from_numpy(tab)) for r, lg, sh, f, ls, Pr, tab in res]
    n_img = P["im_depthmaps"].shape[0]
    assert sum((r[1] for r in res), []) == list(range(len(groups))) and all(r[2] == res[0][2] for r in res) and 0 < len(res[0][2]) <= n_img
    shared = res[0][2]
    touch = AlignShard(groups, n_img, rank=0, world=world).touch
    for rank, local, _, first, losses, Pr, tab in res:
        assert abs(first["loss"] - float(ref_loss)) < 1e-5 * abs(float(ref_loss))
        for k, gref in ref_grads.items():
            if k == "im_depthmaps":      # complete on every TOUCHING rank for the shared images (all-reduce form: on every rank), complete on the owning rank for the others
                rows = [i for i in shared if exchange == "allreduce" or rank in touch[i]]
                assert rows and torch.allclose(first["grads"][k][rows], gref[rows], rtol=1e-4, atol=1e-7), k
            else:
                assert torch.allclose(first["grads"][k], gref, rtol=1e-4, atol=1e-7), k
        assert max(abs(a - b) for a, b in zip(losses, ref_losses)) < 1e-5 * abs(ref_losses[0])
        for k, v in ref_P.items():
            assert torch.allclose(Pr[k], v, rtol=2e-4, atol=2e-6), (rank, k, (Pr[k] - v).abs().max())
        assert torch.equal(tab, torch.arange(len(groups) * 3, dtype=torch.float32).reshape(-1, 3) + 1)
    for k in ref_P:                          # the replicated parameters are bit-identical across ranks (no broadcast needed)
        assert all(torch.equal(res[0][5][k], r[5][k]) for r in res[1:]), k


def test_alignment_shard_falls_back_when_windows_fewer_than_ranks():
    """5 windows on 8 GPUs would hand ranks 5-7 an empty block (the residual kernel refuses an empty slot list while
    ranks 0-4 wait in the per-iteration all-reduce -> hang). make_shard() - what post_optimization builds its shard with - returns None
    (= replicated optimisation, no collective) on EVERY rank in that case, and a shard with a non-empty block on every rank otherwise."""
    from geo4d_amd.align_dist import AlignShard, make_shard
    groups5 = [list(range(4 * w, 4 * w + 16)) for w in range(5)]
    assert all(make_shard(groups5, 32, rank=r, world=8) is None for r in range(8))
    assert [len(AlignShard(groups5, 32, rank=r, world=8).local_groups) for r in range(8)] == [1, 1, 1, 1, 1, 0, 0, 0]   # why
    groups8 = [list(range(4 * w, 4 * w + 16)) for w in range(8)]
    shards = [make_shard(groups8, 44, rank=r, world=8) for r in range(8)]
    assert all(s is not None and len(s.local_groups) == 1 for s in shards)
    assert make_shard(groups8, 44, rank=0, world=1) is None                     # a single rank never shards
    # the aligner itself refuses an empty block loudly instead of handing the kernel a null slot list (no GPU needed to see the message)
    import inspect
    from geo4d_amd import align
    assert "owns no window" in inspect.getsource(align.GroupAligner.__init__)


class _StubScene:
    """Stands in for GroupAligner in the world-2 test of bench.clip_mode: the real one needs the HIP kernels. Its optimisation does what
    the sharded alignment does per iteration - one all-reduce over the ranks."""
    def __init__(self, n_img, H, W):
        self.depth, self.poses = torch.ones((n_img, H * W)), torch.eye(4).repeat(n_img, 1, 1)

    def compute_global_alignment(self, niter, schedule, lr):
        for _ in range(niter):
            t = torch.ones(4)
            if dist.is_initialized() and dist.get_world_size() > 1:
                dist.all_reduce(t)
            self.depth += 0.0 * t[0]

    def get_depthmaps(self):
        return self.depth

    def get_im_poses_matrix(self):
        return self.poses


def _stub_align(slices, maps, traj, args, align=True):
    assert maps.shape[1:3] == (11, 16) and torch.isfinite(maps).all() and traj is not None and traj.shape[:2] == maps.shape[:1] + (16,)
    return _StubScene(1 + max(s.stop - 1 for s in slices), maps.shape[3], maps.shape[4])


def _bench_clip_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    import argparse
    import bench
    from geo4d_amd import dist as gd
    gd.init_from_env(backend="gloo")
    args = argparse.Namespace(clip_frames=22, height=32, width=64, ddim_steps=2, align_iters=5, clip_align_on_noise=False, dtype="bf16x3", no_graph=False)
    res = bench.clip_mode(args, _StubModel, None, torch.device("cpu"), rank, world,
                          clip_kw=dict(synthesize=_stub_synth, decoder=_stub_decoder, with_cameras=False), align_fn=_stub_align)
    # the vote main() takes around the leg: clip_mode returns its record on rank 0 and None elsewhere - every rank must reach the all-reduce
    # (round 6: a `"error" in None` TypeError on rank 1 skipped it and left rank 0 waiting: found on the 2-ranks-on-one-GPU rig)
    voted = bench.clip_leg_vote(res, torch.device("cpu"), world)
    assert voted is res
    peer_failed = bench.clip_leg_vote({"error": "boom"} if rank == 1 else res, torch.device("cpu"), world)
    assert isinstance(peer_failed, dict) and "error" in peer_failed
    q.put((rank, res))
    dist.barrier()
    dist.destroy_process_group()


def test_bench_clip_mode_world2_protocol():
    """bench.py --clip-frames under two gloo ranks on CPU: the SAME function the GPU run calls (window sharding, frame-sharded decode,
    gathers, the untimed synthetic-scene hand-over, barriers on both sides of every phase, MAX over ranks of the phase seconds), with a
    stub denoiser / decoder / aligner in place of the HIP kernels. Rank 0 returns the strong-scaling line, the other rank nothing."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_bench_clip_worker, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in procs]
    res = dict(q.get(timeout=240) for _ in range(2))
    [p.join(timeout=60) for p in procs]
    assert res[1] is None
    line = res[0]
    assert line["scaling"] == "strong" and line["n_gpus"] == 2 and line["config"]["windows"] == 3 and line["config"]["windows_per_rank_max"] == 2
    ph = line["phase_seconds"]
    assert set(ph) == {"denoise_decode_gather", "alignment_init", "alignment_5_iterations", "total"}
    assert abs(ph["total"] - (ph["denoise_decode_gather"] + ph["alignment_init"] + ph["alignment_5_iterations"])) < 0.5 * ph["total"] + 1e-3
    assert abs(line["value"] - 22 / ph["total"]) < 1e-9 and line["alignment_outputs_finite"] is True
    assert "frame-sharded VAE decode" in line["config"]["parallelism"]
