"""Multi-window global alignment on the HIP path (SURVEY.md §8(f) N1, geo4d_amd/align.py + csrc/align.hip) against
tests/golden/align_tiny.pt — loss, gradients and the 40-iteration result of the REFERENCE LightPointCloudGroupOptimizer /
global_alignment_loop (tests/golden/generate.py align) — and against the autograd oracle on a larger scene."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return ((a - b).norm() / b.norm()).item()


@pytest.fixture(scope="module")
def fix():
    return torch.load(os.path.join(G, "align_tiny.pt"), weights_only=False)


def _aligner(g, dev, **kw):
    from geo4d_amd.align import GroupAligner
    a = GroupAligner(g["groups"], g["pred"].to(dev), g["conf"].squeeze(-1).to(dev), shared_focal=True,
                     temporal_smoothing_weight=g["kw"]["temporal_smoothing_weight"], translation_weight=g["kw"]["translation_weight"], **kw)
    for k, v in g["init"].items():
        a.P[k] = v.clone().to(dev)
    return a


@pytest.mark.parametrize("chunk", [256, 1024])
def test_loss_and_gradients_vs_reference(fix, dev, chunk):
    a = _aligner(fix, dev, chunk_pixels=chunk)
    loss, grads = a.loss_and_grads()
    loss2, grads2 = a.loss_and_grads()
    print(f"[align] loss {float(loss):.6f} vs reference {fix['loss0']:.6f}; grad rel errors " +
          " ".join(f"{k}: {rel(grads[k], fix['grads'][k]):.2e}" for k in grads))
    assert abs(float(loss) - fix["loss0"]) < 1e-5 * abs(fix["loss0"])
    for k in grads:
        assert rel(grads[k], fix["grads"][k]) < 1e-4, k
        assert torch.equal(grads[k], grads2[k]), "alignment gradients must be run-to-run deterministic"
    assert torch.equal(loss, loss2)


@pytest.mark.parametrize("graph", [False, True])
def test_alignment_loop_vs_reference(fix, dev, graph):
    """The reference's own 40-iteration global_alignment_loop; eager launches and the captured-iteration replay."""
    a = _aligner(fix, dev)
    final, hist = a.compute_global_alignment(niter=fix["niter"], lr=fix["lr"], lr_min=fix["lr_min"], schedule=fix["schedule"], history=True,
                                             use_graph=graph)
    errs = {k: rel(a.P[k], fix["after"][k]) for k in fix["after"]}
    print(f"[align loop] {fix['niter']} iterations: loss {hist[0]:.4f} -> {hist[-1]:.4f} (reference last evaluated loss {fix['loss_final']:.4f}); "
          + " ".join(f"{k}: {v:.2e}" for k, v in errs.items()))
    assert abs(hist[-1] - fix["loss_final"]) < 2e-3 * abs(fix["loss_final"])
    assert max(errs.values()) < 2e-3, errs


def _scene28():
    """28 frames / 4 windows of 16 (stride 4) at 40x64: ground-truth cameras, per-window predictions in the window's own frame and scale."""
    gen = torch.Generator().manual_seed(4)
    n, S, stride, H, W, f = 28, 16, 4, 40, 64, 55.0
    ys, xs = torch.meshgrid(torch.arange(H), torch.arange(W), indexing="ij")
    grid, pp = torch.stack([xs, ys], -1).float(), torch.tensor([W / 2, H / 2])
    c2w, pts = [], []
    for i in range(n):
        depth = 3.0 + 0.6 * torch.sin(xs / 9.0 + 0.2 * i) + 0.4 * torch.cos(ys / 7.0)
        cam = torch.cat([depth[..., None] * (grid - pp) / f, depth[..., None]], -1)
        a_ = torch.tensor(0.02 * i)
        R = torch.tensor([[torch.cos(a_), 0, torch.sin(a_)], [0, 1, 0], [-torch.sin(a_), 0, torch.cos(a_)]])
        M = torch.eye(4); M[:3, :3] = R; M[:3, 3] = torch.tensor([0.05 * i, 0.0, 0.01 * i])
        c2w.append(M); pts.append(cam @ R.T + M[:3, 3])
    groups = [list(range(s0, s0 + S)) for s0 in range(0, n - S + 1, stride)]
    preds, confs, trajs = [], [], []
    for gi, grp in enumerate(groups):
        w2c = torch.inverse(c2w[grp[0]])
        sc = 0.8 + 0.1 * gi
        preds.append(torch.stack([(pts[i] @ w2c[:3, :3].T + w2c[:3, 3]) * sc for i in grp]) + 0.003 * torch.randn((S, H, W, 3), generator=gen))
        confs.append(1.0 + 3.0 * torch.rand((S, H, W), generator=gen))
        tr = torch.stack([w2c @ c2w[i] for i in grp])
        tr[:, :3, 3] *= sc
        trajs.append(tr)
    rays = torch.cat([(grid - pp) / f, torch.ones(H, W, 1)], -1)
    return dict(groups=groups, preds=preds, confs=confs, trajs=trajs, c2w=c2w, f=f, n=n, S=S, H=H, W=W, rays=rays)


def test_init_from_group_and_convergence_at_window_size(dev):
    """A 28-frame / 4-window scene at 40x64 pixels per frame (the latent-resolution stand-in for 16-frame windows with stride 4):
    registration-chained initialisation from the windows' own cameras, then 60 iterations; checked against the autograd oracle
    evaluated at the same parameters, and for actually aligning the clip."""
    from geo4d_amd.align import GroupAligner
    from oracle import align as oalign
    sc_ = _scene28()
    groups, preds, confs, trajs, f, n, S, H, W = (sc_[k] for k in ("groups", "preds", "confs", "trajs", "f", "n", "S", "H", "W"))
    a = GroupAligner(groups, torch.stack(preds).to(dev), torch.stack(confs).to(dev), temporal_smoothing_weight=0.015, translation_weight=1.0)
    a.init_from_group(torch.stack(trajs).to(dev))
    loss0, grads = a.loss_and_grads()
    # the oracle in fp64: the engine's parameter chain rule runs in fp64 on fp32 gradient sums, an fp32 autograd reference would carry
    # more rounding noise than the thing under test at this near-optimal point (gradients are small differences of large sums)
    data = dict(pred=torch.stack(preds).reshape(-1, H * W, 3).double(), conf=torch.stack(confs).reshape(-1, H * W).double(), H=H, W=W,
                e_all=torch.tensor([i for grp in groups for i in grp]))
    P = {k: v.detach().cpu().double().clone().requires_grad_(True) for k, v in a.P.items()}
    ref = oalign.alignment_loss(P, data, temporal_smoothing_weight=0.015, translation_weight=1.0)
    ref.backward()
    errs = {k: rel(grads[k], P[k].grad) for k in grads}
    print(f"[align 28 frames] init loss {float(loss0):.5f} (oracle {float(ref.detach()):.5f}), focal {float(a.get_focals()[0]):.2f} (true {f}); grads " +
          " ".join(f"{k}: {v:.2e}" for k, v in errs.items()))
    # the focal gradient is ONE scalar summed over 70 k signed per-pixel terms in fp32 (here chunk by chunk, in torch as a tree)
    assert abs(float(loss0) - float(ref.detach())) < 1e-4 * float(ref.detach()) and max(errs.values()) < 3e-3 and errs["im_depthmaps"] < 1e-3, errs
    assert float(loss0) < 0.05 and abs(float(a.get_focals()[0]) - f) < 0.1 * f          # the chained initialisation is already close
    final, hist = a.compute_global_alignment(niter=30, lr=0.003, schedule="linear", history=True)
    print(f"[align 28 frames] from the chained init (already at the noise floor): {hist[0]:.5f} -> {hist[-1]:.5f}")
    assert hist[-1] < 1.25 * hist[0] and torch.isfinite(a.get_depthmaps()).all()      # Adam jitters around an optimum it starts at
    # knock the solution off (window sim(3)s, camera translations, depth scale) and let the loop pull it back
    gen2 = torch.Generator().manual_seed(5)
    a.P["pw_poses"][:, 4:8] += 0.05 * torch.randn((len(groups), 4), generator=gen2).to(dev)
    a.P["im_poses"][:, 4:7] += 0.05 * torch.randn((n, 3), generator=gen2).to(dev)
    a.P["im_depthmaps"] += 0.1
    final, hist = a.compute_global_alignment(niter=150, lr=0.01, schedule="linear", history=True)
    print(f"[align 28 frames] after a perturbation, 150 iterations: {hist[0]:.5f} -> {hist[-1]:.5f}")
    assert hist[0] > 3 * float(loss0) and hist[-1] < 0.35 * hist[0]


# ---- the two late terms (inverse depth, trajectory) ---------------------------------------------------------------------------------
def _late_aligner(g, dev, init_key="init", **kw):
    from geo4d_amd.align import GroupAligner
    d = g["depth_traj"]
    a = GroupAligner(g["groups"], g["pred"].to(dev), g["conf"].squeeze(-1).to(dev), shared_focal=True,
                     temporal_smoothing_weight=g["kw"]["temporal_smoothing_weight"], translation_weight=g["kw"]["translation_weight"],
                     inverse_depth=d["invdepth"].to(dev), traj=d["traj"].to(dev), depth_traj_start_iter=d["start"], **kw)
    src = d[init_key] if isinstance(init_key, str) else init_key
    for k, v in src.items():
        a.P[k] = v.clone().to(dev)
    return a


@pytest.mark.parametrize("n", [1, 2, 7, 768, 4099, 163840])
def test_lower_median_is_torch_median_bit_for_bit(dev, n):
    """Radix select == torch.median (lower median), incl. negatives, duplicates, signed zeros, denormals."""
    import ctypes as C
    from geo4d_amd import _lib, ops
    lib = _lib.load()
    g = torch.Generator().manual_seed(n)
    rows = 5
    x = torch.randn((rows, n), generator=g)
    x[1] = x[1].abs() + 0.05
    x[2] = torch.randint(-3, 4, (n,), generator=g).float()          # many duplicates, +-0
    x[3] = x[3] * 1e-41                                              # denormals
    x[4, : n // 2] = -0.0
    xd = x.to(dev).contiguous()
    out = torch.empty(rows, device=dev)
    ws = torch.empty(rows * 260 * 4, dtype=torch.uint8, device=dev)
    _lib.check(lib.geo4d_lower_median(xd.data_ptr(), rows, n, out.data_ptr(), ws.data_ptr(), ws.numel(), ops._stream()), "geo4d_lower_median")
    want = torch.stack([torch.median(r) for r in x])
    assert torch.equal(out.cpu(), want), (out.cpu(), want)          # value equality (+0 == -0: which zero torch returns is unspecified)


def test_lad_fit_vs_oracle(dev):
    """Per-window least-absolute-deviation fit (the reference's absolute_value_scaling2) for three windows at once: same start
    (median ratio, bit-exact), same Adam; compared through what the fit is FOR — the objective it reaches and the delta < 1.25 score
    — and loosely on (s, t) themselves: a 5000-step Adam at lr 1e-2 on an L1 objective ends within a few steps of jitter of the
    optimum, and that jitter differs with the summation order. Two runs of the HIP fit are bit-identical."""
    import ctypes as C
    from geo4d_amd import _lib, ops
    from oracle import align as oalign
    lib = _lib.load()
    gen = torch.Generator().manual_seed(11)
    G_, n = 3, 4 * 40 * 64
    z = 1.0 + 3.0 * torch.rand((G_, n), generator=gen)
    target = 1.0 / z
    s_true, t_true = torch.tensor([[2.0], [0.7], [1.3]]), torch.tensor([[0.05], [-0.02], [0.0]])
    q = (target - t_true) / s_true + 0.01 * torch.randn((G_, n), generator=gen)
    q[:, :2000] = 0.01 * torch.rand((G_, 2000), generator=gen)                               # 'sky': outliers for the fit
    conf = 0.2 + torch.rand((G_, n), generator=gen)
    qd, td, cd = q.to(dev).contiguous(), target.to(dev).contiguous(), conf.to(dev).contiguous()
    ws = torch.empty(lib.geo4d_lad_workspace(G_, n), dtype=torch.uint8, device=dev)

    def fit(lr, iters, active=None):
        st, info = torch.empty(G_, 2, device=dev), torch.empty(G_, 2, device=dev)
        act = None if active is None else torch.tensor(active, dtype=torch.uint8, device=dev)
        _lib.check(lib.geo4d_lad_fit(qd.data_ptr(), td.data_ptr(), G_, n, None if act is None else act.data_ptr(), lr, iters, 1e-6, st.data_ptr(),
                                     info.data_ptr(), ws.data_ptr(), ws.numel(), ops._stream()), "geo4d_lad_fit")
        counts = torch.zeros(G_, 2, dtype=torch.int32, device=dev)
        _lib.check(lib.geo4d_lad_delta(qd.data_ptr(), td.data_ptr(), cd.data_ptr(), st.data_ptr(), G_, n, 0.5, 10.0, 0.05, counts.data_ptr(), ops._stream()),
                   "geo4d_lad_delta")
        return st.cpu(), info.cpu(), counts.cpu()

    for lr, iters in ((1e-2, 5000), (1e-3, 300)):
        st, info, counts = fit(lr, iters)
        st2, _, counts2 = fit(lr, iters)
        assert torch.equal(st, st2) and torch.equal(counts, counts2), "the fit must be run-to-run deterministic"
        for g in range(G_):
            cm = (conf[g] > 0.5) & (q[g] > 0.05)
            s_o, t_o, d_o = oalign.fit_window_depth(q[g], target[g], cm, lr, iters)
            obj = lambda s, t: float((s * q[g] + t - target[g]).abs().sum())
            d_h = float(counts[g, 0]) / float(counts[g, 1])
            print(f"[lad] lr {lr} window {g}: hip (s, t) = ({float(st[g, 0]):.5f}, {float(st[g, 1]):.5f}) oracle ({s_o:.5f}, {t_o:.5f}); objective "
                  f"{obj(st[g, 0], st[g, 1]):.3f} vs {obj(s_o, t_o):.3f}; delta {d_h:.4f} vs {d_o:.4f}; steps {int(info[g, 0])}")
            assert int(counts[g, 1]) == int(cm.sum())
            assert abs(obj(st[g, 0], st[g, 1]) - obj(s_o, t_o)) < 2e-3 * obj(s_o, t_o)
            assert abs(d_h - d_o) < 5e-3
            # far from the optimum (300 small steps) the two trajectories coincide; at the optimum they differ by Adam's jitter, and the
            # reference's stop test (the loss repeating exactly in fp32) may fire at different steps
            tol_st = 2e-4 if iters == 300 else 4 * lr + 1e-4
            assert abs(float(st[g, 0]) - s_o) < tol_st and abs(float(st[g, 1]) - t_o) < tol_st
            assert 0 < int(info[g, 0]) <= iters and (iters != 300 or int(info[g, 0]) == iters)
    # inactive windows keep their start values (median ratio, 0)
    st, info, _ = fit(1e-2, 50, active=[1, 0, 1])
    assert float(st[1, 1]) == 0.0 and int(info[1, 0]) == 0
    assert float(st[1, 0]) == float(torch.median(target[1]) / torch.median(q[1]))


@pytest.mark.parametrize("chunk", [256, 1024])
def test_late_terms_loss_and_gradients_vs_reference(fix, dev, chunk):
    """Full objective (point maps + inverse depth + trajectory + smoothing) and ALL gradients at the reference loop's end point."""
    d = fix["depth_traj"]
    a = _late_aligner(fix, dev, "after", chunk_pixels=chunk)
    a.set_state(d["invalid_depth_groups"], d["valid_traj_groups"])
    loss, grads = a.loss_and_grads()
    loss2, grads2 = a.loss_and_grads()
    errs = {k: rel(grads[k], d["grads_at_after"][k]) for k in grads}
    print(f"[align late] loss {float(loss):.6f} vs reference {d['loss_at_after']:.6f}; " + " ".join(f"{k}: {v:.2e}" for k, v in errs.items()))
    assert abs(float(loss) - d["loss_at_after"]) < 1e-5 * abs(d["loss_at_after"])
    assert set(grads) == set(d["grads_at_after"])
    for k, e in errs.items():
        assert e < (3e-3 if k == "im_focals" else 2e-4), (k, e)
        assert torch.equal(grads[k], grads2[k])
    # a dropped window contributes nothing to the depth term and gets no (s, t) gradient
    a.set_state([1], d["valid_traj_groups"])
    _, g3 = a.loss_and_grads()
    assert float(g3["s_depth"][1].abs().sum()) == 0 and float(g3["t_depth"][1].abs().sum()) == 0 and float(g3["s_depth"][0].abs().sum()) > 0


def test_late_terms_start_up_vs_reference(fix, dev):
    """_set_st_depth and _set_traj from the state the reference's own routines saw at iteration `start` of its loop."""
    d = fix["depth_traj"]
    su = d["startup"]
    a = _late_aligner(fix, dev, su["at_start"])
    state = a.start_depth_traj()
    s, t = a.P["s_depth"].cpu(), a.P["t_depth"].cpu()
    print("[align start-up] s", s.flatten().tolist(), "ref", su["st"]["s"].flatten().tolist(), "t", t.flatten().tolist(), "ref", su["st"]["t"].flatten().tolist(),
          "delta", a.depth_delta, state)
    assert state["invalid_depth_groups"] == su["st"]["invalid"] and state["valid_traj_groups"] == su["traj"]["valid"]
    # lr 1e-2 windows: within Adam's end-of-run jitter; the retried window (lr 1e-3) much closer
    assert (s - su["st"]["s"]).abs().max() < 4e-2 and (t - su["st"]["t"]).abs().max() < 4e-2
    tq, rq = a.P["traj_align_poses"].cpu(), su["traj"]["poses"]
    sign = torch.sign((tq[:, :4] * rq[:, :4]).sum(1, keepdim=True))
    assert (tq[:, :4] * sign - rq[:, :4]).abs().max() < 1e-5 and (tq[:, 4:] - rq[:, 4:]).abs().max() < 1e-5


@pytest.mark.parametrize("graph", [False, True])
def test_loop_with_late_terms_vs_reference(fix, dev, graph):
    """40 iterations with both terms switched on at iteration 10, against the reference's own loop. Single entries jump by one Adam
    step when an L1 residual changes sign, so the end point is compared robustly (as the oracle is: tests/test_oracle_golden.py)."""
    d = fix["depth_traj"]
    a = _late_aligner(fix, dev, "init")
    final, hist = a.compute_global_alignment(niter=d["niter"], lr=fix["lr"], lr_min=fix["lr_min"], schedule=fix["schedule"], history=True, use_graph=graph)
    print(f"[align late loop] loss {hist[0]:.4f} -> {hist[-1]:.4f} (reference {d['loss_final']:.4f}); state {a.state}")
    assert a.state["invalid_depth_groups"] == d["invalid_depth_groups"] and a.state["valid_traj_groups"] == d["valid_traj_groups"]
    assert abs(hist[-1] - d["loss_final"]) < 1e-2 * d["loss_final"]
    for k, ref in d["after"].items():
        dev_ = (a.P[k].cpu() - ref).abs()
        print("   ", k, "median", float(dev_.median()), "max", float(dev_.max()))
        assert dev_.median() < 3e-3 and dev_.max() < 8e-2 and (dev_.numel() < 64 or (dev_ < 5e-3).float().mean() > 0.95), k


@pytest.mark.parametrize("shared", [True, False])
def test_focal_initialisation_from_ray_maps(fix, dev, shared):
    """init_from_group(raymaps=...): every image's focal = the reference's Weiszfeld estimate on its first ray map, the shared
    focal their mean (init_im_poses.py:133-136, 627-628) - on the device, against the fixture values of dust3r.post_process."""
    from geo4d_amd.align import GroupAligner, estimate_focal_weiszfeld
    fo = fix["focal"]
    assert torch.allclose(estimate_focal_weiszfeld(fo["rays"].to(dev), fo["pp"].to(dev)).cpu(), fo["weiszfeld"], rtol=2e-5)
    groups = [[0, 1], [1, 2]]                                           # three images, two windows of two; ray maps of 24 x 32
    H, W = fo["rays"].shape[1:3]
    gen = torch.Generator().manual_seed(9)
    pred = torch.randn((2, 2, H, W, 3), generator=gen) * 0.1 + torch.tensor([0.0, 0.0, 2.0])
    conf = torch.ones(2, 2, H, W)
    rm = torch.stack([fo["rays"][[0, 1]], fo["rays"][[1, 2]]])          # window 1 sees image 1 again: its first occurrence (window 0) counts
    a = GroupAligner(groups, pred.to(dev), conf.to(dev), shared_focal=shared)
    a.init_from_group(torch.eye(4).repeat(2, 2, 1, 1).to(dev), raymaps=rm.to(dev))
    want = fo["weiszfeld"].mean().reshape(1) if shared else fo["weiszfeld"]
    got = a.get_focals().cpu().flatten()[: want.numel()]
    print("[align focal init]", got.tolist(), want.tolist())
    assert torch.allclose(got, want, rtol=1e-4)


def test_post_optimization_consumes_the_gathered_clip(dev):
    """post_optimization(slices, maps, traj, postprocess args): the consumer of run_clip's output, wired like the script's window
    loop + post_optimization (test_geo4d.py:446-509): decoded maps -> pts3d / inverse confidence / inverse depth / ray maps -> aligner
    with both late terms -> init -> loop. Checked for wiring (shapes, terms started, finite decreasing loss) on a tiny synthetic clip."""
    from geo4d_amd.align import post_optimization
    from geo4d_amd.pipeline import window_slices
    gen = torch.Generator().manual_seed(3)
    T, H, W, n = 4, 16, 24, 7
    slices = window_slices(n, 1, T)
    G_ = len(slices)
    ys, xs = torch.meshgrid(torch.arange(H), torch.arange(W), indexing="ij")
    f = 30.0
    ray = torch.stack([(xs - W / 2) / f, (ys - H / 2) / f, torch.ones(H, W)], 0)
    ray = ray / ray.norm(dim=0, keepdim=True)
    maps = torch.zeros(G_, 11, T, H, W)
    depth = 0.5 + 0.2 * torch.rand((G_, T, H, W), generator=gen)
    maps[:, 0:3] = (ray[None, :, None] * depth[:, None] * 1.2).clamp(-0.9, 0.9)     # point map (normalised bbox coordinates)
    maps[:, 2] = maps[:, 2] * 2 - 1
    maps[:, 3] = 1.0                                                               # confidence logit
    maps[:, 4:7] = ray[None, :, None]
    maps[:, 10] = (1.0 / (1.0 + depth)) * 2 - 1                                     # inverse depth in [-1, 1]
    traj = torch.eye(4).repeat(G_, T, 1, 1)
    traj[:, :, 0, 3] = torch.arange(T).float() * 0.01
    scene = post_optimization(slices, maps.to(dev), traj.to(dev), dict(n_iter=30, pose_schedule="linear", temporal_smoothing_weight=0.015,
                                                                      translation_weight=1.0), depth_traj_start_iter=10)
    assert scene.state is not None and scene.get_depthmaps().shape == (n, H, W) and scene.get_im_poses_matrix().shape == (n, 4, 4)
    assert torch.isfinite(scene.get_depthmaps()).all() and abs(float(scene.get_focals()[0]) - f) < 0.35 * f
    loss, _ = scene.loss_and_grads()
    assert torch.isfinite(loss)
    print("[post_optimization] focal", float(scene.get_focals()[0]), "state", scene.state, "loss", float(loss))
    # a second call continues with the late terms on (their Adam steps count from this call's first iteration)
    final2, hist2 = scene.compute_global_alignment(niter=5, lr=0.003, schedule="linear", history=True)
    assert all(map(lambda x: x == x and x < 10, hist2)) and len(hist2) == 5
    # known intrinsics: preset and frozen (scene.preset_focal(..., requires_grad=False), test_geo4d.py:44-45)
    K = torch.eye(3).repeat(n, 1, 1)
    K[:, 0, 0], K[:, 1, 1] = 41.0, 43.0
    fixed = post_optimization(slices, maps.to(dev), traj.to(dev), dict(n_iter=12, pose_schedule="linear"), intrinsics=K, depth_traj_start_iter=6)
    assert abs(float(fixed.get_focals()[0]) - 42.0) < 1e-3, float(fixed.get_focals()[0])


def test_sharded_residual_kernel_sums_to_the_full_objective(fix, dev):
    """geo4d_amd/align_dist.py on the device: two emulated ranks (window blocks, restricted slot lists, pose-only terms on rank 0, no
    process group: the all-reduce is replaced by adding the two ranks' packed buffers) reproduce the un-sharded loss and EVERY gradient
    of the full objective (point maps + inverse depth + trajectory + smoothing) at the reference loop's end point; the depth-map
    gradient of an image that one rank owns alone is complete on that rank; the sharded start-up fits only its own windows."""
    from geo4d_amd.align_dist import AlignShard
    d = fix["depth_traj"]
    full = _late_aligner(fix, dev, "after")
    full.set_state(d["invalid_depth_groups"], d["valid_traj_groups"])
    loss, grads = full.loss_and_grads()
    n = full.n
    parts = []
    for r in range(2):
        sh = AlignShard(fix["groups"], n, rank=r, world=2, exchange="allreduce")     # (the packed all-reduce form: one buffer per rank to add by hand)
        sh._all_reduce = lambda t: t                         # no process group here: ranks are summed by hand below
        a = _late_aligner(fix, dev, "after", shard=sh)
        a.set_state(d["invalid_depth_groups"], d["valid_traj_groups"])
        l, g = a.loss_and_grads()
        parts.append((sh, float(l), {k: v.clone() for k, v in g.items()}))
    sh0, sh1 = parts[0][0], parts[1][0]
    assert sh0.local_groups + sh1.local_groups == list(range(len(fix["groups"]))) and sh0.shared == sh1.shared and len(sh0.shared) > 0
    assert abs(parts[0][1] + parts[1][1] - float(loss)) < 1e-5 * abs(float(loss))
    for k in grads:
        tot = parts[0][2][k] + parts[1][2][k]
        assert rel(tot, grads[k]) < 2e-5, (k, rel(tot, grads[k]))
    only0 = [i for i in range(n) if sh0.owner[i] == 0 and i not in sh0.shared]
    only1 = [i for i in range(n) if sh0.owner[i] == 1 and i not in sh0.shared]
    assert only0 and only1
    assert rel(parts[0][2]["im_depthmaps"][only0], grads["im_depthmaps"][only0]) < 2e-5 and float(parts[1][2]["im_depthmaps"][only0].abs().max()) == 0
    assert rel(parts[1][2]["im_depthmaps"][only1], grads["im_depthmaps"][only1]) < 2e-5
    # start-up: each emulated rank fits ITS windows; merged (s, t) table == the un-sharded fit
    su = d["startup"]
    ref = _late_aligner(fix, dev, su["at_start"])
    ref.start_depth_traj()
    tabs = []
    for r in range(2):
        sh = AlignShard(fix["groups"], n, rank=r, world=2, exchange="allreduce")
        sh._all_reduce = lambda t: t
        sh.merge_rows = (lambda table, rows, _sh=sh: table * torch.zeros(table.shape[0], 1, device=table.device).index_fill_(0, torch.tensor(list(rows), device=table.device), 1.0))
        a = _late_aligner(fix, dev, su["at_start"], shard=sh)
        a._set_st_depth()
        tabs.append(torch.cat([a.P["s_depth"], a.P["t_depth"]], 1).clone())
    merged = tabs[0] + tabs[1]
    assert (merged - torch.cat([ref.P["s_depth"], ref.P["t_depth"]], 1)).abs().max() < 1e-6


def test_pnp_initialisation_follows_the_reference_recipe(dev):
    """init_from_group(pose_init="pnp"): align_group's recipe (init_im_poses.py:82-214) with the seeded RANSAC-PnP of geo4d_amd/pnp.py -
    chained clouds, one PnP per image at the ray-map focal -/+ 3 % of the image size, shared focal = mean. On the synthetic 28-frame
    scene the cameras come out at the ground truth (up to the first window's frame and scale) and the start is as good as the
    Pluecker-camera start."""
    from geo4d_amd.align import GroupAligner
    sc_ = _scene28()
    groups, preds, confs, trajs, c2w, f, n, H, W = (sc_[k] for k in ("groups", "preds", "confs", "trajs", "c2w", "f", "n", "H", "W"))
    raymaps = sc_["rays"].expand(len(groups), sc_["S"], H, W, 3)
    a = GroupAligner(groups, torch.stack(preds).to(dev), torch.stack(confs).to(dev), temporal_smoothing_weight=0.015, translation_weight=1.0)
    a.init_from_group(None, raymaps=raymaps.to(dev), pose_init="pnp", niter_PnP=50)
    b = GroupAligner(groups, torch.stack(preds).to(dev), torch.stack(confs).to(dev), temporal_smoothing_weight=0.015, translation_weight=1.0)
    b.init_from_group(torch.stack(trajs).to(dev), raymaps=raymaps.to(dev))
    la, lb = float(a.loss_and_grads()[0]), float(b.loss_and_grads()[0])
    M = a.get_im_poses_matrix().cpu()
    w2c0 = torch.inverse(c2w[groups[0][0]])
    scale = float(M[5, :3, 3].norm() / (w2c0 @ c2w[5])[:3, 3].norm())                 # the world is window 0's frame at the normalised scale
    rot_err = max(float((M[i, :3, :3] - (w2c0 @ c2w[i])[:3, :3]).abs().max()) for i in range(n))
    tr_err = max(float((M[i, :3, 3] / scale - (w2c0 @ c2w[i])[:3, 3]).abs().max()) for i in range(n))
    print(f"[pnp init] loss {la:.5f} (trajectory-based start {lb:.5f}); focal {float(a.get_focals()[0]):.2f} (true {f}); max rotation entry error {rot_err:.2e}, translation {tr_err:.2e}")
    assert la < 0.05 and la < 2.0 * lb + 1e-3 and abs(float(a.get_focals()[0]) - f) < 0.05 * f
    assert rot_err < 2e-2 and tr_err < 5e-2
    assert a.init_focals.shape == (n,) and float((a.init_focals - f).abs().max()) < 0.06 * f


def test_pnp_initialisation_vs_the_reference_init_from_group(dev):
    """tests/golden/pnp_init_tiny.pt = the REFERENCE's init_from_group (align_group -> fast_pnp -> init_from_pts3d_group,
    init_im_poses.py:61-214, 569-635) on a 10-image / 4-window scene with ray maps, with cv2.solvePnPRansac stubbed by the same seeded
    restatement this engine calls (generate.py pnp_init). Everything around the solver must agree: chaining without overwrite, focal
    bookkeeping, pairwise poses, scale normalisation, depth maps and camera poses."""
    from geo4d_amd.align import GroupAligner
    g = torch.load(os.path.join(G, "pnp_init_tiny.pt"), weights_only=False)
    Gn, S, H, W, _ = g["pred"].shape
    a = GroupAligner(g["groups"], g["pred"].to(dev), g["conf"].squeeze(-1).to(dev), shared_focal=True, temporal_smoothing_weight=0.015, translation_weight=1.0)
    a.init_from_group(None, raymaps=g["rays"].expand(Gn, S, H, W, 3).to(dev), pose_init="pnp", niter_PnP=g["niter_PnP"])
    ref = g["after_init"]
    got = {k: a.P[k].detach().cpu() for k in ref}
    for k in ("im_poses", "pw_poses"):                      # quaternions are defined up to sign
        sign = torch.sign((got[k][:, :4] * ref[k][:, :4]).sum(1, keepdim=True))
        got[k] = torch.cat([got[k][:, :4] * sign, got[k][:, 4:]], 1)
    errs = {k: float((got[k].reshape(ref[k].shape) - ref[k]).abs().max()) for k in ref}
    loss = float(a.loss_and_grads()[0])
    print(f"[pnp init vs reference] max abs parameter differences {errs}; loss {loss:.5f} (reference {g['loss']:.5f})")
    assert errs["im_focals"] < 1e-3 and errs["pw_poses"] < 2e-3 and errs["im_poses"] < 5e-3 and errs["im_depthmaps"] < 5e-3, errs
    assert abs(loss - g["loss"]) < 0.05 * g["loss"] + 1e-4


@pytest.mark.parametrize("shared", [True, False])
def test_fused_parameter_chain_rule_equals_autograd(fix, dev, shared):
    """csrc/align_small.hip (parameters -> cams / window transforms; gradient sums -> parameter gradients incl. the smoothing term: two
    launches) against round 2's autograd chain over tiny tensors (`_loss_and_grads_torch`), on the full objective with both late terms
    on, shared and per-image focals, and at a point where one translation is exactly 0 (sign'(0) = 0 like torch)."""
    d = fix["depth_traj"]
    a = _late_aligner(fix, dev, "after")
    if not shared:
        a.shared_focal = False
        a.P["im_focals"] = a.P["im_focals"].expand(a.n, 1).clone() + 0.3 * torch.arange(a.n, device=dev).float().reshape(-1, 1)
    a.P["im_poses"][2, 5] = 0.0
    a.P["pw_poses"][1, 4] = 0.0
    a.set_state(d["invalid_depth_groups"], d["valid_traj_groups"])
    loss_t, g_t = a._loss_and_grads_torch()
    g_t = {k: v.clone() for k, v in g_t.items()}
    loss_h, g_h = a.loss_and_grads()
    errs = {k: rel(g_h[k], g_t[k]) for k in g_t}
    print(f"[fused chain rule, shared focal {shared}] loss {float(loss_h):.6f} vs autograd {float(loss_t):.6f}; " + " ".join(f"{k}: {v:.1e}" for k, v in errs.items()))
    assert set(g_h) == set(g_t) and abs(float(loss_h) - float(loss_t)) < 2e-6 * abs(float(loss_t))
    for k, e in errs.items():
        # (the residual kernel sees cams rounded from fp64 here and computed in fp32 by torch there: its outputs differ in the last bits)
        assert e < (2e-4 if k in ("im_focals", "im_depthmaps") else 2e-5), (k, e)
    assert float(g_h["im_poses"][2, 5]) == 0.0 and float(g_h["pw_poses"][1, 4]) == 0.0
    loss2, g2 = a.loss_and_grads()
    assert torch.equal(loss_h, loss2) and all(torch.equal(g_h[k], g2[k]) for k in g_h)


def _oracle_view(scene, requires_grad=True):
    """(P, data, kw) of oracle.align.alignment_loss for a GroupAligner's CURRENT parameters and inputs, on the CPU."""
    P = {k: v.detach().cpu().clone().requires_grad_(requires_grad) for k, v in scene.P.items()}
    G_, S = scene.G, scene.S
    data = dict(pred=scene.pred.cpu(), conf=scene.conf.cpu(), H=scene.H, W=scene.W, e_all=scene.e_all.cpu())
    if scene.invdepth is not None:
        data["invdepth"] = scene.invdepth.cpu()
    if scene.traj is not None:
        data["traj"] = scene.traj.reshape(G_ * S, 4, 4).cpu()
    kw = dict(temporal_smoothing_weight=scene.tsw, translation_weight=scene.tw, base_scale=scene.base_scale, conf_clamp=scene.conf_clamp)
    return P, data, kw


def test_clip_alignment_full_size_vs_oracle(dev):
    """The clip chain at BASELINE clip size against the oracle (the clip mode only asserted finiteness): a 64-frame clip =
    14 sliding windows of 16 at 320x512 (36.7 M residuals, BASELINE.json configs[2]'s window structure) of the consistent synthetic scene
    bench.py's clip mode aligns, through post_optimization exactly as the bench calls it. Checked at full size:
      * the engine's objective and EVERY gradient at a seeded perturbation of the initialised state (the state itself sits on the kinks of
        the L1 objective) == oracle/align.py's alignment_loss + autograd on the CPU;
      * after 40 Adam iterations with both late terms switched on at iteration 20: the oracle objective evaluated at the ENGINE's
        parameters and window state == the engine's own loss there, and that loss fell;
      * the recovered camera track is the scene's (a camera sliding 0.01 per frame along x): consecutive centres equidistant and collinear."""
    import bench
    from conftest import cpu_threads
    from geo4d_amd.align import post_optimization
    from geo4d_amd.pipeline import window_slices
    from oracle import align as oalign
    cpu_threads()
    N, T, H, W = 64, 16, 320, 512
    slices = window_slices(N, 4, T)
    assert len(slices) == 14
    maps, traj = bench.synthetic_scene_maps(slices, T, H, W, dev)
    scene = post_optimization(slices, maps, traj, dict(n_iter=40, pose_schedule="linear", temporal_smoothing_weight=0.015, translation_weight=1.0),
                              align=False, depth_traj_start_iter=20)
    # the scene is consistent, so the initialised state sits ON the kinks of the L1 objective (residuals ~1e-7: the gradient of |r| there is
    # round-off noise in any implementation): compare at a seeded perturbation of every parameter instead
    gen = torch.Generator().manual_seed(17)
    for k, amp in (("im_depthmaps", 2e-2), ("im_poses", 5e-3), ("pw_poses", 5e-3), ("im_focals", 5e-2)):
        scene.P[k] += amp * torch.randn(scene.P[k].shape, generator=gen).to(dev)
    loss0, grads0 = scene.loss_and_grads()
    P, data, kw = _oracle_view(scene)
    lo = oalign.alignment_loss(P, data, state=None, **kw)
    lo.backward()
    errs = {k: rel(grads0[k].cpu(), P[k].grad) for k in grads0}
    print(f"[clip 64x320x512, init] loss engine {float(loss0):.6f} vs oracle {float(lo):.6f}; gradient rel errors " + " ".join(f"{k}: {v:.2e}" for k, v in errs.items()))
    assert abs(float(loss0) - float(lo)) < 2e-5 * abs(float(lo))
    assert max(errs.values()) < 5e-4, errs
    del P, lo
    final, hist = scene.compute_global_alignment(niter=40, schedule="linear", lr=0.03, history=True)
    assert scene.state is not None and all(h == h for h in hist) and hist[-1] < hist[0]
    loss1, _ = scene.loss_and_grads()
    P, data, kw = _oracle_view(scene, requires_grad=False)
    state = dict(invalid_depth_groups=list(scene.state["invalid_depth_groups"]), valid_traj_groups=list(scene.state["valid_traj_groups"]))
    with torch.no_grad():
        l1 = oalign.alignment_loss(P, data, state=state, **kw)
    print(f"[clip 64x320x512, after 40 iterations, late terms on at 20] loss {hist[0]:.5f} -> engine {float(loss1):.6f} vs oracle at the engine's parameters {float(l1):.6f}; "
          f"windows without depth fit {state['invalid_depth_groups']}, windows with trajectory term {len(state['valid_traj_groups'])}")
    assert abs(float(loss1) - float(l1)) < 5e-5 * abs(float(l1))
    c = scene.get_im_poses_matrix()[:, :3, 3].cpu().double()
    d = c[1:] - c[:-1]
    step = d.norm(dim=1)
    cosang = (d[1:] * d[:-1]).sum(1) / (step[1:] * step[:-1])
    print(f"[clip] camera steps: mean {float(step.mean()):.5f}, spread {float(step.std() / step.mean()):.3f}; min cos between consecutive steps {float(cosang.min()):.4f}")
    assert torch.isfinite(c).all() and float(step.std() / step.mean()) < 1.0 and float(cosang.min()) > 0.0       # (a sliding camera, not a random walk)
