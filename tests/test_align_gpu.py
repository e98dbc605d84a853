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


def test_init_from_group_and_convergence_at_window_size(dev):
    """A 28-frame / 4-window scene at 40x64 pixels per frame (the latent-resolution stand-in for 16-frame windows with stride 4):
    registration-chained initialisation from the windows' own cameras, then 60 iterations; checked against the autograd oracle
    evaluated at the same parameters, and for actually aligning the clip."""
    from geo4d_amd.align import GroupAligner
    from oracle import align as oalign
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
    a = GroupAligner(groups, torch.stack(preds).to(dev), torch.stack(confs).to(dev), temporal_smoothing_weight=0.015, translation_weight=1.0)
    a.init_from_group(torch.stack(trajs).to(dev))
    loss0, grads = a.loss_and_grads()
    data = dict(pred=torch.stack(preds).reshape(-1, H * W, 3), conf=torch.stack(confs).reshape(-1, H * W), H=H, W=W,
                e_all=torch.tensor([i for grp in groups for i in grp]))
    P = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in a.P.items()}
    ref = oalign.alignment_loss(P, data, temporal_smoothing_weight=0.015, translation_weight=1.0)
    ref.backward()
    errs = {k: rel(grads[k], P[k].grad) for k in grads}
    print(f"[align 28 frames] init loss {float(loss0):.5f} (oracle {float(ref):.5f}), focal {float(a.get_focals()[0]):.2f} (true {f}); grads " +
          " ".join(f"{k}: {v:.2e}" for k, v in errs.items()))
    # the focal gradient is ONE scalar summed over 70 k signed per-pixel terms in fp32 (here chunk by chunk, in torch as a tree)
    assert abs(float(loss0) - float(ref)) < 1e-4 * float(ref) and max(errs.values()) < 3e-3 and errs["im_depthmaps"] < 1e-3, errs
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
