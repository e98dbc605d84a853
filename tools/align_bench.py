#!/usr/bin/env python3
"""Per-iteration cost of the global-alignment loop (geo4d_amd/align.py + csrc/align.hip) at clip size: G windows of 16 frames at
HxW, stride 4 (BASELINE configs[2]/[3]: 14 windows for 64 frames, 30 for 128). Reports the fused residual kernel's achieved HBM
bandwidth against its algorithmic bytes (20 B read + 4 B written per window-frame pixel, + the depth map once per image).
usage: align_bench.py [n_frames] [H] [W]"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from geo4d_amd.align import GroupAligner
from geo4d_amd.pipeline import window_slices
n, H, W = (int(a) for a in (sys.argv[1:4] + ["128", "320", "512"][len(sys.argv) - 1:]))
dev = torch.device("cuda:0")
groups = [list(range(s.start, s.stop)) for s in window_slices(n, 4, 16)]
G, S = len(groups), 16
g = torch.Generator().manual_seed(0)
pred = (torch.randn((G, S, H, W, 3), generator=g) * 0.3 + torch.tensor([0.0, 0.0, 2.0])).to(dev)
conf = (torch.rand((G, S, H, W), generator=g) * 4 + 0.5).to(dev)
for chunk in (1024, 2048, 4096):
    a = GroupAligner(groups, pred, conf, temporal_smoothing_weight=0.015, translation_weight=1.0, chunk_pixels=chunk)
    a.P["im_depthmaps"] += 0.7
    for _ in range(3):
        a.loss_and_grads()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    it = 20
    e0.record()
    for _ in range(it):
        a.loss_and_grads()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / it
    res = {}
    for graph in (False, True):
        a.compute_global_alignment(niter=5, lr=0.01, schedule="linear", use_graph=graph)      # warm-up (allocator, capture path)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        a.compute_global_alignment(niter=100, lr=0.01, schedule="linear", use_graph=graph)
        torch.cuda.synchronize()
        res[graph] = (time.perf_counter() - t0) * 1e3 / 100
    bytes_alg = G * S * H * W * 20 + a.n * H * W * 8            # residual kernel: pred 12 + conf 4 + depth gradient 4 B per slot-pixel
    bytes_it = bytes_alg + a.n * H * W * 24                      # + fused Adam on the depth maps: param, grad, m, v read, param, m, v written
    print(f"align {n} frames = {G} windows of 16 at {H}x{W}, chunk {chunk}: loss+grads eager {ms:.3f} ms; full Adam iteration eager "
          f"{res[False]:.3f} ms, hipGraph replay {res[True]:.3f} ms = {bytes_it / res[True] / 1e6:.0f} GB/s of {bytes_it / 1e9:.2f} GB algorithmic "
          f"({100 * bytes_it / res[True] / 1e6 / 8000:.1f} % of the 8 TB/s HBM peak)", flush=True)

if os.environ.get("ALIGN_BENCH_LATE", "1") != "0":
    # the two late terms: start-up (LAD fits for all windows + trajectory checks) and the iteration with the inverse-depth term fused in
    invd = (0.5 / pred[..., 2].clamp_min(0.2) + 0.02 * torch.randn((G, S, H, W), generator=g).to(dev)).clamp_min(0.0)
    traj = torch.eye(4).repeat(G, S, 1, 1)
    traj[:, :, 0, 3] = torch.arange(S).float() * 0.05
    a = GroupAligner(groups, pred, conf, temporal_smoothing_weight=0.015, translation_weight=1.0, chunk_pixels=4096, inverse_depth=invd, traj=traj.to(dev),
                     depth_traj_start_iter=0)
    a.P["im_depthmaps"] += 0.7
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    a._set_st_depth()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    a._set_st_depth()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    valid = a._set_traj()
    t3 = time.perf_counter()
    a.set_state([], valid)
    nfit = 1 + 2 * int(any(b < 0.8 for b in a.depth_delta))
    print(f"late terms start-up: LAD fits of {G} windows ({nfit} pass(es) of up to 5000 Adam iterations, {S * H * W * 8 / 1e6:.1f} MB per window and "
          f"iteration; last pass stopped after {min(a.lad_steps):.0f}..{max(a.lad_steps):.0f} steps): {(t2 - t1) * 1e3:.0f} ms (first call {(t1 - t0) * 1e3:.0f} ms); "
          f"trajectory alignment (host) {(t3 - t2) * 1e3:.1f} ms; delta {min(a.depth_delta):.3f}..{max(a.depth_delta):.3f}", flush=True)
    # streaming rate of the fit itself: a fixed number of iterations with the stop test off (tol < 0 never fires)
    import ctypes as C
    from geo4d_amd import _lib, ops
    target = torch.empty(G * S, H * W, device=dev)
    _lib.check(a.lib.geo4d_lad_target(a.P["im_depthmaps"].data_ptr(), a.slot_img.data_ptr(), target.data_ptr(), G * S, H * W, ops._stream()), "lad_target")
    target += 0.3 * torch.rand_like(target)
    ws = torch.empty(a.lib.geo4d_lad_workspace(G, S * H * W), dtype=torch.uint8, device=dev)
    st = torch.empty(G, 2, device=dev)
    for iters in (200, 1000):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        _lib.check(a.lib.geo4d_lad_fit(a.invdepth.data_ptr(), target.data_ptr(), G, S * H * W, None, 1e-2, iters, -1.0, st.data_ptr(), None, ws.data_ptr(),
                                       ws.numel(), ops._stream()), "lad_fit")
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(f"LAD fit, {iters} iterations x {G} windows (stop test off, medians included): {dt * 1e3:.1f} ms = {dt / iters * 1e6:.1f} us per iteration = "
              f"{G * S * H * W * 8 * iters / dt / 1e12:.2f} TB/s", flush=True)
    for graph in (False, True):
        a.compute_global_alignment(niter=5, lr=0.01, schedule="linear", use_graph=graph)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        a.compute_global_alignment(niter=100, lr=0.01, schedule="linear", use_graph=graph)
        torch.cuda.synchronize()
        print(f"full iteration with the inverse-depth + trajectory terms, {'hipGraph replay' if graph else 'eager'}: "
              f"{(time.perf_counter() - t0) * 1e3 / 100:.3f} ms", flush=True)
