"""Multi-window global alignment — the stage the all-gathered clip feeds.

``GroupAligner`` is the core of ``dust3r.cloud_opt.optimizer_group.LightPointCloudGroupOptimizer`` +
``base_opt_group.global_alignment_loop`` (called by ``post_optimization``, scripts/evaluation/test_geo4d.py:30-51,509): the same
parameters (per-image log-depth maps, camera poses as XYZW quaternion + signed-log1p translation, ``focal_break * log f`` focals —
shared by default —, one sim(3) ``pw_poses`` row per window with the reference's scale normalisation), the same loss (the
confidence-weighted L2-norm residual between re-projected depth points and the windows' aligned point maps, confidences clamped
at 10, + the camera temporal-smoothing term) and the same optimiser (Adam, betas (0.9, 0.9), linear / cosine schedule).

How an iteration runs here: ONE fused HIP kernel (csrc/align.hip) reads every window prediction once and produces the loss, the
gradient of the depth maps and 3x3 / 3-vector gradient sums per image and per (window, frame); the chain rule from those sums to
quaternions, log-translations, log-focal and log-scales is a few hundred flops per image and is taken by autograd over tiny device
tensors (no host synchronisation); the depth maps get a fused HIP Adam step. The reference needs ~40 full-size elementwise kernels
per iteration for the same numbers.

Initialisation follows ``init_im_poses.align_group`` / ``init_from_pts3d_group`` (:82-181, :569-635): windows are chained by
confidence-weighted similarity registration (``roma.rigid_points_registration`` = weighted Umeyama, restated here), ``pw_poses``
come from registering every window to the chained cloud, depths are the z of the cloud in each camera. The reference finds every
camera by OpenCV RANSAC-PnP (cv2.solvePnPRansac, absent here): ``init_from_group(pose_init="pnp")`` follows it with the seeded
restatement in geo4d_amd/pnp.py (same candidate focals, threshold and consensus rule; sampler and minimal solver differ and say so);
the DEFAULT start takes the per-window camera-to-world matrices that the Plücker ray maps already give (geo4d_amd/rays.py, N2) —
cheaper (no host loop over images) and equally close to the optimum on the fixtures. DEVIATION (second call): calling
``compute_global_alignment`` again continues with the late terms ON from its first iteration, whereas the reference restarts its
epoch counter (terms off until iteration 150, ``_set_st_depth`` re-run). The focals start, as in the reference,
from the Weiszfeld estimate on every image's ray map (`estimate_focal_weiszfeld`, pinned on dust3r.post_process; the reference then
lets PnP pick among that value and +-3 % of the image size).

From iteration ``depth_traj_start_iter`` (150) on the reference adds two terms (optimizer_group.py:470-512), built here as well:
  * inverse depth: ``2/A sum |1/(depth + 1e-6) - (s_g q + t_g)|`` over the pixels whose predicted inverse depth q exceeds 0.05, fused
    into the SAME residual kernel (4 more bytes per slot-pixel). Its start-up ``_set_st_depth`` (:333-372) — per window a
    5000-iteration Adam least-absolute-deviation fit of (s, t) started at a median ratio, scored by delta < 1.25, retried at two
    smaller learning rates under 80 %, dropped under 30 % — runs as HIP kernels for all windows at once (csrc/align.hip: radix-select
    medians, one launch per Adam iteration, one scoring pass); the reference runs ~6 torch kernels per iteration per window.
  * trajectory: ``0.005 sum relative_pose_loss(T_g [R_k | e^l_g t_k], pose_i)`` over the windows whose predicted trajectory agrees
    with the current cameras within 4 degrees of relative rotation after ``_set_traj`` (:242-268, evo's align_origin + RPE, restated:
    16 4x4 matrices per window, host side); the term itself is tiny-tensor autograd like the smoothing term.
s_depth / t_depth / traj_align_poses join Adam when their terms start (torch skips parameters without gradients, so their moments
and bias corrections count from the start iteration — reproduced with a second hyper-parameter table).
"""
import ctypes as C
import math

import numpy as np
import torch

from . import _lib, ops

FOCAL_BREAK = 20.0
TORCH_CHAIN = __import__("os").environ.get("GEO4D_ALIGN_CHAIN", "hip") == "torch"   # round 2's autograd chain rule instead of align_small.hip


def signed_expm1(x):
    return torch.sign(x) * torch.expm1(torch.abs(x))


def signed_log1p(x):
    return torch.sign(x) * torch.log1p(torch.abs(x))


def quat_to_rotmat(q):
    """XYZW quaternion [..., 4] (normalised here) -> [..., 3, 3] (roma.RigidUnitQuat(...).normalize().to_homogeneous())."""
    x, y, z, w = (q / q.norm(dim=-1, keepdim=True)).unbind(-1)
    return torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w),
                        2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w),
                        2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)], dim=-1).reshape(q.shape[:-1] + (3, 3))


def rotmat_to_quat(R):
    """[3, 3] -> XYZW unit quaternion (branch on the largest diagonal term; w >= 0 convention not enforced, like scipy)."""
    R = R.double()
    m00, m11, m22 = R[0, 0], R[1, 1], R[2, 2]
    tr = m00 + m11 + m22
    if tr > 0:
        s = torch.sqrt(tr + 1.0) * 2
        q = torch.stack([(R[2, 1] - R[1, 2]) / s, (R[0, 2] - R[2, 0]) / s, (R[1, 0] - R[0, 1]) / s, 0.25 * s])
    elif m00 > m11 and m00 > m22:
        s = torch.sqrt(1.0 + m00 - m11 - m22) * 2
        q = torch.stack([0.25 * s, (R[0, 1] + R[1, 0]) / s, (R[0, 2] + R[2, 0]) / s, (R[2, 1] - R[1, 2]) / s])
    elif m11 > m22:
        s = torch.sqrt(1.0 + m11 - m00 - m22) * 2
        q = torch.stack([(R[0, 1] + R[1, 0]) / s, 0.25 * s, (R[1, 2] + R[2, 1]) / s, (R[0, 2] - R[2, 0]) / s])
    else:
        s = torch.sqrt(1.0 + m22 - m00 - m11) * 2
        q = torch.stack([(R[0, 2] + R[2, 0]) / s, (R[1, 2] + R[2, 1]) / s, 0.25 * s, (R[1, 0] - R[0, 1]) / s])
    return (q / q.norm()).float()


def rigid_points_registration(x, y, weights):
    """(s, R, T) minimising sum w |s R x + T - y|^2 (init_im_poses.py:797-800 -> roma.rigid_points_registration(compute_scaling=True)):
    weighted Umeyama, sums in fp64."""
    x, y, w = x.reshape(-1, 3).double(), y.reshape(-1, 3).double(), weights.reshape(-1).double()
    w = w / w.sum()
    xm, ym = (w[:, None] * x).sum(0), (w[:, None] * y).sum(0)
    xc, yc = x - xm, y - ym
    # 3 x 3 weighted covariance as nine fp64 sums: `(w yc)^T @ xc` hands rocBLAS a dgemm with M = N = 3 and K = millions of points, which
    # took 64 ms per window at 16 x 320 x 512 (59 calls = 3.8 of the 3.9 s of a 128-frame clip's alignment init, tools/align_init_profile.py)
    # (row by row: three [N, 3] temporaries instead of one [N, 3, 3] = 190 MB per 16 x 320 x 512 window)
    wy = w[:, None] * yc
    cov = torch.stack([(wy[:, i:i + 1] * xc).sum(0) for i in range(3)])
    U, S, Vt = torch.linalg.svd(cov.cpu())
    d = torch.sign(torch.det(U @ Vt))
    D = torch.diag(torch.stack([torch.ones(()).double(), torch.ones(()).double(), d]))
    R = (U @ D @ Vt).to(x.device)
    s = (S * torch.diagonal(D)).sum().to(x.device) / (w * (xc * xc).sum(-1)).sum()
    return s.float(), R.float(), (ym - s * (R @ xm)).float()


def estimate_focal_weiszfeld(rays, pp=None, iters=10):
    """Focal of each map in `rays` [B, H, W, 3] (point map or ray-direction map) = argmin_f sum | pixel - f (x, y) / z |: closed-form L2
    start, then `iters` rounds of least squares re-weighted by the inverse residual. What align_group uses to initialise the focals
    from the ray maps (init_im_poses.py:133-136 -> estimate_focal :810-817 -> dust3r/post_process.py:12-60, focal_mode='weiszfeld',
    without its optional clipping). Host-side initialisation math: a dozen reductions per clip, any device."""
    B, H, W, _ = rays.shape
    ys, xs = torch.meshgrid(torch.arange(H, device=rays.device), torch.arange(W, device=rays.device), indexing="ij")
    pp = torch.tensor([W / 2, H / 2], device=rays.device).expand(B, 2) if pp is None else pp
    px = torch.stack([xs, ys], -1).reshape(1, H * W, 2).float() - pp.reshape(B, 1, 2)
    r = rays.reshape(B, H * W, 3).float()
    q = (r[..., :2] / r[..., 2:3]).nan_to_num(posinf=0, neginf=0)
    qp, qq = (q * px).sum(-1), q.square().sum(-1)
    f = qp.mean(1) / qq.mean(1)
    for _ in range(iters):
        w = (px - f.view(B, 1, 1) * q).norm(dim=-1).clip(min=1e-8).reciprocal()
        f = (w * qp).mean(1) / (w * qq).mean(1)
    return f


def align_origin_and_rpe(est, ref):
    """The two evo pieces of ``_set_traj`` (optimizer_group.py:242-268 -> dust3r/utils/vo_eval.py:174-266), restated (evo is not
    installed): ``PosePath3D.align_origin`` = left-multiply the estimate by P = ref_0 est_0^-1, and the RPE rotation metric
    (delta = 1 frame, all pairs, rotation_angle_deg): rmse of the angle of (Q_i^-1 Q_{i+1})^-1 (P_i^-1 P_{i+1}).
    est, ref [S, 4, 4] float64 ndarrays -> (P [4, 4], rmse in degrees). Pinned on closed-form cases (tests/test_align_closed_form_cpu.py)."""
    Pm = ref[0] @ np.linalg.inv(est[0])
    al = Pm[None] @ est
    ang = []
    for k in range(len(est) - 1):
        E = np.linalg.inv(np.linalg.inv(ref[k]) @ ref[k + 1]) @ (np.linalg.inv(al[k]) @ al[k + 1])
        ang.append(np.degrees(np.arccos(np.clip((np.trace(E[:3, :3]) - 1) / 2, -1.0, 1.0))))
    return Pm, float(np.sqrt(np.mean(np.square(ang))))


def lr_at(t, schedule, lr_base, lr_min):
    """commons.py:102-110."""
    if schedule == "cosine":
        return lr_min + (lr_base - lr_min) * (1 + np.cos(t * np.pi)) / 2
    if schedule == "linear":
        return lr_base + (lr_min - lr_base) * t
    raise ValueError(f"bad lr schedule={schedule!r}")


class GroupAligner:
    MAX_SLOTS = 6          # csrc/align.hip MAXS: windows per image the fused residual kernel unrolls

    def __init__(self, groups, pred, conf, shared_focal=True, temporal_smoothing_weight=0.0, translation_weight=0.1, base_scale=0.5,
                 conf_clamp=10.0, chunk_pixels=4096, inverse_depth=None, traj=None, depth_traj_start_iter=150, shard=None):
        """groups: list of G lists of S image indices; pred [G, S, H, W, 3], conf [G, S, H, W] fp32 on the HIP device;
        inverse_depth [G, S, H, W(, 1)] (the decoded inverse-depth modality mapped to [0, 1]) and traj [G, S, 4, 4] (every window's
        camera-to-world matrices in its own frame, N2) switch the two late terms on (pred_pts['inverse_depthmap'] / ['traj'],
        scripts/evaluation/test_geo4d.py:498-501)."""
        if not pred.is_cuda:
            raise _lib.Geo4DNativeError("geo4d_amd.align.GroupAligner runs only on a HIP device (there is no CPU fallback)")
        self.lib = _lib.load()
        G, S, H, W, _ = pred.shape
        self.groups, self.G, self.S, self.H, self.W = [list(g) for g in groups], G, S, H, W
        self.n = 1 + max(max(g) for g in self.groups)
        self.dev = pred.device
        self.pred = pred.reshape(G * S, H * W, 3).float().contiguous()
        self.conf = conf.reshape(G * S, H * W).float().contiguous()
        self.shared_focal, self.tsw, self.tw, self.base_scale, self.conf_clamp = shared_focal, temporal_smoothing_weight, translation_weight, base_scale, conf_clamp
        self.norm_pw_scale = True
        e_all = [i for g in self.groups for i in g]
        if any(not any(i == img for i in e_all) for img in range(self.n)):
            raise ValueError("every image must belong to at least one window")
        self.shard = shard
        if shard is not None and not shard.local_groups:
            raise ValueError(f"rank {shard.rank} of {shard.world} owns no window of this {G}-window clip: build the shard with "
                             "align_dist.make_shard (it falls back to the replicated optimisation when windows < ranks)")
        local = set(range(G)) if shard is None else set(shard.local_groups)
        self.local_groups = sorted(local)
        self.primary = shard is None or shard.primary          # evaluates the pose-only terms (temporal smoothing, trajectory)
        slots = [[s for s, i in enumerate(e_all) if i == img and (s // S) in local] for img in range(self.n)]
        self.max_slots = max(len(s) for s in slots)
        if self.max_slots > self.MAX_SLOTS:
            raise NotImplementedError(f"an image belongs to {self.max_slots} windows; the fused residual kernel handles {self.MAX_SLOTS} "
                                      "(window stride >= 3 at 16 frames per window)")
        self.n_local_slots = sum(len(s) for s in slots)
        ptr = np.concatenate([[0], np.cumsum([len(s) for s in slots])]).astype(np.int32)
        self.slot_ptr = torch.from_numpy(ptr).to(self.dev)
        self.slot_idx = torch.tensor([s for lst in slots for s in lst], dtype=torch.int32, device=self.dev)
        self.slot_order = torch.tensor([s for lst in slots for s in lst], dtype=torch.long, device=self.dev)   # CSR position -> slot
        self.slot_group = torch.tensor([s // S for s in range(G * S)], dtype=torch.long, device=self.dev)
        all_slots = [[s for s, i in enumerate(e_all) if i == img] for img in range(self.n)]      # every window's slots (not shard-restricted)
        self._img_slot_ptr = torch.tensor(np.concatenate([[0], np.cumsum([len(x) for x in all_slots])]).astype(np.int32), device=self.dev)
        self._img_slot_idx = torch.tensor([x for lst in all_slots for x in lst], dtype=torch.int32, device=self.dev)
        order = [s for lst in slots for s in lst]                  # CSR entry -> slot; per window: its entries, ascending (align_small.hip)
        by_group = [[e for e, s in enumerate(order) if s // S == g] for g in range(G)]
        self._group_ptr = torch.tensor(np.concatenate([[0], np.cumsum([len(b) for b in by_group])]).astype(np.int32), device=self.dev)
        self._group_entries = torch.tensor([e for b in by_group for e in b] or [0], dtype=torch.int32, device=self.dev)
        self.chunk = chunk_pixels
        f0 = FOCAL_BREAK * math.log(max(H, W))
        z = lambda *s: torch.zeros(s, device=self.dev)
        self.P = {"im_depthmaps": z(self.n, H * W), "im_poses": torch.cat([z(self.n, 3), torch.ones(self.n, 1, device=self.dev), z(self.n, 3)], 1),
                  "im_focals": torch.full((1 if shared_focal else self.n, 1), f0, device=self.dev),
                  "pw_poses": torch.cat([z(G, 3), torch.ones(G, 1, device=self.dev), z(G, 4)], 1)}
        need = self.lib.geo4d_align_workspace(self.n, G * S, H, W, self.chunk)
        self._ws = torch.empty(need, dtype=torch.uint8, device=self.dev)
        self._img_sums, self._slot_sums = z(self.n, 14), z(G * S, 14)
        self._grad_ld = z(self.n, H * W)
        self._pp = torch.tensor([W / 2, H / 2], device=self.dev)
        self.invdepth = None if inverse_depth is None else inverse_depth.reshape(G * S, H * W).float().contiguous()
        self.traj = None if traj is None else traj.reshape(G, S, 4, 4).float().to(self.dev)
        self.start_iter = depth_traj_start_iter
        self.state = None                        # after the start iteration: dict(invalid_depth_groups, valid_traj_groups)
        self.frozen = set()                      # parameter names Adam leaves alone (preset_focal(..., requires_grad=False) -> {"im_focals"})
        self.slot_img = torch.tensor(e_all, dtype=torch.int32, device=self.dev)
        self.e_all = torch.tensor(e_all, dtype=torch.long, device=self.dev)
        if self.invdepth is not None:
            self.P["s_depth"], self.P["t_depth"] = torch.ones(G, 1, device=self.dev), z(G, 1)
        if self.traj is not None:
            self.P["traj_align_poses"] = torch.cat([z(G, 3), torch.ones(G, 1, device=self.dev), z(G, 4)], 1)

    def _late_keys(self):
        """Parameters that receive gradients only once their term is on (torch.optim.Adam skips them until then)."""
        if self.state is None:
            return ()
        keys = ("s_depth", "t_depth") if self.invdepth is not None else ()
        return keys + (("traj_align_poses",) if self.traj is not None and len(self.state["valid_traj_groups"]) else ())

    # ---- parameter -> geometry (tiny device tensors, differentiable) -----------------------------------------------------------
    def get_focals(self):
        lf = self.P["im_focals"]
        return (lf.expand(self.n, 1) / FOCAL_BREAK).exp()

    def get_im_poses(self):
        p = self.P["im_poses"]
        return quat_to_rotmat(p[:, :4]), signed_expm1(p[:, 4:7])

    def get_pw_scale(self):
        ls = self.P["pw_poses"][:, -1]
        scale = ls.exp()
        if self.norm_pw_scale:
            scale = scale * (math.log(self.base_scale) - ls.mean()).exp()
        return scale

    def get_pw_poses(self):
        p = self.P["pw_poses"]
        s = self.get_pw_scale().view(-1, 1, 1)
        return quat_to_rotmat(p[:, :4]) * s, signed_expm1(p[:, 4:7]) * s.view(-1, 1)      # scales rotation AND translation

    def get_depthmaps(self):
        return self.P["im_depthmaps"].exp().reshape(self.n, self.H, self.W)

    def get_im_poses_matrix(self):
        R, t = self.get_im_poses()
        M = torch.eye(4, device=self.dev).repeat(self.n, 1, 1)
        M[:, :3, :3], M[:, :3, 3] = R, t
        return M

    def get_pts3d(self):
        """World points [n, H, W, 3] from depth / pose / focal (optimizer_group.py:407-417)."""
        R, t = self.get_im_poses()
        f = self.get_focals().reshape(self.n, 1, 1)
        ys, xs = torch.meshgrid(torch.arange(self.H, device=self.dev), torch.arange(self.W, device=self.dev), indexing="ij")
        grid = torch.stack([xs, ys], -1).reshape(1, -1, 2).float()
        d = self.P["im_depthmaps"].exp().unsqueeze(-1)
        cam = torch.cat([d * (grid - self._pp) / f, d], -1)
        return (cam @ R.transpose(1, 2) + t[:, None]).reshape(self.n, self.H, self.W, 3)

    # ---- one loss / gradient evaluation ---------------------------------------------------------------------------------------
    def _small(self):
        """The descriptor of the two small-parameter launches (csrc/align_small.hip) for the CURRENT parameter tensors."""
        sm = _lib.AlignSmall()
        P = self.P
        sm.im_poses, sm.im_focals, sm.pw_poses = P["im_poses"].data_ptr(), P["im_focals"].data_ptr(), P["pw_poses"].data_ptr()
        sm.cams, sm.slot_trf = self._cams.data_ptr(), self._trf.data_ptr()
        sm.img_sums, sm.slot_sums, sm.group_sums = self._img_sums.data_ptr(), self._slot_sums.data_ptr(), self._group_sums.data_ptr()
        sm.group_ptr, sm.group_entries = self._group_ptr.data_ptr(), self._group_entries.data_ptr()
        sm.scale_terms = self._scale_terms.data_ptr()
        g = self._small_grads
        sm.grad_im_poses, sm.grad_im_focals, sm.grad_pw_poses, sm.loss = g["im_poses"].data_ptr(), g["im_focals"].data_ptr(), g["pw_poses"].data_ptr(), self._loss.data_ptr()
        sm.n_imgs, sm.n_groups, sm.n_slots, sm.slots_per_group, sm.n_listed_slots = self.n, self.G, self.G * self.S, self.S, self.n_local_slots
        sm.n_focals, sm.norm_pw_scale = P["im_focals"].shape[0], int(self.norm_pw_scale)
        sm.focal_break, sm.base_scale, sm.ppx, sm.ppy = FOCAL_BREAK, self.base_scale, self.W / 2, self.H / 2
        sm.smooth_weight, sm.translation_weight = (self.tsw if self.primary else 0.0), self.tw
        return sm

    def loss_and_grads(self):
        """Returns (loss 0-dim device tensor, dict of gradients shaped like self.P; parameters of terms that are off are absent).
        Five launches: parameters -> cams / window transforms / (s, t) per slot (align_refresh), the fused residual kernel + its two
        fixed-order reductions, gradient sums -> parameter gradients (align_small_grads: the chain rule through quaternions / signed-log
        translations / log focal / normalised log scales, the temporal-smoothing term and the trajectory term, csrc/align_small.hip). `GEO4D_ALIGN_CHAIN=torch` selects round 2's autograd chain instead."""
        if TORCH_CHAIN:
            return self._loss_and_grads_torch()
        late = self._late_keys()
        z = lambda *s: torch.zeros(s, device=self.dev)
        if getattr(self, "_cams", None) is None:
            self._cams, self._trf, self._loss = z(self.n, 16), z(self.G * self.S, 12), z(1)
            self._group_sums, self._scale_terms = torch.zeros(self.G, 14, device=self.dev, dtype=torch.float64), torch.zeros(self.G, device=self.dev, dtype=torch.float64)
            self._small_grads = {"im_poses": z(self.n, 7), "im_focals": torch.zeros_like(self.P["im_focals"]), "pw_poses": z(self.G, 8),
                                 "s_depth": z(self.G, 1), "t_depth": z(self.G, 1)}
        sm = self._small()
        depth_on = "s_depth" in late
        traj_on = "traj_align_poses" in late
        if depth_on:
            if getattr(self, "_slot_st", None) is None:
                self._slot_st = z(self.G * self.S, 3)
            sm.grad_s_depth, sm.grad_t_depth = self._small_grads["s_depth"].data_ptr(), self._small_grads["t_depth"].data_ptr()
            sm.slot_st, sm.s_depth, sm.t_depth, sm.depth_ok = self._slot_st.data_ptr(), self.P["s_depth"].data_ptr(), self.P["t_depth"].data_ptr(), self._depth_ok.data_ptr()
        if traj_on:
            if "traj_align_poses" not in self._small_grads:
                self._small_grads["traj_align_poses"] = z(self.G, 8)
            if self.primary:                                       # pose-only term: one rank evaluates it
                sm.traj, sm.traj_align, sm.traj_valid = self._traj_full.data_ptr(), self.P["traj_align_poses"].data_ptr(), self._traj_valid.data_ptr()
                sm.slot_img, sm.img_slot_ptr, sm.img_slot_idx = self.slot_img.data_ptr(), self._img_slot_ptr.data_ptr(), self._img_slot_idx.data_ptr()
                sm.grad_traj, sm.traj_weight = self._small_grads["traj_align_poses"].data_ptr(), 0.005
        _lib.check(self.lib.geo4d_align_refresh(C.byref(sm), ops._stream()), "geo4d_align_refresh")
        a = _lib.Align()
        a.pred, a.conf, a.logdepth, a.cams, a.slot_trf = self.pred.data_ptr(), self.conf.data_ptr(), self.P["im_depthmaps"].data_ptr(), self._cams.data_ptr(), self._trf.data_ptr()
        a.slot_ptr, a.slot_idx = self.slot_ptr.data_ptr(), self.slot_idx.data_ptr()
        a.grad_logdepth, a.img_sums, a.slot_sums = self._grad_ld.data_ptr(), self._img_sums.data_ptr(), self._slot_sums.data_ptr()
        a.workspace, a.workspace_bytes = self._ws.data_ptr(), self._ws.numel()
        a.n_imgs, a.n_slots, a.H, a.W, a.chunk_pixels, a.max_slots_per_image = self.n, self.G * self.S, self.H, self.W, self.chunk, self.max_slots
        a.conf_clamp, a.inv_area = self.conf_clamp, 1.0 / float(self.G * self.S * self.H * self.W)
        if depth_on:
            a.invdepth, a.slot_st, a.depth_weight = self.invdepth.data_ptr(), self._slot_st.data_ptr(), 2.0 * a.inv_area
        _lib.check(self.lib.geo4d_align_residual(C.byref(a), ops._stream()), "geo4d_align_residual")
        _lib.check(self.lib.geo4d_align_small_grads(C.byref(sm), ops._stream()), "geo4d_align_small_grads")
        loss = self._loss[0]
        grads = {k: self._small_grads[k] for k in ("im_poses", "im_focals", "pw_poses") + (("s_depth", "t_depth") if depth_on else ())
                 + (("traj_align_poses",) if traj_on else ())}
        grads["im_depthmaps"] = self._grad_ld
        if self.shard is not None:
            loss, grads = self.shard.reduce(loss, grads)        # one all-reduce: loss, small gradients, depth gradients of shared images
        return loss, grads

    def _loss_and_grads_torch(self):
        """Round 2's form of loss_and_grads: the chain rule from the gradient sums to the parameters by autograd over tiny tensors
        (~450 launches per iteration). Kept as the independent cross-check of csrc/align_small.hip (tests) and for A/B."""
        late = self._late_keys()
        small = {k: self.P[k].detach().clone().requires_grad_(True) for k in ("im_poses", "im_focals", "pw_poses") + late}
        saved, self.P = self.P, dict(self.P, **small)
        try:
            R, t = self.get_im_poses()
            f = self.get_focals()
            sR, st = self.get_pw_poses()
        finally:
            self.P = saved
        cams = torch.cat([R.reshape(self.n, 9), t, f, self._pp.expand(self.n, 2), torch.zeros(self.n, 1, device=self.dev)], 1).detach().contiguous()
        trf = torch.cat([sR.reshape(self.G, 9), st], 1).detach()[self.slot_group].contiguous()
        a = _lib.Align()
        a.pred, a.conf, a.logdepth, a.cams, a.slot_trf = self.pred.data_ptr(), self.conf.data_ptr(), self.P["im_depthmaps"].data_ptr(), cams.data_ptr(), trf.data_ptr()
        a.slot_ptr, a.slot_idx = self.slot_ptr.data_ptr(), self.slot_idx.data_ptr()
        a.grad_logdepth, a.img_sums, a.slot_sums = self._grad_ld.data_ptr(), self._img_sums.data_ptr(), self._slot_sums.data_ptr()
        a.workspace, a.workspace_bytes = self._ws.data_ptr(), self._ws.numel()
        a.n_imgs, a.n_slots, a.H, a.W, a.chunk_pixels, a.max_slots_per_image = self.n, self.G * self.S, self.H, self.W, self.chunk, self.max_slots
        a.conf_clamp, a.inv_area = self.conf_clamp, 1.0 / float(self.G * self.S * self.H * self.W)
        depth_on = "s_depth" in late
        if depth_on:
            slot_st = torch.cat([small["s_depth"].detach(), small["t_depth"].detach(), self._depth_ok], 1)[self.slot_group].contiguous()
            a.invdepth, a.slot_st, a.depth_weight = self.invdepth.data_ptr(), slot_st.data_ptr(), 2.0 * a.inv_area
        _lib.check(self.lib.geo4d_align_residual(C.byref(a), ops._stream()), "geo4d_align_residual")
        I = self._img_sums
        # slot sums come back in CSR (image-major) order: put them in slot order, then add the S frames of each window
        # (with a shard only this rank's slots are listed: rows past n_local_slots are not written)
        Ssum = torch.zeros_like(self._slot_sums).index_copy_(0, self.slot_order, self._slot_sums[:self.n_local_slots]).reshape(self.G, self.S, -1).sum(1)
        loss = I[:, 13].sum()
        # chain rule through the tiny parameter -> matrix maps: d loss = <dL/dR, dR> + <dL/dt, dt> + dL/df df + <dL/dsR, dsR> + <dL/dst, dst>
        surrogate = (I[:, :9].reshape(self.n, 3, 3) * R).sum() + (I[:, 9:12] * t).sum() + (I[:, 12:13] * f).sum() + \
            (Ssum[:, :9].reshape(self.G, 3, 3) * sR).sum() + (Ssum[:, 9:12] * st).sum()
        if depth_on:
            surrogate = surrogate + (Ssum[:, 12:13] * small["s_depth"]).sum() + (Ssum[:, 13:14] * small["t_depth"]).sum()
        eye = torch.eye(3, device=self.dev)
        if "traj_align_poses" in late and self.primary:
            # optimizer_group.py:496-512: T_g [R_k | e^l t_k] against the cameras of the window's images (rigid inverse written out)
            vg, idx = self._traj_vg, self._traj_idx          # device index tensors made once at the start-up (capturable)
            tap = small["traj_align_poses"].index_select(0, vg)
            Ra, ta, sc = quat_to_rotmat(tap[:, :4]), signed_expm1(tap[:, 4:7]), tap[:, 7].exp()
            Rk, tk = self._traj_R, self._traj_t
            Rm = Ra[:, None] @ Rk
            tm = (Ra[:, None] @ (tk * sc[:, None, None])[..., None])[..., 0] + ta[:, None]
            Rm, tm = Rm.reshape(-1, 3, 3), tm.reshape(-1, 3)
            rel_R = Rm.transpose(1, 2) @ R[idx]
            rel_t = (Rm.transpose(1, 2) @ (t[idx] - tm)[:, :, None])[:, :, 0]
            ltraj = 0.005 * (torch.norm(rel_R - eye, dim=(1, 2)) + torch.norm(rel_t, dim=1) * self.tw).sum()
            surrogate = surrogate + ltraj
            loss = loss + ltraj.detach()
        if self.tsw > 0 and self.n > 1 and self.primary:
            # relative_pose_loss (optimizer_group.py:529-541): inverse(RT1) @ RT2 with RT rigid, so inverse = [R^T | -R^T t]
            # (no batched LU on the device: keeps the iteration capturable)
            Rt = R[:-1].transpose(1, 2)
            rel_R, rel_t = Rt @ R[1:], (Rt @ (t[1:] - t[:-1])[:, :, None])[:, :, 0]
            smooth = (torch.norm(rel_R - eye, dim=(1, 2)) + torch.norm(rel_t, dim=1) * self.tw).sum()
            surrogate = surrogate + self.tsw * smooth
            loss = loss + self.tsw * smooth.detach()
        surrogate.backward()
        grads = {k: (v.grad if v.grad is not None else torch.zeros_like(v)) for k, v in small.items()}
        grads["im_depthmaps"] = self._grad_ld
        if self.shard is not None:
            loss, grads = self.shard.reduce(loss, grads)        # one all-reduce: loss, small gradients, depth gradients of shared images
        return loss, grads

    # ---- start-up of the two late terms -------------------------------------------------------------------------------------------
    @torch.no_grad()
    def _lad_fit(self, target, active, lr, iters, ws):
        G, n = self.G, self.S * self.H * self.W
        st, counts = torch.empty(G, 2, device=self.dev), torch.zeros(G, 2, dtype=torch.int32, device=self.dev)
        info = torch.empty(G, 2, device=self.dev)
        act = None if active is None else torch.tensor(active, dtype=torch.uint8, device=self.dev)
        _lib.check(self.lib.geo4d_lad_fit(self.invdepth.data_ptr(), target.data_ptr(), G, n, None if act is None else act.data_ptr(), lr, iters,
                                          1e-6, st.data_ptr(), info.data_ptr(), ws.data_ptr(), ws.numel(), ops._stream()), "geo4d_lad_fit")
        self.lad_steps = info[:, 0].tolist()                 # Adam steps each window took before the reference's stop test fired
        _lib.check(self.lib.geo4d_lad_delta(self.invdepth.data_ptr(), target.data_ptr(), self.conf.data_ptr(), st.data_ptr(), G, n, 0.5,
                                            self.conf_clamp, 0.05, counts.data_ptr(), ops._stream()), "geo4d_lad_delta")
        c = counts.double().cpu()
        return st, (c[:, 0] / c[:, 1]).tolist()

    @torch.no_grad()
    def _set_st_depth(self):
        """optimizer_group.py:333-372. Sets s_depth / t_depth; returns the windows dropped from the term."""
        G, S, HW = self.G, self.S, self.H * self.W
        target = torch.empty(G * S, HW, device=self.dev)
        _lib.check(self.lib.geo4d_lad_target(self.P["im_depthmaps"].data_ptr(), self.slot_img.data_ptr(), target.data_ptr(), G * S, HW, ops._stream()),
                   "geo4d_lad_target")
        ws = torch.empty(self.lib.geo4d_lad_workspace(G, S * HW), dtype=torch.uint8, device=self.dev)
        mine = None if self.shard is None else [1 if g in set(self.local_groups) else 0 for g in range(G)]    # a rank fits its own windows
        st, best = self._lad_fit(target, mine, 1e-2, 5000, ws)
        retry = [1 if (b < 0.8 and (mine is None or mine[g])) else 0 for g, b in enumerate(best)]
        if any(retry):
            for lr in (1e-4, 1e-3):
                st2, d2 = self._lad_fit(target, retry, lr, 3000, ws)
                for g in range(G):
                    if retry[g] and d2[g] > best[g]:
                        st[g], best[g] = st2[g], d2[g]
        if self.shard is not None:
            tab = torch.cat([st, torch.tensor(best, device=self.dev, dtype=st.dtype).reshape(G, 1).nan_to_num()], 1)
            tab = self.shard.merge_rows(tab.nan_to_num(), self.local_groups)
            st, best = tab[:, :2].contiguous(), tab[:, 2].tolist()
        self.P["s_depth"].copy_(st[:, :1])
        self.P["t_depth"].copy_(st[:, 1:])
        self.depth_delta = best
        return [g for g in range(G) if best[g] < 0.3]

    @torch.no_grad()
    def _set_traj(self):
        """optimizer_group.py:242-268 (evo's align_origin and RPE rotation restated: oracle/align.py header). Host side: G x S 4x4."""
        R, t = self.get_im_poses()
        im = np.tile(np.eye(4), (self.n, 1, 1))
        im[:, :3, :3], im[:, :3, 3] = R.double().cpu().numpy(), t.double().cpu().numpy()
        scale = self.get_pw_scale().double().cpu().numpy()
        traj = self.traj.double().cpu().numpy()
        valid = []
        for g, grp in enumerate(self.groups):
            est = traj[g].copy()
            est[:, :3, 3] *= scale[g]
            ref = im[grp]
            Pm, rpe = align_origin_and_rpe(est, ref)
            self.P["traj_align_poses"][g, :4] = rotmat_to_quat(torch.from_numpy(Pm[:3, :3])).to(self.dev)
            self.P["traj_align_poses"][g, 4:7] = signed_log1p(torch.from_numpy(Pm[:3, 3])).float().to(self.dev)
            self.P["traj_align_poses"][g, 7] = float(np.log(scale[g]))
            if rpe < 4:
                valid.append(g)
        return valid

    def start_depth_traj(self):
        """What forward() does at epoch == depth_traj_start_iter (optimizer_group.py:470-503)."""
        self.set_state(self._set_st_depth() if self.invdepth is not None else [], self._set_traj() if self.traj is not None else [])
        return self.state

    def set_state(self, invalid_depth_groups, valid_traj_groups):
        """Which windows take part in the late terms (normally decided by start_depth_traj)."""
        self.state = dict(invalid_depth_groups=list(invalid_depth_groups), valid_traj_groups=list(valid_traj_groups))
        ok = torch.ones(self.G, 1)
        ok[self.state["invalid_depth_groups"]] = 0
        self._depth_ok = ok.to(self.dev)
        if self.traj is not None and len(valid_traj_groups):
            vg = torch.tensor(self.state["valid_traj_groups"], dtype=torch.long, device=self.dev)
            self._traj_vg = vg
            valid = torch.zeros(self.G, dtype=torch.int32)
            valid[self.state["valid_traj_groups"]] = 1
            self._traj_valid = valid.to(self.dev)
            self._traj_full = self.traj.reshape(self.G * self.S, 16).float().contiguous()
            self._traj_idx = self.e_all.reshape(self.G, self.S).index_select(0, vg).reshape(-1)
            self._traj_R, self._traj_t = self.traj.index_select(0, vg)[:, :, :3, :3].contiguous(), self.traj.index_select(0, vg)[:, :, :3, 3].contiguous()

    # ---- optimisation loop (base_opt_group.py:553-626) --------------------------------------------------------------------------
    def compute_global_alignment(self, niter=300, lr=0.01, lr_min=1e-3, schedule="cosine", history=False, use_graph=True):
        """Adam (betas 0.9 / 0.9, eps 1e-8, bias-corrected: torch.optim.Adam's arithmetic) under the reference's schedule. One
        iteration = fused residual kernel + tiny chain rule + fused Adam on the depth maps + Adam on the small parameters, with
        lr and the bias corrections read from device tables indexed by a device counter — so ONE captured hipGraph is replayed
        with no host work in between (`use_graph`; the eager loop runs the same kernels). With the late terms the loop is two such
        phases around the start-up at iteration `depth_traj_start_iter` (which synchronises with the host once)."""
        b1 = b2 = 0.9
        eps = 1e-8
        if self.shard is not None and self.shard.active:
            use_graph = False                                  # the per-iteration all-reduce runs on RCCL's stream: eager launches
        steps = torch.arange(1, niter + 1, dtype=torch.float64)
        lrs = torch.tensor([lr_at(it / niter, schedule, lr, lr_min) for it in range(niter)], dtype=torch.float64)
        table = torch.stack([lrs, 1 - b1 ** steps, (1 - b2 ** steps).sqrt()], 1).float().to(self.dev)            # [niter, 3]
        has_late = self.invdepth is not None or self.traj is not None
        # iteration at which the late terms (and their parameters' Adam moments) start; 0 when a previous call already started them
        start = 0 if self.state is not None else (min(self.start_iter, niter) if has_late else niter)
        lsteps = (steps - start).clamp_min(1)                                  # a late parameter's own step count
        table_late = torch.stack([lrs, 1 - b1 ** lsteps, (1 - b2 ** lsteps).sqrt()], 1).float().to(self.dev)
        idx = torch.zeros((1,), dtype=torch.long, device=self.dev)
        hyper, hyper_late = torch.zeros(3, device=self.dev), torch.zeros(3, device=self.dev)
        mom = {k: (torch.zeros_like(v), torch.zeros_like(v)) for k, v in self.P.items()}
        losses = torch.zeros(niter, device=self.dev)

        def iteration():
            hyper.copy_(table.index_select(0, idx)[0])
            hyper_late.copy_(table_late.index_select(0, idx)[0])
            loss, grads = self.loss_and_grads()
            losses.index_copy_(0, idx, loss.reshape(1))
            d, (m, v) = self.P["im_depthmaps"], mom["im_depthmaps"]
            _lib.check(self.lib.geo4d_adam_step_dev(d.data_ptr(), grads["im_depthmaps"].data_ptr(), m.data_ptr(), v.data_ptr(), d.numel(),
                                                    hyper.data_ptr(), b1, b2, eps, ops._stream()), "geo4d_adam_step_dev")
            late = self._late_keys()
            for k in ("im_poses", "im_focals", "pw_poses") + late:   # a few dozen numbers each: plain tensor ops, same formula
                if k in self.frozen:
                    continue
                g, (m, v), h = grads[k].contiguous(), mom[k], hyper_late if k in late else hyper
                _lib.check(self.lib.geo4d_adam_step_dev(self.P[k].data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), self.P[k].numel(),
                                                        h.data_ptr(), b1, b2, eps, ops._stream()), "geo4d_adam_step_dev")
            idx.add_(1)

        def run(count):
            done = 0
            if use_graph and count > 2:
                try:
                    iteration()                             # eager first iteration (allocator warm-up), then capture the second
                    done = 1
                    g = torch.cuda.CUDAGraph()
                    side = torch.cuda.Stream()
                    side.wait_stream(torch.cuda.current_stream())
                    with torch.cuda.stream(side):
                        with torch.cuda.graph(g, stream=side, capture_error_mode="thread_local"):
                            iteration()
                    torch.cuda.current_stream().wait_stream(side)
                    for _ in range(count - 1):          # capturing records, it does not run
                        g.replay()
                    done = count
                except RuntimeError:                        # an op that cannot be captured on this build: finish eagerly
                    torch.cuda.synchronize()
                    done = None
            return done

        def run_phase(first, last):
            done = run(last - first)
            if done is None:
                done = int(idx.item()) - first
            for _ in range(done, last - first):
                iteration()

        if self.state is None:
            run_phase(0, start)
            if has_late and start < niter:
                self.start_depth_traj()
                run_phase(start, niter)
        else:                                               # terms already started (a second call continues with them on)
            run_phase(0, niter)
        if self.shard is not None:                             # every image's depth map from the rank that owns it
            self.P["im_depthmaps"].copy_(self.shard.gather_depthmaps(self.P["im_depthmaps"]))
        hist = losses.tolist() if history else None
        return float(losses[-1]) if niter else float("inf"), hist

    # ---- initialisation (init_im_poses.py:82-181, 569-635) ----------------------------------------------------------------------
    @torch.no_grad()
    def init_from_group(self, traj, focal=None, raymaps=None, pose_init="traj", niter_PnP=100, pnp_seed=0):
        """pose_init = "pnp": the reference's initialisation (init_im_poses.align_group :82-214): windows chained by registration
        WITHOUT overwriting an image's first estimate, every camera from seeded RANSAC-PnP of its chained point map against the
        pixel grid (geo4d_amd/pnp.py, confidence > 0.5), tried at the image's ray-map focal and -/+ 3 % of the image size; images of
        the FIRST window take the winning candidate as their focal, the others keep the ray-map estimate (as the reference's
        `if im_focals[img_idx] is None` leaves them); the shared focal is the mean. `traj` is then only used by the trajectory term.
        pose_init = "traj" (default): cameras from the Plücker ray maps instead (below).
        traj [G, S, 4, 4]: camera-to-world of every frame in its window's own frame (the Plücker cameras of N2);
        focal: pixels; raymaps [G, S, H, W, 3] (pred_pts['raydir']): when given and `focal` is None, every image's focal is the
        Weiszfeld estimate on its first ray map and the shared focal their mean, as align_group / init_from_pts3d_group do
        (init_im_poses.py:133-136, 183-185, 627-628); with neither, the focal is estimated from window 0's first point map."""
        G, S, H, W = self.G, self.S, self.H, self.W
        pred = self.pred.reshape(G, S, H * W, 3)
        conf = self.conf.reshape(G, S, H * W)
        pts3d, conf_list, im_poses = [None] * self.n, [None] * self.n, [None] * self.n
        done = set()
        if pose_init == "pnp":
            return self._init_pnp(pred, conf, focal, raymaps, niter_PnP, pnp_seed)
        for k, i in enumerate(self.groups[0]):
            pts3d[i], conf_list[i], im_poses[i] = pred[0, k].clone(), conf[0, k].clone(), traj[0, k].clone().float()
            done.add(i)
        for g in range(1, G):
            grp = self.groups[g]
            seen = [k for k, i in enumerate(grp) if i in done]
            assert seen and grp[0] in done, "the first image of every window must belong to an earlier window"
            s, R, T = rigid_points_registration(pred[g, seen], torch.stack([pts3d[grp[k]] for k in seen]),
                                                torch.stack([conf[g, k] * conf_list[grp[k]] for k in seen]))
            for k, i in enumerate(grp):       # later windows overwrite earlier estimates, as the reference does
                pts3d[i], conf_list[i] = s * (pred[g, k] @ R.t()) + T, conf[g, k].clone()
                done.add(i)
                M = torch.eye(4, device=self.dev)
                M[:3, :3] = R @ traj[g, k, :3, :3].float()
                M[:3, 3] = s * (R @ traj[g, k, :3, 3].float()) + T
                im_poses[i] = M
        # pairwise poses: register every window onto the chained cloud
        for g, grp in enumerate(self.groups):
            s, R, T = rigid_points_registration(pred[g], torch.stack([pts3d[i] for i in grp]), torch.stack([conf[g, k] * conf_list[i] for k, i in enumerate(grp)]))
            self.P["pw_poses"][g, :4] = rotmat_to_quat(R).to(self.dev)
            self.P["pw_poses"][g, 4:7] = signed_log1p(T / s)
            self.P["pw_poses"][g, 7] = torch.log(s)
        sf = float((math.log(self.base_scale) - self.P["pw_poses"][:, -1].mean()).exp()) if self.norm_pw_scale else 1.0
        if focal is None and raymaps is not None:
            first = {}
            for g, grp in enumerate(self.groups):
                for k, i in enumerate(grp):
                    first.setdefault(i, (g, k))
            rm = raymaps.reshape(G, S, H, W, 3)
            per_image = estimate_focal_weiszfeld(torch.stack([rm[g, k] for g, k in (first[i] for i in range(self.n))]).to(self.dev))
            per_image = torch.where(torch.isfinite(per_image) & (per_image > 0), per_image, torch.full_like(per_image, max(H, W) / (2.0 * math.tan(math.radians(30.0)))))
            self.init_focals = per_image
            focal = float(per_image.mean()) if self.shared_focal else per_image
        if focal is None:
            p0, ys, xs = pred[0, 0], *torch.meshgrid(torch.arange(H, device=self.dev), torch.arange(W, device=self.dev), indexing="ij")
            u, v = (xs.reshape(-1).float() - W / 2), (ys.reshape(-1).float() - H / 2)
            ok = (p0[:, 2] > 1e-6) & (conf[0, 0] > 0.5)
            fx = (u * p0[:, 2] / p0[:, 0])[ok & (u.abs() > W / 8)]
            fy = (v * p0[:, 2] / p0[:, 1])[ok & (v.abs() > H / 8)]
            cand = torch.cat([fx[torch.isfinite(fx)], fy[torch.isfinite(fy)]])
            focal = float(cand.median()) if cand.numel() else float(max(H, W))
        # degenerate predictions (e.g. the synthetic noise maps of bench.py's clip mode) can yield a non-positive / non-finite estimate, whose
        # log the reference would carry into the optimisation as NaN: fall back to the 60-degree field of view dust3r's estimator is scaled by
        fbase = max(H, W) / (2.0 * math.tan(math.radians(30.0)))
        if torch.is_tensor(focal):
            focal = torch.where(torch.isfinite(focal) & (focal > 0), focal, torch.full_like(focal, fbase))
            self.P["im_focals"][:, 0] = FOCAL_BREAK * torch.log(focal)
        else:
            if not (math.isfinite(focal) and focal > 0):
                focal = fbase
            self.P["im_focals"][:] = FOCAL_BREAK * math.log(focal)
        sky = 0.0
        for i in range(self.n):
            M = im_poses[i].clone()
            M[:3, 3] *= sf
            Rw, tw = M[:3, :3], M[:3, 3]
            depth = ((pts3d[i] * sf - tw) @ Rw)[:, 2]                      # z of R^T (X - t)
            skym = conf_list[i] < 1e-4
            if i == 0:
                sky = depth.max()
            depth = torch.where(skym, torch.as_tensor(sky, device=self.dev), depth)
            self.P["im_depthmaps"][i] = depth.clamp_min(1e-6).log().nan_to_num(neginf=0)
            self.P["im_poses"][i, :4] = rotmat_to_quat(Rw).to(self.dev)
            self.P["im_poses"][i, 4:7] = signed_log1p(tw)
        return self


def _init_pnp(self, pred, conf, focal, raymaps, niter_PnP, seed):
    """GroupAligner.init_from_group(pose_init="pnp"): align_group + init_from_pts3d_group of the reference (see the docstring there)."""
    from . import pnp
    G, S, H, W = self.G, self.S, self.H, self.W
    pts3d, conf_list, im_poses, im_focals = [None] * self.n, [None] * self.n, [None] * self.n, [None] * self.n
    rm = None if raymaps is None else raymaps.reshape(G, S, H, W, 3)

    def ray_focal(g, k):
        return None if rm is None else float(estimate_focal_weiszfeld(rm[g, k][None].to(self.dev))[0])

    def run_pnp(i, g, k):
        msk = (conf[g, k] > 0.5).reshape(H, W).cpu().numpy()
        return pnp.fast_pnp(pts3d[i].reshape(H, W, 3).double().cpu().numpy(), im_focals[i], msk, niter_PnP=niter_PnP, seed=seed)
    done = set()
    for k, i in enumerate(self.groups[0]):                                  # the first window is the world frame
        pts3d[i], conf_list[i] = pred[0, k].clone(), conf[0, k].clone()
        im_focals[i] = ray_focal(0, k)
        res = run_pnp(i, 0, k)
        if res:
            im_focals[i], im_poses[i] = res[0], torch.from_numpy(res[1]).float().to(self.dev)
        if im_poses[i] is None:
            im_poses[i] = torch.eye(4, device=self.dev)
        done.add(i)
    for g in range(1, G):
        grp = self.groups[g]
        assert grp[0] in done, "the first image of every window must belong to an earlier window"
        seen = [k for k, i in enumerate(grp) if i in done]
        s, R, T = rigid_points_registration(pred[g, seen], torch.stack([pts3d[grp[k]] for k in seen]),
                                            torch.stack([conf[g, k] * conf_list[grp[k]] for k in seen]))
        for k, i in enumerate(grp):
            if pts3d[i] is None:                                            # an image keeps its FIRST chained estimate
                pts3d[i], conf_list[i] = s * (pred[g, k] @ R.t()) + T, conf[g, k].clone()
                done.add(i)
            if im_focals[i] is None:
                im_focals[i] = ray_focal(g, k)
            # the PnP result is only used where pose / focal are still unset (both assignments below are `is None`-guarded, as in the
            # reference) and the solver's sampler is re-seeded per call: skipping the call for already-initialised images (~3 of 4
            # occurrences at stride 4) changes nothing but the start-up time (host-side numpy RANSAC)
            res = run_pnp(i, g, k) if (im_poses[i] is None or im_focals[i] is None) else None
            if res:
                if im_poses[i] is None:
                    im_poses[i] = torch.from_numpy(res[1]).float().to(self.dev)
                if im_focals[i] is None:
                    im_focals[i] = res[0]
            if im_poses[i] is None:
                im_poses[i] = torch.eye(4, device=self.dev)
    for g, grp in enumerate(self.groups):                                   # pairwise poses: every window onto the chained cloud
        s, R, T = rigid_points_registration(pred[g], torch.stack([pts3d[i] for i in grp]), torch.stack([conf[g, k] * conf_list[i] for k, i in enumerate(grp)]))
        self.P["pw_poses"][g, :4] = rotmat_to_quat(R).to(self.dev)
        self.P["pw_poses"][g, 4:7] = signed_log1p(T / s)
        self.P["pw_poses"][g, 7] = torch.log(s)
    sf = float((math.log(self.base_scale) - self.P["pw_poses"][:, -1].mean()).exp()) if self.norm_pw_scale else 1.0
    self.init_focals = torch.tensor([f if f is not None else float(max(H, W)) for f in im_focals], device=self.dev)
    if focal is not None:
        self.P["im_focals"][:] = FOCAL_BREAK * (torch.log(focal) if torch.is_tensor(focal) else math.log(focal))
    elif self.shared_focal:
        self.P["im_focals"][:] = FOCAL_BREAK * math.log(float(self.init_focals.mean()))
    else:
        self.P["im_focals"][:, 0] = FOCAL_BREAK * torch.log(self.init_focals)
    sky = 0.0
    for i in range(self.n):
        M = im_poses[i].clone()
        M[:3, 3] *= sf
        Rw, tw = M[:3, :3], M[:3, 3]
        depth = ((pts3d[i] * sf - tw) @ Rw)[:, 2]
        skym = conf_list[i] < 1e-4
        if i == 0:
            sky = depth.max()
        depth = torch.where(skym, torch.as_tensor(sky, device=self.dev), depth)
        self.P["im_depthmaps"][i] = depth.clamp_min(1e-6).log().nan_to_num(neginf=0)
        self.P["im_poses"][i, :4] = rotmat_to_quat(Rw).to(self.dev)
        self.P["im_poses"][i, 4:7] = signed_log1p(tw)
    return self


GroupAligner._init_pnp = _init_pnp


def post_optimization(slices, maps, traj, args=None, conf_optimize=True, lr=0.03, align=True, intrinsics=None,
                      use_raymap=True, use_inverse_depthmap=True, use_traj=True, pointmap_vae_used=True, depth_traj_start_iter=150,
                      sharded=None, pose_init="traj"):
    """The consumer of the gathered clip: ``post_optimization`` of scripts/evaluation/test_geo4d.py:30-51 with the pred_list its
    window loop builds (:446-501). ``slices`` / ``maps [n_windows, 11, T, H, W]`` / ``traj [n_windows, T, 4, 4]`` are what
    ``pipeline.run_clip(..., with_cameras=True)`` returns; ``args`` = the config's ``postprocess`` tree (a dict or any object with
    n_iter / pose_schedule / temporal_smoothing_weight / translation_weight / not_shared_focal / use_gt_focal attributes; None =
    the shipped values). ``sharded``: None = shard the optimisation over the ranks of the default process group when there is one
    (align_dist.AlignShard: every rank evaluates its block of windows, one all-reduce per iteration), False = replicate it.
    Returns the optimised ``GroupAligner`` (``get_depthmaps`` / ``get_im_poses_matrix`` / ``get_focals``).
    ``intrinsics [n_images, 3, 3]`` presets the focals and freezes them (scene.preset_focal(..., requires_grad=False) in the script)."""
    from .pipeline import postprocess_window
    get = (lambda k, d: args.get(k, d)) if isinstance(args, dict) else (lambda k, d: getattr(args, k, d))
    if args is None:
        get = lambda k, d: d
    post = [postprocess_window(m[None], pointmap_vae_used=pointmap_vae_used) for m in maps]
    groups = [list(range(s.start, s.stop)) for s in slices]
    conf = torch.stack([p["conf"][..., 0] for p in post])
    if not conf_optimize:
        conf = torch.ones_like(conf)
    shard = None
    if sharded or (sharded is None and torch.distributed.is_available() and torch.distributed.is_initialized()
                   and torch.distributed.get_world_size() > 1):
        from .align_dist import make_shard
        shard = make_shard(groups, 1 + max(max(g) for g in groups))     # None when the clip has fewer windows than ranks (replicated run)
    scene = GroupAligner(groups, torch.stack([p["pts3d"] for p in post]), conf, shard=shard,
                         shared_focal=not get("not_shared_focal", False) and not get("use_gt_focal", False),
                         temporal_smoothing_weight=get("temporal_smoothing_weight", 0.015), translation_weight=get("translation_weight", 1.0),
                         inverse_depth=torch.stack([p["inverse_depthmap"] for p in post]) if use_inverse_depthmap else None,
                         traj=traj if use_traj else None, depth_traj_start_iter=depth_traj_start_iter)
    focal = None
    if intrinsics is not None:
        per_image = (intrinsics[:, 0, 0] + intrinsics[:, 1, 1]) / 2.0
        focal = per_image.to(scene.dev).float() if not scene.shared_focal else float(per_image.float().mean())
    scene.init_from_group(traj, focal=focal, raymaps=torch.stack([p["raymap"] for p in post]) if use_raymap else None, pose_init=pose_init)
    if intrinsics is not None:
        scene.frozen.add("im_focals")
    if align:
        scene.compute_global_alignment(niter=get("n_iter", 500), schedule=get("pose_schedule", "linear"), lr=lr)
    return scene
