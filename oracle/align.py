"""TEST INFRASTRUCTURE ONLY — torch-autograd restatement of the multi-window global alignment core.

Follows dust3r/cloud_opt/optimizer_group.py (LightPointCloudGroupOptimizer: parameterisation :57-75, get_focals :188-193,
get_principal_points :205-206, depth_to_pts3d :407-417 with _fast_depthmap_to_pts3d :551-558, forward :440-525 —
the confidence-weighted point-map term `li` and the camera temporal-smoothing term — and relative_pose_loss :529-541),
dust3r/cloud_opt/base_opt_group.py (_get_poses :262-267, get_pw_scale / get_pw_poses / get_pw_norm_scale_factor :310-327,
get_adaptors :254-259, global_alignment_loop / global_alignment_iter :553-626: Adam(betas 0.9, 0.9) under a linear or cosine
learning-rate schedule) and dust3r/cloud_opt/commons.py (signed_expm1 :97-99, l1_dist :86-87, schedules :102-110).

`roma` (unpinned in requirements.txt; absent here) provides three things to that code; they are restated from its published
definitions: RigidUnitQuat(q, t).normalize().to_homogeneous() (unit quaternion in XYZW order -> rotation matrix),
rotmat_to_unitquat, and rigid_points_registration(x, y, weights, compute_scaling=True) = weighted Umeyama / Kabsch.
PINNED: tests/golden/align_tiny.pt holds the loss, its gradients and the result of the reference's own
global_alignment_loop, produced by the reference classes imported from /root/reference with ONLY `roma` replaced by these
restatements (and unrelated absent packages mocked) — tests/golden/generate.py align.

The two terms the reference switches on at iteration `depth_traj_start_iter` (optimizer_group.py:470-512) are restated below:
  * inverse-depth term + its `_set_st_depth` start-up (:333-372): per window, a least-absolute-deviation fit of (s, t) by
    dust3r/depth_eval.py's `absolute_value_scaling2` (:112-145, Adam on sum |s p + t - g|, median-ratio start :218-221) and the
    delta < 1.25 acceptance metric (:283-304). PINNED on the reference's own `depth_evaluation` and on `forward` (align_tiny.pt).
  * trajectory term + `_set_traj` (:242-268): needs `evo` (unpinned in requirements.txt, ABSENT here) for PoseTrajectory3D.align_origin
    and the RPE rotation metric; both are restated from evo's published definitions (align_origin: P = ref_0 est_0^-1 applied on
    the left of every pose; RPE(delta = 1 frame, all pairs) rotation_angle_deg: angle of (Q_i^-1 Q_{i+1})^-1 (P_i^-1 P_{i+1}), rmse).
    PARITY UNPINNED for those two functions; the loss term itself (relative_pose_loss on the transformed trajectory) is pinned
    through `forward` with the restated functions substituted for evo's.
Not restated (and not built in geo4d_amd/align.py): the RANSAC-PnP based initialisation (cv2.solvePnPRansac is not reproducible;
geo4d_amd initialises poses from the Plücker cameras of N2 instead).
"""
import math

import numpy as np
import torch


# ---- roma restatements --------------------------------------------------------------------------------------------------------
def unitquat_to_rotmat(q):
    """XYZW unit quaternion [..., 4] -> [..., 3, 3]."""
    x, y, z, w = q.unbind(-1)
    return torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w),
                        2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w),
                        2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)], dim=-1).reshape(q.shape[:-1] + (3, 3))


def rotmat_to_unitquat(R):
    """[..., 3, 3] -> XYZW unit quaternion (Shepperd's method through scipy, as roma's CPU path effectively does)."""
    from scipy.spatial.transform import Rotation
    q = Rotation.from_matrix(R.detach().cpu().double().reshape(-1, 3, 3).numpy()).as_quat()     # scipy: XYZW
    return torch.as_tensor(q, dtype=R.dtype, device=R.device).reshape(R.shape[:-2] + (4,))


def rigid_points_registration(x, y, weights=None, compute_scaling=False):
    """Weighted Kabsch / Umeyama: (R, t, s) minimising sum w |s R x + t - y|^2 (roma.rigid_points_registration)."""
    x, y = x.double(), y.double()
    w = torch.ones(x.shape[0], dtype=torch.float64) if weights is None else weights.double()
    w = w / w.sum()
    xm, ym = (w[:, None] * x).sum(0), (w[:, None] * y).sum(0)
    xc, yc = x - xm, y - ym
    M = (w[:, None] * yc).t() @ xc
    U, S, Vt = torch.linalg.svd(M)
    d = torch.sign(torch.det(U @ Vt))
    D = torch.diag(torch.tensor([1.0, 1.0, float(d)], dtype=torch.float64))
    R = U @ D @ Vt
    s = torch.tensor(1.0, dtype=torch.float64)
    if compute_scaling:
        s = (S * torch.diagonal(D)).sum() / (w * (xc * xc).sum(-1)).sum()
    t = ym - s * (R @ xm)
    return (R.float(), t.float(), s.float()) if compute_scaling else (R.float(), t.float())


class RigidUnitQuat:
    def __init__(self, linear, translation):
        self.linear, self.translation = linear, translation

    def normalize(self):
        return RigidUnitQuat(self.linear / self.linear.norm(dim=-1, keepdim=True), self.translation)

    def to_homogeneous(self):
        R = unitquat_to_rotmat(self.linear)
        H = torch.zeros(self.linear.shape[:-1] + (4, 4), dtype=R.dtype, device=R.device)
        H[..., :3, :3] = R
        H[..., :3, 3] = self.translation
        H[..., 3, 3] = 1
        return H


# ---- the loss --------------------------------------------------------------------------------------------------------------------
def signed_expm1(x):
    return torch.sign(x) * torch.expm1(torch.abs(x))


def signed_log1p(x):
    return torch.sign(x) * torch.log1p(torch.abs(x))


def poses_to_matrix(poses):
    """[n, >= 7] (XYZW quaternion, signed-log1p translation, ...) -> cam-to-world [n, 4, 4] (base_opt_group.py:262-267)."""
    return RigidUnitQuat(poses[:, :4], signed_expm1(poses[:, 4:7])).normalize().to_homogeneous()


def relative_pose_loss(RT1, RT2, translation_weight):
    rel = torch.matmul(torch.inverse(RT1), RT2)
    rot = torch.norm(rel[:, :3, :3] - torch.eye(3, device=RT1.device), dim=(1, 2))
    return rot + torch.norm(rel[:, :3, 3], dim=1) * translation_weight


def rigid_inverse(M):
    R, t = M[..., :3, :3], M[..., :3, 3]
    out = torch.zeros_like(M)
    out[..., :3, :3] = R.transpose(-1, -2)
    out[..., :3, 3] = -(R.transpose(-1, -2) @ t[..., None])[..., 0]
    out[..., 3, 3] = 1
    return out


def alignment_loss(P, data, temporal_smoothing_weight=0.0, translation_weight=0.1, focal_break=20.0, base_scale=0.5,
                   norm_pw_scale=True, conf_clamp=10.0, state=None, local_groups=None, pose_terms=True):
    """P: dict(im_depthmaps [n, HW] (log depth), im_poses [n, 7], im_focals [1 or n, 1] (focal_break * log f), pw_poses [G, 8]
    [, s_depth [G, 1], t_depth [G, 1], traj_align_poses [G, 8]]);
    data: dict(pred [G*S, HW, 3], conf [G*S, HW], e_all int64 [G*S] image of every (group, slot), H, W
    [, invdepth [G*S, HW] predicted inverse depth, traj [G*S, 4, 4] per-window camera-to-world]);
    state: None before `depth_traj_start_iter`; afterwards dict(invalid_depth_groups, valid_traj_groups) from start_depth_traj.
    local_groups / pose_terms (checker of geo4d_amd/align_dist.py): the part of the objective ONE rank of a sharded run evaluates -
    the window terms of `local_groups` only (normalised by the GLOBAL area, scales normalised over ALL windows) and the pose-only
    terms (temporal smoothing, trajectory) only where `pose_terms`; the sum over a partition of the windows is the full loss."""
    n, HW = P["im_depthmaps"].shape
    H, W = data["H"], data["W"]
    G = P["pw_poses"].shape[0]
    S = data["pred"].shape[0] // G
    dev = P["im_depthmaps"].device
    ys, xs = torch.meshgrid(torch.arange(H, device=dev), torch.arange(W, device=dev), indexing="ij")
    grid = torch.stack([xs, ys], -1).reshape(1, HW, 2).float()
    pp = torch.tensor([W / 2, H / 2], device=dev).reshape(1, 1, 2)
    lf = P["im_focals"] if P["im_focals"].shape[0] == n else P["im_focals"][:1].expand(n, 1)
    focals = (lf / focal_break).exp().reshape(n, 1, 1)
    depth = P["im_depthmaps"].exp().unsqueeze(-1)
    rel = torch.cat((depth * (grid - pp) / focals, depth), dim=-1)                       # camera frame
    im_poses = poses_to_matrix(P["im_poses"])
    pts = rel @ im_poses[:, :3, :3].transpose(1, 2) + im_poses[:, None, :3, 3]            # world frame [n, HW, 3]
    scale = P["pw_poses"][:, -1].exp()
    if norm_pw_scale:
        scale = scale * (math.log(base_scale) - P["pw_poses"][:, -1].mean()).exp()
    pw = poses_to_matrix(P["pw_poses"])
    pw = torch.cat([pw[:, :3] * scale.view(-1, 1, 1), pw[:, 3:]], dim=1)                  # scales rotation AND translation
    pw = pw.unsqueeze(1).repeat(1, S, 1, 1).reshape(-1, 4, 4)
    aligned = data["pred"] @ pw[:, :3, :3].transpose(1, 2) + pw[:, None, :3, 3]
    wgt = data["conf"].clamp(max=conf_clamp)
    total_area = float(data["pred"].shape[0] * HW)
    gmask = torch.ones(G, device=dev)
    if local_groups is not None:
        gmask = torch.zeros(G, device=dev)
        gmask[list(local_groups)] = 1
    smask = gmask.repeat_interleave(S).reshape(G * S, 1)
    li = ((pts[data["e_all"]] - aligned).norm(dim=-1) * wgt * smask).sum() / total_area
    loss = li
    if state is not None and data.get("invdepth") is not None:
        # optimizer_group.py:470-494: | 1 / (depth + 1e-6) - (s_g q + t_g) | where q > 0.05 and the window's fit was accepted, x 2
        inv = 1.0 / (P["im_depthmaps"].exp() + 1e-6)
        s = P["s_depth"].unsqueeze(1).repeat(1, S, 1).reshape(-1, 1)
        t = P["t_depth"].unsqueeze(1).repeat(1, S, 1).reshape(-1, 1)
        w = (data["invdepth"] > 0.05).float().reshape(G, S, HW).clone()
        if len(state["invalid_depth_groups"]):
            w[state["invalid_depth_groups"]] = 0
        loss = loss + 2 * ((inv[data["e_all"]] - (data["invdepth"] * s + t)).abs() * w.reshape(G * S, HW) * smask).sum() / total_area
    if pose_terms and state is not None and data.get("traj") is not None and len(state["valid_traj_groups"]):
        # optimizer_group.py:496-512
        vg = state["valid_traj_groups"]
        sc = P["traj_align_poses"][:, -1].exp()[vg]
        RT = poses_to_matrix(P["traj_align_poses"])[vg]
        tr = data["traj"].reshape(G, S, 4, 4)[vg]
        moved = torch.cat([torch.cat([tr[:, :, :3, :3], tr[:, :, :3, 3:] * sc.reshape(-1, 1, 1, 1)], -1), tr[:, :, 3:]], -2)
        moved = (RT[:, None] @ moved).reshape(-1, 4, 4)
        idx = data["e_all"].reshape(G, S)[vg].reshape(-1)
        loss = loss + 0.005 * relative_pose_loss(moved, im_poses[idx], translation_weight).sum()
    if pose_terms and temporal_smoothing_weight > 0:
        loss = loss + temporal_smoothing_weight * relative_pose_loss(im_poses[:-1], im_poses[1:], translation_weight).sum()
    return loss


# ---- start-up of the inverse-depth term -------------------------------------------------------------------------------------------
def lad_fit(pred, gt, lr, max_iters, tol=1e-6):
    """depth_eval.py:218-221 + absolute_value_scaling2 :112-145: (s, t) minimising sum |s pred + t - gt| by Adam (default betas),
    started at s = median(gt) / median(pred) (torch.median = the LOWER median), t = 0; stops when the loss repeats within tol."""
    s = torch.tensor([(torch.median(gt) / torch.median(pred)).item()], requires_grad=True, dtype=pred.dtype)
    t = torch.tensor([0.0], requires_grad=True, dtype=pred.dtype)
    opt = torch.optim.Adam([s, t], lr=lr)
    prev = None
    with torch.enable_grad():
        for _ in range(max_iters):
            opt.zero_grad()
            loss = torch.sum(torch.abs(s * pred + t - gt))
            loss.backward()
            opt.step()
            if prev is not None and torch.abs(prev - loss) < tol:
                break
            prev = loss.item()
    return s.detach().item(), t.detach().item()


def delta_125(pred_aligned, gt):
    """depth_eval.py:297-301: share of pixels whose ratio to the target is below 1.25 (prediction clamped at 1e-5 first)."""
    p = torch.clamp(pred_aligned, min=1e-5)
    return torch.mean((torch.maximum(p / gt, gt / p) < 1.25).float()).item()


def fit_window_depth(q, g, custom_mask, lr, max_iters):
    """depth_evaluation(q, g, max_depth=None, align_with_lad2=True, custom_mask=..., return_st=True) (depth_eval.py:147-330) reduced
    to what _set_st_depth reads: fit over g > 0, delta < 1.25 over the custom mask inside it."""
    m = g > 0
    s, t = lad_fit(q[m], g[m], lr, max_iters)
    mm = custom_mask[m]
    return s, t, delta_125((s * q[m] + t)[mm], g[m][mm])


@torch.no_grad()
def set_st_depth(P, data, conf_clamp=10.0):
    """optimizer_group.py:333-372: per window, fit (s, t) of its predicted inverse depth onto the inverse of the CURRENT depth maps;
    retry at two smaller learning rates when fewer than 80 % of the pixels agree; windows under 30 % are dropped from the term."""
    G = P["pw_poses"].shape[0]
    inv = (1.0 / (P["im_depthmaps"].exp() + 1e-6))[data["e_all"]].reshape(G, -1)
    q = data["invdepth"].reshape(G, -1)
    cm = (data["conf"].clamp(max=conf_clamp).reshape(G, -1) > 0.5) & (q > 0.05)
    invalid = []
    for i in range(G):
        s, t, best = fit_window_depth(q[i], inv[i], cm[i], 1e-2, 5000)
        if best < 0.8:
            for lr in (1e-4, 1e-3):
                s2, t2, d2 = fit_window_depth(q[i], inv[i], cm[i], lr, 3000)
                if d2 > best:
                    s, t, best = s2, t2, d2
        P["s_depth"].data[i], P["t_depth"].data[i] = s, t
        if best < 0.3:
            invalid.append(i)
    return invalid


# ---- start-up of the trajectory term (evo restated: see the header) ------------------------------------------------------------------
def rotation_angle_deg(R):
    return np.degrees(np.arccos(np.clip((np.trace(R) - 1) / 2, -1.0, 1.0)))


def align_origin_and_rpe(est, ref):
    """est, ref: [S, 4, 4] float64 numpy. Returns (P, rpe_rot): P = ref_0 est_0^-1 (evo PosePath3D.align_origin), rpe_rot = rmse over
    consecutive pairs of the angle (degrees) of (Q_i^-1 Q_{i+1})^-1 (P_i^-1 P_{i+1}) (evo main_rpe, rotation_angle_deg, delta 1)."""
    P = ref[0] @ np.linalg.inv(est[0])
    al = P[None] @ est
    ang = []
    for i in range(len(est) - 1):
        E = np.linalg.inv(np.linalg.inv(ref[i]) @ ref[i + 1]) @ (np.linalg.inv(al[i]) @ al[i + 1])
        ang.append(rotation_angle_deg(E[:3, :3]))
    return P, float(np.sqrt(np.mean(np.square(ang))))


@torch.no_grad()
def set_traj(P, data, base_scale=0.5, norm_pw_scale=True):
    """optimizer_group.py:242-268: move every window's predicted trajectory (translations scaled by the window's current scale) onto
    the current cameras at its first frame; traj_align_poses_g = (that transform, log scale); windows whose relative-rotation error
    stays below 4 degrees take part in the term."""
    G = P["pw_poses"].shape[0]
    S = data["pred"].shape[0] // G
    im = poses_to_matrix(P["im_poses"])
    scale = P["pw_poses"][:, -1].exp()
    if norm_pw_scale:
        scale = scale * (math.log(base_scale) - P["pw_poses"][:, -1].mean()).exp()
    valid = []
    for g in range(G):
        tr = data["traj"].reshape(G, S, 4, 4)[g].clone()
        tr[:, :3, 3] = tr[:, :3, 3] * scale[g]
        ref = im[data["e_all"].reshape(G, S)[g]]
        Pm, rpe_rot = align_origin_and_rpe(tr.double().numpy(), ref.double().numpy())
        P["traj_align_poses"].data[g, :4] = rotmat_to_unitquat(torch.from_numpy(Pm[:3, :3])).float()
        P["traj_align_poses"].data[g, 4:7] = signed_log1p(torch.from_numpy(Pm[:3, 3])).float()
        P["traj_align_poses"].data[g, 7] = float(np.log(float(scale[g])))
        if rpe_rot < 4:
            valid.append(g)
    return valid


def lr_at(t, schedule, lr_base, lr_min):
    if schedule == "cosine":
        return lr_min + (lr_base - lr_min) * (1 + np.cos(t * np.pi)) / 2
    if schedule == "linear":
        return lr_base + (lr_min - lr_base) * t
    raise ValueError(schedule)


def alignment_loop(P, data, niter, lr=0.01, lr_min=1e-3, schedule="cosine", depth_traj_start_iter=150, **loss_kw):
    """global_alignment_loop (base_opt_group.py:553-626) on the restated loss: optimises P in place, returns the loss history.
    Parameters that have not received a gradient yet (s_depth, t_depth, traj_align_poses before the start iteration) are skipped by
    torch.optim.Adam, so their moments and bias corrections start when their terms do - as in the reference."""
    params = [p for p in P.values() if p.requires_grad]
    opt = torch.optim.Adam(params, lr=lr, betas=(0.9, 0.9))
    hist = []
    state = None
    has_extra = data.get("invdepth") is not None or data.get("traj") is not None
    for it in range(niter):
        for g in opt.param_groups:
            g["lr"] = lr_at(it / niter, schedule, lr, lr_min)
        opt.zero_grad()
        if has_extra and it == depth_traj_start_iter:
            state = dict(invalid_depth_groups=set_st_depth(P, data, loss_kw.get("conf_clamp", 10.0)) if data.get("invdepth") is not None else [],
                         valid_traj_groups=set_traj(P, data, loss_kw.get("base_scale", 0.5), loss_kw.get("norm_pw_scale", True))
                         if data.get("traj") is not None else [])
        loss = alignment_loss(P, data, state=state, **loss_kw)
        loss.backward()
        opt.step()
        hist.append(float(loss))
    return hist
