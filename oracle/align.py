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

Not restated (and not built in geo4d_amd/align.py): the inverse-depth and trajectory terms that the reference switches on at
iteration `depth_traj_start_iter` (they need its 5000-iteration LAD fit and evo's trajectory alignment), and the RANSAC-PnP based
initialisation (cv2.solvePnPRansac is not reproducible; geo4d_amd initialises poses from the Plücker cameras of N2 instead).
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


def alignment_loss(P, data, temporal_smoothing_weight=0.0, translation_weight=0.1, focal_break=20.0, base_scale=0.5,
                   norm_pw_scale=True, conf_clamp=10.0):
    """P: dict(im_depthmaps [n, HW] (log depth), im_poses [n, 7], im_focals [1 or n, 1] (focal_break * log f), pw_poses [G, 8]);
    data: dict(pred [G*S, HW, 3], conf [G*S, HW], e_all int64 [G*S] image of every (group, slot), H, W)."""
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
    li = ((pts[data["e_all"]] - aligned).norm(dim=-1) * wgt).sum() / total_area
    loss = li
    if temporal_smoothing_weight > 0:
        loss = loss + temporal_smoothing_weight * relative_pose_loss(im_poses[:-1], im_poses[1:], translation_weight).sum()
    return loss


def lr_at(t, schedule, lr_base, lr_min):
    if schedule == "cosine":
        return lr_min + (lr_base - lr_min) * (1 + np.cos(t * np.pi)) / 2
    if schedule == "linear":
        return lr_base + (lr_min - lr_base) * t
    raise ValueError(schedule)


def alignment_loop(P, data, niter, lr=0.01, lr_min=1e-3, schedule="cosine", **loss_kw):
    """global_alignment_loop (base_opt_group.py:553-626) on the restated loss: optimises P in place, returns the loss history."""
    params = [p for p in P.values() if p.requires_grad]
    opt = torch.optim.Adam(params, lr=lr, betas=(0.9, 0.9))
    hist = []
    for it in range(niter):
        for g in opt.param_groups:
            g["lr"] = lr_at(it / niter, schedule, lr, lr_min)
        opt.zero_grad()
        loss = alignment_loss(P, data, **loss_kw)
        loss.backward()
        opt.step()
        hist.append(float(loss))
    return hist
