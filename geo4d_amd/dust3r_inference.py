"""Call-surface shims for the names of ``dust3r/inference.py``.

None of these are on the Geo4D inference path (the entry scripts never import ``dust3r.inference``); the north-star asks
for the names to exist. They are plain-PyTorch host utilities with the reference signatures — no kernels, no hot path.
`inference()` is the reference's batching loop over a CALLER-SUPPLIED pairwise model (round 4; tests/test_dust3r_inference_cpu.py).
"""
import torch


def _interleave(a, b):
    out = {}
    for k, va in a.items():
        vb = b[k]
        if isinstance(va, torch.Tensor) and va.ndim == vb.ndim:
            out[k] = torch.stack((va, vb), dim=1).flatten(0, 1)
        else:
            out[k] = [x for pair in zip(va, vb) for x in pair]
    return out


def make_batch_symmetric(batch):
    """dust3r/inference.py:27-30."""
    v1, v2 = batch
    return _interleave(v1, v2), _interleave(v2, v1)


def check_if_same_size(pairs):
    """dust3r/inference.py:104-107."""
    s1 = [a["img"].shape[-2:] for a, _ in pairs]
    s2 = [b["img"].shape[-2:] for _, b in pairs]
    return all(s1[0] == s for s in s1) and all(s2[0] == s for s in s2)


def loss_of_one_batch(batch, model, criterion, device, symmetrize_batch=False, use_amp=False, ret=None):
    """dust3r/inference.py:58-79 (pairwise DUSt3R model call; the DUSt3R network itself is not part of Geo4D)."""
    view1, view2 = batch
    skip = {"depthmap", "dataset", "label", "instance", "idx", "true_shape", "rng"}
    for view in batch:
        for name in list(view.keys()):
            if name not in skip and isinstance(view[name], torch.Tensor):
                view[name] = view[name].to(device, non_blocking=True)
    if symmetrize_batch:
        view1, view2 = make_batch_symmetric(batch)
    pred1, pred2 = model(view1, view2)
    loss = criterion(view1, view2, pred1, pred2) if criterion is not None else None
    result = dict(view1=view1, view2=view2, pred1=pred1, pred2=pred2, loss=loss)
    return result[ret] if ret else result


def to_cpu(x):
    """dust3r/utils/device.py:10-44 (`todevice(x, 'cpu')`): every tensor of a nested dict / list / tuple moved to the host, numpy arrays
    become tensors, containers keep their type, everything else passes through."""
    import numpy as np
    if isinstance(x, dict):
        return {k: to_cpu(v) for k, v in x.items()}
    if isinstance(x, (tuple, list)):
        return type(x)(to_cpu(v) for v in x)
    if isinstance(x, np.ndarray):
        x = torch.from_numpy(x)
    return x.to("cpu") if torch.is_tensor(x) else x


def collate_with_cat(items, lists=False):
    """dust3r/utils/device.py:47-75: merge a list of per-batch results into one. Dicts are merged key by key and tuples position by
    position (recursively); tensors / arrays are concatenated along dim 0, or flattened into one python list of samples when `lists`
    (images of different sizes cannot be stacked); scalars / strings stay the list they are; nested lists are chained; a list that
    starts with None collapses to None."""
    import numpy as np
    if isinstance(items, dict):
        return {k: collate_with_cat(v, lists=lists) for k, v in items.items()}
    if not isinstance(items, (tuple, list)):
        return items
    if len(items) == 0:
        return items
    first, kind = items[0], type(items)
    if first is None:
        return None
    if isinstance(first, (bool, int, float, str)):
        return items
    if isinstance(first, tuple):
        return kind(collate_with_cat(col, lists=lists) for col in zip(*items))
    if isinstance(first, dict):
        return {k: collate_with_cat([it[k] for it in items], lists=lists) for k in first}
    if isinstance(first, (torch.Tensor, np.ndarray)):
        if lists:
            return [sample for batch in items for sample in batch]
        return torch.cat([torch.from_numpy(t) if isinstance(t, np.ndarray) else t for t in items])
    out = kind()
    for it in items:              # lists of lists: chain
        out = out + it
    return out


@torch.no_grad()
def inference(pairs, model, device, batch_size=8, verbose=True):
    """dust3r/inference.py:82-101: run a pairwise model (any callable `model(view1, view2) -> (pred1, pred2)`; the reference passes its
    DUSt3R network, which Geo4D does not ship) over `pairs` = [(view1_dict, view2_dict), ...] in batches: collate the pairs of a batch,
    call the model through loss_of_one_batch (no criterion), bring the result to the host, and collate the per-batch results into one
    dict(view1, view2, pred1, pred2, loss). Pairs whose images differ in size force batch size 1 and list-valued (unstacked) outputs."""
    if verbose:
        print(f">> Inference with model on {len(pairs)} image pairs")
    ragged = not check_if_same_size(pairs)
    if ragged:
        batch_size = 1
    chunks = []
    for lo in range(0, len(pairs), batch_size):
        out = loss_of_one_batch(collate_with_cat(pairs[lo:lo + batch_size]), model, None, device)
        chunks.append(to_cpu(out))
        if verbose:
            print(f"   pairs {lo}..{min(lo + batch_size, len(pairs)) - 1} done")
    return collate_with_cat(chunks, lists=ragged)


def _rigid_apply(pose, pts):
    """dust3r/utils/geometry.py:40-112 `geotrf` for the case get_pred_pts3d needs: `pose` [..., 3|4, 3|4] applied to points
    [..., h, w, 3] (leading batch dimensions equal): x -> R x + t."""
    d = pts.shape[-1]
    pose = pose.to(pts.dtype)
    lead = pose.shape[:-2]
    if tuple(pts.shape[:len(lead)]) != tuple(lead):
        raise ValueError(f"camera_pose batch {tuple(lead)} does not match the points {tuple(pts.shape)}")
    flat = pts.reshape(*lead, -1, d)
    out = flat @ pose[..., :d, :d].transpose(-1, -2)
    if pose.shape[-1] == d + 1:
        out = out + pose[..., None, :d, d]
    elif pose.shape[-1] != d:
        raise ValueError(f"bad transform shape {tuple(pose.shape)} for {d}-D points")
    return out.reshape(pts.shape)


def depthmap_to_pts3d(depth, pseudo_focal, pp=None, **_):
    """dust3r/utils/geometry.py:114-162: unproject a z-depth map [B,H,W] (or [B,H,W,n]) with per-pixel pseudo focals ([B,H,W],
    [B,1,H,W] or [B,2,H,W]) around the principal point `pp` [B,2] (default: the image centre (W-1)/2, (H-1)/2)."""
    B, H, W = depth.shape[:3]
    if pseudo_focal.ndim == 3:
        fx = fy = pseudo_focal
    elif pseudo_focal.ndim == 4:
        fx = pseudo_focal[:, 0]
        fy = pseudo_focal[:, 1] if pseudo_focal.shape[1] == 2 else fx
    else:
        raise NotImplementedError("Error, unknown input focal shape format.")
    assert fx.shape == depth.shape[:3] and fy.shape == depth.shape[:3]
    u = torch.arange(W, device=depth.device).view(1, 1, W).expand(1, H, W)
    v = torch.arange(H, device=depth.device).view(1, H, 1).expand(1, H, W)
    if pp is None:
        u, v = u - (W - 1) / 2, v - (H - 1) / 2
    else:
        u, v = u - pp[:, 0, None, None], v - pp[:, 1, None, None]
    if depth.ndim == 3:
        pts = torch.empty((B, H, W, 3), device=depth.device)
        pts[..., 0], pts[..., 1], pts[..., 2] = depth * u / fx, depth * v / fy, depth
    else:
        pts = torch.empty((B, H, W, 3, depth.shape[3]), device=depth.device)
        pts[..., 0, :], pts[..., 1, :], pts[..., 2, :] = depth * (u / fx)[..., None], depth * (v / fy)[..., None], depth
    return pts


def get_pred_pts3d(gt, pred, use_pose=False):
    """dust3r/inference.py:110-132, all three branches: depth + pseudo_focal heads are unprojected (principal point from
    gt['camera_intrinsics'] when present), a `pts3d` head is returned as is, and either goes through pred['camera_pose'] when
    `use_pose`; `pts3d_in_other_view` is already in the other camera's frame (use_pose must be set)."""
    if "depth" in pred and "pseudo_focal" in pred:
        try:
            pp = gt["camera_intrinsics"][..., :2, 2]
        except KeyError:
            pp = None
        pts3d = depthmap_to_pts3d(**pred, pp=pp)
    elif "pts3d" in pred:
        pts3d = pred["pts3d"]
    elif "pts3d_in_other_view" in pred:
        assert use_pose is True
        return pred["pts3d_in_other_view"]
    else:
        raise KeyError("pred holds none of depth + pseudo_focal, pts3d, pts3d_in_other_view")
    if use_pose:
        camera_pose = pred.get("camera_pose")
        assert camera_pose is not None
        pts3d = _rigid_apply(camera_pose, pts3d)
    return pts3d


def find_opt_scaling(gt_pts1, gt_pts2, pr_pts1, pr_pts2=None, fit_mode="weiszfeld_stop_grad", valid1=None, valid2=None):
    """dust3r/inference.py:135-179: scale s minimising |pr - s*gt| (avg / median / Weiszfeld IRLS)."""
    def nan_invalid(x, valid):
        if x is None:
            return None
        x = x.clone()
        if valid is not None:
            x[~valid] = float("nan")
        return x.flatten(1, 2)
    g = [t for t in (nan_invalid(gt_pts1, valid1), nan_invalid(gt_pts2, valid2)) if t is not None]
    p = [t for t in (nan_invalid(pr_pts1, valid1), nan_invalid(pr_pts2, valid2)) if t is not None]
    all_gt, all_pr = torch.cat(g, dim=1), torch.cat(p, dim=1)
    dot_gp, dot_gg = (all_pr * all_gt).sum(-1), all_gt.square().sum(-1)
    if fit_mode.startswith("avg"):
        s = dot_gp.nanmean(1) / dot_gg.nanmean(1)
    elif fit_mode.startswith("median"):
        s = (dot_gp / dot_gg).nanmedian(1).values
    elif fit_mode.startswith("weiszfeld"):
        s = dot_gp.nanmean(1) / dot_gg.nanmean(1)
        for _ in range(10):
            w = (all_pr - s.view(-1, 1, 1) * all_gt).norm(dim=-1).clip_(min=1e-8).reciprocal()
            s = (w * dot_gp).nanmean(1) / (w * dot_gg).nanmean(1)
    else:
        raise ValueError(f"bad {fit_mode=}")
    if fit_mode.endswith("stop_grad"):
        s = s.detach()
    return s.clip(min=1e-3)
