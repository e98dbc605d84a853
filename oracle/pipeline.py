"""TEST INFRASTRUCTURE ONLY — restatement of the per-window glue of scripts/evaluation/test_geo4d.py.

window_slices: :417-423 (16-frame windows, stride 4, tail window always appended — `slice(T-16, T)` never equals
`slice(T-16, T, 1)`, so the tail is duplicated when (T-16) % stride == 0).
decode_modalities: :248-258 with decode_pm_confhead :291-312 and LatentDiffusion.decode_core (ddpm3d.py:802-823).
postprocess_window: :446-501 (channel split, softplus confidence, sky / far masks, denormalize_pc_bbox2 :84-89).
"""
import torch
import torch.nn.functional as F

from . import vae as ovae


def window_slices(T, stride=4, length=16):
    out = [(s, s + length) for s in range(0, T - length + 1, stride)]
    out.append((T - length, T))   # the reference's membership test compares slice(a,b) with slice(a,b,1): never equal
    return out


@torch.no_grad()
def decode_modalities(first_stage_sd, pointmap_sd, ddconfig, adaptorconfig, samples, scale_factor=0.18215):
    """samples [B,16,T,h,w] -> [B,11,T,H,W]: xyz+conf (pointmap VAE) | ray (3) | cross (3) | depth mean (1)."""
    b, _, t, h, w = samples.shape

    def frames(z):
        return z.permute(0, 2, 1, 3, 4).reshape(b * t, 4, h, w) / scale_factor

    def back(y):
        return y.reshape(b, t, y.shape[1], y.shape[2], y.shape[3]).permute(0, 2, 1, 3, 4)

    pc = back(ovae.decode_with_conf_adaptor(pointmap_sd, ddconfig, adaptorconfig, frames(samples[:, 0:4])))
    ray = back(ovae.decode(first_stage_sd, ddconfig, frames(samples[:, 4:8])))
    cross = back(ovae.decode(first_stage_sd, ddconfig, frames(samples[:, 8:12])))
    depth = back(ovae.decode(first_stage_sd, ddconfig, frames(samples[:, 12:16]))).mean(dim=1, keepdim=True)
    return torch.cat([pc, ray, cross, depth], dim=1)


@torch.no_grad()
def postprocess_window(batch_samples):
    """batch_samples [1,11,T,H,W] -> dict(pts3d [T,H,W,3], conf [T,H,W,1] (inverse confidence), inverse_depthmap,
    raymap, crossmap, invalid mask)."""
    x = batch_samples[0].permute(1, 0, 2, 3)            # t c h w
    raymap = x[:, 4:7].permute(0, 2, 3, 1)
    crossmap = x[:, 7:10].permute(0, 2, 3, 1)
    inv_depth = (x[:, 10:11].permute(0, 2, 3, 1) + 1.0) / 2.0
    conf = F.softplus(x[:, 3:4]).permute(0, 2, 3, 1)
    pts = x[:, 0:3].permute(0, 2, 3, 1)
    lo, hi = 1.05 - 0.35, 1.05 + 0.35
    sky = ((pts > lo) & (pts < hi)).all(dim=-1, keepdim=True)
    far = (pts.abs() > 1.99).any(dim=-1, keepdim=True)
    invalid = sky | far
    conf = torch.where(invalid, torch.full_like(conf, 999.0), conf)
    inv_conf = torch.where(invalid, torch.zeros_like(conf), 1.0 / conf)
    pts = torch.stack([pts[..., 0] / 2.0, pts[..., 1] / 2.0, (pts[..., 2] + 1) / 2], dim=-1)
    return dict(pts3d=pts, conf=inv_conf, inverse_depthmap=inv_depth, raymap=raymap, crossmap=crossmap, invalid=invalid)
