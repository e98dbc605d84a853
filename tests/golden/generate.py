#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by running the REFERENCE ITSELF (/root/reference) on CPU.

Run in the build container only (`python tests/golden/generate.py`); the GPU box has no /root/reference and only
reads the committed .pt files. What is imported from the reference, unmodified:
  lvdm.models.ddpm3d.LatentDiffusion (schedule, v-param helpers, apply_model, DiffusionWrapper 'hybrid',
  decode_first_stage), lvdm.modules.networks.openaimodel3d.UNetModel, lvdm.models.samplers.ddim.DDIMSampler,
  lvdm.models.autoencoder.AutoencoderKL (+ Decoder, VAEDecoderadaptor), utils.utils.instantiate_from_config,
  and — because scripts/evaluation/test_geo4d.py cannot be imported (decord / omegaconf / pytorch3d absent) — the
  SOURCE TEXT of its window loop (:417-423), mask helpers (:84-89, :276-287) and post-decode block (:447-503),
  exec'd verbatim from the file at generation time (never copied into this repo).
Missing third-party modules are stubbed: cv2 (unused on this path), pytorch_lightning.LightningModule -> nn.Module,
torchvision.utils.make_grid (logging only). Weights: oracle.params (name-keyed seeded fill), so fixtures hold only
inputs and outputs.

Modes (each writes its own file and leaves the others untouched):
  (no argument)  schedule.pt, unet_tiny.pt, unet_full.pt, vae_tiny.pt, ddim_tiny.pt, glue.pt   - the §8(a)-(e) path
  encode         vae_encode_tiny.pt   AutoencoderKL.encode / encode_with_adaptor, LatentDiffusion.encode_first_stage (seeded)
  rays           rays.pt              raymap_to_camera_matrix -> utils/rays.py cameras_from_plucker (pytorch3d's PerspectiveCameras
                                      stubbed as a plain container of R / T / focal_length)
  timesteps      timesteps.pt         make_ddim_timesteps for every spacing method x 16 step counts
  ddim_eta       ddim_eta_tiny.pt     DDIMSampler at eta = 1 with a seeded CPU noise stream
  ddim_cfg       ddim_cfg_tiny.pt     2-way (ddim.py) and 3-way (ddim_multiplecond.py) classifier-free guidance + guidance_rescale
"""
import ast
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)
sys.path.insert(0, REF)

from oracle.params import fill_module_  # noqa: E402


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


class _LM(nn.Module):
    @property
    def device(self):
        return next(self.parameters()).device


_stub("cv2")
_pl = _stub("pytorch_lightning", LightningModule=_LM)
_pl.utilities = _stub("pytorch_lightning.utilities", rank_zero_only=lambda f: f)
_tv = _stub("torchvision")
_tv.utils = _stub("torchvision.utils", make_grid=None)


class AD(dict):
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)


def ad(x):
    if isinstance(x, dict):
        return AD({k: ad(v) for k, v in x.items()})
    if isinstance(x, list):
        return [ad(v) for v in x]
    return x


def tiny_configs(mc=64, ctx=128, vae_ch=64):
    dd = dict(double_z=True, z_channels=4, resolution=256, in_channels=3, out_ch=3, ch=vae_ch, ch_mult=[1, 2, 4, 4],
              num_res_blocks=2, attn_resolutions=[], dropout=0.0)
    adp = dict(double_z=True, z_channels=4, resolution=256, in_channels=3, out_ch=1, ch=vae_ch, ch_mult=[1],
               num_res_blocks=1, attn_resolutions=[], dropout=0.0)
    unet = dict(in_channels=20, out_channels=16, model_channels=mc, attention_resolutions=[4, 2, 1], num_res_blocks=2,
                channel_mult=[1, 2, 4, 4], dropout=0.1, num_head_channels=64, transformer_depth=1, context_dim=ctx,
                use_linear=True, use_checkpoint=False, temporal_conv=True, temporal_attention=True,
                temporal_selfatt_only=True, use_relative_position=False, use_causal_attention=False, temporal_length=16,
                addition_attention=True, image_cross_attention=True, default_fs=24, fs_condition=True)
    return unet, dd, adp


def build_reference(unet, dd, adp):
    from lvdm.models.ddpm3d import LatentDiffusion
    fs = dict(target="lvdm.models.autoencoder.AutoencoderKL",
              params=dict(embed_dim=4, monitor="val/rec_loss", ddconfig=dd, lossconfig=dict(target="torch.nn.Identity"),
                          adaptorconfig=adp))
    m = LatentDiffusion(
        first_stage_config=ad(fs), cond_stage_config=ad(dict(target="torch.nn.Identity")),
        unet_config=ad(dict(target="lvdm.modules.networks.openaimodel3d.UNetModel", params=unet)),
        rescale_betas_zero_snr=True, parameterization="v", linear_start=0.00085, linear_end=0.012, num_timesteps_cond=1,
        timesteps=1000, modality="pc_ray_cross_depth", first_stage_key="normed_allpts", cond_stage_key="video",
        cond_stage_trainable=False, conditioning_key="hybrid", image_size=[32, 64], channels=16, scale_by_std=False,
        scale_factor=0.18215, use_ema=False, uncond_type="empty_seq", use_dynamic_rescale=True, base_scale=0.7,
        fps_condition_type="fps", perframe_ae=True)
    m.eval()
    fill_module_(m.model.diffusion_model)
    fill_module_(m.first_stage_model)
    return m


def randn(shape, seed):
    return torch.randn(shape, generator=torch.Generator().manual_seed(seed))


def script_source(first, last):
    with open(os.path.join(REF, "scripts/evaluation/test_geo4d.py")) as f:
        lines = [l.rstrip("\r\n") + "\n" for l in f.readlines()]   # the script has CRLF line endings
    return lines[first - 1:last]


def script_functions(names):
    with open(os.path.join(REF, "scripts/evaluation/test_geo4d.py")) as f:
        src = f.read().replace("\r\n", "\n")
    tree = ast.parse(src)
    ns = {"torch": torch}
    for node in tree.body:
        if isinstance(node, ast.FunctionDef) and node.name in names:
            exec(compile(ast.Module([node], []), "test_geo4d.py", "exec"), ns)
    return ns


def encode_fixtures():
    """`python tests/golden/generate.py encode` -> vae_encode_tiny.pt: the reference AutoencoderKL.encode /
    encode_with_adaptor moments and LatentDiffusion.encode_first_stage (seeded posterior sampling) on a tiny config."""
    torch.manual_seed(0)
    torch.set_num_threads(8)
    unet_cfg, dd, adp = tiny_configs()
    model = build_reference(unet_cfg, dd, adp)
    vae = model.first_stage_model
    x = torch.rand((2, 3, 48, 64), generator=torch.Generator().manual_seed(310)) * 2 - 1
    video = torch.rand((1, 3, 3, 48, 64), generator=torch.Generator().manual_seed(311)) * 2 - 1
    with torch.no_grad():
        mom = vae.encode(x).parameters
        mom_a = vae.encode_with_adaptor(x).parameters
        torch.manual_seed(4242)
        z = model.encode_first_stage(video)          # perframe_ae=True: one CPU torch.randn per frame
    print("encode", tuple(mom.shape), tuple(mom_a.shape), tuple(z.shape), float(mom.std()), float((mom_a - mom).abs().max()))
    torch.save(dict(ddconfig=dd, adaptorconfig=adp, x=x, moments=mom, moments_adaptor=mom_a, video=video, seed=4242,
                    scale_factor=model.scale_factor, perframe_ae=model.perframe_ae, z_first_stage=z,
                    shapes={k: tuple(v.shape) for k, v in vae.state_dict().items()}), os.path.join(HERE, "vae_encode_tiny.pt"))


def synthetic_ray_maps(T, H, W, seed, noise=0.01):
    """Plücker maps of a smooth synthetic camera path (directions through a pinhole, moments = c x d) + noise: the kind of
    input the decoder hands to raymap_to_camera_matrix, with a well-defined answer."""
    g = torch.Generator().manual_seed(seed)
    ys, xs = torch.meshgrid(torch.linspace(-0.5, 0.5, H), torch.linspace(-0.8, 0.8, W), indexing="ij")
    cam_dirs = torch.nn.functional.normalize(torch.stack([xs, ys, torch.ones_like(xs)], -1), dim=-1)       # [H, W, 3]
    rays, moms = [], []
    for t in range(T):
        ang = torch.tensor([0.05 * t, -0.08 * t, 0.03 * t])
        cx, cy, cz = torch.cos(ang); sx, sy, sz = torch.sin(ang)
        Rx = torch.tensor([[1, 0, 0], [0, cx, -sx], [0, sx, cx]]); Ry = torch.tensor([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
        Rz = torch.tensor([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]])
        R = (Rz @ Ry @ Rx).float()
        c = torch.tensor([0.1 * t, 0.02 * t * t, -0.05 * t])
        d = cam_dirs @ R.T
        m = torch.cross(c.expand_as(d), d, dim=-1)
        rays.append(d * (1.0 + 0.3 * torch.rand((H, W, 1), generator=g)) + noise * torch.randn(d.shape, generator=g))   # un-normalised, like a decoder output
        moms.append(m + noise * torch.randn(d.shape, generator=g))
    to = lambda L: torch.stack(L).permute(3, 0, 1, 2)[None].contiguous()      # [1, 3, T, H, W]
    return to(rays), to(moms)


def rays_fixtures():
    """`python tests/golden/generate.py rays` -> rays.pt: the reference's raymap_to_camera_matrix (test_geo4d.py:539-557 ->
    utils/rays.py cameras_from_plucker) on synthetic Plücker maps. pytorch3d is absent here; its PerspectiveCameras is used by
    that code only as a container of R / T / focal_length, so it is stubbed as one."""
    for name in ("ipdb",):
        sys.modules.setdefault(name, types.ModuleType(name))
    p3, rend, tr = types.ModuleType("pytorch3d"), types.ModuleType("pytorch3d.renderer"), types.ModuleType("pytorch3d.transforms")

    class PerspectiveCameras:
        def __init__(self, focal_length=None, R=None, T=None, device=None, **kw):
            n = len(focal_length)
            self.focal_length = focal_length
            self.R = torch.eye(3).repeat(n, 1, 1) if R is None else R
            self.T = torch.zeros(n, 3) if T is None else T

        def __len__(self):
            return self.R.shape[0]

        def clone(self):
            return PerspectiveCameras(self.focal_length, self.R.clone(), self.T.clone())
    rend.PerspectiveCameras, rend.RayBundle = PerspectiveCameras, object
    tr.Rotate = tr.Translate = object
    sys.modules.update({"pytorch3d": p3, "pytorch3d.renderer": rend, "pytorch3d.transforms": tr})
    from utils.rays import cameras_from_plucker
    ns = script_functions({"raymap_to_camera_matrix"})
    ns["cameras_from_plucker"] = cameras_from_plucker
    cases = {}
    for name, (T, H, W, seed) in {"wide_4x16x24": (4, 16, 24, 1), "tall_3x24x16": (3, 24, 16, 2), "wide_16x8x20": (16, 8, 20, 3)}.items():   # H == W raises UnboundLocalError in the reference (rays.py:399-417)
        ray, mom = synthetic_ray_maps(T, H, W, seed)
        with torch.no_grad():
            P = ns["raymap_to_camera_matrix"](ray, mom)
        print(name, tuple(P.shape), P[-1, :3, 3].tolist())
        cases[name] = dict(raymap=ray, crossmap=mom, P_c2w=P)
    torch.save(cases, os.path.join(HERE, "rays.pt"))


def timestep_fixtures():
    """`python tests/golden/generate.py timesteps` -> timesteps.pt: the reference's make_ddim_timesteps for every spacing
    method it implements (utils_diffusion.py:56-78) over a sweep of step counts - the integer tables of SURVEY.md §8 (a1)."""
    from lvdm.models.utils_diffusion import make_ddim_timesteps
    out = {}
    for method in ("uniform", "quad", "uniform_trailing"):
        for S in (1, 2, 3, 5, 7, 10, 20, 25, 50, 100, 200, 250, 333, 500, 999, 1000):
            out[f"{method}/{S}"] = torch.from_numpy(np.asarray(make_ddim_timesteps(method, S, 1000, verbose=False)).astype(np.int64))
    torch.save(out, os.path.join(HERE, "timesteps.pt"))
    print(len(out), "tables")


def ddim_eta_fixtures():
    """`python tests/golden/generate.py ddim_eta` -> ddim_eta_tiny.pt: the reference DDIMSampler at eta = 1 (the "better visual
    results" setting of scripts/infer_geo4d.sh) on the tiny LatentDiffusion. The per-step noise comes from torch.randn on the
    model's device (ddim.py:271, noise_like) = the CPU global generator here, seeded right before sample()."""
    torch.manual_seed(0)
    torch.set_num_threads(8)
    from lvdm.models.samplers.ddim import DDIMSampler

    class CpuSampler(DDIMSampler):  # ddim.py:18-22 hard-codes torch.device("cuda")
        def register_buffer(self, name, attr):
            setattr(self, name, attr)
    unet_cfg, dd, adp = tiny_configs()
    model = build_reference(unet_cfg, dd, adp)
    B, T, h, w = 1, 16, 8, 8
    x_T = randn((B, 16, T, h, w), 500)
    cond = {"c_crossattn": [randn((B, 77 + 16 * T, unet_cfg["context_dim"]), 501)], "c_concat": [randn((B, 4, T, h, w), 502)]}
    fs = torch.tensor([24])
    with torch.no_grad():
        torch.manual_seed(777)
        samples, _ = CpuSampler(model).sample(S=5, conditioning=cond, batch_size=B, shape=[16, T, h, w], verbose=False,
                                              unconditional_guidance_scale=1.0, unconditional_conditioning=None, eta=1.0, cfg_img=None,
                                              mask=None, x0=None, fs=fs, x_T=x_T, timestep_spacing="uniform_trailing",
                                              guidance_rescale=0.7, unconditional_conditioning_img_nonetext=None)
    print("ddim eta=1", float(samples.std()))
    torch.save(dict(unet_config=unet_cfg, x_T=x_T, context=cond["c_crossattn"][0], c_concat=cond["c_concat"][0], fs=fs, S=5, eta=1.0,
                    seed=777, samples=samples), os.path.join(HERE, "ddim_eta_tiny.pt"))


def ddim_cfg_fixtures():
    """`python tests/golden/generate.py ddim_cfg` -> ddim_cfg_tiny.pt: the reference samplers with classifier-free guidance on:
    lvdm.models.samplers.ddim.DDIMSampler (2-way, scale 7.5, guidance_rescale 0.7) and
    lvdm.models.samplers.ddim_multiplecond.DDIMSampler (3-way, scale 7.5, cfg_img 2.0) on the tiny LatentDiffusion."""
    torch.manual_seed(0)
    torch.set_num_threads(8)
    from lvdm.models.samplers.ddim import DDIMSampler
    from lvdm.models.samplers.ddim_multiplecond import DDIMSampler as DDIMSamplerMulti

    def cpu(cls):
        class Cpu(cls):  # ddim.py:18-22 hard-codes torch.device("cuda")
            def register_buffer(self, name, attr):
                setattr(self, name, attr)
        return Cpu
    unet_cfg, dd, adp = tiny_configs()
    model = build_reference(unet_cfg, dd, adp)
    B, T, h, w = 1, 4, 8, 8
    x_T = randn((B, 16, T, h, w), 600)
    zc = randn((B, 4, T, h, w), 601)
    ctx = [randn((B, 77 + 16 * T, unet_cfg["context_dim"]), 602 + i) for i in range(3)]      # cond, uncond, image-yes / text-""
    mk = lambda c: {"c_crossattn": [c], "c_concat": [zc]}
    fs = torch.tensor([24])
    common = dict(S=3, conditioning=mk(ctx[0]), batch_size=B, shape=[16, T, h, w], verbose=False, eta=0.0, mask=None, x0=None, fs=fs,
                  x_T=x_T, timestep_spacing="uniform_trailing", guidance_rescale=0.7, unconditional_guidance_scale=7.5,
                  unconditional_conditioning=mk(ctx[1]))
    with torch.no_grad():
        two, _ = cpu(DDIMSampler)(model).sample(cfg_img=None, unconditional_conditioning_img_nonetext=None, **common)
        three, _ = cpu(DDIMSamplerMulti)(model).sample(cfg_img=2.0, unconditional_conditioning_img_nonetext=mk(ctx[2]), **common)
    print("cfg", float(two.std()), float(three.std()), float((two - three).abs().max()))
    torch.save(dict(unet_config=unet_cfg, x_T=x_T, c_concat=zc, contexts=ctx, fs=fs, S=3, scale=7.5, cfg_img=2.0, guidance_rescale=0.7,
                    samples_2way=two, samples_3way=three), os.path.join(HERE, "ddim_cfg_tiny.pt"))


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "ddim_cfg":
        return ddim_cfg_fixtures()
    if len(sys.argv) > 1 and sys.argv[1] == "ddim_eta":
        return ddim_eta_fixtures()
    if len(sys.argv) > 1 and sys.argv[1] == "timesteps":
        return timestep_fixtures()
    if len(sys.argv) > 1 and sys.argv[1] == "encode":
        return encode_fixtures()
    if len(sys.argv) > 1 and sys.argv[1] == "rays":
        return rays_fixtures()
    torch.manual_seed(0)
    torch.set_num_threads(8)
    from lvdm.models.samplers.ddim import DDIMSampler

    class CpuSampler(DDIMSampler):  # ddim.py:18-22 hard-codes torch.device("cuda")
        def register_buffer(self, name, attr):
            setattr(self, name, attr)

    unet_cfg, dd, adp = tiny_configs()
    model = build_reference(unet_cfg, dd, adp)
    out = {}

    # ---- 1. schedule tables ------------------------------------------------------------------------------
    sched = dict(alphas_cumprod=model.alphas_cumprod.clone(), betas=model.betas.clone(),
                 sqrt_alphas_cumprod=model.sqrt_alphas_cumprod.clone(),
                 sqrt_one_minus_alphas_cumprod=model.sqrt_one_minus_alphas_cumprod.clone(),
                 alphas_cumprod_prev=model.alphas_cumprod_prev.clone(), scale_arr=model.scale_arr.clone())
    for S in (5, 50):
        s = CpuSampler(model)
        s.make_schedule(ddim_num_steps=S, ddim_discretize="uniform_trailing", ddim_eta=0.0, verbose=False)
        sched[f"ddim_timesteps_{S}"] = torch.from_numpy(np.ascontiguousarray(s.ddim_timesteps)).long()
        sched[f"ddim_alphas_{S}"] = torch.as_tensor(np.asarray(s.ddim_alphas), dtype=torch.float32)
        sched[f"ddim_alphas_prev_{S}"] = torch.as_tensor(np.asarray(s.ddim_alphas_prev, dtype=np.float64), dtype=torch.float64)
        sched[f"ddim_scale_arr_{S}"] = s.ddim_scale_arr.clone()
        sched[f"ddim_scale_arr_prev_{S}"] = s.ddim_scale_arr_prev.clone()
    torch.save(sched, os.path.join(HERE, "schedule.pt"))

    # ---- 2. U-Net forward (reference UNetModel through DiffusionWrapper 'hybrid') -------------------------
    cases = {}
    for name, (B, T, h, w) in {"t16_8x8": (1, 16, 8, 8), "b2_t5_8x16": (2, 5, 8, 16)}.items():
        x = randn((B, 16, T, h, w), 100)
        cc = randn((B, 4, T, h, w), 101)
        ctx = randn((B, 77 + 16 * T, unet_cfg["context_dim"]), 102)
        t = torch.tensor([999, 419][:B], dtype=torch.long)
        fs = torch.tensor([24, 8][:B], dtype=torch.long)
        with torch.no_grad():
            y = model.apply_model(x, t, {"c_crossattn": [ctx], "c_concat": [cc]}, fs=fs, cfg_img=None,
                                  unconditional_conditioning_img_nonetext=None)
        cases[name] = dict(x=x, c_concat=cc, context=ctx, t=t, fs=fs, out=y)
        print("unet", name, tuple(y.shape), float(y.std()))
    shapes = {k: tuple(v.shape) for k, v in model.model.diffusion_model.state_dict().items()}
    torch.save(dict(unet_config=unet_cfg, cases=cases, shapes=shapes), os.path.join(HERE, "unet_tiny.pt"))

    # ---- 3. DDIM sampling (reference DDIMSampler driving the reference LatentDiffusion) ------------------
    B, T, h, w = 1, 16, 8, 8
    x_T = randn((B, 16, T, h, w), 200)
    cond = {"c_crossattn": [randn((B, 77 + 16 * T, unet_cfg["context_dim"]), 201)], "c_concat": [randn((B, 4, T, h, w), 202)]}
    fs = torch.tensor([24], dtype=torch.long)
    visited = []
    orig = model.apply_model

    def spy(x, t, c, **kw):
        visited.append((int(t[0]), sorted(kw.keys())))
        return orig(x, t, c, **kw)
    model.apply_model = spy
    with torch.no_grad():
        samples, _ = CpuSampler(model).sample(S=4, conditioning=cond, batch_size=B, shape=[16, T, h, w], verbose=False,
                                              unconditional_guidance_scale=1.0, unconditional_conditioning=None, eta=0.0,
                                              cfg_img=None, mask=None, x0=None, fs=fs, x_T=x_T,
                                              timestep_spacing="uniform_trailing", guidance_rescale=0.7,
                                              unconditional_conditioning_img_nonetext=None)
    model.apply_model = orig
    print("ddim visited", visited, float(samples.std()))
    ddim = dict(unet_config=unet_cfg, x_T=x_T, context=cond["c_crossattn"][0], c_concat=cond["c_concat"][0], fs=fs, S=4,
                samples=samples, visited_t=torch.tensor([v[0] for v in visited]), forwarded_kwargs=visited[0][1])

    # ---- 4. VAE decode paths (reference AutoencoderKL) + 4-modality decode of the sampled latent ----------
    vae = model.first_stage_model
    z = randn((2, 4, 6, 8), 300)
    with torch.no_grad():
        dec = vae.decode(z)
        dec_conf = vae.decode_with_conf_adaptor(z)
        ray = model.decode_first_stage(samples[:, 4:8])                  # ddpm3d.py:802-823, perframe_ae=True
    print("vae", tuple(dec.shape), tuple(dec_conf.shape), tuple(ray.shape))
    torch.save(dict(ddconfig=dd, adaptorconfig=adp, z=z, decode=dec, decode_with_conf_adaptor=dec_conf,
                    shapes={k: tuple(v.shape) for k, v in vae.state_dict().items()}), os.path.join(HERE, "vae_tiny.pt"))
    ddim["decode_first_stage_4_8"] = ray
    torch.save(ddim, os.path.join(HERE, "ddim_tiny.pt"))

    # ---- 5. window indices + post-decode block: exec the script's own source lines ------------------------
    win_src = "".join(l[8:] if l.startswith("        ") else l for l in script_source(417, 422))
    windows = {}
    for Tn in (16, 17, 19, 20, 50, 64, 128):
        for stride in (4, 3):
            ns = {"T": Tn, "args": types.SimpleNamespace(stride=stride)}
            exec(win_src, ns)
            windows[(Tn, stride)] = [(s.start, s.stop) for s in ns["slice_list"]]
    print("windows T=64:", len(windows[(64, 4)]), windows[(64, 4)][-3:])

    fns = script_functions({"get_sky_mask", "get_far_away_mask", "denormalize_pc_bbox2"})
    from einops import rearrange
    post_src = "".join(l[12:] if l.startswith("            ") else l for l in script_source(447, 503))
    Tn, H, W = 16, 12, 10
    bs = randn((1, 11, Tn, H, W), 400) * 1.2
    bs[0, 0:3, 2, 3:6, 2:5] = 1.05   # sky pixels
    bs[0, 0, 5, 0:2, 0:2] = 2.5      # far pixels
    ns = dict(fns)
    ns.update(torch=torch, rearrange=rearrange, batch_samples=bs.clone(), model=types.SimpleNamespace(modality="pc_ray_cross_depth"),
              use_raymap=False, use_crossmap=False, use_inverse_depthmap=True, use_traj=True, pointmap_vae=object(),
              raymap_to_camera_matrix=lambda r, c: None, pnt_valid_mask=torch.ones((Tn, H, W, 1)) > 0, sl=slice(0, Tn, 1),
              pred_list=[])
    exec(post_src, ns)
    pred = ns["pred_list"][0]
    post = dict(batch_samples=bs, pts3d=pred["pts3d"], conf=pred["conf"], inverse_depthmap=pred["inverse_depthmap"],
                pnt_valid_mask=ns["pnt_valid_mask"])
    print("post", {k: tuple(v.shape) for k, v in post.items()}, int((~ns["pnt_valid_mask"]).sum()))
    torch.save(dict(windows=windows, post=post), os.path.join(HERE, "glue.pt"))

    # ---- 6. the REAL yaml config: state_dict census + one forward at 8x8 latents ---------------------------------
    import yaml
    with open(os.path.join(REF, "configs/inference_geo4d.yaml")) as f:
        ycfg = yaml.safe_load(f)
    full_cfg = dict(ycfg["model"]["params"]["unet_config"]["params"])
    full_cfg["use_checkpoint"] = False   # test_geo4d.py:321-322
    from lvdm.modules.networks.openaimodel3d import UNetModel
    from lvdm.models.autoencoder import AutoencoderKL
    full = UNetModel(**full_cfg).eval()
    fill_module_(full)
    x = randn((1, 20, 16, 8, 8), 500)
    ctx = randn((1, 77 + 16 * 16, 1024), 501)
    with torch.no_grad():
        y = full(x, torch.tensor([639]), context=ctx, fs=torch.tensor([24]))
    print("unet full", tuple(y.shape), float(y.std()), sum(p.numel() for p in full.parameters()) / 1e6, "M params")
    vparams = ycfg["model"]["params"]["first_stage_config"]["params"]
    with torch.device("meta"):
        fvae = AutoencoderKL(**{k: (ad(v) if isinstance(v, dict) else v) for k, v in vparams.items()})
    torch.save(dict(unet_config=full_cfg, x=x, context=ctx, t=torch.tensor([639]), fs=torch.tensor([24]), out=y,
                    shapes={k: tuple(v.shape) for k, v in full.state_dict().items()},
                    vae_shapes={k: tuple(v.shape) for k, v in fvae.state_dict().items()},
                    ddconfig=vparams["ddconfig"], adaptorconfig=vparams["adaptorconfig"]),
               os.path.join(HERE, "unet_full.pt"))

    for f in sorted(os.listdir(HERE)):
        if f.endswith(".pt"):
            print(f, os.path.getsize(os.path.join(HERE, f)) // 1024, "KiB")


if __name__ == "__main__":
    main()
