"""Latent diffusion glue — the slice of ``lvdm.models.ddpm3d`` that inference touches, rebuilt around
the HIP U-Net and VAE: noise schedule (register_schedule, ddpm3d.py:162-225), v-parameterisation helpers (:278-290),
dynamic-rescale table (:585-590), ``apply_model`` (:1002-1017), ``DiffusionWrapper`` 'hybrid' (:2529-2544) and
``decode_first_stage`` / ``decode_core`` (:802-823, 935-936). Class and attribute names follow the reference so that
``scripts/evaluation/test_geo4d.py`` style code (``model.model.diffusion_model.out_channels``, ``model.scale_factor``,
``model.perframe_ae``, ``model.decode_first_stage`` ...) works unchanged. The ~2400 training / logging lines of the
reference class are out of scope.

The conditioning encoders of SURVEY.md §8(f) N3 — OpenCLIP text / image towers and the Resampler (geo4d_amd/encoders.py), next
to the VAE encode — are built LAZILY on first use of ``cond_stage_model`` / ``embedder`` / ``image_proj_model`` (or by
``build_frontend()``): together they are ~1 B parameters that the denoise + decode hot path never touches, and callers that
inject precomputed ``cond`` tensors (bench.py, run_clip) never pay for them.
"""
import numpy as np
import torch
import torch.nn as nn

from .registry import instantiate_from_config


class DiffusionWrapper(nn.Module):
    def __init__(self, diff_model_config, conditioning_key):
        super().__init__()
        self.diffusion_model = instantiate_from_config(diff_model_config)
        self.conditioning_key = conditioning_key

    def forward(self, x, t, c_concat=None, c_crossattn=None, **kwargs):
        if self.conditioning_key != "hybrid":
            raise NotImplementedError(f"conditioning_key={self.conditioning_key!r}: Geo4D inference uses 'hybrid' only")
        cc = c_crossattn[0] if len(c_crossattn) == 1 else torch.cat(c_crossattn, 1)
        extra = c_concat[0] if len(c_concat) == 1 else torch.cat(c_concat, 1)
        # the channel concat of ddpm3d.py:2542 happens inside the token-layout kernel (two sources)
        return self.diffusion_model(x, t, context=cc, c_concat=extra, **kwargs)

    def prepare(self, c_crossattn, T):
        """Project (or refresh in place) the cross-attention K/V of this conditioning outside a captured step; returns the
        U-Net's validity token (see UNetModel.prepare_context)."""
        cc = c_crossattn[0] if len(c_crossattn) == 1 else torch.cat(c_crossattn, 1)
        return self.diffusion_model.prepare_context(cc, T)


def _beta_schedule(n, linear_start, linear_end, zero_snr):
    betas = (torch.linspace(linear_start ** 0.5, linear_end ** 0.5, n, dtype=torch.float64, device="cpu") ** 2).numpy()
    if zero_snr:  # utils_diffusion.py:112-144
        s = np.sqrt(np.cumprod(1.0 - betas, axis=0))
        s0, sT = s[0].copy(), s[-1].copy()
        s = (s - sT) * (s0 / (s0 - sT))
        abar = s ** 2
        betas = 1 - np.concatenate([abar[0:1], abar[1:] / abar[:-1]])
    return betas


class LatentDiffusion(nn.Module):
    def __init__(self, unet_config, first_stage_config=None, cond_stage_config=None, timesteps=1000, beta_schedule="linear",
                 linear_start=1e-4, linear_end=2e-2, parameterization="eps", conditioning_key=None,
                 rescale_betas_zero_snr=False, scale_factor=1.0, scale_by_std=False, use_dynamic_rescale=False,
                 base_scale=0.7, turning_step=400, perframe_ae=False, encoder_type="2d", modality="rgb",
                 uncond_type="empty_seq", channels=3, image_size=256, first_stage_key="image", cond_stage_key="caption",
                 use_ema=False, **ignored):
        super().__init__()
        if beta_schedule != "linear":
            raise NotImplementedError("only the 'linear' beta schedule of the Geo4D config is built")
        self.parameterization = parameterization
        self.model = DiffusionWrapper(unet_config, conditioning_key or "crossattn")
        self.first_stage_model = instantiate_from_config(first_stage_config) if first_stage_config is not None else None
        self.cond_stage_config = cond_stage_config      # front-end modules are built lazily: see __getattr__ / build_frontend
        self.scale_factor, self.scale_by_std = scale_factor, scale_by_std
        self.use_dynamic_rescale, self.perframe_ae, self.encoder_type = use_dynamic_rescale, perframe_ae, encoder_type
        self.modality, self.uncond_type, self.channels, self.image_size = modality, uncond_type, channels, image_size
        self.first_stage_key, self.cond_stage_key = first_stage_key, cond_stage_key
        self.cross_attention = False      # LatentVisualDiffusion default (ddpm3d.py:1333): conditioning image is zeroed
        self.temporal_length = unet_config["params"].get("temporal_length")
        betas = _beta_schedule(timesteps, linear_start, linear_end, rescale_betas_zero_snr)
        ac = np.cumprod(1.0 - betas, axis=0)
        f32 = lambda a: torch.tensor(a, dtype=torch.float32, device="cpu")
        self.num_timesteps = int(timesteps)
        self.register_buffer("betas", f32(betas))
        self.register_buffer("alphas_cumprod", f32(ac))
        self.register_buffer("alphas_cumprod_prev", f32(np.append(1.0, ac[:-1])))
        self.register_buffer("sqrt_alphas_cumprod", f32(np.sqrt(ac)))
        self.register_buffer("sqrt_one_minus_alphas_cumprod", f32(np.sqrt(1.0 - ac)))
        if use_dynamic_rescale:
            arr = np.concatenate((np.linspace(1.0, base_scale, turning_step), np.full(self.num_timesteps, base_scale)))
            self.register_buffer("scale_arr", f32(arr))
        self.register_load_state_dict_post_hook(lambda mod, inc: mod._after_load(inc))

    @property
    def device(self):
        return self.betas.device

    # ---- v-parameterisation (ddpm3d.py:278-290) -------------------------------------------------------------------
    def _gather(self, a, t, x):
        return a.gather(-1, t).reshape(t.shape[0], *((1,) * (x.dim() - 1)))

    def predict_start_from_z_and_v(self, x_t, t, v):
        return self._gather(self.sqrt_alphas_cumprod, t, x_t) * x_t - self._gather(self.sqrt_one_minus_alphas_cumprod, t, x_t) * v

    def predict_eps_from_z_and_v(self, x_t, t, v):
        return self._gather(self.sqrt_alphas_cumprod, t, x_t) * v + self._gather(self.sqrt_one_minus_alphas_cumprod, t, x_t) * x_t

    def q_sample(self, x_start, t, noise=None):
        """ddpm3d.py:344-355: the forward process at step t (used by the samplers' mask / x0 blending)."""
        noise = torch.randn_like(x_start) if noise is None else noise
        return self._gather(self.sqrt_alphas_cumprod, t, x_start) * x_start + self._gather(self.sqrt_one_minus_alphas_cumprod, t, x_start) * noise

    # ---- denoiser (ddpm3d.py:1002-1017) -------------------------------------------------------------------------------
    def apply_model(self, x_noisy, t, cond, **kwargs):
        if not isinstance(cond, dict):
            cond = {"c_crossattn": cond if isinstance(cond, list) else [cond]}
        out = self.model(x_noisy, t, **cond, **kwargs)
        return out[0] if isinstance(out, tuple) else out

    def prepare_conditioning(self, cond, T):
        """Hook used by the samplers before replaying a captured step (geo4d_amd/ddim.py)."""
        if not isinstance(cond, dict):
            cond = {"c_crossattn": cond if isinstance(cond, list) else [cond]}
        return self.model.prepare(cond["c_crossattn"], T)

    # ---- first stage (ddpm3d.py:802-870, 935-936) -----------------------------------------------------------------
    def _decode(self, fn, z):
        reshape_back = self.encoder_type == "2d" and z.dim() == 5
        if reshape_back:
            b, c, t, h, w = z.shape
            z = z.permute(0, 2, 1, 3, 4).reshape(b * t, c, h, w)
        out = fn(z * (1.0 / self.scale_factor))       # all frames in one batch: identical to the per-frame loop
        if reshape_back:
            out = out.reshape(b, t, *out.shape[1:]).permute(0, 2, 1, 3, 4)
        return out

    @torch.no_grad()
    def decode_first_stage(self, z, **kwargs):
        return self._decode(self.first_stage_model.decode, z)

    decode_core = differentiable_decode_first_stage = decode_first_stage

    @torch.no_grad()
    def decode_first_stage_confhead(self, z, **kwargs):
        return self._decode(self.first_stage_model.decode_with_conf_adaptor, z)

    decode_core_confhead = decode_first_stage_confhead

    # ---- conditioning front-end (N3): lazily built submodules with the reference's attribute names ---------------------------
    _FRONTEND = {"cond_stage_model": "cond_stage_config", "embedder": "img_cond_stage_config", "image_proj_model": "image_proj_stage_config"}

    def __getattr__(self, name):
        if name in LatentDiffusion._FRONTEND and "_modules" in self.__dict__ and name not in self._modules:
            try:
                self.build_frontend(only=name)
            except NotImplementedError as e:          # hasattr(model, "embedder") must answer False, not throw
                raise AttributeError(f"{type(self).__name__}.{name}: {e}") from None
            return self._modules[name]
        return super().__getattr__(name)

    def _frontend_configured(self, name):
        cfg = getattr(self, LatentDiffusion._FRONTEND[name], None)
        return isinstance(cfg, dict) and str(cfg.get("target", "")).startswith("geo4d_amd.")

    def build_frontend(self, only=None):
        """Instantiate ``cond_stage_model`` (text), ``embedder`` (image) and ``image_proj_model`` (Resampler) from their yaml
        sections, in the U-Net's compute mode, on this model's device. Idempotent."""
        unet = self.model.diffusion_model
        for name, cfg_attr in LatentDiffusion._FRONTEND.items():
            if (only is not None and name != only) or name in self._modules:
                continue
            cfg = getattr(self, cfg_attr, None)
            if not isinstance(cfg, dict) or not str(cfg.get("target", "")).startswith("geo4d_amd."):
                raise NotImplementedError(f"{name}: no geo4d_amd front-end configured ({cfg_attr} = {cfg!r}); pass precomputed `cond` tensors "
                                          "or point the yaml section at geo4d_amd.encoders.*")
            cfg = {"target": cfg["target"], "params": dict(cfg.get("params") or {}, compute_dtype=getattr(unet, "compute_dtype", None))}
            self._modules[name] = instantiate_from_config(cfg).to(self.device)     # (add_module would probe hasattr -> __getattr__ -> here)
            # a lazily built encoder holds its constructor's random weights until a state_dict that carries its tensors is loaded:
            # remembered, and said out loud when a context is computed from it (context_for / get_learned_conditioning)
            self.__dict__.setdefault("_frontend_unloaded", set()).add(name)
            self.__dict__.pop("_geo4d_context_cache", None)
        return self

    def _warn_unloaded(self, *names):
        bad = sorted(n for n in names if n in self.__dict__.get("_frontend_unloaded", ()))
        if bad:
            import warnings
            warnings.warn(f"geo4d_amd: {', '.join(bad)} built lazily and NO checkpoint weights loaded into it since: the cross-attention "
                          "context comes from randomly initialised encoders (call build_frontend() before load_state_dict, or use "
                          "load_reference_state_dict, which does)", RuntimeWarning, stacklevel=3)

    def _after_load(self, incompatible=None):
        """Bookkeeping after ANY load_state_dict: encoders whose tensors were all present count as loaded; cached contexts computed
        from the old weights are dropped (pipeline.image_guided_synthesis caches the step-constant context per model)."""
        missing = set(getattr(incompatible, "missing_keys", ()) or ())
        un = self.__dict__.setdefault("_frontend_unloaded", set())
        for name in list(un):
            mod = self._modules.get(name)
            if mod is not None and not any(k.startswith(name + ".") for k in missing):
                un.discard(name)
        self.__dict__.pop("_geo4d_context_cache", None)

    def get_learned_conditioning(self, c):
        """ddpm3d.py:640-651: prompts (or token ids) -> text context [B, 77, 1024]."""
        enc = self.cond_stage_model
        self._warn_unloaded("cond_stage_model")
        return enc.encode(c)

    def get_first_stage_encoding(self, encoder_posterior, noise=None):
        """ddpm3d.py:674-681: sample the posterior (or pass a tensor through) and apply scale_factor."""
        if isinstance(encoder_posterior, torch.Tensor):
            z = encoder_posterior
        elif hasattr(encoder_posterior, "sample"):
            z = encoder_posterior.sample(noise=noise)
        else:
            raise NotImplementedError(f"encoder_posterior of type '{type(encoder_posterior)}' not yet implemented")
        return self.scale_factor * z

    def _encode(self, fn, x):
        reshape_back = self.encoder_type == "2d" and x.dim() == 5
        if reshape_back:
            b, c, t, h, w = x.shape
            x = x.permute(0, 2, 1, 3, 4).reshape(b * t, c, h, w)
        post = fn(x)                                   # all frames in one batch through the HIP encoder
        if self.perframe_ae:
            # the reference loops over frames and samples each posterior separately (ddpm3d.py:696-702): draw the noise
            # frame by frame in that order so a seeded run consumes the CPU generator exactly as the reference does
            noise = torch.cat([torch.randn((1,) + tuple(post.mean.shape[1:])) for _ in range(post.mean.shape[0])], 0)
        else:
            noise = None                               # one torch.randn of the whole batch (distributions.py:35-40)
        z = self.get_first_stage_encoding(post, noise=noise).detach()
        if reshape_back:
            z = z.reshape(b, t, *z.shape[1:]).permute(0, 2, 1, 3, 4)
        return z

    @torch.no_grad()
    def encode_first_stage(self, x):
        """ddpm3d.py:683-707: frames [b,3,t,H,W] (or [n,3,H,W]) in [-1,1] -> scale_factor * sampled latent [b,4,t,H/8,W/8]."""
        return self._encode(self.first_stage_model.encode, x)

    @torch.no_grad()
    def encode_first_stage_adaptor(self, x):
        """ddpm3d.py:775-798: the same through encoder_adaptor."""
        return self._encode(self.first_stage_model.encode_with_adaptor, x)

    @torch.no_grad()
    def context_for(self, prompts, image=None, frames=None):
        """The cross-attention context of test_geo4d.py:118-158 for modality 'pc_ray_cross_depth': text tokens (77) followed by
        the Resampler's image tokens (16 per frame). ``cross_attention`` False (the shipped setting): ONE zero image per sample ->
        [B, 77 + 16*T, 1024]; True: every frame of ``frames`` [b, c, t, h, w] is embedded."""
        self.embedder, self.image_proj_model             # build both before the check below
        self._warn_unloaded("embedder", "image_proj_model")
        if self.cross_attention:
            b, c, t, h, w = frames.shape
            emb = self.embedder(frames.permute(0, 2, 1, 3, 4).reshape(b * t, c, h, w))
            img_emb = self.image_proj_model(emb.reshape(b, t, emb.shape[1], emb.shape[2]))
        else:
            img_emb = self.image_proj_model(self.embedder(torch.zeros_like(image)))
        return torch.cat([self.get_learned_conditioning(prompts).to(img_emb.device), img_emb], dim=1)

    # ---- checkpoints (test_geo4d.py:54-81): reference keys model.diffusion_model.* / first_stage_model.* ---------
    def load_reference_state_dict(self, state_dict):
        sd = state_dict.get("state_dict", state_dict)
        # the conditioning encoders are built lazily: a checkpoint that carries their tensors gets them built FIRST (when the yaml
        # points those sections at geo4d_amd.encoders.*), so its cond_stage_model.* / embedder.* / image_proj_model.* tensors are
        # loaded instead of being reported as skipped and replaced by random weights at first use
        for name in LatentDiffusion._FRONTEND:
            if name not in self._modules and self._frontend_configured(name) and any(k.startswith(name + ".") for k in sd):
                self.build_frontend(only=name)
        own = self.state_dict()
        picked = {k: v for k, v in sd.items() if k in own}
        missing = [k for k in own if k not in picked]
        if missing:
            raise KeyError(f"checkpoint lacks {len(missing)} tensors of the hot path, e.g. {missing[:3]}")
        skipped = sorted({k.split(".")[0] for k in sd if k not in own})
        self.load_state_dict(picked, strict=True)     # (the post hook marks the encoders loaded and drops cached contexts)
        return skipped   # e.g. ['cond_stage_model', 'embedder', 'image_proj_model']: N3 components


class LatentVisualDiffusion(LatentDiffusion):
    """yaml ``model.target`` (configs/inference_geo4d.yaml:40); img_cond_stage_config / image_proj_stage_config are N3."""

    def __init__(self, img_cond_stage_config=None, image_proj_stage_config=None, freeze_embedder=True,
                 image_proj_model_trainable=True, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.img_cond_stage_config, self.image_proj_stage_config = img_cond_stage_config, image_proj_stage_config
