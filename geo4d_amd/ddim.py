"""DDIM sampler — drop-in for ``lvdm.models.samplers.ddim.DDIMSampler`` (constructed by name in
``image_guided_synthesis``, test_geo4d.py:120) with the same ``sample(...)`` signature and return value.

What changes is how a step is executed. The reference's ``p_sample_ddim`` (ddim.py:206-279) creates half a dozen
``torch.full`` tensors and ~10 elementwise kernels per step on top of ~2000 eager launches of the U-Net. Here one step =
[gather t from a device table] + [U-Net kernels] + [one fused DDIM-update kernel reading its coefficients from a device
table indexed by a device-side step counter] + [counter -= 1]; nothing in it depends on host state, so the step is
captured ONCE into a hipGraph and replayed S times (``use_graph=True``, default when eta == 0; with classifier-free
guidance the 2 (or 3, ``multicond``) U-Net evaluations and their combination are part of the captured step).
"""
import os

import numpy as np
import torch

from . import ops


def make_ddim_timesteps(method, num_ddim, num_ddpm):
    """utils_diffusion.py:56-76 — integer table, must be bit-exact."""
    if method == "uniform":
        c = num_ddpm // num_ddim
        return np.asarray(list(range(0, num_ddpm, c))) + 1
    if method == "uniform_trailing":
        c = num_ddpm / num_ddim
        return np.flip(np.round(np.arange(num_ddpm, 0, -c))).astype(np.int64) - 1
    if method == "quad":
        return ((np.linspace(0, np.sqrt(num_ddpm * .8), num_ddim)) ** 2).astype(int) + 1
    raise NotImplementedError(f'There is no ddim discretization method called "{method}"')


class DDIMSampler(object):
    multicond = False     # True in geo4d_amd.ddim_multiplecond.DDIMSampler (3-way guidance)

    def __init__(self, model, schedule="linear", use_graph=True, batch_cfg=None, **kwargs):
        self.model = model
        self.ddpm_num_timesteps = model.num_timesteps
        self.schedule = schedule
        self.use_graph = use_graph
        # classifier-free guidance: the 2 (3) U-Net evaluations of a step as ONE forward of batch 2B (3B) - the evaluations are independent, and at
        # batch B the U-Net's levels 1-3 cannot fill the chip (round 6: two windows per step take 1.64x the time of one, profiles/r06_window_batch.md;
        # the same holds for two conditionings of one window). GEO4D_CFG_BATCH=0 / batch_cfg=False: one forward per conditioning, as before.
        self.batch_cfg = (os.environ.get("GEO4D_CFG_BATCH", "1") != "0") if batch_cfg is None else bool(batch_cfg)
        self._static = None       # captured step + its static buffers (state, step counter, copies of the conditioning)
        self._graph_key = None

    # ---- schedule (ddim.py:24-57, utils_diffusion.py:79-91) -----------------------------------------------------
    def make_schedule(self, ddim_num_steps, ddim_discretize="uniform", ddim_eta=0., verbose=True):
        m = self.model
        self.ddim_timesteps = make_ddim_timesteps(ddim_discretize, ddim_num_steps, self.ddpm_num_timesteps)
        ac = m.alphas_cumprod.detach().float().cpu()
        assert ac.shape[0] == self.ddpm_num_timesteps, 'alphas have to be defined for each timestep'
        ts = self.ddim_timesteps
        alphas = ac.numpy()[ts]
        alphas_prev = np.asarray([ac.numpy()[0]] + ac.numpy()[ts[:-1]].tolist())
        sigmas = ddim_eta * np.sqrt((1 - alphas_prev) / (1 - alphas) * (1 - alphas / alphas_prev))
        self.ddim_alphas, self.ddim_alphas_prev, self.ddim_sigmas = alphas, alphas_prev, sigmas
        S = len(ts)
        if m.use_dynamic_rescale:
            sc = m.scale_arr.detach().float().cpu()[ts]
            self.ddim_scale_arr, self.ddim_scale_arr_prev = sc, torch.cat([sc[0:1], sc[:-1]])
            rescale = self.ddim_scale_arr_prev / self.ddim_scale_arr           # fp32 division, as ddim.py:262-266
        else:
            rescale = torch.ones(S)
        sa = m.sqrt_alphas_cumprod.detach().float().cpu()[ts]
        s1 = m.sqrt_one_minus_alphas_cumprod.detach().float().cpu()[ts]
        a_prev = torch.tensor(alphas_prev, dtype=torch.float32)                # torch.full(size, alphas_prev[index]) -> fp32
        sig = torch.tensor(sigmas, dtype=torch.float32)
        coef = torch.stack([sa, s1, rescale, a_prev.sqrt(), (1. - a_prev - sig ** 2).sqrt(), sig], dim=1).contiguous()
        dev = m.device
        table = torch.from_numpy(np.ascontiguousarray(ts)).to(torch.int64)
        old = getattr(self, "coef", None)
        if old is not None and old.shape == coef.shape and old.device == dev:
            self.coef.copy_(coef)              # in place: a captured step graph keeps reading these device tables
            self.ts_table.copy_(table)
        else:
            self.coef = coef.to(dev)           # [S, 6], row = ddim index
            self.ts_table = table.to(dev)
            self._static, self._graph_key = None, None

    # ---- public API (ddim.py:60-132) ------------------------------------------------------------------------------
    @staticmethod
    def _cond_signature(c):
        """Structure + shapes + dtypes of a conditioning dict (NOT its pointers: the captured step reads sampler-owned
        static copies, so a new window's tensors of the same shape reuse the graph)."""
        if not isinstance(c, dict):
            return None if c is None else ("tensor", tuple(c.shape), c.dtype)
        return tuple((k, tuple((tuple(t.shape), t.dtype) for t in (v if isinstance(v, (list, tuple)) else [v])))
                     for k, v in sorted(c.items()))

    @staticmethod
    def _clone_cond(c):
        if not isinstance(c, dict):
            return None if c is None else c.detach().clone()
        return {k: [t.detach().clone() for t in v] if isinstance(v, (list, tuple)) else v.detach().clone() for k, v in c.items()}

    @staticmethod
    def _copy_cond(dst, src):
        if dst is None:
            return
        if not isinstance(dst, dict):
            dst.copy_(src)
            return
        for k, v in dst.items():
            if isinstance(v, list):
                for d, t in zip(v, src[k]):
                    d.copy_(t)
            else:
                v.copy_(src[k])

    @staticmethod
    def _fill_cat(cat, parts, fs):
        """Refresh the batched conditioning buffers IN PLACE from the per-conditioning static copies (a captured step keeps their pointers)."""
        B = next(iter(parts[0].values()))[0].shape[0]
        for k, v in cat["cond"].items():
            for j, dst in enumerate(v):
                for r, c in enumerate(parts):
                    dst[r * B:(r + 1) * B].copy_(c[k][j])
        if cat["fs"] is not None and fs is not None:
            cat["fs"].copy_(fs.repeat(len(parts)))

    def _prepare(self, conds, T):
        """Project / refresh the cross-attention K/V of every conditioning OUTSIDE the captured step and return the model's
        validity token (weights generation + K/V buffer identity). Models without the hook (stubs) return None."""
        prep = getattr(self.model, "prepare_conditioning", None)
        if prep is None:
            return None
        return tuple(prep(c, T) for c in conds if c is not None)

    @torch.no_grad()
    def sample(self, S, batch_size, shape, conditioning=None, callback=None, normals_sequence=None, img_callback=None,
               quantize_x0=False, eta=0., mask=None, x0=None, temperature=1., noise_dropout=0., score_corrector=None,
               corrector_kwargs=None, verbose=True, schedule_verbose=False, x_T=None, log_every_t=100,
               unconditional_guidance_scale=1., unconditional_conditioning=None, precision=None, fs=None,
               timestep_spacing='uniform', guidance_rescale=0.0, **kwargs):
        """Same signature and return value as the reference. Two extra, optional keywords (popped, never forwarded to the
        model): ``noise_generator`` — a device ``torch.Generator`` for x_T (when not given) and the eta > 0 step noise, so a
        window's result does not depend on what ran before it; ``strict_rng`` — draw one ``randn`` per step even at eta == 0
        like ``noise_like`` in ddim.py:271 does (keeps the global RNG stream aligned with the reference across calls; forces
        the eager path). ``precision`` is accepted and ignored (the compute mode is a property of the model here)."""
        if score_corrector is not None or quantize_x0 or noise_dropout > 0.:
            raise NotImplementedError("score_corrector / quantize_x0 / noise_dropout are not used by Geo4D "
                                      "inference (test_geo4d.py:212-227) and have no HIP path")
        if (mask is None) != (x0 is None):
            raise ValueError("mask and x0 go together (ddim.py:174-175)")
        if self.model.parameterization != "v":
            raise NotImplementedError("only the v-parameterisation of configs/inference_geo4d.yaml:43 is built")
        gen = kwargs.pop("noise_generator", None)
        strict_rng = bool(kwargs.pop("strict_rng", False))
        self.make_schedule(ddim_num_steps=S, ddim_discretize=timestep_spacing, ddim_eta=eta, verbose=schedule_verbose)
        size = (batch_size,) + tuple(shape)
        dev = self.model.device
        cfg = not (unconditional_conditioning is None or unconditional_guidance_scale == 1.)
        total = len(self.ddim_timesteps)
        clean_cond = bool(kwargs.pop("clean_cond", False))
        graph_ok = (self.use_graph and eta == 0. and not strict_rng and callback is None and img_callback is None and total > 2
                    and mask is None)
        # 3-way guidance of ddim_multiplecond.py:229-234 (image yes / text "" as a third evaluation)
        uc_img = kwargs.get("unconditional_conditioning_img_nonetext") if (self.multicond and cfg) else None
        if self.multicond and cfg and uc_img is None:
            raise ValueError("DDIMSampler (multiple cond): unconditional_conditioning_img_nonetext is required when CFG is on")
        cfg_img = kwargs.get("cfg_img")
        cfg_img = unconditional_guidance_scale if cfg_img is None else cfg_img
        T_frames = size[2] if len(size) == 5 else 1
        # The captured step reads SAMPLER-OWNED static copies of x / t / fs / every conditioning tensor; a later call with the
        # same shapes (the next window of a clip) copies its values in, refreshes the context K/V in place and replays.
        sig = self._cond_signature
        key = (size, S, timestep_spacing, sig(conditioning),
               (sig(unconditional_conditioning), sig(uc_img), float(unconditional_guidance_scale), float(cfg_img),
                float(guidance_rescale)) if cfg else (),
               None if fs is None else (tuple(fs.shape), fs.dtype), tuple(sorted(kwargs)))
        st = self._static if (graph_ok and self._graph_key == key) else None
        if st is not None:
            self._copy_cond(st["cond"], conditioning)
            if cfg:
                self._copy_cond(st["uc"], unconditional_conditioning)
                self._copy_cond(st["uc_img"], uc_img)
            if fs is not None:
                st["fs"].copy_(fs)
            if st.get("cat") is not None:
                self._fill_cat(st["cat"], [c for c in (st["cond"], st["uc"], st["uc_img"]) if c is not None], st["fs"])
            if self._prepare([st["cat"]["cond"]] if st.get("cat") is not None else [st["cond"], st["uc"], st["uc_img"]], T_frames) != st["token"]:
                st = None                      # weights re-packed / K/V buffers replaced since the capture: capture again
        if st is None:
            own = graph_ok                       # eager runs use the caller's tensors directly
            st = {"g": None,
                  "img": torch.empty(size, device=dev, dtype=torch.float32),
                  "ts": torch.empty((batch_size,), dtype=torch.int64, device=dev),
                  "idx": torch.empty((1,), dtype=torch.int32, device=dev),
                  "cond": self._clone_cond(conditioning) if own else conditioning,
                  "uc": (self._clone_cond(unconditional_conditioning) if own else unconditional_conditioning) if cfg else None,
                  "uc_img": (self._clone_cond(uc_img) if own else uc_img) if cfg else None,
                  "fs": None if fs is None else (fs.detach().clone() if own else fs)}
            st["pred_x0"] = torch.empty_like(st["img"])
            st["cat"] = None
            parts = [c for c in (st["cond"], st["uc"], st["uc_img"]) if c is not None]
            if cfg and self.batch_cfg and all(isinstance(c, dict) and set(c) == set(parts[0]) for c in parts):
                st["cat"] = {"cond": {k: [torch.cat([c[k][j] for c in parts], 0) for j in range(len(v))] for k, v in parts[0].items()},
                             "fs": None if st["fs"] is None else st["fs"].repeat(len(parts)), "n": len(parts)}
        img, ts, idx, pred_x0 = st["img"], st["ts"], st["idx"], st["pred_x0"]
        c_cond, c_uc, c_img, c_fs = st["cond"], st["uc"], st["uc_img"], st["fs"]
        cat = st.get("cat")
        prep_list = [cat["cond"]] if cat is not None else [c_cond, c_uc, c_img]
        mkw = dict(kwargs)
        if "unconditional_conditioning_img_nonetext" in mkw and c_img is not None:
            mkw["unconditional_conditioning_img_nonetext"] = c_img   # forwarded (and ignored) like ddim.py:217 does
        img.copy_(torch.randn(size, device=dev, generator=gen) if x_T is None else x_T.to(dev).float())
        idx.fill_(total - 1)
        intermediates = {'x_inter': [img.clone()], 'pred_x0': [img.clone()]}

        def model_out():
            if not cfg:
                return self.model.apply_model(img, ts, c_cond, fs=c_fs, **mkw)
            if cat is not None:        # the evaluations of this step as one batch: rows [0, B) conditional, [B, 2B) unconditional, [2B, 3B) image-only
                n = cat["n"]
                e = self.model.apply_model(img.repeat((n,) + (1,) * (img.dim() - 1)), ts.repeat(n), cat["cond"], fs=cat["fs"], **mkw).float().contiguous()
                return ops.cfg_combine(e[:batch_size], e[batch_size:2 * batch_size], e[2 * batch_size:] if n == 3 else None,
                                       scale=unconditional_guidance_scale, cfg_img=cfg_img, guidance_rescale=guidance_rescale)
            e_c = self.model.apply_model(img, ts, c_cond, fs=c_fs, **mkw)
            e_u = self.model.apply_model(img, ts, c_uc, fs=c_fs, **mkw)
            e_i = self.model.apply_model(img, ts, c_img, fs=c_fs, **mkw) if c_img is not None else None
            # ddim.py:216-229 / ddim_multiplecond.py:229-236 + rescale_noise_cfg (utils_diffusion.py:147-158): one fused HIP op
            f = lambda t: None if t is None else t.float().contiguous()
            return ops.cfg_combine(f(e_c), f(e_u), f(e_i), scale=unconditional_guidance_scale, cfg_img=cfg_img,
                                   guidance_rescale=guidance_rescale)

        def step(noise=None):
            ops.gather_timestep(idx, self.ts_table, ts)
            v = model_out()
            ops.ddim_step(img, v.float().contiguous(), self.coef, idx, noise=noise, pred_x0=pred_x0)
            ops.advance_index(idx, -1)

        def log(i):          # ddim.py:195-197: index = total - i - 1
            index = total - i - 1
            if index % log_every_t == 0 or index == total - 1:
                intermediates['x_inter'].append(img.clone())
                intermediates['pred_x0'].append(pred_x0.clone())

        if graph_ok and st["g"] is not None:
            for i in range(total):
                st["g"].replay()
                log(i)
        elif graph_ok:
            step()                                   # eager first step: packs weights, fills the context K/V cache
            log(0)
            g = torch.cuda.CUDAGraph()
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                # thread_local: a process-group watchdog thread (N > 1: RCCL) may query its events while this thread captures
                with torch.cuda.graph(g, stream=side, capture_error_mode="thread_local"):
                    step()                           # recorded, not executed
            torch.cuda.current_stream().wait_stream(side)
            for i in range(1, total):
                g.replay()
                log(i)
            st["g"] = g
            st["token"] = self._prepare(prep_list, T_frames)   # K/V already cached: returns the identity the graph baked in
            self._static, self._graph_key = st, key
        else:
            for i in range(total):
                if mask is not None:
                    # ddim.py:173-180: blend the (noised) original latent back in before every step - inpainting-style sampling
                    t_now = torch.full((batch_size,), int(self.ddim_timesteps[total - 1 - i]), device=dev, dtype=torch.long)
                    x0d, md = x0.to(dev).float(), mask.to(dev).float()
                    img_orig = x0d if clean_cond else self.model.q_sample(x0d, t_now, noise=torch.randn(x0d.shape, device=dev, generator=gen))
                    img.copy_(img_orig * md + (1. - md) * img)
                noise = None
                if eta > 0.:
                    noise = torch.randn(size, device=dev, generator=gen) * temperature
                elif strict_rng:
                    torch.randn(size, device=dev, generator=gen)          # drawn and multiplied by sigma = 0 in the reference
                step(noise)
                if callback:
                    callback(i)
                if img_callback:
                    img_callback(pred_x0, i)
                log(i)
        # the static buffers are overwritten by the next call: hand out copies
        return (img.clone() if graph_ok else img), intermediates
