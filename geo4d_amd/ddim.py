"""DDIM sampler — drop-in for ``lvdm.models.samplers.ddim.DDIMSampler`` (constructed by name in
``image_guided_synthesis``, test_geo4d.py:120) with the same ``sample(...)`` signature and return value.

What changes is how a step is executed. The reference's ``p_sample_ddim`` (ddim.py:206-279) creates half a dozen
``torch.full`` tensors and ~10 elementwise kernels per step on top of ~2000 eager launches of the U-Net. Here one step =
[gather t from a device table] + [U-Net kernels] + [one fused DDIM-update kernel reading its coefficients from a device
table indexed by a device-side step counter] + [counter -= 1]; nothing in it depends on host state, so the step is
captured ONCE into a hipGraph and replayed S times (``use_graph=True``, default when eta == 0; with classifier-free
guidance the 2 (or 3, ``multicond``) U-Net evaluations and their combination are part of the captured step).
"""
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

    def __init__(self, model, schedule="linear", use_graph=True, **kwargs):
        self.model = model
        self.ddpm_num_timesteps = model.num_timesteps
        self.schedule = schedule
        self.use_graph = use_graph
        self._static = None
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
    @torch.no_grad()
    def sample(self, S, batch_size, shape, conditioning=None, callback=None, normals_sequence=None, img_callback=None,
               quantize_x0=False, eta=0., mask=None, x0=None, temperature=1., noise_dropout=0., score_corrector=None,
               corrector_kwargs=None, verbose=True, schedule_verbose=False, x_T=None, log_every_t=100,
               unconditional_guidance_scale=1., unconditional_conditioning=None, precision=None, fs=None,
               timestep_spacing='uniform', guidance_rescale=0.0, **kwargs):
        if mask is not None or x0 is not None or score_corrector is not None or quantize_x0 or noise_dropout > 0.:
            raise NotImplementedError("mask / x0 / score_corrector / quantize_x0 / noise_dropout are not used by Geo4D "
                                      "inference (test_geo4d.py:212-227) and have no HIP path")
        if self.model.parameterization != "v":
            raise NotImplementedError("only the v-parameterisation of configs/inference_geo4d.yaml:43 is built")
        self.make_schedule(ddim_num_steps=S, ddim_discretize=timestep_spacing, ddim_eta=eta, verbose=schedule_verbose)
        size = (batch_size,) + tuple(shape)
        dev = self.model.device
        cfg = not (unconditional_conditioning is None or unconditional_guidance_scale == 1.)
        total = len(self.ddim_timesteps)
        kwargs.pop("clean_cond", None)
        graph_ok = self.use_graph and eta == 0. and callback is None and img_callback is None and total > 2
        # 3-way guidance of ddim_multiplecond.py:229-234 (image yes / text "" as a third evaluation)
        uc_img = kwargs.get("unconditional_conditioning_img_nonetext") if (self.multicond and cfg) else None
        if self.multicond and cfg and uc_img is None:
            raise ValueError("DDIMSampler (multiple cond): unconditional_conditioning_img_nonetext is required when CFG is on")
        cfg_img = kwargs.get("cfg_img")
        cfg_img = unconditional_guidance_scale if cfg_img is None else cfg_img
        # static buffers: a captured step graph is reused across sample() calls with the same shapes / conditioning
        ptrs = lambda c: tuple((t.data_ptr(), t._version) for v in (c or {}).values() for t in (v if isinstance(v, (list, tuple)) else [v])) if isinstance(c, dict) else ()   # _version: an in-place edit of a conditioning tensor invalidates the captured step (its context K/V are baked in)
        cond_ptrs = ptrs(conditioning) + ((ptrs(unconditional_conditioning), ptrs(uc_img), float(unconditional_guidance_scale),
                                          float(cfg_img), float(guidance_rescale)) if cfg else ())
        key = (size, S, timestep_spacing, cond_ptrs, None if fs is None else fs.data_ptr(), tuple(sorted(kwargs)))
        cached = self._static if (graph_ok and self._graph_key == key) else None
        if cached is not None:
            g, img, ts, idx, pred_x0 = cached
        else:
            g = None
            img = torch.empty(size, device=dev, dtype=torch.float32)
            ts = torch.empty((batch_size,), dtype=torch.int64, device=dev)
            idx = torch.empty((1,), dtype=torch.int32, device=dev)
            pred_x0 = torch.empty_like(img)
        img.copy_(torch.randn(size, device=dev) if x_T is None else x_T.to(dev).float())
        idx.fill_(total - 1)
        intermediates = {'x_inter': [img.clone()], 'pred_x0': [img.clone()]}

        def model_out():
            if not cfg:
                return self.model.apply_model(img, ts, conditioning, fs=fs, **kwargs)
            e_c = self.model.apply_model(img, ts, conditioning, fs=fs, **kwargs)
            e_u = self.model.apply_model(img, ts, unconditional_conditioning, fs=fs, **kwargs)
            e_i = self.model.apply_model(img, ts, uc_img, fs=fs, **kwargs) if uc_img is not None else None
            # ddim.py:216-229 / ddim_multiplecond.py:229-236 + rescale_noise_cfg (utils_diffusion.py:147-158): one fused HIP op
            f = lambda t: None if t is None else t.float().contiguous()
            return ops.cfg_combine(f(e_c), f(e_u), f(e_i), scale=unconditional_guidance_scale, cfg_img=cfg_img,
                                   guidance_rescale=guidance_rescale)

        def step(noise=None):
            ops.gather_timestep(idx, self.ts_table, ts)
            v = model_out()
            ops.ddim_step(img, v.float().contiguous(), self.coef, idx, noise=noise, pred_x0=pred_x0)
            ops.advance_index(idx, -1)

        if graph_ok and g is not None:
            for _ in range(total):
                g.replay()
        elif graph_ok:
            step()                                   # eager first step: packs weights, fills the context K/V cache
            g = torch.cuda.CUDAGraph()
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                with torch.cuda.graph(g, stream=side):
                    step()                           # recorded, not executed
            torch.cuda.current_stream().wait_stream(side)
            for _ in range(total - 1):
                g.replay()
            self._static, self._graph_key = (g, img, ts, idx, pred_x0), key
            self._keepalive = (conditioning, unconditional_conditioning, uc_img, fs)   # keeps the captured device pointers valid and unique
        else:
            for i in range(total):
                noise = None
                if eta > 0.:
                    noise = torch.randn(size, device=dev) * temperature
                step(noise)
                if callback:
                    callback(i)
                if img_callback:
                    img_callback(pred_x0, i)
        if graph_ok:
            img, pred_x0 = img.clone(), pred_x0.clone()   # the static buffers are overwritten by the next call
        intermediates['x_inter'].append(img)
        intermediates['pred_x0'].append(pred_x0)
        return img, intermediates
