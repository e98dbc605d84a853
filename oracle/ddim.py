"""TEST INFRASTRUCTURE ONLY — restatement of the noise schedule and the DDIM sampler.

Follows lvdm/models/ddpm3d.py (register_schedule :162-225, predict_start_from_z_and_v / predict_eps_from_z_and_v
:278-290, scale_arr :585-590), lvdm/models/utils_diffusion.py (make_beta_schedule 'linear' :31-36,
make_ddim_timesteps :56-76, make_ddim_sampling_parameters :79-91, rescale_zero_terminal_snr :112-144) and
lvdm/models/samplers/ddim.py (make_schedule :24-57, ddim_sampling :135-203, p_sample_ddim :206-279) for the shipped
settings: v-parameterisation, eta given, CFG scale 1 (single U-Net evaluation per step), dynamic rescale on.
"""
import numpy as np
import torch


def make_schedule(timesteps=1000, linear_start=0.00085, linear_end=0.012, zero_snr=True):
    betas = (torch.linspace(linear_start ** 0.5, linear_end ** 0.5, timesteps, dtype=torch.float64) ** 2).numpy()
    if zero_snr:
        abar_sqrt = np.sqrt(np.cumprod(1.0 - betas, axis=0))
        a0, aT = abar_sqrt[0].copy(), abar_sqrt[-1].copy()
        abar_sqrt -= aT
        abar_sqrt *= a0 / (a0 - aT)
        abar = abar_sqrt ** 2
        alphas = np.concatenate([abar[0:1], abar[1:] / abar[:-1]])
        betas = 1 - alphas
    alphas_cumprod = np.cumprod(1.0 - betas, axis=0)
    f32 = lambda a: torch.tensor(a, dtype=torch.float32)
    return dict(betas=f32(betas), alphas_cumprod=f32(alphas_cumprod),
                alphas_cumprod_prev=f32(np.append(1.0, alphas_cumprod[:-1])),
                sqrt_alphas_cumprod=f32(np.sqrt(alphas_cumprod)),
                sqrt_one_minus_alphas_cumprod=f32(np.sqrt(1.0 - alphas_cumprod)))


def make_scale_arr(num_timesteps=1000, base_scale=0.7, turning_step=400):
    arr = np.concatenate((np.linspace(1.0, base_scale, turning_step), np.full(num_timesteps, base_scale)))
    return torch.tensor(arr, dtype=torch.float32)


def ddim_timesteps(S, num_ddpm=1000, spacing="uniform_trailing"):
    if spacing == "uniform_trailing":
        c = num_ddpm / S
        return np.flip(np.round(np.arange(num_ddpm, 0, -c))).astype(np.int64) - 1
    if spacing == "uniform":
        c = num_ddpm // S
        return np.asarray(list(range(0, num_ddpm, c))) + 1
    raise NotImplementedError(spacing)


@torch.no_grad()
def ddim_sample(apply_model, sched, scale_arr, S, x_T, eta=0.0, spacing="uniform_trailing", noise_fn=None):
    """apply_model(x, t_long[B]) -> v.  Returns the final latent."""
    ts = ddim_timesteps(S, sched["alphas_cumprod"].shape[0], spacing)
    ac = sched["alphas_cumprod"]
    alphas = ac[ts].double().numpy() if False else ac.numpy()[ts]                     # ddim.py:82 (numpy float32 view of the buffer)
    alphas_prev = np.asarray([ac.numpy()[0]] + ac.numpy()[ts[:-1]].tolist())
    sigmas = eta * np.sqrt((1 - alphas_prev) / (1 - alphas) * (1 - alphas / alphas_prev))
    sc = scale_arr[ts]
    sc_prev = torch.cat([sc[0:1], sc[:-1]])
    x = x_T
    b = x.shape[0]
    shape1 = (b,) + (1,) * (x.dim() - 1)
    for i, step in enumerate(np.flip(ts)):
        index = len(ts) - i - 1
        t = torch.full((b,), int(step), dtype=torch.long)
        v = apply_model(x, t)
        sa = sched["sqrt_alphas_cumprod"][t].reshape(shape1)
        s1 = sched["sqrt_one_minus_alphas_cumprod"][t].reshape(shape1)
        e_t = sa * v + s1 * x
        pred_x0 = sa * x - s1 * v
        pred_x0 = pred_x0 * (torch.full(shape1, sc_prev[index]) / torch.full(shape1, sc[index]))
        a_prev = torch.full(shape1, alphas_prev[index])
        sigma_t = torch.full(shape1, sigmas[index])
        dir_xt = (1.0 - a_prev - sigma_t ** 2).sqrt() * e_t
        noise = sigma_t * (noise_fn(x.shape) if (noise_fn is not None and eta > 0) else torch.zeros_like(x))
        x = a_prev.sqrt() * pred_x0 + dir_xt + noise
    return x


def guided_output(e_c, e_u, scale, guidance_rescale=0.0, e_i=None, cfg_img=None):
    """Classifier-free guidance as the reference samplers combine it: 2-way (ddim.py:216-229) or, with the image-yes /
    text-"" evaluation e_i, 3-way (ddim_multiplecond.py:229-236); then rescale_noise_cfg (utils_diffusion.py:147-158)."""
    if e_i is None:
        out = e_u + scale * (e_c - e_u)
    else:
        out = e_u + (scale if cfg_img is None else cfg_img) * (e_i - e_u) + scale * (e_c - e_i)
    if guidance_rescale > 0.0:
        dims = list(range(1, out.ndim))
        resc = out * (e_c.std(dim=dims, keepdim=True) / out.std(dim=dims, keepdim=True))
        out = guidance_rescale * resc + (1 - guidance_rescale) * out
    return out
