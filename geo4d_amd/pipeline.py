"""Per-window inference glue — the slice of ``scripts/evaluation/test_geo4d.py`` on the hot path:
window index generation (:417-423), ``image_guided_synthesis`` (:118-274), ``decode_pm_confhead`` (:291-312), the
4-modality decode (:248-258) and the post-decode tensor math (:446-501). File / video I/O, CLIP conditioning and the
global alignment are out of scope here.
"""
import torch
import torch.nn.functional as F

from .ddim import DDIMSampler
from .ddim_multiplecond import DDIMSampler as DDIMSamplerMulticond


DECODE_PIXEL_BUDGET = 30 * 1000 * 1000     # output pixels per decoder pass (decode_modalities)


def window_slices(T, stride=4, length=16):
    """test_geo4d.py:417-423. The reference tests ``slice(T-16, T) not in slice_list`` against entries built as
    ``slice(start, start+16, 1)``; ``slice(a, b) != slice(a, b, 1)``, so the tail window is ALWAYS appended — and is a
    duplicate whenever (T-16) % stride == 0 (T = 64 -> 14 windows, last two both (48, 64)). Reproduced bit-exactly."""
    out = [slice(s, s + length, 1) for s in range(0, T - length + 1, stride)]
    out.append(slice(T - length, T, 1))
    return out


@torch.no_grad()
def decode_pm_confhead(z, model, pointmap_vae):
    """test_geo4d.py:291-312 — point map + confidence through the fine-tuned VAE; frames batched, not looped."""
    reshape_back = model.encoder_type == "2d" and z.dim() == 5
    if reshape_back:
        b, c, t, h, w = z.shape
        z = z.permute(0, 2, 1, 3, 4).reshape(b * t, c, h, w)
    out = pointmap_vae.decode_with_conf_adaptor(z * (1.0 / model.scale_factor))
    if reshape_back:
        out = out.reshape(b, t, *out.shape[1:]).permute(0, 2, 1, 3, 4)
    return out


@torch.no_grad()
def decode_modalities(model, samples, pointmap_vae=None):
    """samples [B,16,T,h,w] -> [B,11,T,H,W] = xyz+conf | ray | cross | mean depth (test_geo4d.py:248-258).

    MI355X path: the three modalities that share the first-stage decoder go through it as ONE batch of 3*B*T frames, the
    per-modality heads write straight into their channel range of the NCTHW result, and the depth channel mean is folded
    into the head's weights (exact: the head is linear)."""
    b, c, t, h, w = samples.shape
    assert c == 16 and model.modality == "pc_ray_cross_depth"
    vae = model.first_stage_model
    pvae = pointmap_vae if pointmap_vae is not None else vae
    inv = 1.0 / model.scale_factor
    H, W = 8 * h, 8 * w
    out = torch.empty((b, 11, t, H, W), device=samples.device, dtype=torch.float32)
    # clips per pass: the 3-modality batch of one pass holds 3 * cb * t frames of [H*W, 128] feature maps (7.5 GB per bf16 tensor
    # at the budget): 16 x 320x512 -> all clips at once; 4 x 16 x 576x1024 (BASELINE configs[4]) -> one clip per pass
    cb = max(1, min(b, DECODE_PIXEL_BUDGET // max(1, 3 * t * H * W)))
    for b0 in range(0, b, cb):
        s_, o_ = samples[b0:b0 + cb], out[b0:b0 + cb]
        n = s_.shape[0]

        def frames(z):
            return (z.permute(0, 2, 1, 3, 4).reshape(n * t, 4, h, w) * inv).contiguous()

        # point map + confidence
        feat, _, _ = pvae.decoder_features(frames(s_[:, 0:4]))
        Pp = pvae._packed
        pvae._head(Pp["head"], feat, n * t, H, W, o_[:, 0:], t, 11)
        pvae._head(Pp["adaptor_head"], pvae._conf(Pp, feat, n * t, H, W), n * t, H, W, o_[:, 3:], t, 11)
        # ray | cross | depth share the first-stage decoder: one 3*n*t-frame batch through the trunk
        z3 = torch.cat([frames(s_[:, 4:8]), frames(s_[:, 8:12]), frames(s_[:, 12:16])], 0)
        feat, _, _ = vae.decoder_features(z3)
        Pv = vae._packed
        px = n * t * H * W
        vae._head(Pv["head"], feat[0:px], n * t, H, W, o_[:, 4:], t, 11)
        vae._head(Pv["head"], feat[px:2 * px], n * t, H, W, o_[:, 7:], t, 11)
        vae._head(Pv["head_mean"], feat[2 * px:3 * px], n * t, H, W, o_[:, 10:], t, 11)
    return out


@torch.no_grad()
def decode_modalities_sharded(model, samples, pointmap_vae=None, rank=None, world=None, group=None, decoder=None):
    """Frame-sharded 4-modality decode of ONE window whose latent every rank holds (north_star: "per-frame VAE decode ...
    shard over the 8 GPUs ... RCCL all-gather ... to reassemble the clip"; the per-frame independence is ddpm3d.py:810-819):
    rank r decodes frames ``dist.frame_shard(T, r, world)`` of all four modalities, one all-gather along the frame axis
    returns [B, 11, T, H, W] on every rank. T = 16 on 8 GPUs = 2 frames (8 frame-modalities) per GPU."""
    from . import dist as gdist
    if world is None:
        world = torch.distributed.get_world_size(group) if torch.distributed.is_initialized() else 1
    if rank is None:
        rank = torch.distributed.get_rank(group) if torch.distributed.is_initialized() else 0
    decoder = decoder or decode_modalities
    if world == 1:
        return decoder(model, samples, pointmap_vae)
    T = samples.shape[2]
    lo, hi = gdist.frame_shard(T, rank, world)
    if hi > lo:
        part = decoder(model, samples[:, :, lo:hi].contiguous(), pointmap_vae)
    else:                                         # more ranks than frames: this rank only takes part in the collective
        b, _, _, h, w = samples.shape
        part = samples.new_zeros((b, 11, 0, 8 * h, 8 * w), dtype=torch.float32)
    return gdist.all_gather_frames(part, T, rank=rank, world=world, group=group, dim=2)


@torch.no_grad()
def get_latent_z(model, videos):
    """test_geo4d.py:110-115: videos [b,3,t,H,W] in [-1,1] -> z_video [b,4,t,H/8,W/8] (the c_concat conditioning)."""
    return model.encode_first_stage(videos)


@torch.no_grad()
def image_guided_synthesis(model, prompts, videos, noise_shape, n_samples=1, ddim_steps=50, ddim_eta=1.,
                           unconditional_guidance_scale=1.0, cfg_img=None, fs=None, text_input=False,
                           multiple_cond_cfg=False, loop=False, interp=False, timestep_spacing='uniform',
                           guidance_rescale=0.0, pointmap_vae=None, cond=None, x_T=None, decode=True, **kwargs):
    """test_geo4d.py:118-274 for modality 'pc_ray_cross_depth'. ``cond`` = {"c_crossattn": [ctx [B,77+16T,1024]]} may be
    supplied; otherwise it is computed like the reference does (OpenCLIP text tower on the prompts + Resampler over the OpenCLIP
    image tokens of a zero image, geo4d_amd/encoders.py). ``c_concat`` is taken from ``cond`` if present, otherwise computed
    from ``videos`` [B,3,T,H,W] by the VAE encoder like the reference does. ``multiple_cond_cfg`` selects the 3-way guidance
    sampler. Returns [B, n_samples, 11, T, H, W] like the reference; with ``decode=False`` the denoised latents
    [B, n_samples, 16, T, h, w] instead (the caller decodes them, e.g. frame-sharded across GPUs)."""
    if loop or interp:
        raise NotImplementedError("loop / interp raise in the reference too (test_geo4d.py:161-162)")
    batch_size = noise_shape[0]
    fs_t = torch.tensor([fs] * batch_size, dtype=torch.long, device=model.device)
    if cond is None:
        # test_geo4d.py:124-158: text prompts are blanked unless text_input; the image branch embeds a ZERO image unless
        # model.cross_attention (then every frame). Built lazily from geo4d_amd.encoders and cached: constant across windows.
        if not text_input:
            prompts = [""] * batch_size
        # keyed on the compute mode too; the cache is dropped by LatentDiffusion._after_load / build_frontend (weights changed)
        key = (tuple(prompts), bool(model.cross_attention), tuple(videos.shape) if model.cross_attention else tuple(videos.shape[-2:]),
               getattr(model.model.diffusion_model, "compute_dtype", None))
        cache = model.__dict__.setdefault("_geo4d_context_cache", {})
        ctx = cache.get(key) if not model.cross_attention else None
        if ctx is None:
            ctx = model.context_for(prompts, image=videos[:, :, 0], frames=videos)
            if not model.cross_attention:
                cache.clear()
                cache[key] = ctx
        cond = {"c_crossattn": [ctx]}
    if "c_concat" not in cond and model.model.conditioning_key == "hybrid":
        cond = dict(cond, c_concat=[get_latent_z(model, videos)])       # test_geo4d.py:159-170 (modality != img_vidpc)
    def with_latent(c):
        # test_geo4d.py:184-195: the unconditional dicts carry the SAME video latent (uc['c_concat'] = [img_cat_cond])
        if isinstance(c, dict) and "c_concat" not in c and "c_concat" in cond:
            c = dict(c, c_concat=cond["c_concat"])
        return c
    uc = None
    if unconditional_guidance_scale != 1.0:
        uc = kwargs.pop("unconditional_conditioning", None)
        if uc is None:
            raise NotImplementedError("CFG needs precomputed unconditional conditioning (front-end is N3)")
        uc = with_latent(uc)
    if multiple_cond_cfg and cfg_img != 1.0 and uc is not None:
        if kwargs.get("unconditional_conditioning_img_nonetext") is None:
            raise NotImplementedError("multiple_cond_cfg needs precomputed unconditional_conditioning_img_nonetext (front-end is N3)")
        kwargs["unconditional_conditioning_img_nonetext"] = with_latent(kwargs["unconditional_conditioning_img_nonetext"])
    else:
        kwargs.update({"unconditional_conditioning_img_nonetext": None})
    # one sampler per (model, class): its captured step graph and static conditioning buffers are reused by every later window
    cls = DDIMSamplerMulticond if multiple_cond_cfg else DDIMSampler
    cache = model.__dict__.setdefault("_geo4d_samplers", {})
    sampler = cache.get(cls)
    if sampler is None:
        sampler = cache[cls] = cls(model)
    variants = []
    for _ in range(n_samples):
        samples, _ = sampler.sample(S=ddim_steps, conditioning=cond, batch_size=batch_size, shape=noise_shape[1:], verbose=False,
                                    unconditional_guidance_scale=unconditional_guidance_scale, unconditional_conditioning=uc,
                                    eta=ddim_eta, cfg_img=cfg_img, mask=None, x0=None, fs=fs_t, x_T=x_T,
                                    timestep_spacing=timestep_spacing, guidance_rescale=guidance_rescale, **kwargs)
        variants.append(decode_modalities(model, samples, pointmap_vae) if decode else samples)
    return torch.stack(variants).permute(1, 0, 2, 3, 4, 5)


@torch.no_grad()
def run_clip(model, videos_all, context, *, pointmap_vae=None, stride=4, video_length=16, seed=123, ddim_steps=50, ddim_eta=0.0,
             unconditional_guidance_scale=1.0, fs=24, timestep_spacing="uniform_trailing", guidance_rescale=0.7,
             prompts=("Output a video that assigns each 3D location in the world a consistent color.",),
             synthesize=None, gather=True, with_cameras=False, decode="local", decoder=None, window_batch=1, **kwargs):
    """The window loop of ``run_inference`` (test_geo4d.py:396-443), data-parallel over windows.

    ``videos_all`` [1,3,T,H,W] in [-1,1]; ``context`` = cross-attention context [1, 77+16*video_length, D] (the OpenCLIP /
    Resampler front-end is N3; the shipped config feeds a fixed prompt and a zero image, so it is window-independent) or a
    callable ``context(window_frames) -> tensor``. Windows come from ``window_slices`` (tail window always appended), window
    ``w`` runs on rank ``w % world`` and ONE all-gather returns every window's decoded maps on every rank:
    ``(slices, maps [n_windows, 11, video_length, H, W])``; with ``with_cameras`` also the per-window camera-to-world
    matrices ``traj [n_windows, video_length, 4, 4]`` from the ray / ray-moment maps (test_geo4d.py:455-458, computed on the
    device by ``geo4d_amd.rays``, no host sync in the loop). The initial noise, the posterior sampling of the VAE encode and
    (eta > 0) the per-step noise are all seeded PER WINDOW (``seed``, window index), so the result does not depend on the
    number of GPUs or on which windows a rank ran before — unlike the reference's single sequential RNG stream, which
    cannot be reproduced across a sharded loop.

    ``decode``: "local" — every rank decodes the windows it denoised and ONE all-gather of the decoded maps follows the loop;
    "sharded" — windows are processed in rounds of ``world``; each round's latents are broadcast by their owners (2.6 MB each)
    and every window's 4 x T frame-modalities are decoded FRAME-SHARDED over all ranks with an all-gather along the frame axis
    (decode_modalities_sharded), which keeps all GPUs busy in a ragged last round (14 windows on 8 GPUs) and for a single window.
    Every rank ends up with every window either way.

    ``window_batch`` (round 6): a rank denoises this many of ITS windows as ONE batch (eta = 0 only). Windows are independent, and at B = 1
    the U-Net's levels 1-3 leave most of the 256 CUs idle (tiles < CUs): two windows per DDIM step take 1.64x the time of one on an
    MI355X (profiles/r06_window_batch.md: 7.78 vs 6.52 denoised frames/s). Noise, VAE-encode sampling and conditioning stay seeded /
    computed PER WINDOW, so a window's result does not depend on what it was batched with beyond fp32 round-off (GEMM tile choice follows M).
    In the sharded-decode mode a round then covers world x window_batch windows, rank r owning the r-th run of window_batch."""
    from . import dist as gdist
    synthesize = synthesize or image_guided_synthesis
    B, C, T, H, W = videos_all.shape
    if B != 1:
        raise ValueError("run_clip: one clip at a time (the reference asserts bs == 1, test_geo4d.py:354-356)")
    slices = window_slices(T, stride, video_length)
    world = torch.distributed.get_world_size() if torch.distributed.is_initialized() else 1
    rank = torch.distributed.get_rank() if torch.distributed.is_initialized() else 0
    channels = model.model.diffusion_model.out_channels
    noise_shape = [B, channels, video_length, H // 8, W // 8]
    local, traj = [], []

    def cameras(maps):
        from .rays import raymap_to_camera_matrix
        return raymap_to_camera_matrix(maps[:, 4:7], maps[:, 7:10])[None]

    def one_window(wi, decode_here):
        videos = videos_all[:, :, slices[wi]].clone()
        wseed = (int(seed) * 1000003 + wi) % (2 ** 63 - 1)
        x_T = torch.randn(noise_shape, generator=torch.Generator().manual_seed(wseed)).to(videos_all.device)
        if ddim_eta > 0.0 and videos_all.is_cuda:   # per-window device stream for the stochastic step noise (eta > 0)
            kwargs["noise_generator"] = torch.Generator(device=videos_all.device).manual_seed(wseed)
        ctx = context(videos) if callable(context) else context
        extra = {} if decode_here else {"decode": False}
        with torch.random.fork_rng(devices=[]):
            torch.manual_seed(wseed)                                   # posterior sampling noise of the VAE encode
            maps = synthesize(model, list(prompts), videos, noise_shape, n_samples=1, ddim_steps=ddim_steps, ddim_eta=ddim_eta,
                              unconditional_guidance_scale=unconditional_guidance_scale, fs=fs, timestep_spacing=timestep_spacing,
                              guidance_rescale=guidance_rescale, pointmap_vae=pointmap_vae, cond={"c_crossattn": [ctx]},
                              x_T=x_T, **extra, **kwargs)
        assert maps.shape[1] == 1, "only support variants size = 1"
        return maps[:, 0]

    wb = 1 if ddim_eta > 0.0 else max(1, int(window_batch))

    def window_group(wis, decode_here):
        """Several of this rank's windows as ONE batch through the sampler (and the decoder): [len(wis), 11 | 16, T, ...]."""
        if len(wis) == 1:
            return one_window(wis[0], decode_here)
        vids, xs, zs, ctxs = [], [], [], []
        for wi in wis:
            videos = videos_all[:, :, slices[wi]].clone()
            wseed = (int(seed) * 1000003 + wi) % (2 ** 63 - 1)
            xs.append(torch.randn(noise_shape, generator=torch.Generator().manual_seed(wseed)))
            ctxs.append(context(videos) if callable(context) else context)
            if model.model.conditioning_key == "hybrid":
                with torch.random.fork_rng(devices=[]):
                    torch.manual_seed(wseed)                           # the window's own posterior-sampling noise, as in one_window
                    zs.append(get_latent_z(model, videos))
            vids.append(videos)
        cond = {"c_crossattn": [torch.cat(ctxs, 0)]}
        if zs:
            cond["c_concat"] = [torch.cat(zs, 0)]
        extra = {} if decode_here else {"decode": False}
        maps = synthesize(model, list(prompts) * len(wis), torch.cat(vids, 0), [len(wis)] + noise_shape[1:], n_samples=1, ddim_steps=ddim_steps,
                          ddim_eta=ddim_eta, unconditional_guidance_scale=unconditional_guidance_scale, fs=fs, timestep_spacing=timestep_spacing,
                          guidance_rescale=guidance_rescale, pointmap_vae=pointmap_vae, cond=cond, x_T=torch.cat(xs, 0).to(videos_all.device),
                          **extra, **kwargs)
        assert maps.shape[1] == 1, "only support variants size = 1"
        return maps[:, 0]

    if decode == "sharded" and world > 1:
        if B != 1:
            raise ValueError("sharded decode handles one clip at a time")
        nwin = len(slices)
        wb = min(wb, -(-nwin // world))          # never more windows per batch than a balanced deal gives a rank (14 windows on 8 ranks: 2)
        per_round = world * wb
        for k in range(0, nwin, per_round):
            mine = [k + rank * wb + j for j in range(wb) if k + rank * wb + j < nwin]
            lat = window_group(mine, False).float().contiguous() if mine else None
            for wi in range(k, min(k + per_round, nwin)):
                owner, j = divmod(wi - k, wb)
                buf = lat[j:j + 1].contiguous() if owner == rank else videos_all.new_empty(noise_shape, dtype=torch.float32)
                gdist.broadcast_from(buf, owner)
                maps = decode_modalities_sharded(model, buf, pointmap_vae, rank=rank, world=world, decoder=decoder)
                local.append(maps)
                if with_cameras:
                    traj.append(cameras(maps))
        out = [torch.cat(local, 0)]
        if with_cameras:
            out.append(torch.cat(traj, 0))
        return (slices, *out)
    if decode not in ("local", "sharded"):
        raise ValueError(f"run_clip: decode={decode!r} (expected 'local' or 'sharded')")
    mine = gdist.shard_windows(len(slices), rank, world)
    for i in range(0, len(mine), wb):
        maps = window_group(mine[i:i + wb], True)
        local.append(maps)
        if with_cameras:
            traj.extend(cameras(maps[j:j + 1]) for j in range(maps.shape[0]))
    like = videos_all.new_zeros((0, 11, video_length, H, W), dtype=torch.float32)
    local = torch.cat(local, 0) if local else like
    out = [local]
    if with_cameras:
        out.append(torch.cat(traj, 0) if traj else videos_all.new_zeros((0, video_length, 4, 4), dtype=torch.float32))
    if gather:
        out = [gdist.all_gather_windows(o, len(slices), rank=rank, world=world) for o in out]
    return (slices, *out)


def get_sky_mask(x, sky_value=1.05, eps=0.05):
    lo, hi = sky_value - eps, sky_value + eps
    return ((x > lo) & (x < hi)).all(dim=-1, keepdim=True)


def get_far_away_mask(x, far_away_value=1.5):
    return (x.abs() > far_away_value).any(dim=-1, keepdim=True)


def denormalize_pc_bbox2(pc, alpha=1.0, beta=1.0):
    return torch.stack([pc[..., 0] / alpha, pc[..., 1] / beta, (pc[..., 2] + 1) / 2], dim=-1)


@torch.no_grad()
def postprocess_window(batch_samples, pointmap_vae_used=True):
    """test_geo4d.py:446-501: [1,11,T,H,W] -> dict(pts3d [T,H,W,3], conf (inverse confidence) [T,H,W,1],
    inverse_depthmap, raymap, crossmap, valid mask)."""
    x = batch_samples[0].permute(1, 2, 3, 0)                      # t h w c
    raymap, crossmap = x[..., 4:7], x[..., 7:10]
    inverse_depthmap = (x[..., 10:11] + 1.0) / 2.0
    conf = F.softplus(x[..., 3:4]) if pointmap_vae_used else torch.ones_like(x[..., 3:4])
    pts = x[..., 0:3]
    invalid = get_sky_mask(pts, sky_value=1.05, eps=0.35) | get_far_away_mask(pts, far_away_value=1.99)
    conf = torch.where(invalid, torch.full_like(conf, 999.0), conf)
    inv_conf = torch.where(invalid, torch.zeros_like(conf), 1.0 / conf)
    return dict(pts3d=denormalize_pc_bbox2(pts, alpha=2.0, beta=2.0), conf=inv_conf, inverse_depthmap=inverse_depthmap,
                raymap=raymap, crossmap=crossmap, valid=~invalid)
