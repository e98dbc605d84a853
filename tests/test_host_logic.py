"""CPU tests of the host side: state_dict compatibility with the reference (names + shapes from fixtures produced by the
reference classes), config registry, integer tables (bit-exact), window slicing, C-ABI surface."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "tests", "golden")


def load(name):
    return torch.load(os.path.join(G, name), weights_only=False)


def _shapes(m):
    return {k: tuple(v.shape) for k, v in m.state_dict().items()}


def test_unet_state_dict_matches_reference_tiny():
    from geo4d_amd.unet import UNetModel
    g = load("unet_tiny.pt")
    assert _shapes(UNetModel(**g["unet_config"])) == g["shapes"]


def test_unet_state_dict_matches_reference_full_config():
    """The shipped yaml: 1516 tensors, 1438.9 M parameters, incl. `temopral_conv` and the Conv1d init_attn projections."""
    from geo4d_amd.registry import load_config
    from geo4d_amd.unet import UNetModel
    g = load("unet_full.pt")
    cfg = load_config(os.path.join(ROOT, "configs", "inference_geo4d.yaml"))
    with torch.device("meta"):
        m = UNetModel(**cfg.model.params.unet_config.params)
    s = _shapes(m)
    assert s == g["shapes"] and len(s) == 1516
    assert sum(int(np.prod(v)) for v in s.values()) == 1438924112
    assert s["input_blocks.1.0.temopral_conv.conv1.2.weight"] == (320, 320, 3, 1, 1)
    assert s["init_attn.0.proj_in.weight"] == (512, 320, 1)
    assert s["input_blocks.1.1.transformer_blocks.0.attn2.to_k_ip.weight"] == (320, 1024)


def test_vae_state_dict_matches_reference():
    from geo4d_amd.registry import load_config
    from geo4d_amd.vae import AutoencoderKL
    g = load("vae_tiny.pt")
    assert _shapes(AutoencoderKL(ddconfig=g["ddconfig"], lossconfig={"target": "torch.nn.Identity"}, embed_dim=4,
                                 adaptorconfig=g["adaptorconfig"])) == g["shapes"]
    cfg = load_config(os.path.join(ROOT, "configs", "inference_geo4d.yaml"))
    with torch.device("meta"):
        m = AutoencoderKL(**cfg.pointmap_vae_config.params)
    assert _shapes(m) == load("unet_full.pt")["vae_shapes"]


def test_registry_builds_the_engine_from_yaml():
    from geo4d_amd.registry import instantiate_from_config, load_config
    cfg = load_config(os.path.join(ROOT, "configs", "inference_geo4d.yaml"))
    model_cfg = cfg.pop("model")
    model_cfg["params"]["unet_config"]["params"]["use_checkpoint"] = False   # as test_geo4d.py:321-322
    with torch.device("meta"):
        model = instantiate_from_config(model_cfg)
        pvae = instantiate_from_config(cfg.pop("pointmap_vae_config"))
    assert type(model).__name__ == "LatentVisualDiffusion" and model.model.diffusion_model.out_channels == 16
    assert model.parameterization == "v" and model.scale_factor == 0.18215 and model.perframe_ae and model.use_dynamic_rescale
    assert type(pvae).__name__ == "AutoencoderKL" and cfg.postprocess.n_iter == 500
    with pytest.raises(KeyError):
        instantiate_from_config({"params": {}})
    assert instantiate_from_config("__is_unconditional__") is None


def test_schedule_buffers_and_ddim_tables_match_reference():
    from geo4d_amd.ddim import DDIMSampler, make_ddim_timesteps
    from geo4d_amd.diffusion import LatentDiffusion
    g = load("schedule.pt")
    u = load("unet_tiny.pt")["unet_config"]
    m = LatentDiffusion(unet_config={"target": "geo4d_amd.unet.UNetModel", "params": u}, parameterization="v",
                        conditioning_key="hybrid", rescale_betas_zero_snr=True, linear_start=0.00085, linear_end=0.012,
                        use_dynamic_rescale=True, base_scale=0.7, scale_factor=0.18215)
    for k in ("betas", "alphas_cumprod", "alphas_cumprod_prev", "sqrt_alphas_cumprod", "sqrt_one_minus_alphas_cumprod", "scale_arr"):
        assert torch.equal(getattr(m, k), g[k]), k
    for S in (5, 50):
        assert np.array_equal(make_ddim_timesteps("uniform_trailing", S, 1000), g[f"ddim_timesteps_{S}"].numpy())
        s = DDIMSampler(m)
        s.make_schedule(S, "uniform_trailing", 0.0, verbose=False)
        assert np.array_equal(s.ddim_alphas_prev, g[f"ddim_alphas_prev_{S}"].numpy())
        assert torch.equal(s.ddim_scale_arr, g[f"ddim_scale_arr_{S}"]) and torch.equal(s.ddim_scale_arr_prev, g[f"ddim_scale_arr_prev_{S}"])
        assert s.coef.shape == (S, 6) and torch.equal(s.ts_table.cpu(), g[f"ddim_timesteps_{S}"])
        assert torch.all(s.coef[:, 5] == 0)


def test_window_slices_bit_exact():
    from geo4d_amd.pipeline import window_slices
    for (T, stride), ref in load("glue.pt")["windows"].items():
        assert [(s.start, s.stop) for s in window_slices(T, stride)] == ref
        assert all(s.step == 1 for s in window_slices(T, stride))


def test_postprocess_matches_reference_cpu():
    from geo4d_amd.pipeline import postprocess_window
    g = load("glue.pt")["post"]
    o = postprocess_window(g["batch_samples"])
    assert torch.equal(o["valid"], g["pnt_valid_mask"])
    for k in ("pts3d", "conf", "inverse_depthmap"):
        assert torch.allclose(o[k], g[k], rtol=1e-6, atol=1e-7), k


def test_compute_path_refuses_cpu():
    """No silent fallback: a CPU tensor / CPU module must raise, not run somewhere else."""
    from geo4d_amd import _lib, ops
    from geo4d_amd.unet import UNetModel
    with pytest.raises(_lib.Geo4DNativeError):
        ops.layernorm(torch.zeros(4, 64), torch.ones(64), torch.zeros(64))
    m = UNetModel(**load("unet_tiny.pt")["unet_config"])
    with pytest.raises(_lib.Geo4DNativeError):
        m(torch.zeros(1, 20, 2, 8, 8), torch.zeros(1, dtype=torch.long), context=torch.zeros(1, 77 + 32, 128))


def test_c_abi_exports_every_declared_symbol():
    """include/geo4d_hip.h <-> libgeo4d_hip.so <-> ctypes table (no kernel is launched)."""
    from geo4d_amd import _lib
    with open(os.path.join(ROOT, "include", "geo4d_hip.h")) as f:
        hdr = f.read()
    declared = set(re.findall(r"\b(geo4d_[a-z_0-9]+)\s*\(", hdr))
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    lib = _lib.load()
    for name in declared:
        assert hasattr(lib, name)
    assert lib.geo4d_abi_version() == _lib.ABI_VERSION == int(re.search(r"#define GEO4D_ABI_VERSION (\d+)", hdr).group(1))
    assert ctypes.sizeof(_lib.ConvGemm) == 9 * 8 + 9 * 8 + 27 * 4 + 4 + 3 * 4 + 4 + 8 + 8, ctypes.sizeof(_lib.ConvGemm)   # + o_split (+ pad), gn_colsum, sat_count (ABI 8)
    assert lib.geo4d_groupnorm_workspace(16, 2560, 32, 1) == (16 * 64 * 32 * 3 + 16 * 32 * 2) * 4
    # argument validation happens on the host before any launch: bad descriptors return -EINVAL with a message
    p = _lib.ConvGemm()
    assert lib.geo4d_conv_gemm(ctypes.byref(p), None) == -22 and b"conv_gemm" in lib.geo4d_last_error()


def test_tuning_table_uses_only_known_tile_hints():
    """geo4d_amd/tuning/gfx950.json is consulted inside hipGraph capture, where a -EINVAL from geo4d_conv_gemm cannot be
    recovered by re-tuning: every entry must name a tile hint the C ABI documents and a power-of-two split."""
    import json
    import re
    from geo4d_amd import ops
    here = os.path.dirname(os.path.abspath(ops.__file__))
    table = json.load(open(os.path.join(here, "tuning", "gfx950.json")))
    header = open(os.path.join(os.path.dirname(here), "include", "geo4d_hip.h")).read()
    documented = {0, 1, 2, 3, 4, 5} | {int(x) for x in re.findall(r"\b([1237][0-9]) = \d+x\d+", header)}
    assert {11, 13, 16} <= documented
    assert len(table) > 100
    for key, (tile, split) in table.items():
        assert tile in documented, (key, tile)
        assert split in (0, 1, 2, 3, 4, 5, 6, 8, 9, 10, 12, 16), (key, split)       # tile hints 71..74 need an EVEN split of the K slabs: 3, 5, 6, 9, 10, 12 occur
        assert re.match(r"^\d/\d\|\d+x\d+x\d+\|c\d+\|t\d{3}s\du\d\|a\dr\dn\d\|b\d+(\|x[01][01]o?)?$", key), key   # |xAW: bf16x3 pre-split operand flags
    assert all(t in documented for t, _ in ops._CANDIDATES)


def test_weight_relayout_is_the_same_linear_map():
    """geo4d_amd/pack.py (host logic, runs once per load): the packed matrices, applied as the kernel applies them
    (out[m][n] = sum_k A_gather[m][k] W[n][k], taps outer / channels inner, padded channels zero), reproduce F.conv2d,
    F.conv3d (3,1,1), F.linear and GEGLU (attention.py:415-422) on CPU."""
    import torch.nn.functional as F
    from geo4d_amd import pack
    g = torch.Generator().manual_seed(3)
    dt = torch.float32
    # conv2d 3x3, Cin = 5 padded to the f32 K alignment
    x = torch.randn((2, 5, 6, 7), generator=g)
    w = torch.randn((4, 5, 3, 3), generator=g)
    wp = pack.pack_conv2d(w, dt)
    cp = wp.shape[1] // 9
    assert cp >= 5 and cp % 4 == 0                                                # f32: padded to one 32-element K slab
    cols = F.unfold(F.pad(x, (0, 0, 0, 0, 0, cp - 5)), 3, padding=1)             # [2, cp*9, 42], channel-major / tap-minor
    cols = cols.reshape(2, cp, 9, 42).permute(0, 3, 2, 1).reshape(2 * 42, 9 * cp)  # rows = pixels, K = tap-major / channel-minor
    assert torch.allclose((cols @ wp.t()).reshape(2, 6, 7, 4).permute(0, 3, 1, 2), F.conv2d(x, w, padding=1), atol=1e-5)
    # temporal conv (3,1,1)
    xt = torch.randn((1, 8, 5, 2, 3), generator=g)
    wt = torch.randn((6, 8, 3, 1, 1), generator=g)
    wtp = pack.pack_conv3d_t(wt, dt)
    xp = F.pad(xt, (0, 0, 0, 0, 1, 1))
    rows = torch.stack([xp[:, :, k:k + 5] for k in range(3)], dim=1)              # [1, tap, C, T, h, w]
    rows = rows.permute(0, 3, 4, 5, 1, 2).reshape(-1, 3 * 8)
    assert torch.allclose((rows @ wtp.t()).reshape(1, 5, 2, 3, 6).permute(0, 4, 1, 2, 3), F.conv3d(xt, wt, padding=(1, 0, 0)), atol=1e-5)
    # linear with K padding
    wl = torch.randn((7, 10), generator=g)
    wlp = pack.pack_linear(wl, dt)
    xl = torch.randn((3, 10), generator=g)
    assert wlp.shape[1] % 4 == 0 and torch.allclose(F.pad(xl, (0, wlp.shape[1] - 10)) @ wlp.t(), xl @ wl.t(), atol=1e-6)
    # GEGLU: value / gate rows interleaved in blocks of 32 -> out[:, 32j + i] = value * gelu(gate) of column 32j + i
    inner = 64
    wg, bg = torch.randn((2 * inner, 12), generator=g), torch.randn((2 * inner,), generator=g)
    wgp, bgp = pack.pack_geglu(wg, bg, dt)
    xg = torch.randn((5, 12), generator=g)
    h = F.pad(xg, (0, wgp.shape[1] - 12)) @ wgp.t() + bgp
    blocks = h.reshape(5, inner // 32, 2, 32)
    got = (blocks[:, :, 0] * F.gelu(blocks[:, :, 1])).reshape(5, inner)
    ref = xg @ wg.t() + bg
    assert torch.allclose(got, ref[:, :inner] * F.gelu(ref[:, inner:]), atol=1e-5)
    perm = pack.geglu_perm(inner)
    assert sorted(perm.tolist()) == list(range(2 * inner))


def test_binding_refuses_a_library_with_another_struct_layout(monkeypatch):
    """The ctypes mirror of every parameter struct is checked against the library's own sizeof at load time (and the version):
    a stale binding fails at import instead of handing the GPU a struct with shifted fields."""
    import ctypes
    from geo4d_amd import _lib
    lib = _lib.load()
    for which, struct in enumerate((_lib.ConvGemm, _lib.GroupNorm, _lib.Attention, _lib.Align)):
        assert lib.geo4d_abi_struct_size(which) == ctypes.sizeof(struct)
    assert lib.geo4d_abi_struct_size(99) == 0

    class Grown(ctypes.Structure):
        _fields_ = list(_lib.Align._fields_) + [("extra", ctypes.c_void_p)]
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "Align", Grown)
    with pytest.raises(_lib.Geo4DNativeError, match="ABI mismatch: .* bytes in the library"):
        _lib.load()
    monkeypatch.undo()
    assert _lib.load() is not None


def test_library_path_override_fails_loudly(monkeypatch, tmp_path):
    """GEO4D_HIP_LIB points the binding at another build; a missing file raises (no fallback to the default library)."""
    import importlib
    from geo4d_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(_lib.Geo4DNativeError, match="not found"):
        _lib.load()
    monkeypatch.undo()
    assert _lib.load() is not None


def _tiny_lvd(frontend=True):
    from geo4d_amd.diffusion import LatentVisualDiffusion
    g = torch.load(os.path.join(ROOT, "tests", "golden", "unet_tiny.pt"), weights_only=False)
    v = torch.load(os.path.join(ROOT, "tests", "golden", "vae_tiny.pt"), weights_only=False)
    fe = dict(cond_stage_config={"target": "geo4d_amd.encoders.FrozenOpenCLIPEmbedder", "params": dict(layer="penultimate", width=64, layers=2, heads=1, vocab_size=600)},
              img_cond_stage_config={"target": "geo4d_amd.encoders.FrozenOpenCLIPImageEmbedderV2", "params": dict(width=64, layers=1, heads=1, image_size=28, patch_size=14, embed_dim=32)},
              image_proj_stage_config={"target": "geo4d_amd.encoders.Resampler", "params": dict(dim=64, depth=1, dim_head=64, heads=1, num_queries=2, embedding_dim=64, output_dim=128, video_length=4)})
    if not frontend:
        fe = dict(cond_stage_config={"target": "lvdm.modules.encoders.condition.FrozenOpenCLIPEmbedder", "params": {}})
    return LatentVisualDiffusion(unet_config={"target": "geo4d_amd.unet.UNetModel", "params": dict(g["unet_config"])},
                                 first_stage_config={"target": "geo4d_amd.vae.AutoencoderKL", "params": dict(ddconfig=v["ddconfig"], lossconfig=None, embed_dim=4, adaptorconfig=v["adaptorconfig"])},
                                 parameterization="v", conditioning_key="hybrid", rescale_betas_zero_snr=True, linear_start=0.00085, linear_end=0.012,
                                 use_dynamic_rescale=True, base_scale=0.7, scale_factor=0.18215, perframe_ae=True, modality="pc_ray_cross_depth", channels=16, **fe)


def test_reference_checkpoint_loads_the_lazily_built_encoders():
    """instantiate -> load checkpoint -> synthesize must not run on randomly initialised encoders. A checkpoint
    that carries cond_stage_model.* / embedder.* / image_proj_model.* builds those modules before loading; encoders built lazily
    AFTER a load are flagged and warned about; hasattr() on an unconfigured front-end answers False instead of throwing."""
    import warnings
    donor = _tiny_lvd().build_frontend()
    for p in donor.parameters():
        torch.nn.init.normal_(p, std=0.5)
    ckpt = {"state_dict": {k: v.clone() for k, v in donor.state_dict().items()}}
    fresh = _tiny_lvd()
    assert "cond_stage_model" not in fresh._modules                            # still lazy
    skipped = fresh.load_reference_state_dict(ckpt)
    assert skipped == [] and all(n in fresh._modules for n in ("cond_stage_model", "embedder", "image_proj_model"))
    for k, v in fresh.state_dict().items():
        assert torch.equal(v, ckpt["state_dict"][k]), k
    assert not fresh.__dict__.get("_frontend_unloaded")
    # a hot-path-only checkpoint: the encoders stay lazy, and using them later warns that their weights are random
    hot = {k: v for k, v in ckpt["state_dict"].items() if k.split(".")[0] in ("model", "first_stage_model") or "." not in k}
    late = _tiny_lvd()
    late.load_reference_state_dict(hot)
    late.__dict__["_geo4d_context_cache"] = {"stale": 1}
    late.build_frontend(only="cond_stage_model")
    assert "cond_stage_model" in late.__dict__["_frontend_unloaded"] and "_geo4d_context_cache" not in late.__dict__
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        late._warn_unloaded("cond_stage_model")
    assert any("randomly initialised" in str(x.message) for x in w)
    late.__dict__["_geo4d_context_cache"] = {"stale": 1}
    late.load_state_dict(ckpt["state_dict"], strict=False)                    # any later load marks them loaded + drops cached contexts
    assert "cond_stage_model" not in late.__dict__["_frontend_unloaded"] and "_geo4d_context_cache" not in late.__dict__
    foreign = _tiny_lvd(frontend=False)
    assert not hasattr(foreign, "cond_stage_model") and not hasattr(foreign, "embedder")


def test_third_generation_tile_hints_validate_on_the_host():
    """Tile hints 71..74 (gemm_kernel_v3.h): what they cannot serve is refused with -EINVAL and a message before any launch (the
    descriptor's pointers are never dereferenced on the host), and unknown hints next to them stay unknown."""
    from geo4d_amd import _lib
    lib = _lib.load()
    buf = (ctypes.c_char * 4096)()
    addr = ctypes.addressof(buf)

    def desc(dtype, tile, **kw):
        p = _lib.ConvGemm()
        p.A = p.W = p.O = p.zeros = p.workspace = addr
        p.workspace_bytes = 4096
        p.M, p.N, p.K, p.Cin, p.batch = 256, 256, 512, 512, 1
        p.lda = p.ldw = 512
        p.ldo = 256
        p.T = p.Hin = p.Win = p.Hout = p.Wout = p.KT = p.KH = p.KW = p.stride = p.ups = 1
        p.Hin = p.Hout = 256                      # 256 "pixels" of one frame: M = T * Hout * Wout
        p.dtype = p.out_dtype = dtype
        p.alpha = 1.0
        p.tile_hint = tile
        for k, v in kw.items():
            setattr(p, k, v)
        return p
    F32, BF16 = 0, 1
    for tile in (71, 72, 73, 74):
        assert lib.geo4d_conv_gemm(ctypes.byref(desc(F32, tile)), None) == -22
        assert b"71..74" in lib.geo4d_last_error(), lib.geo4d_last_error()
        assert lib.geo4d_conv_gemm(ctypes.byref(desc(BF16, tile, out_nchw=1, ldo=256)), None) == -22
        assert b"NCTHW" in lib.geo4d_last_error(), lib.geo4d_last_error()
    for tile in (70, 75, 79):
        assert lib.geo4d_conv_gemm(ctypes.byref(desc(BF16, tile)), None) == -22
        assert b"tile_hint" in lib.geo4d_last_error(), lib.geo4d_last_error()


def test_split_act_is_rejected_outside_gemm_operands():
    """a SplitAct (bf16 [rows, 2K] hi | lo image) reaching a kernel that expects plain activations would be silently
    reinterpreted; `ops._dev` refuses it everywhere except conv_gemm's operands, and producers only write whole contiguous matrices."""
    import pytest
    import torch
    from geo4d_amd import ops

    class FakeCuda(ops.SplitAct):          # a SplitAct that claims to live on a HIP device (no GPU in this test)
        is_cuda = True
    t = torch.zeros((4, 32), dtype=torch.bfloat16).as_subclass(FakeCuda)
    with pytest.raises(TypeError, match="only GEMM operands"):
        ops._dev(t, "q")
    assert ops._dev(t, "A", True) is t
    with pytest.raises(AssertionError, match="no column-offset views"):
        ops._split_out_ok(torch.zeros((4, 64), dtype=torch.bfloat16)[:, 16:48].as_subclass(ops.SplitAct), 4)
    ops._split_out_ok(ops.SplitAct.wrap(torch.zeros((4, 32), dtype=torch.bfloat16)), 4)


def test_bench_clip_mode_synthetic_scene_is_a_valid_alignment_input():
    """bench.py --clip-frames runs its alignment phases on `synthetic_scene_maps` (random-init weights decode to noise): the maps must
    survive the post-decode math of the reference script (no sky / far-away masks, positive depth, inverse depth inside its range) and
    carry the focal they were built with."""
    import bench
    from geo4d_amd.align import estimate_focal_weiszfeld
    from geo4d_amd.pipeline import postprocess_window, window_slices
    H, W = 40, 64
    sl = window_slices(24, 4, 16)
    maps, traj = bench.synthetic_scene_maps(sl, 16, H, W, torch.device("cpu"))
    assert maps.shape == (len(sl), 11, 16, H, W) and traj.shape == (len(sl), 16, 4, 4) and torch.isfinite(maps).all()
    for g in range(len(sl)):
        p = postprocess_window(maps[g][None])
        assert p["valid"].all() and (p["pts3d"][..., 2] > 0.3).all() and (p["conf"] > 0).all()
        assert (p["inverse_depthmap"] > 0.4).all() and (p["inverse_depthmap"] < 0.8).all()
        f = estimate_focal_weiszfeld(p["raymap"][:2])
        assert torch.allclose(f, torch.full_like(f, 1.2 * W), rtol=1e-4)
    assert torch.allclose(traj[1, 3, :3, 3], torch.tensor([0.03, 0.0, 0.0]))      # the camera slides 0.01 per frame inside a window


def test_two_pass_f16_host_side_contracts():
    """Round 5 (bf16x3m): what the host decides without a GPU - the weight split represents w to ~22 bits for any scale and carries the
    exact inverse of its power-of-two scale; the operand-format helpers tell the formats apart; the mode's class list is the documented
    one and bf16x3 never takes a class."""
    import torch
    from geo4d_amd import ops, pack
    from geo4d_amd.precision import TWO_PASS_CLASSES, resolve
    g = torch.Generator().manual_seed(3)
    for scale in (1e-5, 2e-3, 0.7, 40.0, 3e4):
        w = torch.randn((24, 64), generator=g) * scale
        wp = pack.split_f16(w)
        assert wp.dtype == torch.float16 and wp.shape == (24, 128) and bool(torch.isfinite(wp.float()).all())
        e = wp._x2_alpha
        assert e > 0 and abs(torch.log2(torch.tensor(e)).item() - round(torch.log2(torch.tensor(e)).item())) < 1e-9      # an exact power of two
        halves = wp.reshape(24, 8, 2, 8).double()
        seen = (halves[:, :, 0] + halves[:, :, 1]).reshape(24, 64) * e
        assert ((seen - w.double()).norm() / w.double().norm()).item() < 2e-6
        assert halves[:, :, 0].abs().max().item() < 40000                                                                    # hi stays well inside the f16 range
    z = pack.split_f16(torch.zeros((8, 16)))
    assert z._x2_alpha == 1.0 and float(z.abs().max()) == 0.0
    w = pack.split_f16(torch.randn((8, 32), generator=g))
    x32, a16 = torch.randn((4, 32), generator=g), ops.new_split(4, 32, "cpu", "f16")
    assert ops.is_x2_weight(w) and ops.kdim(w, a16) == 32 and ops.act_k(a16) == 32
    # round 6: the scale travels with the tensor through copies (X2Weight subclass) and is dropped - loudly, at the launch - by anything that changes the layout
    for moved in (w.clone(), w.detach(), w.to(torch.float16), w.contiguous(), w.to("cpu")):
        assert ops.is_x2_weight(moved) and moved._x2_alpha == w._x2_alpha and torch.equal(moved.as_subclass(torch.Tensor), w.as_subclass(torch.Tensor))
    for broken in (w[:4], w.reshape(16, -1), w.float(), w + 0):
        assert not ops.is_x2_weight(broken)
    assert a16.dtype == torch.float16 and tuple(a16.shape) == (4, 32) and not isinstance(a16, ops.SplitAct)       # plain f16 rows: 2 bytes per element
    assert ops.new_split(4, 32, "cpu").dtype == torch.bfloat16 and tuple(ops.new_split(4, 32, "cpu").shape) == (4, 64)
    assert ops.split_fmt(False) == (0, None) and ops.split_fmt(True) == (1, "bf16") and ops.split_fmt("f16") == (2, "f16")
    import pytest
    with pytest.raises(ValueError):
        ops.split_fmt("fp8")
    assert {"conv3x3", "vae3x3", "tconv", "ln", "ff"} <= set(TWO_PASS_CLASSES) or "GEO4D_TWO_PASS" in os.environ
    m, s = resolve("bf16x3m"), resolve("bf16x3")
    assert m.x3 and m.two_pass_conv and m.storage == torch.float32 and m != s and resolve("mixed") == m
    assert not any(s.two_pass(c) for c in ("conv3x3", "vae3x3", "tconv", "ln", "ff", "proj_in"))
    assert all(m.two_pass(c) == (c in TWO_PASS_CLASSES) for c in ("conv3x3", "tconv", "ln", "ff", "proj_in", "proj_out"))
