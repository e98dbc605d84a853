"""CPU evidence for the compute-mode choice (geo4d_amd/precision.py): with the engine's rounding points emulated on the
oracle (tests/precision_sim.py), NO single pass over 16-bit operands reaches the 1e-3 point-map bar on this network —
rounding only the weights to f16 already exceeds it, so does rounding only the GEMM inputs — while the operand the bf16x3
scheme sees (bf16 hi + bf16 lo) leaves > 10x margin. One window = 3-step DDIM + 4-modality decode of the tiny golden
config, i.e. the setting of tests/test_parity_gpu.py::test_window_end_to_end_vs_oracle."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def test_single_pass_16bit_cannot_meet_the_bar_but_split_bf16_does():
    import precision_sim as ps
    from oracle import ddim as oddim
    from oracle.params import seeded_state_dict
    G = os.path.join(ps.ROOT, "tests", "golden")
    u = torch.load(os.path.join(G, "unet_tiny.pt"), weights_only=False)
    v = torch.load(os.path.join(G, "vae_tiny.pt"), weights_only=False)
    usd, vsd = seeded_state_dict(u["shapes"]), seeded_state_dict(v["shapes"])
    psd = seeded_state_dict(dict(v["shapes"]), gain=0.9)
    cfg = u["unet_config"]
    gen = torch.Generator().manual_seed(777)
    B, T, h, w = 1, 16, 8, 8
    x_T = torch.randn((B, 16, T, h, w), generator=gen)
    ctx = torch.randn((B, 77 + 16 * T, cfg["context_dim"]), generator=gen)
    zc = torch.randn((B, 4, T, h, w), generator=gen)
    fs = torch.tensor([24])

    def pts(sc):
        S_u = ps.S_(usd, sc)
        am = lambda x, t: ps.unet_forward(S_u, cfg, torch.cat([x, zc], 1), t, ctx, fs, sc)
        lat = oddim.ddim_sample(am, oddim.make_schedule(), oddim.make_scale_arr(), 3, x_T, eta=0.0)
        return ps.decode_modalities(vsd, psd, v["ddconfig"], v["adaptorconfig"], lat, sc)[:, :3]
    ref = pts(ps.Scheme())
    e_w = ps.rel(pts(ps.Scheme("f16", "f32", "f32")), ref)          # weights rounded once to f16, everything else exact
    e_a = ps.rel(pts(ps.Scheme("f32", "f16", "f32")), ref)          # GEMM inputs rounded to f16, fp32 residual streams, exact weights
    e_3 = ps.rel(pts(ps.Scheme("bf16x2", "bf16x2", "f32")), ref)    # what the bf16x3 mode multiplies
    print(f"point-map rel L2: f16 weights only {e_w:.2e}, f16 GEMM inputs only {e_a:.2e}, bf16 hi+lo operands {e_3:.2e}")
    assert e_w > 1e-3 and e_a > 1e-3, "a single f16 pass would meet the bar after all: revisit geo4d_amd/precision.py"
    assert e_3 < 1e-4
