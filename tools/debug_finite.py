#!/usr/bin/env python3
"""Find the first layer producing non-finite values at the bench configuration (monkeypatches the layer executors)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from geo4d_amd import ops, unet as U
dev = torch.device("cuda:0")
model, pvae = bench.build(sys.argv[1] if len(sys.argv) > 1 else "bf16", dev)
net = model.model.diffusion_model
bad = []
def wrap(name):
    orig = getattr(ops, name)
    def f(*a, **k):
        out = orig(*a, **k)
        t = out[0] if isinstance(out, tuple) else out
        if not bad and not torch.isfinite(t.float()).all():
            ins = [x for x in a if isinstance(x, torch.Tensor)]
            bad.append(name)
            print("FIRST NON-FINITE from", name, "out", tuple(t.shape), "inputs finite:", [bool(torch.isfinite(x.float()).all()) for x in ins],
                  "in absmax", [float(x.float().abs().max()) for x in ins][:4], {kk: vv for kk, vv in k.items() if not isinstance(vv, torch.Tensor)})
        return out
    setattr(ops, name, f)
for n in ("linear", "conv2d", "conv_temporal", "groupnorm", "layernorm", "attention", "temporal_attention", "linear_t_batched", "linear_t", "concat_channels", "linear_small"):
    wrap(n)
T, h, w = 16, 40, 64
g = torch.Generator().manual_seed(123)
ctx = torch.randn((1, 77 + 16 * T, 1024), generator=g).to(dev); zc = torch.randn((1, 4, T, h, w), generator=g).to(dev)
x = torch.randn((1, 16, T, h, w), generator=torch.Generator().manual_seed(2001)).to(dev)
from geo4d_amd.ddim import DDIMSampler
s = DDIMSampler(model, use_graph=False)
lat, _ = s.sample(S=int(sys.argv[2]) if len(sys.argv) > 2 else 6, conditioning={"c_crossattn": [ctx], "c_concat": [zc]}, batch_size=1, shape=[16, T, h, w], verbose=False, eta=0.0,
                  fs=torch.tensor([24], device=dev), x_T=x, timestep_spacing="uniform_trailing", unconditional_conditioning_img_nonetext=None)
print("latent finite:", bool(torch.isfinite(lat).all()), float(lat.abs().max()))
from geo4d_amd.pipeline import decode_modalities
out = decode_modalities(model, lat, pvae)
print("decoded finite:", bool(torch.isfinite(out).all()))
