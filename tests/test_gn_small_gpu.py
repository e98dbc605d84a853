"""The one-launch GroupNorm for small statistics (norm.hip gn_small_kernel, round 5): one workgroup per (statistic, group), no workspace,
no producer sums. Against torch.nn.functional.group_norm in fp64 and against the three-launch form of the same library (`small=2`), for
the shapes the U-Net's 20x32 / 10x16 / 5x8 levels and the tiny test configs produce: 4-D (per frame) and 5-D (across T frames)
statistics, SiLU, plain / bf16-split / f16-split outputs, strided inputs, every storage type. Run-to-run determinism is asserted."""
import pytest
import torch
import torch.nn.functional as TF

from test_gemm_v2_gpu import rel, rnd

pytestmark = pytest.mark.gpu


def ref_gn(x, gamma, beta, F, HW, fps, eps, silu):
    C = x.shape[1]
    t = x.double().reshape(F // fps, fps * HW, C).permute(0, 2, 1)                # [stat, C, rows]
    y = TF.group_norm(t, 32, gamma.double(), beta.double(), eps).permute(0, 2, 1).reshape(F * HW, C)
    return TF.silu(y) if silu else y


@pytest.mark.parametrize("F,HW,C,fps", [(16, 640, 640, 1), (16, 160, 1280, 1), (16, 40, 1280, 1), (16, 40, 1280, 16), (16, 40, 2560, 16), (16, 160, 1920, 1),
                                        (8, 64, 128, 1), (8, 64, 128, 8), (4, 36, 256, 2)])
@pytest.mark.parametrize("silu", [False, True])
def test_one_launch_groupnorm_f32_all_output_formats(dev, F, HW, C, fps, silu):
    from geo4d_amd import ops
    from test_f16x2_gpu import split_f16_act, split_halves
    x = rnd((F * HW, C), dev, 1) * 1.7 + rnd((1, C), dev, 2) * 3.0                 # channel offsets: mean >> std in some groups
    gamma, beta = rnd((C,), dev, 3) + 1.0, rnd((C,), dev, 4)
    ref = ref_gn(x, gamma, beta, F, HW, fps, 1e-5, silu)
    kw = dict(F=F, HW=HW, eps=1e-5, frames_per_stat=fps, silu=silu)
    one = ops.groupnorm(x, gamma, beta, small=1, **kw)
    three = ops.groupnorm(x, gamma, beta, small=2, **kw)
    assert rel(one, ref) < 2e-6 and rel(three, ref) < 2e-6, (rel(one, ref), rel(three, ref))
    assert torch.equal(one, ops.groupnorm(x, gamma, beta, small=1, **kw)), "one-launch GroupNorm is not deterministic"
    b16 = ops.groupnorm(x, gamma, beta, small=1, split_out=True, **kw)
    hi, lo = split_halves(b16)
    assert b16.dtype == torch.bfloat16 and rel(hi + lo, one.double()) < 1e-5
    f16 = ops.groupnorm(x, gamma, beta, small=1, split_out="f16", **kw)
    assert torch.equal(f16.view(torch.int16), split_f16_act(one).view(torch.int16))
    auto = ops.groupnorm(x, gamma, beta, **kw)                                      # the library's own choice: one of the two, same values to round-off
    assert rel(auto, ref) < 2e-6


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_one_launch_groupnorm_16bit_storage_and_strided_rows(dev, dtype):
    from geo4d_amd import ops
    F, HW, C = 16, 40, 1280
    wide = (rnd((F * HW, C + 64), dev, 5) * 2.0 + 0.5).to(dtype)
    x = wide[:, :C]                                                                  # row pitch > C
    gamma, beta = rnd((C,), dev, 6) + 1.0, rnd((C,), dev, 7)
    for fps in (1, 16):
        ref = ref_gn(x.float(), gamma, beta, F, HW, fps, 1e-6, True)
        one = ops.groupnorm(x, gamma, beta, F=F, HW=HW, eps=1e-6, frames_per_stat=fps, silu=True, small=1)
        three = ops.groupnorm(x, gamma, beta, F=F, HW=HW, eps=1e-6, frames_per_stat=fps, silu=True, small=2)
        tol = 6e-3 if dtype == torch.bfloat16 else 8e-4
        assert one.dtype == dtype and rel(one.float(), ref) < tol and rel(one.float(), three.float()) < tol
