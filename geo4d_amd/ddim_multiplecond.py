"""3-way classifier-free guidance sampler — drop-in for ``lvdm.models.samplers.ddim_multiplecond.DDIMSampler``
(``DDIMSampler_multicond`` in test_geo4d.py:120, selected by ``--multiple_cond_cfg``).

Per step, with guidance on (ddim_multiplecond.py:229-236):
    out = e(uncond) + cfg_img * (e(img, text="") - e(uncond)) + scale * (e(cond) - e(img, text=""))
followed by the same ``rescale_noise_cfg`` as the 2-way sampler. The third conditioning arrives as the
``unconditional_conditioning_img_nonetext`` keyword (test_geo4d.py:191-197); ``cfg_img`` defaults to the text scale.
Everything else (schedule tables, fused update kernel, hipGraph-captured step) is ``geo4d_amd.ddim.DDIMSampler``.
"""
from .ddim import DDIMSampler as _Base


class DDIMSampler(_Base):
    multicond = True
