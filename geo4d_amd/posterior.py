"""Posterior object returned by ``AutoencoderKL.encode`` — the surface of ``lvdm/distributions.py:24-65``
(DiagonalGaussianDistribution) that callers of the hot path touch: attributes ``parameters, mean, logvar, std, var,
deterministic`` and ``sample(noise=None)`` / ``mode()`` / ``kl()`` / ``nll()``.

RNG contract kept from the reference (distributions.py:35-40): when no noise is passed, it is drawn with the CPU global
generator in the shape of ``mean`` and then moved to the device, so a seeded script consumes the RNG stream exactly as the
reference does. The moments come from HIP kernels; the handful of elementwise ops here act on a [n, 2*z, h, w] tensor.
"""
import math

import torch


class DiagonalGaussianDistribution:
    def __init__(self, parameters, deterministic=False):
        mean, logvar = parameters.chunk(2, dim=1)
        logvar = logvar.clamp(-30.0, 20.0)
        self.parameters, self.deterministic = parameters, deterministic
        self.mean, self.logvar = mean, logvar
        if deterministic:
            self.std = self.var = torch.zeros_like(mean)
        else:
            self.std, self.var = (0.5 * logvar).exp(), logvar.exp()

    def sample(self, noise=None):
        eps = torch.randn(self.mean.shape) if noise is None else noise
        return torch.addcmul(self.mean, self.std, eps.to(self.parameters.device))

    def mode(self):
        return self.mean

    def kl(self, other=None):
        if self.deterministic:
            return torch.zeros(1)
        if other is None:
            t = self.mean.square() + self.var - 1.0 - self.logvar
        else:
            t = (self.mean - other.mean).square() / other.var + self.var / other.var - 1.0 - self.logvar + other.logvar
        return 0.5 * t.sum(dim=[1, 2, 3])

    def nll(self, sample, dims=(1, 2, 3)):
        if self.deterministic:
            return torch.zeros(1)
        t = math.log(2.0 * math.pi) + self.logvar + (sample - self.mean).square() / self.var
        return 0.5 * t.sum(dim=list(dims))
