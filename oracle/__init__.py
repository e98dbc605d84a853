"""oracle/ — TEST INFRASTRUCTURE ONLY.

A plain-PyTorch, fp32, CPU restatement of the reference algorithm for the Geo4D hot path (U-Net forward, DDIM
sampling, VAE decode, window/post-processing glue). It exists only as the CHECKER: `tests/`, `__graft_entry__.smoke()`
and the `cpu_baseline` leg of `bench.py` may import it; nothing under `geo4d_amd/` does, and the product path raises
when the HIP library is missing instead of falling back to this code.

Parity pin: every function here is checked by `tests/test_oracle_golden.py` against fixtures under `tests/golden/`
that were produced by the REFERENCE ITSELF (`/root/reference` imported on CPU, see `tests/golden/generate.py`):
the real `UNetModel`, `LatentDiffusion` + `DDIMSampler`, `AutoencoderKL` classes with seeded weights. The reference
ships no tests or golden vectors of its own, so these generated fixtures are the pin.
"""
