"""geo4d_amd — MI355X-native (gfx950 / CDNA4) engine for the Geo4D denoise + decode hot path.

Python host code (state_dict-compatible modules, the reference's config registry and sampler call surface)
driving hand-written HIP kernels through the C ABI in include/geo4d_hip.h. See DESIGN.md.
"""
__version__ = "0.1.0"
