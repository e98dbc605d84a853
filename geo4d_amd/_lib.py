"""ctypes binding of libgeo4d_hip.so (include/geo4d_hip.h).

The library is the product: there is NO fallback. If the shared object is missing or an entry point is absent the
import of any compute path raises immediately (``Geo4DNativeError``) instead of silently running something else.
"""
import ctypes as C
import os

import torch  # noqa: F401  -- MUST precede loading the .so: PyTorch ships its own libamdhip64.so; loading ours first would
#                              bind it to /opt/rocm's copy and the process would hold two HIP runtimes (streams from one
#                              are invalid in the other: "no ROCm-capable device is detected" at the first launch).

_HERE = os.path.dirname(os.path.abspath(__file__))
# GEO4D_HIP_LIB: load another build of the SAME library (A/B builds of a kernel: tools/gpu_r2k.sh); the ABI handshake below still applies
LIB_PATH = os.environ.get("GEO4D_HIP_LIB") or os.path.join(_HERE, "csrc", "libgeo4d_hip.so")
ABI_VERSION = 8

F32, BF16, F16, BF16X3, F16X2 = 0, 1, 2, 3, 4


class Geo4DNativeError(RuntimeError):
    pass


class ConvGemm(C.Structure):
    _fields_ = [
        ("A", C.c_void_p), ("W", C.c_void_p), ("O", C.c_void_p),
        ("bias", C.c_void_p), ("rowbias", C.c_void_p), ("R", C.c_void_p),
        ("zeros", C.c_void_p), ("workspace", C.c_void_p), ("workspace_bytes", C.c_size_t),
        ("lda", C.c_long), ("ldw", C.c_long), ("ldo", C.c_long), ("ldr", C.c_long), ("ldrb", C.c_long),
        ("a_bs", C.c_long), ("w_bs", C.c_long), ("o_bs", C.c_long), ("r_bs", C.c_long),
        ("M", C.c_int), ("N", C.c_int), ("K", C.c_int), ("batch", C.c_int),
        ("Cin", C.c_int),
        ("T", C.c_int), ("Hin", C.c_int), ("Win", C.c_int), ("Hout", C.c_int), ("Wout", C.c_int),
        ("KT", C.c_int), ("KH", C.c_int), ("KW", C.c_int), ("pt", C.c_int), ("ph", C.c_int), ("pw", C.c_int),
        ("stride", C.c_int), ("ups", C.c_int),
        ("rowbias_div", C.c_int), ("bias_per_row", C.c_int), ("act", C.c_int),
        ("dtype", C.c_int), ("out_dtype", C.c_int), ("out_nchw", C.c_int), ("tile_hint", C.c_int),
        ("split_k", C.c_int), ("debug_ablate", C.c_int), ("alpha", C.c_float),
        ("a_split", C.c_int), ("w_split", C.c_int), ("o_split", C.c_int), ("gn_colsum", C.c_void_p), ("sat_count", C.c_void_p),
    ]


class GroupNorm(C.Structure):
    _fields_ = [
        ("x", C.c_void_p), ("y", C.c_void_p), ("gamma", C.c_void_p), ("beta", C.c_void_p),
        ("workspace", C.c_void_p), ("workspace_bytes", C.c_size_t),
        ("ldx", C.c_long), ("ldy", C.c_long),
        ("F", C.c_int), ("HW", C.c_int), ("C", C.c_int), ("groups", C.c_int), ("frames_per_stat", C.c_int),
        ("act", C.c_int), ("dtype", C.c_int), ("eps", C.c_float), ("colsum", C.c_void_p), ("split_out", C.c_int), ("colsum_rows", C.c_int),
        ("sat_count", C.c_void_p),
    ]


class Attention(C.Structure):
    _fields_ = [
        ("q", C.c_void_p), ("o", C.c_void_p),
        ("k", C.c_void_p * 2), ("vt", C.c_void_p * 2), ("zeros", C.c_void_p),
        ("ldq", C.c_long), ("ldo", C.c_long), ("ldk", C.c_long * 2), ("ldvt", C.c_long * 2), ("vt_bs", C.c_long * 2),
        ("Nk", C.c_int * 2), ("kv_div", C.c_int * 2),
        ("B", C.c_int), ("H", C.c_int), ("Nq", C.c_int), ("nseg", C.c_int), ("head_dim", C.c_int), ("dtype", C.c_int),
        ("scale", C.c_float), ("split_out", C.c_int), ("variant", C.c_int), ("qkv_split", C.c_int),
    ]


class Align(C.Structure):
    _fields_ = [
        ("pred", C.c_void_p), ("conf", C.c_void_p), ("logdepth", C.c_void_p), ("cams", C.c_void_p), ("slot_trf", C.c_void_p),
        ("slot_ptr", C.c_void_p), ("slot_idx", C.c_void_p),
        ("grad_logdepth", C.c_void_p), ("img_sums", C.c_void_p), ("slot_sums", C.c_void_p),
        ("workspace", C.c_void_p), ("workspace_bytes", C.c_size_t),
        ("img_part", C.c_void_p), ("slot_part", C.c_void_p),
        ("invdepth", C.c_void_p), ("slot_st", C.c_void_p),
        ("n_imgs", C.c_int), ("n_slots", C.c_int), ("H", C.c_int), ("W", C.c_int), ("chunk_pixels", C.c_int), ("max_slots_per_image", C.c_int),
        ("conf_clamp", C.c_float), ("inv_area", C.c_float), ("depth_weight", C.c_float),
    ]


class AlignSmall(C.Structure):
    _fields_ = [
        ("im_poses", C.c_void_p), ("im_focals", C.c_void_p), ("pw_poses", C.c_void_p), ("cams", C.c_void_p), ("slot_trf", C.c_void_p),
        ("img_sums", C.c_void_p), ("slot_sums", C.c_void_p), ("group_ptr", C.c_void_p), ("group_entries", C.c_void_p), ("group_sums", C.c_void_p),
        ("scale_terms", C.c_void_p),
        ("grad_im_poses", C.c_void_p), ("grad_im_focals", C.c_void_p), ("grad_pw_poses", C.c_void_p), ("grad_s_depth", C.c_void_p),
        ("grad_t_depth", C.c_void_p), ("loss", C.c_void_p),
        ("slot_st", C.c_void_p), ("s_depth", C.c_void_p), ("t_depth", C.c_void_p), ("depth_ok", C.c_void_p),
        ("traj", C.c_void_p), ("traj_align", C.c_void_p), ("traj_valid", C.c_void_p), ("slot_img", C.c_void_p), ("img_slot_ptr", C.c_void_p),
        ("img_slot_idx", C.c_void_p), ("grad_traj", C.c_void_p),
        ("n_imgs", C.c_int), ("n_groups", C.c_int), ("n_slots", C.c_int), ("slots_per_group", C.c_int), ("n_listed_slots", C.c_int),
        ("n_focals", C.c_int), ("norm_pw_scale", C.c_int),
        ("focal_break", C.c_float), ("base_scale", C.c_float), ("ppx", C.c_float), ("ppy", C.c_float), ("smooth_weight", C.c_float),
        ("translation_weight", C.c_float), ("traj_weight", C.c_float),
    ]


# name -> (restype, argtypes); checked against include/geo4d_hip.h by tests/test_host_logic.py::test_c_abi_exports_every_declared_symbol
SIGNATURES = {
    "geo4d_conv_gemm": (C.c_int, [C.POINTER(ConvGemm), C.c_void_p]),
    "geo4d_conv_gemm_colsum_rows": (C.c_int, [C.POINTER(ConvGemm)]),
    "geo4d_groupnorm_workspace": (C.c_size_t, [C.c_int, C.c_int, C.c_int, C.c_int]),
    "geo4d_groupnorm": (C.c_int, [C.POINTER(GroupNorm), C.c_void_p]),
    "geo4d_layernorm": (C.c_int, [C.c_void_p, C.c_long, C.c_void_p, C.c_long, C.c_int, C.c_int, C.c_float, C.c_void_p,
                                  C.c_void_p, C.c_int, C.c_void_p]),
    "geo4d_layernorm_split": (C.c_int, [C.c_void_p, C.c_long, C.c_void_p, C.c_long, C.c_int, C.c_int, C.c_float, C.c_void_p,
                                        C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "geo4d_softmax_rows": (C.c_int, [C.c_void_p, C.c_long, C.c_void_p, C.c_long, C.c_long, C.c_int, C.c_float, C.c_int,
                                     C.c_void_p]),
    "geo4d_softmax_rows_causal": (C.c_int, [C.c_void_p, C.c_long, C.c_void_p, C.c_long, C.c_long, C.c_int, C.c_float, C.c_int,
                                            C.c_int, C.c_void_p]),
    "geo4d_attention": (C.c_int, [C.POINTER(Attention), C.c_void_p]),
    "geo4d_temporal_attention": (C.c_int, [C.c_void_p, C.c_long, C.c_void_p, C.c_long, C.c_void_p, C.c_long, C.c_void_p,
                                           C.c_long, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int,
                                           C.c_void_p]),
    "geo4d_temporal_attention2": (C.c_int, [C.c_void_p, C.c_long, C.c_void_p, C.c_long, C.c_void_p, C.c_long, C.c_void_p,
                                            C.c_long, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int,
                                            C.c_int, C.c_void_p]),
    "geo4d_tokens_from_ncthw": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                          C.c_int, C.c_int, C.c_void_p]),
    "geo4d_cast_rows_f16": (C.c_int, [C.c_void_p, C.c_long, C.c_void_p, C.c_long, C.c_long, C.c_int, C.c_void_p, C.c_void_p]),
    "geo4d_split_rows_bf16": (C.c_int, [C.c_void_p, C.c_long, C.c_void_p, C.c_long, C.c_long, C.c_int, C.c_void_p]),
    "geo4d_concat_channels": (C.c_int, [C.c_void_p, C.c_long, C.c_int, C.c_void_p, C.c_long, C.c_int, C.c_void_p, C.c_long,
                                        C.c_long, C.c_int, C.c_void_p]),
    "geo4d_timestep_embedding": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "geo4d_linear_small": (C.c_int, [C.c_void_p, C.c_long, C.c_void_p, C.c_long, C.c_void_p, C.c_void_p, C.c_long,
                                     C.c_void_p, C.c_long, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "geo4d_ddim_step": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_long,
                                  C.c_void_p]),
    "geo4d_cfg_combine_workspace": (C.c_size_t, [C.c_int]),
    "geo4d_cfg_combine": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_long, C.c_float, C.c_float,
                                    C.c_float, C.c_void_p, C.c_size_t, C.c_void_p]),
    "geo4d_embed_tokens": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_long, C.c_long, C.c_int, C.c_int, C.c_int,
                                     C.c_int, C.c_void_p]),
    "geo4d_advance_index": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p]),
    "geo4d_gather_timestep": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "geo4d_plucker_cameras_workspace": (C.c_size_t, [C.c_int, C.c_int, C.c_int]),
    "geo4d_plucker_cameras": (C.c_int, [C.c_void_p, C.c_void_p, C.c_long, C.c_long, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                        C.c_size_t, C.c_void_p, C.c_void_p]),
    "geo4d_align_workspace": (C.c_size_t, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "geo4d_align_residual": (C.c_int, [C.POINTER(Align), C.c_void_p]),
    "geo4d_adam_step": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_long, C.c_float, C.c_float, C.c_float, C.c_float,
                                  C.c_int, C.c_void_p]),
    "geo4d_adam_step_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_long, C.c_void_p, C.c_float, C.c_float, C.c_float,
                                      C.c_void_p]),
    "geo4d_lad_workspace": (C.c_size_t, [C.c_int, C.c_long]),
    "geo4d_lad_target": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "geo4d_lower_median": (C.c_int, [C.c_void_p, C.c_int, C.c_long, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "geo4d_lad_fit": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_long, C.c_void_p, C.c_float, C.c_int, C.c_float, C.c_void_p, C.c_void_p,
                                C.c_void_p, C.c_size_t, C.c_void_p]),
    "geo4d_lad_delta": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_long, C.c_float, C.c_float, C.c_float,
                                  C.c_void_p, C.c_void_p]),
    "geo4d_last_error": (C.c_char_p, []),
    "geo4d_align_refresh": (C.c_int, [C.POINTER(AlignSmall), C.c_void_p]),
    "geo4d_align_small_grads": (C.c_int, [C.POINTER(AlignSmall), C.c_void_p]),
    "geo4d_abi_version": (C.c_int, []),
    "geo4d_abi_struct_size": (C.c_size_t, [C.c_int]),
}

_lib = None


def load():
    """Load the shared object once; raise Geo4DNativeError (never fall back) when it cannot be used."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise Geo4DNativeError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` (or `make`). "
            "geo4d_amd has no CPU/PyTorch fallback for its compute path.")
    try:
        lib = C.CDLL(LIB_PATH)
    except OSError as e:  # pragma: no cover - depends on the host
        raise Geo4DNativeError(f"cannot load {LIB_PATH}: {e}") from e
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise Geo4DNativeError(f"{LIB_PATH} does not export {name}; rebuild the library") from e
        fn.restype = res
        fn.argtypes = args
    if lib.geo4d_abi_version() != ABI_VERSION:
        raise Geo4DNativeError(f"ABI mismatch: library {lib.geo4d_abi_version()} vs binding {ABI_VERSION}; rebuild")
    for which, struct in enumerate((ConvGemm, GroupNorm, Attention, Align, AlignSmall)):
        if lib.geo4d_abi_struct_size(which) != C.sizeof(struct):
            raise Geo4DNativeError(f"ABI mismatch: {struct.__name__} is {lib.geo4d_abi_struct_size(which)} bytes in the library, "
                                   f"{C.sizeof(struct)} in the binding; rebuild")
    _lib = lib
    return lib


def check(rc, what):
    if rc != 0:
        msg = load().geo4d_last_error()
        raise RuntimeError(f"{what} failed (rc={rc}): {msg.decode() if msg else ''}")
