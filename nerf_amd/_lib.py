"""ctypes binding of libnerf_amd.so (include/nerf_amd.h).  There is NO fallback: if the HIP library
is missing or fails to load, importing this module raises -- the product never silently runs a
torch/CPU substitute."""
import ctypes as C
import os

import torch  # noqa: F401  -- MUST precede the dlopen below: torch bundles its own libamdhip64; loading ours first
#                             would bring up a second HIP runtime that cannot see torch's device allocations.

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("NERF_AMD_LIB") or os.path.join(_HERE, "libnerf_amd.so")   # env override: diagnostic builds only

F32, BF16 = 0, 1
BF16_F8 = 2            # NERF_AMD_BF16_F8: bf16 arithmetic, training dumps of the hidden layers in scaled e4m3 (training entry points only)
EXPECTED_VERSION = 125  # nerf_amd_version() of the library these signatures were written against
NET_PROPOSAL, NET_MIP, NET_REF, NET_PROPOSAL_128, NET_MIP_128 = 0, 1, 2, 3, 4
FINE_W128 = 0x200     # layout flag: the fine-network blob is a NET_MIP_128 blob
PROP_W128 = 0x100     # layout flag OR-ed into `precision`: packed_prop is a NET_PROPOSAL_128 blob
ACT_RELU, ACT_IDENTITY, ACT_SOFTPLUS = 0, 1, 2

c_float_p = C.POINTER(C.c_float)
c_void = C.c_void_p
i64 = C.c_int64


class Samples(C.Structure):
    """struct nerf_amd_samples"""
    _fields_ = [("mode", C.c_int32), ("S", C.c_int32), ("M", C.c_int64), ("pts", c_void), ("pts_stride", C.c_int32),
                ("z_stride", C.c_int32), ("rays", c_void), ("z", c_void), ("z_base", c_void), ("u", c_void),
                ("z_jitter", C.c_float), ("H", C.c_int32), ("W", C.c_int32), ("fx", C.c_float), ("fy", C.c_float),
                ("pose", C.c_float * 12), ("contract", C.c_int32), ("ipe", C.c_int32), ("ipe_radius", C.c_float),
                ("ipe_dir_norm", c_void), ("rng_seed", C.c_uint64), ("rng_ray_offset", C.c_int64)]


# name -> (restype, argtypes); mirrors include/nerf_amd.h one to one (tests check the two agree)
SIGNATURES = {
    "nerf_amd_last_error": (C.c_char_p, []),
    "nerf_amd_version": (C.c_int, []),
    "nerf_amd_device_info": (C.c_int, [C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "nerf_amd_packed_bytes": (C.c_size_t, [C.c_int, C.c_int]),
    "nerf_amd_pack_weights": (C.c_int, [C.c_int, C.c_int, C.POINTER(c_void), C.POINTER(c_void), C.c_int, c_void, c_void]),
    "nerf_amd_proposal_forward": (C.c_int, [c_void, C.c_int, C.POINTER(Samples), c_void, c_void]),
    "nerf_amd_mip_forward": (C.c_int, [c_void, C.c_int, C.POINTER(Samples), c_void, c_void]),
    "nerf_amd_mip_forward_composite": (C.c_int, [c_void, C.c_int, C.POINTER(Samples), C.c_int, C.c_float, C.c_float, c_void, c_void, c_void,
                                                 c_void]),
    "nerf_amd_ref_forward": (C.c_int, [c_void, C.c_int, C.POINTER(Samples), C.c_int, c_void, c_void, c_void]),
    "nerf_amd_ref_forward_train": (C.c_int, [c_void, C.c_int, C.POINTER(Samples), C.c_int, c_void, c_void, c_void, c_void]),
    "nerf_amd_positional_encoding": (C.c_int, [c_void, i64, C.c_int, c_void, c_void]),
    "nerf_amd_ipe_feature": (C.c_int, [c_void, c_void, i64, C.c_int, C.c_int, C.c_float, c_void, c_void, c_void, c_void, c_void]),
    "nerf_amd_ipe_feature_contracted": (C.c_int, [c_void, c_void, i64, C.c_int, C.c_int, C.c_float, c_void, c_void, c_void, c_void, c_void]),
    "nerf_amd_cone_parameters": (C.c_int, [c_void, i64, C.c_int, C.c_float, c_void, c_void, c_void, c_void]),
    "nerf_amd_dirs_norm": (C.c_int, [c_void, i64, c_void, c_void]),
    "nerf_amd_generate_rays": (C.c_int, [c_float_p, C.c_int, C.c_int, C.c_float, C.c_float, i64, i64, c_void, c_void]),
    "nerf_amd_length2pts": (C.c_int, [c_void, c_void, i64, C.c_int, c_void, c_void]),
    "nerf_amd_sigma_to_weights": (C.c_int, [c_void, c_void, c_void, i64, C.c_int, C.c_int, c_void, c_void]),
    "nerf_amd_max_blur": (C.c_int, [c_void, i64, C.c_int, C.c_float, c_void, c_void]),
    "nerf_amd_inverse_sample": (C.c_int, [c_void, c_void, c_void, i64, C.c_int, C.c_int, C.c_int, c_void, c_void, c_void]),
    "nerf_amd_sample_pdf": (C.c_int, [c_void, c_void, c_void, i64, C.c_int, C.c_int, c_void, c_void, c_void, c_void]),
    "nerf_amd_pixel_rays": (C.c_int, [c_float_p, C.c_float, C.c_float, c_void, i64, c_void, c_void]),
    "nerf_amd_sample_training_rays": (C.c_int, [c_void, c_void, i64, c_float_p, C.c_float, C.c_float, C.c_float, C.c_float, i64, C.c_int, C.c_uint64,
                                               c_void, c_void, c_void, c_void, c_void]),
    "nerf_amd_sample_training_rays_dev": (C.c_int, [c_void, c_void, i64, c_void, C.c_float, C.c_float, C.c_float, C.c_float, i64, C.c_int, c_void,
                                                   c_void, c_void, c_void, c_void, c_void]),
    "nerf_amd_philox_uniforms": (C.c_int, [c_void, i64, C.c_int, C.c_uint64, c_void, c_void]),
    "nerf_amd_philox_stream": (C.c_int, [c_void, i64, C.c_int, C.c_uint64, c_void, i64, C.c_int, c_void]),
    "nerf_amd_advance_seed": (C.c_int, [c_void, c_void]),
    "nerf_amd_stratified_points": (C.c_int, [c_void, c_void, c_void, C.c_float, i64, C.c_int, c_void, c_void, c_void]),
    "nerf_amd_resample": (C.c_int, [c_void, c_void, c_void, c_void, C.c_float, c_void, C.c_int, c_void, i64, C.c_int,
                                    C.c_int, C.c_int, C.c_float, C.c_uint64, i64, c_void, c_void, c_void, c_void, c_void]),
    "nerf_amd_composite": (C.c_int, [c_void, c_void, C.c_int, c_void, C.c_int, i64, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float,
                                     C.c_float, c_void, c_void, c_void, c_void, c_void, c_void, c_void]),
    "nerf_amd_train_dump_rows_mask_partials": (i64, []),
    "nerf_amd_train_dump_rows_mask": (C.c_int, [c_void, C.c_int, C.c_int, i64, C.c_int, C.c_int, c_void, c_void, c_void, c_void]),
    "nerf_amd_render_ref_workspace_bytes": (C.c_size_t, [i64, C.c_int]),
    "nerf_amd_render_rays_ref": (C.c_int, [c_void, c_void, C.c_int, C.c_int, c_void, C.POINTER(Samples), i64, c_void, c_void, c_void, i64, C.c_int,
                                          C.c_float, C.c_float, C.c_int, c_void, c_void, c_void, c_void, c_void, c_void]),
    "nerf_amd_encode_rows": (C.c_int, [c_void, C.c_int, i64, C.c_int, C.c_int, C.c_int, c_void, c_void]),
    "nerf_amd_merge_depths": (C.c_int, [c_void, c_void, i64, C.c_int, C.c_int, c_void, c_void]),
    "nerf_amd_merge_depths_order": (C.c_int, [c_void, c_void, c_void, i64, C.c_int, C.c_int, c_void, c_void, c_void, c_void]),
    "nerf_amd_coarse_grad_select": (C.c_int, [c_void, c_void, i64, C.c_int, C.c_int, C.c_int, c_void, c_void]),
    "nerf_amd_mfma_stream": (C.c_int, [C.c_int, C.c_int, C.c_int, c_void, c_void]),
    "nerf_amd_weighted_dot_loss": (C.c_int, [c_void, c_void, c_void, i64, C.c_int, C.c_float, c_void, c_void, c_void]),
    "nerf_amd_weighted_dot_loss_backward": (C.c_int, [c_void, c_void, c_void, c_void, i64, C.c_int, C.c_float, c_void, c_void, c_void, c_void]),
    "nerf_amd_get_bounds": (C.c_int, [c_void, c_void, i64, C.c_int, C.c_int, c_void, c_void]),
    "nerf_amd_train_dump_bytes": (C.c_size_t, [C.c_int, C.c_int, i64]),
    "nerf_amd_proposal_forward_train": (C.c_int, [c_void, C.c_int, C.POINTER(Samples), c_void, c_void, c_void]),
    "nerf_amd_mip_forward_train": (C.c_int, [c_void, C.c_int, C.POINTER(Samples), c_void, c_void, c_void]),
    "nerf_amd_train_dump_to_rows": (C.c_int, [c_void, C.c_int, C.c_int, i64, C.c_int, C.c_int, c_void, c_void]),
    "nerf_amd_relu_mask": (C.c_int, [c_void, c_void, C.c_int, i64, c_void]),
    "nerf_amd_relu_mask_bias_partials": (i64, [C.c_int, i64, C.c_int]),
    "nerf_amd_relu_mask_bias": (C.c_int, [c_void, c_void, C.c_int, i64, C.c_int, c_void, c_void]),
    "nerf_amd_packed_backward_bytes": (C.c_size_t, [C.c_int, C.c_int]),
    "nerf_amd_pack_weights_backward": (C.c_int, [C.c_int, C.c_int, C.POINTER(c_void), C.c_int, c_void, c_void]),
    "nerf_amd_proposal_backward_chain": (C.c_int, [c_void, C.c_int, c_void, i64, c_void, c_void, c_void]),
    "nerf_amd_mip_backward_chain": (C.c_int, [c_void, C.c_int, c_void, c_void, i64, c_void, c_void, c_void]),
    "nerf_amd_weight_grads_workspace_bytes": (C.c_size_t, [C.c_int, C.c_int, i64]),
    "nerf_amd_proposal_weight_grads": (C.c_int, [C.c_int, i64, c_void, c_void, C.POINTER(c_void), C.POINTER(c_void), c_void, c_void]),
    "nerf_amd_mip_weight_grads": (C.c_int, [C.c_int, i64, c_void, c_void, C.POINTER(c_void), C.POINTER(c_void), C.POINTER(c_void),
                                           C.POINTER(c_void), c_void, c_void]),
    "nerf_amd_ref_forward_train_dump": (C.c_int, [c_void, C.c_int, C.POINTER(Samples), C.c_int, c_void, c_void, c_void, c_void, c_void, c_void]),
    "nerf_amd_ref_forward_train_dump_rng": (C.c_int, [c_void, C.c_int, C.POINTER(Samples), C.c_int, C.c_uint64, c_void, C.c_float, c_void, c_void, c_void,
                                                      c_void, c_void]),
    "nerf_amd_philox_normal": (C.c_int, [c_void, i64, C.c_uint64, c_void, C.c_float, i64, c_void]),
    "nerf_amd_density_grad_workspace_bytes": (C.c_size_t, [C.c_int, C.c_int, i64]),
    "nerf_amd_density_grad": (C.c_int, [C.c_int, c_void, C.c_int, i64, c_void, c_void, C.c_int, c_void, C.c_int, c_void, c_void, c_void]),
    "nerf_amd_ref_backward_workspace_bytes": (C.c_size_t, [C.c_int, i64]),
    "nerf_amd_ref_backward": (C.c_int, [c_void, C.c_int, C.c_int, i64, c_void, c_void, c_void, C.c_int, c_void, C.c_int, c_void, C.POINTER(c_void),
                                       C.POINTER(c_void), c_void, c_void]),
    "nerf_amd_adam_step": (C.c_int, [C.POINTER(c_void), C.POINTER(c_void), C.POINTER(c_void), C.POINTER(c_void), C.POINTER(i64), C.c_int, c_void,
                                    C.c_double, c_void, C.c_double, C.c_double, C.c_double, C.c_float, c_void]),
    "nerf_amd_sigma_to_weights_backward": (C.c_int, [c_void, c_void, c_void, i64, C.c_int, C.c_int, c_void, c_void, c_void]),
    "nerf_amd_composite_backward": (C.c_int, [c_void, c_void, C.c_int, c_void, C.c_int, i64, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float,
                                              C.c_float, c_void, c_void, c_void, c_void, c_void]),
    "nerf_amd_max_blur_backward": (C.c_int, [c_void, c_void, i64, C.c_int, c_void, c_void]),
    "nerf_amd_get_bounds_backward": (C.c_int, [c_void, c_void, i64, C.c_int, C.c_int, c_void, c_void]),
    "nerf_amd_render_workspace_bytes": (C.c_size_t, [i64, C.c_int]),
    "nerf_amd_render_rays": (C.c_int, [c_void, c_void, C.c_int, c_void, C.POINTER(Samples), i64, c_void, c_void, c_void, i64,
                                       C.c_int, C.c_float, C.c_float, C.c_int, c_void, c_void, c_void, c_void, c_void]),
    "nerf_amd_gemm_workspace_bytes": (C.c_size_t, [i64, i64, i64]),
    "nerf_amd_gemm": (C.c_int, [C.c_int, i64, i64, i64, c_void, i64, i64, c_void, i64, i64, c_void, i64, c_void, C.c_int, c_void, i64, c_void, c_void]),
    "nerf_amd_sigmoid_backward": (C.c_int, [c_void, i64, c_void, i64, i64, C.c_int, c_void, i64, c_void]),
    "nerf_amd_ref_dir_inputs": (C.c_int, [c_void, i64, c_void, i64, i64, C.c_int, c_void, c_void, i64, c_void, c_void]),
    "nerf_amd_ref_dir_inputs_backward": (C.c_int, [c_void, i64, c_void, i64, i64, C.c_int, c_void, c_void, i64, c_void, i64, c_void, i64, c_void]),
    "nerf_amd_ref_combine": (C.c_int, [c_void, i64, c_void, i64, i64, C.c_int, c_void, c_void]),
    "nerf_amd_ref_combine_backward": (C.c_int, [c_void, i64, c_void, i64, c_void, i64, i64, C.c_int, c_void, i64, c_void, i64, c_void]),
    "nerf_amd_positional_encoding_backward": (C.c_int, [c_void, i64, c_void, i64, i64, C.c_int, C.c_int, c_void, c_void]),
    "nerf_amd_contract_positions": (C.c_int, [c_void, i64, i64, c_void, i64, c_void, c_void]),
    "nerf_amd_add_rows": (C.c_int, [c_void, i64, c_void, i64, i64, C.c_int, c_void]),
    "nerf_amd_rows_gemm": (C.c_int, [i64, i64, i64, c_void, i64, c_void, i64, i64, c_void, C.c_int, c_void, i64, C.c_int, c_void]),
    "nerf_amd_rows_to_bf16": (C.c_int, [c_void, i64, i64, i64, C.c_int, C.c_int, c_void, i64, c_void]),
}


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "nerf_amd: %s is missing. Build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C nerf_amd/csrc`). There is no CPU/torch fallback." % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)                 # AttributeError if the .so does not export it
        fn.restype = res
        fn.argtypes = args
    got = lib.nerf_amd_version()
    if got != EXPECTED_VERSION and not os.environ.get("NERF_AMD_LIB"):      # a stale git-ignored .so would be called with shifted arguments
        raise ImportError("nerf_amd: %s is ABI version %d, this package expects %d -- rebuild it (`make -C nerf_amd/csrc`)" % (LIB_PATH, got, EXPECTED_VERSION))
    return lib


lib = _load()


class NerfAmdError(RuntimeError):
    pass


def check(rc: int, what: str = ""):
    if rc != 0:
        msg = lib.nerf_amd_last_error()
        raise NerfAmdError("%s failed (%d): %s" % (what or "nerf_amd call", rc, msg.decode() if msg else "?"))
