"""ctypes binding of libvae_hip.so (the C ABI declared in include/vae_hip.h).

There is no fallback: if the shared library is missing or fails to load, importing the
kernels raises, and every non-zero return code becomes a RuntimeError carrying
vae_last_error().
"""
from __future__ import annotations

import ctypes
import os
from ctypes import c_char_p, c_double, c_float, c_int, c_long, c_uint, c_ulonglong, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libvae_hip.so")

P = c_void_p

# name -> (restype, argtypes); restype None means "int status, raise on non-zero"
SIGNATURES = {
    "vae_version": (c_int, []),
    "vae_last_error": (c_char_p, []),
    "vae_set_tuning": (None, [c_char_p, c_int]),
    "vae_set_debug_buffer": (None, [P]),
    "vae_prof_start": (None, [c_int]),
    "vae_prof_stop": (c_int, [P, P, P, P, P, c_int]),
    "vae_recon_loss_grid": (c_int, [c_long]),
    "vae_recon_loss_fwd_bwd": (None, [P, P, P, P, c_long, c_int, c_float, P]),
    "vae_recon_loss_static_max_elems": (c_long, []),
    "vae_recon_loss_fwd_bwd_static": (None, [P, P, P, P, c_long, c_int, c_float, P]),
    "vae_elbo_combine": (None, [P, c_int, P, c_int, c_float, c_float, P, P]),
    "vae_reparam_grid": (c_int, [c_long]),
    "vae_reparam_kl_fwd": (None, [P, P, P, P, P, P, P, c_long, c_ulonglong, c_uint, c_uint, P]),
    "vae_reparam_kl_fwd_slabs": (None, [P, c_int, P, P, P, P, P, P, P, P, c_long, c_int, c_ulonglong, c_uint, c_uint, P]),
    "vae_reparam_kl_bwd_slabs": (None, [P, c_int, P, P, P, P, P, c_float, P, P, P, c_long, c_int, c_ulonglong, c_uint,
                                        c_uint, P]),
    "vae_reparam_kl_bwd": (None, [P, P, P, P, P, P, c_float, P, P, P, c_long, c_int, c_ulonglong,
                                  c_uint, c_uint, P]),
    "vae_philox_fill": (None, [P, c_long, c_int, c_double, c_ulonglong, c_uint, c_uint, c_uint, P]),
    "vae_dropout_threshold": (c_uint, [c_double]),
    "vae_gemm_rows_per_block": (c_int, []),
    "vae_gemm_stat_rows": (c_int, [c_long, c_int]),
    "vae_conv3x3_c8_stat_rows": (c_int, [c_int, c_int, c_int, c_int]),
    "vae_conv3x3_c8_twice_ok": (c_int, [c_int, c_int, c_int, c_int]),
    "vae_conv3x3_c8_stats": (None, [P, P, P, P, P, c_int, c_int, c_int, c_int, P]),
    "vae_conv3x3_c8_bn_act": (None, [P, P, P, P, P, P, P, c_int, c_int, c_int, c_int, P]),
    "vae_convt2x2_stat_rows": (c_int, [c_int, c_int, c_int, c_int, c_int]),
    "vae_convt2x2_stat_fold": (c_int, [c_int, c_int, c_int, c_int, c_int]),
    "vae_convt2x2_bwd_fused_ok": (c_int, [c_int, c_int, c_int, c_int, c_int]),
    "vae_convt2x2_bwd_fused_slabs": (c_int, [c_int, c_int, c_int]),
    "vae_convt2x2_bwd_fused": (None, [P, P, P, P, P, c_int, c_int, c_int, c_int, c_int, P]),
    "vae_conv3x3_stat_rows": (c_int, [c_int, c_int, c_int, c_int, c_int]),
    "vae_conv3x3_gemm": (None, [P, P, P, P, P, P, c_int, c_int, c_int, c_int, c_int, P]),
    "vae_convt2x2_fwd_gemm": (None, [P, P, P, P, P, P, c_int, c_int, c_int, c_int, c_int, P]),
    "vae_convt2x2_dgrad_gemm": (None, [P, P, P, c_int, c_int, c_int, c_int, c_int, P]),
    "vae_linear_gemm": (None, [P, P, P, P, P, P, P, c_int, c_int, c_int, c_int, P]),
    "vae_linear_splitk": (c_int, [c_int, c_int, c_int]),
    "vae_linear_gemm_splitk": (None, [P, P, P, P, P, P, c_int, c_int, c_int, c_int, c_int, P]),
    "vae_conv3x3_c8_gemm": (None, [P, P, P, P, P, P, c_int, c_int, c_int, c_int, P]),
    "vae_wgrad_ksplit": (c_int, [c_int, c_int, c_int, c_long]),
    "vae_wgrad_gemm": (None, [P, P, P, c_int, c_int, c_long, c_int, c_int, c_int, c_int, c_int, c_int, P]),
    "vae_conv3_wgrad_patch_ok": (c_int, [c_int, c_int, c_int, c_int]),
    "vae_conv3_wgrad_patch_ksplit": (c_int, [c_int, c_int, c_int, c_int, c_int]),
    "vae_conv3_wgrad_patch": (None, [P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, P]),
    "vae_conv3_wgrad_multi_plan": (None, [P, c_int, c_int, P]),
    "vae_conv3_wgrad_patch_multi": (None, [P, c_int, P]),
    "vae_reduce_slabs": (None, [P, c_int, c_long, P, P, P, c_long, c_int, P]),
    "vae_pack_bf16": (None, [P, P, P, P, c_long, P]),
    "vae_bn_finalize": (None, [P, P, c_int, c_int, c_int, c_double, P, P, P, P, c_float, c_float, c_int,
                               P, P, P, P, P]),
    "vae_sum_partials": (None, [P, c_int, c_int, P, c_float, P]),
    "vae_bn_act_fwd": (None, [P, P, P, P, c_int, c_int, c_int, c_int, c_int, c_double, P, c_ulonglong,
                              c_uint, c_uint, c_uint, P]),
    "vae_bn_act_fwd_route": (None, [P, P, P, P, P, c_int, c_int, c_int, c_int, c_int, c_double, P, c_ulonglong,
                                    c_uint, c_uint, c_uint, P]),
    "vae_bn_act_bwd_route": (None, [P, P, P, P, P, P, P, P, P, P, c_int, c_int, c_int, c_int, c_int, c_double, P,
                                    c_ulonglong, c_uint, c_uint, c_uint, P]),
    "vae_bn_act_fwd_route_y": (None, [P, P, P, P, P, P, c_int, c_int, c_int, c_int, c_int, c_double, P, c_ulonglong,
                                      c_uint, c_uint, c_uint, P]),
    "vae_bn_act_bwd_route_y": (None, [P, P, P, P, P, P, P, P, P, P, P, c_int, c_int, c_int, c_int, c_int, c_double, P,
                                      c_ulonglong, c_uint, c_uint, c_uint, P]),
    "vae_bn_act_bwd_blocks": (c_int, [c_int, c_int, c_int, c_int, c_int]),
    "vae_bn_bwd_conv1_scratch_floats": (c_long, [c_int, c_int, c_int]),
    "vae_bn_bwd_conv1": (None, [P, P, P, P, P, P, P, P, P, P, c_int, c_int, c_int, c_int, P]),
    "vae_bn_bwd_conv1_bits": (None, [P, P, P, P, P, P, P, P, P, P, P, c_int, c_int, c_int, c_int, P]),
    "vae_colsum_blocks": (c_int, [c_long, c_int]),
    "vae_colsum_bf16": (None, [P, c_long, c_int, P, P, P]),
    "vae_conv1x1_out_fwd": (None, [P, P, P, P, c_int, c_int, c_int, c_int, P]),
    "vae_conv1x1_out_bwd_blocks": (c_int, [c_int, c_int, c_int]),
    "vae_conv1x1_out_bwd": (None, [P, P, P, P, P, P, c_int, c_int, c_int, c_int, P]),
    "vae_clamp_f32": (None, [P, P, c_float, c_float, P, c_long, P]),
    "vae_maxpool2_fwd": (None, [P, P, P, c_int, c_int, c_int, c_int, P]),
    "vae_maxpool2_bwd": (None, [P, P, P, P, c_int, c_int, c_int, c_int, P]),
    "vae_gn_act_fwd": (None, [P, P, P, P, P, c_int, c_int, c_int, c_int, c_int, c_float, c_int, c_double, P, c_ulonglong,
                              c_uint, c_uint, c_uint, P]),
    "vae_gn_act_bwd": (None, [P, P, P, P, P, P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, c_double, P,
                              c_ulonglong, c_uint, c_uint, c_uint, P]),
    "vae_bn_act_bwd": (None, [P, P, P, P, P, P, P, P, P, c_int, c_int, c_int, c_int, c_int, c_double, P,
                              c_ulonglong, c_uint, c_uint, c_uint, P]),
    "vae_head_fwd": (None, [P, P, P, P, P, P, c_int, c_int, c_int, c_int, P]),
    "vae_head_bwd_blocks": (c_int, [c_int, c_int, c_int]),
    "vae_head_bwd": (None, [P, P, P, P, P, P, P, P, P, P, P, P, c_int, c_int, c_int, c_int, P]),
    "vae_head_bwd_sums": (None, [P, P, P, P, P, P, P, P, P, P, P, P, c_int, c_int, c_int, c_int, P]),
    "vae_head_convt_bwd_fused": (None, [P, P, P, P, P, P, P, P, c_int, P, P, P, P, c_int, c_int, c_int, P]),
    "vae_reduce_slabs_scatter": (None, [P, c_int, c_long, c_long, P, P, P, P, c_long, c_int, P]),
    "vae_multi_pack": (None, [P, c_int, P]),
    "vae_multi_reduce_slabs_scatter": (None, [P, c_int, P]),
    "vae_tanh_bwd_rows": (c_int, [c_int]),
    "vae_tanh_bwd": (None, [P, P, P, P, c_int, c_int, P]),
    "vae_nchw_to_nhwc8": (None, [P, P, c_int, c_int, c_int, c_int, P]),
    "vae_nchw_to_nhwc8_count": (None, [P, P, c_int, c_int, c_int, c_int, P, c_int, P]),
    "vae_feed_u8": (None, [P, c_long, P, P, c_int, c_int, c_int, c_int, P]),
    "vae_adamw_grid": (c_int, [c_long]),
    "vae_adamw_clip_step": (None, [P, P, P, P, c_long, P, c_float, c_float, c_double, c_double, c_double,
                                   c_double, c_double, c_int, P, P]),
    "vae_adamw_clip_step_elbo": (None, [P, P, P, P, c_long, P, c_float, c_float, c_double, c_double, c_double,
                                        c_double, c_double, c_int, P, P, c_int, P, c_int, c_float, c_float, P, P]),
    "vae_permute_f32": (None, [P, P, P, P, c_long, P]),
    "vae_cast_bf16": (None, [P, P, c_long, P]),
    "vae_model_param_count": (c_long, [P]),
    "vae_model_param_offset": (c_long, [P, c_int]),
    "vae_model_bn_count": (c_long, [P]),
    "vae_model_wpack_bytes": (c_long, [P]),
    "vae_model_workspace_bytes": (c_long, [P]),
    "vae_model_pack_weights": (None, [P, P, P, P]),
    "vae_model_forward": (None, [P, P, P, P, P, P, P, P, P, P, P, P]),
    "vae_model_encode": (None, [P, P, P, P, P, P, P, P, P, P]),
    "vae_model_decode": (None, [P, P, P, P, P, P, P, P]),
    "vae_model_kl_partials": (None, [P, P, P, P]),
    "vae_model_backward": (None, [P, P, P, P, P, P, P, P, P, P, P, P, c_float, P, c_int, P]),
}


class ModelCfg(ctypes.Structure):
    """mirror of struct vae_model_cfg (include/vae_hip.h)."""
    _fields_ = [("struct_size", c_uint), ("batch", c_int), ("in_channels", c_int), ("out_channels", c_int), ("latent", c_int),
                ("image_size", c_int), ("train", c_int), ("p_dropout", c_double),
                ("seed", c_ulonglong), ("step", c_uint), ("rank", c_uint), ("side_stream", c_void_p),
                ("precision", c_int), ("batches_tracked", c_void_p)]

    def __init__(self, *args, **kw):
        super().__init__(*args, **kw)
        self.struct_size = ctypes.sizeof(ModelCfg)      # the library compares it with its own sizeof(vae_model_cfg)


class WgradJob(ctypes.Structure):
    """mirror of struct vae_wgrad_job (include/vae_hip.h)."""
    _fields_ = [("dy", c_void_p), ("x", c_void_p), ("slab", c_void_p), ("n_img", c_int), ("H", c_int), ("W", c_int),
                ("Cin", c_int), ("Cout", c_int), ("nsplit", c_int)]


ABI_VERSION = 3                          # VAE_HIP_VERSION of the include/vae_hip.h these signatures were written for
PRECISIONS = {"bf16": 0, "fp32": 1}      # VAE_PRECISION_* of include/vae_hip.h


def long4(*v):
    """a C array of 4 longs (dims / strides arguments)."""
    return (c_long * 4)(*v)

_lib = None


def load():
    """Load libvae_hip.so once and attach the prototypes.  Raises if it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} not found: build it with `make -C vae_amd/csrc` "
            "(or python -c 'import __graft_entry__ as g; g.build()'). There is no CPU fallback.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the symbol is missing
        fn.restype = c_int if res is None else res
        fn.argtypes = args
    built = lib.vae_version()
    if built != ABI_VERSION:
        raise RuntimeError(f"{LIB_PATH} reports ABI revision {built}, this binding was written for {ABI_VERSION}: "
                           "rebuild the library (make -C vae_amd/csrc) from the same tree as vae_amd/_lib.py")
    _lib = lib
    return lib


def ptr(t):
    """device pointer of a tensor (None -> NULL)."""
    return None if t is None else t.data_ptr()


def call(name, *args):
    lib = load()
    rc = getattr(lib, name)(*args)
    if SIGNATURES[name][0] is None and rc != 0:
        raise RuntimeError(f"{name} failed ({rc}): {lib.vae_last_error().decode()}")
    return rc


def current_stream(device=None):
    """raw hipStream_t of torch's current stream (on `device`, default: the current device)."""
    import torch
    return torch.cuda.current_stream(device).cuda_stream


def check_device(t):
    """The C entry points launch on the CURRENT device's stream and never switch devices (one process per GPU,
    torch.cuda.set_device(local_rank) once): a tensor that lives on another GPU would be handed to the wrong
    device's queue.  Raise instead."""
    import torch
    if not t.is_cuda:
        raise RuntimeError("this path runs on an MI355X only (no CPU execution path): move the tensor to 'cuda'")
    cur = torch.cuda.current_device()
    if t.device.index != cur:
        raise RuntimeError(f"tensor lives on cuda:{t.device.index} but the current device is cuda:{cur}: call "
                           f"torch.cuda.set_device({t.device.index}) (one process per GPU) or wrap the call in "
                           f"torch.cuda.device({t.device.index})")
