"""ctypes binding of libmfvi_hip.so (include/mfvi_hip.h).  There is no CPU fallback: if the
HIP library is missing the import of anything that computes fails loudly."""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MFVI_LIB_PATH") or os.path.join(HERE, "libmfvi_hip.so")      # MFVI_LIB_PATH: an experimental build (A/B timing)

OP_CONV, OP_CONCAT_UP, OP_CONV_LRT = 1, 2, 3
DOMAIN_EPS, DOMAIN_INPUT, DOMAIN_INIT, DOMAIN_UNIFORM, DOMAIN_SGLD, DOMAIN_DROPOUT, DOMAIN_ROUND, DOMAIN_LRT = 0, 1, 2, 3, 4, 5, 6, 7
PARAM_F32, PARAM_BF16 = 0, 1


class TensorDesc(C.Structure):
    _fields_ = [("C", C.c_int32), ("H", C.c_int32), ("W", C.c_int32), ("has_bn", C.c_int32), ("has_act", C.c_int32),
                ("slope", C.c_float), ("eps", C.c_float), ("drop_p", C.c_float), ("bn_off", C.c_int64)]


class OpDesc(C.Structure):
    _fields_ = [("type", C.c_int32), ("in0", C.c_int32), ("in1", C.c_int32), ("out", C.c_int32), ("ksize", C.c_int32),
                ("stride", C.c_int32), ("layer_id", C.c_int32), ("up_mode", C.c_int32), ("w_off", C.c_int64), ("b_off", C.c_int64)]


class MfviError(RuntimeError):
    pass


_lib = None

# name -> (restype, argtypes); every symbol declared in include/mfvi_hip.h
_P, _I, _I64, _U32, _U64, _F = C.c_void_p, C.c_int, C.c_int64, C.c_uint32, C.c_uint64, C.c_float
SIGNATURES = {
    "mfvi_plan_create": (_I, [_P, _I, _P, _I, _I, _I, _I64, _I64, _I, _P]),
    "mfvi_plan_destroy": (None, [_P]),
    "mfvi_plan_workspace_bytes": (_I64, [_P]),
    "mfvi_plan_set_dropout": (_I, [_P, _I]),
    "mfvi_plan_set_param_dtype": (_I, [_P, _I]),
    "mfvi_plan_bn_update_running": (_I, [_P, _P, _I, _F, _P, _P]),
    "mfvi_plan_set_bn_eval": (_I, [_P, _P]),
    "mfvi_plan_set_side_stream": (_I, [_P, _I]),
    "mfvi_plan_set_step_source": (_I, [_P, _P]),
    "mfvi_plan_set_capture_mode": (_I, [_P, _I]),
    "mfvi_plan_set_grad_split": (_I, [_P, _I, _P]),
    "mfvi_plan_grad_split_offset": (_I, [_P, _I, _P]),
    "mfvi_forward": (_I, [_P, _P, _P, _P, _P, _U64, _U32, _U32, _I, _I, _P, _P, _P]),
    "mfvi_backward": (_I, [_P, _P, _P, _P, _P, _U64, _U32, _U32, _I, _I, _P, _P, _P, _P, _P, _P, _P]),
    "mfvi_plan_read_tensor": (_I, [_P, _P, _I, _I, _I, _P, _P]),
    "mfvi_plan_profile": (_I, [_P, _I, _I, _I]),
    "mfvi_plan_profile_read": (_I, [_P, _I, _P, _P, _P, _P]),
    "mfvi_plan_autotune": (_I, [_P, _P, _P, _P, _P, _I, _P, _P, _P, _P]),
    "mfvi_plan_get_tune": (_I, [_P, _I, _I]),
    "mfvi_plan_last_kernel": (_I, [_P, _I, _I]),
    "mfvi_plan_set_tune": (_I, [_P, _I, _I, _I]),
    "mfvi_gaussian_nll": (_I, [_P, _P, _I, _I, _I, _I, _F, _P, _P, _P]),
    "mfvi_gaussian_nll_inpainting": (_I, [_P, _P, _P, _I, _I, _I, _I, _F, _P, _P, _P]),
    "mfvi_gaussian_nll_tensors": (_I, [_P, _P, _P, _P, _I, _I, _I, _I64, _I, _P, _P]),
    "mfvi_gaussian_nll_tensors_backward": (_I, [_P, _P, _P, _P, _I, _I, _I, _I64, _I, _P, _P, _P, _P]),
    "mfvi_radon_mse": (_I, [_P, _P, _P, _I, _I, _I, _I, _F, _P, _P, _P, _P]),
    "mfvi_radon_forward": (_I, [_P, _P, _I, _I, _I, _I, _P, _P]),
    "mfvi_radon_adjoint": (_I, [_P, _P, _I, _I, _I, _I, _P, _P]),
    "mfvi_kl": (_I, [_P, _P, _I64, _F, _F, _P, _P]),
    "mfvi_kl_backward": (_I, [_P, _P, _I64, _F, _F, _F, _P, _P, _P]),
    "mfvi_adam_step": (_I, [_P, _P, _P, _P, _I64, _F, _F, _F, _F, _I, _P]),
    "mfvi_elbo_update_scratch_bytes": (_I64, []),
    "mfvi_elbo_update": (_I, [_P, _P, _P, _P, _I64, _I64, _F, _F, _F, _F, _F, _F, _F, _I, _P, _P, _P]),
    "mfvi_elbo_update_guarded": (_I, [_P, _P, _P, _P, _I64, _I64, _F, _F, _F, _F, _F, _F, _F, _P, _P, _P, _P, _P, _P]),
    "mfvi_adamw_step_guarded": (_I, [_P, _P, _P, _P, _I64, _F, _F, _F, _F, _P, _P, _P, _F, _P]),
    "mfvi_step_advance": (_I, [_P, _P, _P, _P]),
    "mfvi_decimate": (_I, [_P, _I, _I, _I, _P, _P]),
    "mfvi_elbo_update_bf16": (_I, [_P, _P, _P, _P, _P, _P, _I64, _I64, _F, _F, _F, _F, _F, _F, _F, _I, _U64, _P, _P, _P]),
    "mfvi_bf16_to_f32": (_I, [_P, _I64, _P, _P]),
    "mfvi_f32_to_bf16": (_I, [_P, _I64, _P, _P]),
    "mfvi_adamw_step": (_I, [_P, _P, _P, _P, _I64, _F, _F, _F, _F, _I, _F, _P]),
    "mfvi_mse_sigmoid_masked": (_I, [_P, _P, _P, _I, _I, _I, _I, _F, _P, _P, _P]),
    "mfvi_mse_channel": (_I, [_P, _P, _I, _I, _I, _I, _I, _I, _F, _P, _P, _P]),
    "mfvi_uniform_fill_range": (_I, [_U64, _U32, _U32, _U32, _I64, _F, _F, _P, _P]),
    "mfvi_add_normal": (_I, [_P, _U64, _U32, _U32, _I64, _F, _P]),
    "mfvi_normal_fill": (_I, [_U64, _U32, _U32, _U32, _U32, _I64, _F, _F, _P, _P]),
    "mfvi_uniform_fill": (_I, [_U64, _U32, _U32, _U32, _I64, _F, _P, _P]),
    "mfvi_perturb_input": (_I, [_P, _U64, _U32, _I64, _F, _P, _P]),
    "mfvi_perturb_input_dev": (_I, [_P, _U64, _P, _U32, _I64, _F, _P, _P]),
    "mfvi_sq_err_sum": (_I, [_P, _P, _I64, _P, _P]),
    "mfvi_ssim_sum": (_I, [_P, _P, _I, _I, _P, _P]),
    "mfvi_bookkeep": (_I, [_P, _I, _I, _I, _I, _P, _F, _I, _P, _P, _P, _P, _P, _P]),
    "mfvi_bookkeep_inpainting": (_I, [_P, _I, _I, _I, _P, _P, _I, _P, _F, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "mfvi_ring_stats": (_I, [_P, _I, _I, _I, _P, _P, _P]),
    "mfvi_post_step": (_I, [_P, _I, _I, _I, _I, _P, _F, _I, _P]),
    "mfvi_last_error": (C.c_char_p, []),
    "mfvi_abi_version": (_I, []),
}


def lib():
    """Load the HIP library (once).  Raises MfviError if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise MfviError("libmfvi_hip.so is missing (%s): build it with `python __graft_entry__.py` or "
                            "`python mfvi-dip-mia_amd/_build.py`; there is no CPU fallback" % LIB_PATH)
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype, fn.argtypes = res, args
        _lib = L
    return _lib


def check(rc):
    if rc != 0:
        raise MfviError("libmfvi_hip error %d: %s" % (rc, lib().mfvi_last_error().decode()))


def ptr(t):
    """Device pointer of a torch tensor (or None)."""
    return None if t is None else C.c_void_p(t.data_ptr())


def stream_ptr():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)
