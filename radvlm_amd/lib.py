"""ctypes binding of libradvlm_hip.so (the C ABI declared in include/radvlm_hip.h).

The product path has no fallback: if the shared library is missing or a kernel returns an error code, this
module raises.  Tensors are plain torch CUDA(HIP) tensors used only as device memory; every call is
asynchronous on the current torch stream.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libradvlm_hip.so")

_c_void_p = ctypes.c_void_p
_i64 = ctypes.c_int64
_i32 = ctypes.c_int
_f32 = ctypes.c_float

ACT_NONE, ACT_QUICK_GELU, ACT_GELU, ACT_GELU_TANH = 0, 1, 2, 3

_SIGS = {
    "rv_gemm_nt_bf16": [_c_void_p, _i64, _c_void_p, _i64, _c_void_p, _i64, _c_void_p, _c_void_p, _i64, _i32, _i32, _i32,
                        _i32, _i32, _i32, _c_void_p, _c_void_p],
    "rv_gemm_bf16": [_c_void_p, _i64, _c_void_p, _i64, _c_void_p, _i64, _c_void_p, _c_void_p, _i64, _i32, _i32, _i32, _i32, _i32,
                     _f32, _i32, _i32, _i32, _c_void_p, _c_void_p],
    "rv_gemm_bf16_ex": [_c_void_p, _i64, _c_void_p, _i64, _c_void_p, _i64, _c_void_p, _c_void_p, _i64, _i32, _i32, _i32, _i32, _i32,
                        _f32, _i32, _i32, _i32, _c_void_p, _i64, _c_void_p, _i64, _i32, _c_void_p, _i64, _c_void_p, _c_void_p],
    "rv_gemm_rope_bf16": [_c_void_p, _i64, _c_void_p, _i64, _c_void_p, _i64, _c_void_p, _i32, _i32, _i32, _c_void_p, _c_void_p, _i32, _i32, _i32,
                          _c_void_p, _i64, _c_void_p, _c_void_p],
    "rv_gemm_swiglu_fwd_bf16": [_c_void_p, _i64, _c_void_p, _i64, _c_void_p, _i64, _c_void_p, _i64, _i32, _i32, _i32, _c_void_p, _i64, _c_void_p,
                                _c_void_p],
    "rv_gemm_swiglu_bwd_bf16": [_c_void_p, _i64, _c_void_p, _i64, _c_void_p, _i64, _c_void_p, _i64, _c_void_p, _i64, _i32, _i32, _i32, _c_void_p,
                                _i64, _c_void_p, _c_void_p],
    "rv_dropout_bf16": [_c_void_p, _c_void_p, _i64, _f32, ctypes.c_uint64, _c_void_p],
    "rv_dropout_add_bf16": [_c_void_p, _c_void_p, _i64, _f32, ctypes.c_uint64, _c_void_p],
    "rv_gemm_dropout_add_bf16": [_c_void_p, _i64, _c_void_p, _i64, _c_void_p, _i64, _i32, _i32, _i32, _i32, _f32, _f32, ctypes.c_uint64, _i32, _c_void_p,
                                 _c_void_p],
    "rv_lora_a_grad_bf16": [_c_void_p, _i64, _c_void_p, _i64, _c_void_p, _i64, _i32, _i32, _i32, _f32, ctypes.c_uint64, _i32, _c_void_p, _i64, _c_void_p],
    "rv_lora_down_bf16": [_c_void_p, _i64, _c_void_p, _i64, _c_void_p, _i64, _i32, _i32, _i32, _f32, _f32, ctypes.c_uint64, _c_void_p, _c_void_p],
    "rv_transpose_bf16": [_c_void_p, _i64, _i64, _i64, _c_void_p, _i64, _i64, _i64, _i32, _i32, _i32, _i32, _i32, _i32, _c_void_p],
    "rv_rmsnorm_fwd": [_c_void_p, _c_void_p, _c_void_p, _c_void_p, _i32, _i32, _f32, _c_void_p],
    "rv_rmsnorm_bwd": [_c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_void_p, _i32, _c_void_p, _i32, _i32, _i32, _c_void_p],
    "rv_layernorm_fwd": [_c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_void_p, _i32, _i32, _f32, _c_void_p],
    "rv_layernorm_bwd": [_c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_void_p, _i32, _c_void_p, _i32, _i32, _i32, _c_void_p],
    "rv_quick_gelu_fwd": [_c_void_p, _c_void_p, _i64, _c_void_p],
    "rv_quick_gelu_bwd": [_c_void_p, _c_void_p, _c_void_p, _i64, _c_void_p],
    "rv_colsum_f32": [_c_void_p, _i32, _i32, _c_void_p, _i32, _c_void_p],
    "rv_colsum_partial_bf16": [_c_void_p, _i64, _i32, _i32, _c_void_p, _i32, _c_void_p],
    "rv_rope_inplace": [_c_void_p, _i64, _c_void_p, _i32, _i32, _i32, _i32, _i32, _i32, _c_void_p],
    "rv_attn_fwd": [_c_void_p, _i64, _c_void_p, _i64, _c_void_p, _c_void_p, _i64, _c_void_p, _c_void_p, _i32, _i32, _i32, _i32,
                    _i32, _i32, _f32, _c_void_p, _c_void_p],
    "rv_attn_bwd": [_c_void_p, _i64, _c_void_p, _i64, _c_void_p, _i64, _c_void_p, _i64, _c_void_p, _i64, _c_void_p, _c_void_p,
                    _c_void_p, _c_void_p, _c_void_p, _c_void_p, _i64, _c_void_p, _i64, _c_void_p, _i64, _c_void_p, _i32, _i32,
                    _i32, _i32, _i32, _i32, _f32, _c_void_p, _c_void_p],
    "rv_attn_fwd_gqa": [_c_void_p, _i64, _c_void_p, _i64, _c_void_p, _c_void_p, _i64, _c_void_p, _c_void_p, _c_void_p, _i32, _i32, _i32,
                        _i32, _i32, _i32, _i32, _f32, _c_void_p, _c_void_p],
    "rv_attn_bwd_gqa": [_c_void_p, _i64, _c_void_p, _i64, _c_void_p, _i64, _c_void_p, _i64, _c_void_p, _i64, _c_void_p, _c_void_p,
                        _c_void_p, _c_void_p, _c_void_p, _c_void_p, _i64, _c_void_p, _i64, _c_void_p, _i64, _c_void_p, _c_void_p, _i32,
                        _i32, _i32, _i32, _i32, _i32, _i32, _i32, _f32, _c_void_p, _i64, _c_void_p, _c_void_p],
    "rv_attn_fwd_nat": [_c_void_p, _i64, _c_void_p, _i64, _c_void_p, _i64, _c_void_p, _i64, _c_void_p, _c_void_p, _c_void_p, _i32, _i32, _i32,
                        _i32, _i32, _i32, _i32, _f32, _c_void_p, _c_void_p],
    "rv_attn_bwd_nat": [_c_void_p, _i64, _c_void_p, _i64, _c_void_p, _i64, _c_void_p, _i64, _c_void_p, _i64, _c_void_p, _c_void_p, _c_void_p, _i64,
                        _c_void_p, _i64, _c_void_p, _i64, _c_void_p, _c_void_p, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _f32, _c_void_p, _i64,
                        _c_void_p, _c_void_p, _c_void_p, _c_void_p],
    "rv_attn_bwd_gqa_rope": [_c_void_p, _i64, _c_void_p, _i64, _c_void_p, _i64, _c_void_p, _i64, _c_void_p, _i64, _c_void_p, _c_void_p,
                             _c_void_p, _c_void_p, _c_void_p, _c_void_p, _i64, _c_void_p, _i64, _c_void_p, _i64, _c_void_p, _c_void_p, _i32,
                             _i32, _i32, _i32, _i32, _i32, _i32, _i32, _f32, _c_void_p, _i64, _c_void_p, _c_void_p, _c_void_p, _c_void_p],
    "rv_transpose_bf16_varlen": [_c_void_p, _i64, _c_void_p, _i64, _c_void_p, _i64, _i64, _i64, _i32, _i32, _i32, _i32, _i32, _i32,
                                 _c_void_p],
    "rv_rope_inplace_pos": [_c_void_p, _i64, _c_void_p, _c_void_p, _i32, _i32, _i32, _i32, _i32, _c_void_p],
    "rv_gelu_tanh_fwd": [_c_void_p, _c_void_p, _i64, _c_void_p],
    "rv_gelu_tanh_bwd": [_c_void_p, _c_void_p, _c_void_p, _i64, _c_void_p],
    "rv_weighted_segment_sum_rows": [_c_void_p, _i64, _c_void_p, _c_void_p, _c_void_p, _c_void_p, _i32, _c_void_p, _i64, _i32,
                                     _c_void_p],
    "rv_max4_rows_fwd": [_c_void_p, _i64, _c_void_p, _c_void_p, _i32, _c_void_p, _i64, _c_void_p, _i32, _c_void_p],
    "rv_max4_rows_bwd": [_c_void_p, _i64, _c_void_p, _c_void_p, _i32, _c_void_p, _c_void_p, _i64, _i32, _c_void_p],
    "rv_add_pos_rows": [_c_void_p, _c_void_p, _i32, _i32, _i32, _c_void_p],
    "rv_swiglu_fwd": [_c_void_p, _i64, _c_void_p, _i64, _i32, _i32, _c_void_p],
    "rv_swiglu_bwd": [_c_void_p, _i64, _c_void_p, _i64, _c_void_p, _i64, _i32, _i32, _c_void_p],
    "rv_gelu_fwd": [_c_void_p, _c_void_p, _i64, _c_void_p],
    "rv_gelu_bwd": [_c_void_p, _c_void_p, _c_void_p, _i64, _c_void_p],
    "rv_cross_entropy": [_c_void_p, _i64, _c_void_p, _c_void_p, _c_void_p, _i64, _i32, _i32, _f32, _c_void_p],
    "rv_sum_f32": [_c_void_p, _i64, _f32, _c_void_p, _c_void_p],
    "rv_gather_rows": [_c_void_p, _i64, _c_void_p, _i64, _c_void_p, _i64, _c_void_p, _i32, _i32, _c_void_p],
    "rv_segment_sum_rows": [_c_void_p, _i64, _c_void_p, _c_void_p, _c_void_p, _i32, _c_void_p, _i64, _i32, _c_void_p],
    "rv_normalize_tiles_u8": [_c_void_p, _c_void_p, _i32, _i32, _i32, _i32, _i32, ctypes.c_double, ctypes.POINTER(_f32), ctypes.POINTER(_f32), _c_void_p],
    "rv_im2col_patches": [_c_void_p, _c_void_p, _i32, _i32, _i32, _i32, _i32, _c_void_p],
    "rv_clip_embed": [_c_void_p, _c_void_p, _c_void_p, _c_void_p, _i32, _i32, _i32, _c_void_p],
    "rv_adamw": [_c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_void_p, _i64, _f32, _f32, _f32, _f32, _f32, _f32, _f32,
                 _c_void_p, _c_void_p],
    "rv_sumsq_partial_bf16": [_c_void_p, _i64, _c_void_p, _i32, _c_void_p],
    "rv_clip_coef": [_c_void_p, _i32, _f32, _c_void_p, _c_void_p],
    "rv_cast_f32_to_bf16": [_c_void_p, _c_void_p, _i64, _c_void_p],
    "rv_cast_bf16_to_f32": [_c_void_p, _c_void_p, _i64, _c_void_p],
    "rv_add_bf16": [_c_void_p, _c_void_p, _c_void_p, _i64, _c_void_p],
}

EXPORTED_SYMBOLS = ["rv_version", "rv_gemm_select_kernel", "rv_gemm_set_cu_budget", "rv_attn_select_kernel", "rv_attn_fwd_nat_pairs"] + sorted(_SIGS)

_lib = None


class RadvlmHipError(RuntimeError):
    pass


def load():
    """Load the shared library (no GPU needed to load); raises if it has not been built."""
    global _lib
    if _lib is None:
        path = os.environ.get("RADVLM_HIP_LIB", _LIB_PATH)     # override: A/B of two builds (tools/), never a fallback
        if not os.path.exists(path):
            raise RadvlmHipError(f"{path} not found: build it with radvlm_amd/csrc/build.sh "
                                 "(or __graft_entry__.build()); there is no CPU fallback")
        lib = ctypes.CDLL(path)
        lib.rv_version.restype = ctypes.c_char_p
        for name, sig in _SIGS.items():
            fn = getattr(lib, name)
            fn.argtypes = sig
            fn.restype = _i32
        _lib = lib
    return _lib


def _ptr(t):
    if t is None:
        return None
    if isinstance(t, int):
        return t
    assert t.is_cuda, "radvlm_amd kernels take device tensors only"
    return t.data_ptr()


def _stream():
    return torch.cuda.current_stream().cuda_stream


def call(name, *args):
    lib = load()
    rc = getattr(lib, name)(*[(_ptr(a) if (a is None or isinstance(a, torch.Tensor)) else a) for a in args], _stream())
    if rc != 0:
        raise RadvlmHipError(f"{name} failed with code {rc}")


_zeros = {}


def zeros16(device=None):
    device = torch.device(device or torch.cuda.current_device())
    key = (device.type, device.index)
    if key not in _zeros:
        _zeros[key] = torch.zeros(64, dtype=torch.uint8, device=device)
    return _zeros[key]
