"""ctypes binding of ``libtransformer_engine.so`` (the ``nvte_*`` C ABI).

This is the Python twin of what the reference does in
``transformer_engine/common/__init__.py:358-360`` (load the core library RTLD_GLOBAL) plus
the part of ``pytorch/csrc`` that turns framework tensors into ``NVTETensor`` handles
(``TensorWrapper``, ref ``common/include/transformer_engine/transformer_engine.h:738-792``).

The library is REQUIRED: importing this module raises when the shared object is missing;
there is no CPU fallback anywhere in the package.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional, Sequence

_HERE = os.path.dirname(os.path.abspath(__file__))
# TE_AMD_LIBRARY: load another build of the same library (diagnostic builds under scratch/)
LIB_PATH = os.environ.get("TE_AMD_LIBRARY") or os.path.join(_HERE, "lib", "libtransformer_engine.so")

# ---- enums (ABI values, include/transformer_engine/transformer_engine.h) -------------
kByte, kInt16, kInt32, kInt64, kFloat32, kFloat16, kBFloat16, kFloat8E4M3, kFloat8E5M2, kFloat8E8M0, kFloat4E2M1 = range(11)
(kRowwiseData, kColumnwiseData, kScale, kAmax, kRowwiseScaleInv, kColumnwiseScaleInv,
 kColumnwiseAmax, kWithGEMMSwizzledScales, kRowScaledNVFP4, kNVFP4E4M3Max) = range(10)
NVTE_DELAYED_TENSOR_SCALING = 0
NVTE_MXFP8_1D_SCALING = 1
(kQForcePow2Scales, kQAmaxEpsilon, kQNoopTensor) = (0, 1, 2)
(kMBiasTensor, kMDBiasTensor, kMWithGELU, kMWithDGELU, kMEpilogueAux, kMUseSplitAccumulator,
 kMSMCount) = range(7)


class NVTEShape(C.Structure):
    _fields_ = [("data", C.c_size_t * 15), ("ndim", C.c_size_t)]


class NVTEBasicTensor(C.Structure):
    _fields_ = [("data_ptr", C.c_void_p), ("dtype", C.c_int), ("shape", NVTEShape)]


class NVTETensorPack(C.Structure):
    _fields_ = [("tensors", C.c_void_p * 10), ("size", C.c_size_t)]


_SHAPES = {}


def make_shape(dims: Sequence[int]) -> NVTEShape:
    """NVTEShape of `dims`.  The same few shapes recur every step: the structs are built once and reused (the
    library copies them, and no caller here writes into one)."""
    key = tuple(dims)
    s = _SHAPES.get(key)
    if s is None:
        if len(key) > 15:
            raise ValueError("NVTEShape holds at most 15 dims")
        s = NVTEShape()
        for i, d in enumerate(key):
            s.data[i] = int(d)
        s.ndim = len(key)
        if len(_SHAPES) < 4096:
            _SHAPES[key] = s
    return s


class NVTEError(RuntimeError):
    """A failed NVTE_CHECK inside the library (same text the reference would throw)."""


def _load() -> C.CDLL:
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C transformer_engine_amd/csrc`). There is no CPU fallback.")
    # torch (if used) must be imported first so that its bundled libamdhip64.so.7 is the one
    # HIP runtime in the process; the dynamic loader then resolves our NEEDED entry to it.
    try:
        import torch  # noqa: F401
    except Exception:  # pragma: no cover - torch-less use is fine
        pass
    return C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)


_L = _load()

_V, _P, _SZ, _I, _F, _B = None, C.c_void_p, C.c_size_t, C.c_int, C.c_float, C.c_bool

_SIGS = {
    # amd_ext.h
    "nvte_amd_set_error_mode": (_V, [_I]),
    "nvte_amd_get_error_mode": (_I, []),
    "nvte_amd_last_error": (C.c_char_p, []),
    "nvte_amd_clear_error": (_V, []),
    "nvte_amd_error_counter": (C.POINTER(C.c_ulonglong), []),
    "nvte_amd_version": (C.c_char_p, []),
    "nvte_amd_multi_gemm": (_V, [_P, _P, _P, _P, _P, _I, _I, _P, _P, _P]),
    "nvte_amd_set_columnwise_transposed": (_V, [_P, _I]),
    "nvte_amd_set_grouped_columnwise_transposed": (_V, [_P, _I]),
    "nvte_swizzle_grouped_scaling_factors": (_V, [_P, _P, _P]),
    "nvte_unswizzle_grouped_scaling_factors": (_V, [_P, _P, _P]),
    "nvte_group_gelu": (_V, [_P, _P, _P]),
    "nvte_group_dgelu": (_V, [_P, _P, _P, _P]),
    "nvte_group_quantize_dbias_dgelu": (_V, [_P, _P, _P, _P, _P, _P]),
    "nvte_group_silu": (_V, [_P, _P, _P]),
    "nvte_group_dsilu": (_V, [_P, _P, _P, _P]),
    "nvte_group_quantize_dbias_dsilu": (_V, [_P, _P, _P, _P, _P, _P]),
    "nvte_group_relu": (_V, [_P, _P, _P]),
    "nvte_group_drelu": (_V, [_P, _P, _P, _P]),
    "nvte_group_quantize_dbias_drelu": (_V, [_P, _P, _P, _P, _P, _P]),
    "nvte_group_qgelu": (_V, [_P, _P, _P]),
    "nvte_group_dqgelu": (_V, [_P, _P, _P, _P]),
    "nvte_group_quantize_dbias_dqgelu": (_V, [_P, _P, _P, _P, _P, _P]),
    "nvte_group_srelu": (_V, [_P, _P, _P]),
    "nvte_group_dsrelu": (_V, [_P, _P, _P, _P]),
    "nvte_group_quantize_dbias_dsrelu": (_V, [_P, _P, _P, _P, _P, _P]),
    "nvte_group_quantize_dbias": (_V, [_P, _P, _P, _P, _P, _P]),
    "nvte_amd_get_grouped_columnwise_transposed": (_I, [_P]),
    "nvte_amd_get_columnwise_transposed": (_I, [_P]),
    "nvte_amd_delayed_scaling_after_reduction": (_V, [_P, C.POINTER(_P), C.POINTER(_P), _SZ, C.c_char_p, _I, _F, _P]),
    "nvte_amd_delayed_scaling_after_reduction_multi": (_V, [_P, C.POINTER(_P), C.POINTER(_P), C.POINTER(_I), _SZ, C.c_char_p, _F, _P]),
    # transformer_engine.h
    "nvte_create_tensor": (_P, [_I]),
    "nvte_create_tensors": (_V, [_I, C.POINTER(_P), _SZ]),
    "nvte_destroy_tensor": (_V, [_P]),
    "nvte_destroy_tensors": (_V, [C.POINTER(_P), _SZ]),
    "nvte_tensor_data": (_P, [_P]),
    "nvte_tensor_columnwise_data": (_P, [_P]),
    "nvte_make_shape": (NVTEShape, [C.POINTER(_SZ), _SZ]),
    "nvte_tensor_shape": (NVTEShape, [_P]),
    "nvte_tensor_columnwise_shape": (NVTEShape, [_P]),
    "nvte_tensor_ndims": (_SZ, [_P]),
    "nvte_tensor_size": (_SZ, [_P, _SZ]),
    "nvte_tensor_size_bytes": (_SZ, [_P]),
    "nvte_tensor_numel": (_SZ, [_P]),
    "nvte_tensor_element_size": (_SZ, [_P]),
    "nvte_tensor_element_size_bits": (_SZ, [_P]),
    "nvte_tensor_type": (_I, [_P]),
    "nvte_tensor_amax": (_P, [_P]),
    "nvte_tensor_scale": (_P, [_P]),
    "nvte_tensor_scale_inv": (_P, [_P]),
    "nvte_tensor_scale_inv_shape": (NVTEShape, [_P]),
    "nvte_zero_tensor": (_V, [_P, _P]),
    "nvte_set_tensor_param": (_V, [C.POINTER(_P), _I, C.POINTER(NVTEBasicTensor)]),
    "nvte_get_tensor_param": (NVTEBasicTensor, [_P, _I]),
    "nvte_set_tensor_param_v2": (_V, [_P, _I, _P, _SZ]),
    "nvte_get_tensor_param_v2": (_V, [_P, _I, _P, _SZ, C.POINTER(_SZ)]),
    "nvte_tensor_scaling_mode": (_I, [_P]),
    "nvte_tensor_pack_create": (_V, [C.POINTER(NVTETensorPack)]),
    "nvte_tensor_pack_destroy": (_V, [C.POINTER(NVTETensorPack)]),
    "nvte_create_quantization_config": (_P, []),
    "nvte_get_quantization_config_attribute": (_V, [_P, _I, _P, _SZ, C.POINTER(_SZ)]),
    "nvte_set_quantization_config_attribute": (_V, [_P, _I, _P, _SZ]),
    "nvte_destroy_quantization_config": (_V, [_P]),
    "nvte_is_non_tn_fp8_gemm_supported": (_I, []),
    "nvte_memset": (_V, [_P, _I, _SZ, _P]),
    # cast.h
    "nvte_quantize": (_V, [_P, _P, _P]),
    "nvte_quantize_noop": (_V, [_P, _P, _P, _P]),
    "nvte_quantize_v2": (_V, [_P, _P, _P, _P]),
    "nvte_quantize_dbias": (_V, [_P, _P, _P, _P, _P]),
    "nvte_dequantize": (_V, [_P, _P, _P]),
    "nvte_multi_tensor_quantize": (_V, [C.POINTER(_P), C.POINTER(_P), _P, _SZ, _P]),
    # transpose.h / cast_transpose_noop.h
    "nvte_cast_transpose": (_V, [_P, _P, _P]),
    "nvte_transpose": (_V, [_P, _P, _P]),
    "nvte_cast_transpose_dbias": (_V, [_P, _P, _P, _P, _P]),
    "nvte_fp8_transpose_dbias": (_V, [_P, _P, _P, _P, _P]),
    "nvte_multi_cast_transpose": (_V, [_SZ, C.POINTER(_P), C.POINTER(_P), _P]),
    "nvte_transpose_with_noop": (_V, [_P, _P, _P, _P]),
    "nvte_cast_transpose_with_noop": (_V, [_P, _P, _P, _P]),
    # recipe.h
    "nvte_delayed_scaling_recipe_amax_and_scale_update": (_V, [_P, _P, _P, _P, C.c_char_p, _I, _F, _P]),
    "nvte_compute_amax": (_V, [_P, _P, _P]),
    "nvte_compute_amax_with_config": (_V, [_P, _P, _P, _P]),
    "nvte_compute_scale_from_amax": (_V, [_P, _P, _P]),
    # gemm.h
    "nvte_create_matmul_config": (_P, []),
    "nvte_get_matmul_config_attribute": (_V, [_P, _I, _P, _SZ, C.POINTER(_SZ)]),
    "nvte_set_matmul_config_attribute": (_V, [_P, _I, _P, _SZ]),
    "nvte_destroy_matmul_config": (_V, [_P]),
    "nvte_cublas_gemm": (_V, [_P, _P, _P, _P, _P, _B, _B, _B, _P, _B, _B, _I, _P]),
    "nvte_cublas_gemm_v2": (_V, [_I, _I, C.POINTER(_F), _P, _P, C.POINTER(_F), _P, _P, _P, _P, _P]),
    "nvte_cublas_gemm_scaled": (_V, [_P, _P, _P, _P, _P, _B, _B, _B, _P, _F, _F, _B, _I, _P]),
    "nvte_multi_tensor_gemm": (_V, [C.POINTER(_P), C.POINTER(_P), C.POINTER(_P), C.POINTER(_P),
                                    C.POINTER(_P), _I, _B, _B, _B, C.POINTER(_P), _B, _B, _I, _P]),
    "nvte_cublas_handle_init": (_V, []),
    # grouped tensors / grouped quantize / grouped GEMM
    "nvte_splits_to_offsets": (_V, [_P, _P, _SZ, C.c_int64, _P]),
    "nvte_splits_to_offsets_2d": (_V, [_P, _P, _P, _P]),
    "nvte_splits_to_offsets_multi": (_V, [_P, C.POINTER(_P), C.POINTER(C.c_int64), C.POINTER(_I), _SZ, _P]),
    "nvte_create_grouped_tensor": (_P, [_I, _SZ, NVTEShape]),
    "nvte_destroy_grouped_tensor": (_V, [_P]),
    "nvte_set_grouped_tensor_param": (_V, [_P, _I, _P, _SZ]),
    "nvte_get_grouped_tensor_param": (_V, [_P, _I, _P, _SZ, C.POINTER(_SZ)]),
    "nvte_grouped_tensor_num_tensors": (_SZ, [_P]),
    "nvte_grouped_tensor_type": (_I, [_P]),
    "nvte_grouped_tensor_scaling_mode": (_I, [_P]),
    "nvte_get_grouped_tensor_logical_shape": (NVTEShape, [_P]),
    "nvte_group_quantize": (_V, [_P, _P, _P, _P]),
    "nvte_group_dequantize": (_V, [_P, _P, _P]),
    "nvte_group_compute_amax_with_config": (_V, [_P, _P, _P, _P]),
    "nvte_group_compute_scale_from_amax": (_V, [_P, _P, _P]),
    "nvte_create_grouped_matmul_config": (_P, []),
    "nvte_get_grouped_matmul_config_attribute": (_V, [_P, _I, _P, _SZ, C.POINTER(_SZ)]),
    "nvte_set_grouped_matmul_config_attribute": (_V, [_P, _I, _P, _SZ]),
    "nvte_destroy_grouped_matmul_config": (_V, [_P]),
    "nvte_get_grouped_gemm_setup_workspace_size": (_SZ, [_SZ]),
    "nvte_convert_int32_to_int64": (_V, [_P, _P, _SZ, _P]),
    "nvte_convert_int32_to_int64_with_multiplier": (_V, [_P, _P, _SZ, C.c_int64, _P]),
    "nvte_compute_grouped_tensor_offsets": (_V, [_P, _P, _SZ, C.c_int64, _P]),
    "nvte_grouped_gemm": (_V, [_P, _I, _P, _I, _P, _P, _P, _P, _P, _P, _P, _P]),
    "nvte_amd_grouped_gemm_noalloc": (_V, [_P, _I, _P, _I, _P, _P, _P, _P, _P, _P]),
    "nvte_amd_grouped_gemm_noalloc_max_tensors": (_SZ, []),
    "nvte_grouped_gemm_with_discrete_inputA": (_V, [C.POINTER(_P), _SZ, _I, _P, _I, _P, _P, _P, _P, _P, _P, _P, _P]),
    "nvte_grouped_gemm_with_discrete_out": (_V, [_P, _I, _P, _I, C.POINTER(_P), _SZ, C.POINTER(_P), _SZ, _P, _P,
                                                 _P, _P, _P, _P]),
    "nvte_grouped_bias_add": (_V, [_P, _P, _P]),
    "nvte_grouped_scaled_bias_add": (_V, [_P, _P, _P, _P]),
    # normalization.h
    "nvte_layernorm_fwd": (_V, [_P, _P, _P, _F, _P, _P, _P, _P, _I, _B, _P]),
    "nvte_layernorm_bwd": (_V, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _B, _P]),
    "nvte_rmsnorm_fwd": (_V, [_P, _P, _F, _P, _P, _P, _I, _B, _P]),
    "nvte_rmsnorm_bwd": (_V, [_P, _P, _P, _P, _P, _P, _P, _I, _B, _P]),
    "nvte_rmsnorm_bwd_add": (_V, [_P, _P, _P, _P, _P, _P, _P, _P, _I, _B, _P]),
    "nvte_enable_cudnn_norm_fwd": (_V, [_B]),
    "nvte_enable_cudnn_norm_bwd": (_V, [_B]),
    "nvte_enable_zero_centered_gamma_in_weight_dtype": (_V, [_B]),
    # activation.h + fused d-act entry points of cast.h / transpose.h
    "nvte_gelu": (_V, [_P, _P, _P]),
    "nvte_silu": (_V, [_P, _P, _P]),
    "nvte_relu": (_V, [_P, _P, _P]),
    "nvte_qgelu": (_V, [_P, _P, _P]),
    "nvte_srelu": (_V, [_P, _P, _P]),
    "nvte_glu": (_V, [_P, _P, _P]),
    "nvte_geglu": (_V, [_P, _P, _P]),
    "nvte_swiglu": (_V, [_P, _P, _P]),
    "nvte_reglu": (_V, [_P, _P, _P]),
    "nvte_qgeglu": (_V, [_P, _P, _P]),
    "nvte_sreglu": (_V, [_P, _P, _P]),
    "nvte_clamped_swiglu": (_V, [_P, _P, _F, _F, _P]),
    "nvte_clamped_swiglu_v2": (_V, [_P, _P, _F, _F, _F, _P]),
    "nvte_clamped_dswiglu": (_V, [_P, _P, _P, _F, _F, _P]),
    "nvte_clamped_dswiglu_v2": (_V, [_P, _P, _P, _F, _F, _F, _P]),
    "nvte_scaled_swiglu": (_V, [_P, _P, _P, C.c_int64, _P]),
    "nvte_scaled_clamped_swiglu": (_V, [_P, _P, _P, _F, _F, _F, C.c_int64, _P]),
    "nvte_scaled_dswiglu": (_V, [_P, _P, _P, _P, _P, C.c_int64, _P]),
    "nvte_scaled_clamped_dswiglu": (_V, [_P, _P, _P, _P, _P, _F, _F, _F, C.c_int64, _P]),
    "nvte_scaled_srelu": (_V, [_P, _P, _P, _P]),
    "nvte_scaled_dsrelu": (_V, [_P, _P, _P, _P, _P, _P]),
    "nvte_dgelu": (_V, [_P, _P, _P, _P]),
    "nvte_dsilu": (_V, [_P, _P, _P, _P]),
    "nvte_drelu": (_V, [_P, _P, _P, _P]),
    "nvte_dqgelu": (_V, [_P, _P, _P, _P]),
    "nvte_dsrelu": (_V, [_P, _P, _P, _P]),
    "nvte_dglu": (_V, [_P, _P, _P, _P]),
    "nvte_dgeglu": (_V, [_P, _P, _P, _P]),
    "nvte_dswiglu": (_V, [_P, _P, _P, _P]),
    "nvte_dreglu": (_V, [_P, _P, _P, _P]),
    "nvte_dqgeglu": (_V, [_P, _P, _P, _P]),
    "nvte_dsreglu": (_V, [_P, _P, _P, _P]),
    "nvte_dgeglu_cast_transpose": (_V, [_P, _P, _P, _P]),
    "nvte_dswiglu_cast_transpose": (_V, [_P, _P, _P, _P]),
    "nvte_dreglu_cast_transpose": (_V, [_P, _P, _P, _P]),
    "nvte_dqgeglu_cast_transpose": (_V, [_P, _P, _P, _P]),
    "nvte_dsreglu_cast_transpose": (_V, [_P, _P, _P, _P]),
    "nvte_quantize_dbias_dgelu": (_V, [_P, _P, _P, _P, _P, _P]),
    "nvte_quantize_dbias_dsilu": (_V, [_P, _P, _P, _P, _P, _P]),
    "nvte_quantize_dbias_drelu": (_V, [_P, _P, _P, _P, _P, _P]),
    "nvte_quantize_dbias_dqgelu": (_V, [_P, _P, _P, _P, _P, _P]),
    "nvte_quantize_dbias_dsrelu": (_V, [_P, _P, _P, _P, _P, _P]),
    "nvte_cast_transpose_dbias_dgelu": (_V, [_P, _P, _P, _P, _P, _P]),
    "nvte_cast_transpose_dbias_dsilu": (_V, [_P, _P, _P, _P, _P, _P]),
    "nvte_cast_transpose_dbias_drelu": (_V, [_P, _P, _P, _P, _P, _P]),
    "nvte_cast_transpose_dbias_dqgelu": (_V, [_P, _P, _P, _P, _P, _P]),
    "nvte_cast_transpose_dbias_dsrelu": (_V, [_P, _P, _P, _P, _P, _P]),
    # multi_tensor.h + the shard (partial) cast of recipe.h: the FP8 weight-update path
    "nvte_multi_tensor_scale_cuda": (_V, [_I, _P, C.POINTER(C.POINTER(_P)), _SZ, _SZ, _F, _P]),
    "nvte_multi_tensor_scale_tensor_cuda": (_V, [_I, _P, C.POINTER(C.POINTER(_P)), _SZ, _SZ, _P, _P]),
    "nvte_multi_tensor_compute_scale_and_scale_inv_cuda": (_V, [_I, _P, C.POINTER(C.POINTER(_P)), _SZ, _SZ,
                                                                _F, _I, _F, _P]),
    "nvte_multi_tensor_compute_scale_inv_e8m0_cuda": (_V, [_I, C.POINTER(C.POINTER(_P)), _SZ, _SZ, _P]),
    "nvte_multi_tensor_adam_fp8_cuda": (_V, [_I, _P, C.POINTER(C.POINTER(_P)), _SZ, _SZ, _F, _F, _F, _F,
                                             _I, _I, _I, _F, _I, _P]),
    "nvte_mxfp8_scaling_compute_partial_amax": (_V, [_P, _P, _P, _I, _I, _SZ, _P]),
    "nvte_mxfp8_scaling_partial_cast": (_V, [_P, _P, _P, _P, _P, _I, _I, _SZ, _P]),
    # swizzle.h
    "nvte_swizzle_scaling_factors": (_V, [_P, _P, _P]),
    "nvte_multi_tensor_swizzle_scaling_factors": (_V, [C.POINTER(_P), C.POINTER(_P), _SZ, _P]),
    "nvte_unswizzle_scaling_factors": (_V, [_P, _P, _P]),
    "nvte_multi_tensor_unswizzle_scaling_factors": (_V, [C.POINTER(_P), C.POINTER(_P), _SZ, _P]),
    # multi_stream.h
    "nvte_get_num_compute_streams": (_I, []),
    "nvte_get_compute_stream": (_P, [_I]),
    "nvte_get_compute_stream_event": (_P, [_I]),
}

EXPORTED = tuple(_SIGS.keys())


class _Api:
    """Attribute access returns a checked callable: after every call the thread's recorded
    error (if any) is raised as NVTEError - the ctypes stand-in for the C++ exception the
    reference lets propagate into pybind11."""

    def __init__(self, lib):
        self._lib = lib
        for name, (res, args) in _SIGS.items():
            fn = getattr(lib, name)  # raises AttributeError if the symbol is not exported
            fn.restype = res
            fn.argtypes = args
        lib.nvte_amd_set_error_mode(1)
        self._last_error = lib.nvte_amd_last_error
        self._clear_error = lib.nvte_amd_clear_error
        self._errors = lib.nvte_amd_error_counter()      # process-wide count of recorded errors
        self._cache = {}

    def __getattr__(self, name):
        if name.startswith("_"):
            raise AttributeError(name)
        fn = self._cache.get(name)
        if fn is None:
            raw = getattr(self._lib, name)
            last_error, clear_error, errors = self._last_error, self._clear_error, self._errors

            def fn(*a, _raw=raw):
                before = errors[0]
                r = _raw(*a)
                if errors[0] != before:          # some thread recorded an error during the call: ours?
                    e = last_error()
                    if e is not None:
                        clear_error()
                        raise NVTEError(e.decode("utf-8", "replace"))
                return r

            fn.__name__ = name
            self._cache[name] = fn
            setattr(self, name, fn)     # later accesses find the attribute and skip __getattr__
        return fn


api = _Api(_L)
raw_lib = _L


def exported_symbols() -> Sequence[str]:
    return EXPORTED


def version() -> str:
    return raw_lib.nvte_amd_version().decode()
