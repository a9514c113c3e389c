"""Host-side mirror of the reference's quantized-tensor / quantizer classes.

Same names, constructor arguments and attribute names as the reference's
``transformer_engine/pytorch/tensor/float8_tensor.py:41,185`` (Float8Quantizer,
Float8CurrentScalingQuantizer, Float8Tensor), ``tensor/mxfp8_tensor.py`` (MXFP8Quantizer,
MXFP8Tensor) and the C++ mirrors in ``pytorch/csrc/quantizer.cpp:317-1027`` whose call
sequences they reproduce (pointer wiring, which layouts exist, amax nulled before the
current-scaling cast).  PyTorch is used for device memory and streams only; all math
runs in the HIP library through the C ABI.
"""
from __future__ import annotations

import ctypes as C
import threading
import enum
from typing import Optional, Sequence, Tuple

import torch

from . import _lib
from ._lib import api


class DType(enum.IntEnum):
    """ABI values of NVTEDType (what the reference exposes as ``tex.DType``)."""
    kByte = 0
    kInt16 = 1
    kInt32 = 2
    kInt64 = 3
    kFloat32 = 4
    kFloat16 = 5
    kBFloat16 = 6
    kFloat8E4M3 = 7
    kFloat8E5M2 = 8
    kFloat8E8M0 = 9
    kFloat4E2M1 = 10

    @classmethod
    def cast(cls, v) -> "DType":
        return v if isinstance(v, cls) else cls(int(v))


_TORCH_TO_TE = {
    torch.float32: DType.kFloat32,
    torch.float16: DType.kFloat16,
    torch.bfloat16: DType.kBFloat16,
    torch.uint8: DType.kByte,
    torch.int32: DType.kInt32,
    torch.int64: DType.kInt64,
}
for _name, _te in (("float8_e4m3fn", DType.kFloat8E4M3), ("float8_e5m2", DType.kFloat8E5M2),
                   ("float8_e8m0fnu", DType.kFloat8E8M0)):
    if hasattr(torch, _name):
        _TORCH_TO_TE[getattr(torch, _name)] = _te


def te_dtype(t: torch.dtype) -> DType:
    try:
        return _TORCH_TO_TE[t]
    except KeyError:
        raise TypeError(f"unsupported torch dtype {t}") from None


def fp8_max(dtype: DType) -> float:
    return 448.0 if DType.cast(dtype) == DType.kFloat8E4M3 else 57344.0


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def current_stream_ptr() -> int:
    """The hipStream_t PyTorch would launch on (``torch.cuda.current_stream().cuda_stream``), through the
    raw accessor when this torch has it (the Stream object costs ~5 us to build, seven times per step)."""
    if _raw_stream is not None:
        return int(_raw_stream(torch.cuda.current_device()))
    return int(torch.cuda.current_stream().cuda_stream)


def non_tn_fp8_gemm_supported() -> bool:
    return bool(api.nvte_is_non_tn_fp8_gemm_supported())


class TensorWrapper:
    """RAII NVTETensor handle (ref transformer_engine.h:738-792).  Keeps the torch tensors it
    points at alive until it is destroyed."""

    def __init__(self, scaling_mode: int = _lib.NVTE_DELAYED_TENSOR_SCALING):
        self.handle = api.nvte_create_tensor(int(scaling_mode))
        self._keep = []

    # -- setters -------------------------------------------------------------------
    def _set(self, param: int, ptr: int, dtype: DType, shape: Sequence[int]):
        bt = _lib.NVTEBasicTensor(ptr, int(dtype), _lib.make_shape(shape))
        api.nvte_set_tensor_param_v2(self.handle, param, C.byref(bt), C.sizeof(bt))

    def set_tensor(self, param: int, t: Optional[torch.Tensor], dtype: Optional[DType] = None,
                   shape: Optional[Sequence[int]] = None):
        if t is None:
            return self
        if not t.is_contiguous():
            raise ValueError("NVTETensor buffers must be contiguous")
        self._keep.append(t)
        self._set(param, t.data_ptr(), dtype if dtype is not None else te_dtype(t.dtype),
                  tuple(t.shape) if shape is None else shape)
        return self

    def set_rowwise_data(self, t, dtype=None, shape=None):
        return self.set_tensor(_lib.kRowwiseData, t, dtype, shape)

    def set_columnwise_data(self, t, dtype=None, shape=None):
        return self.set_tensor(_lib.kColumnwiseData, t, dtype, shape)

    def set_scale(self, t):
        return self.set_tensor(_lib.kScale, t)

    def set_amax(self, t):
        return self.set_tensor(_lib.kAmax, t)

    def set_rowwise_scale_inv(self, t, dtype=None):
        return self.set_tensor(_lib.kRowwiseScaleInv, t, dtype)

    def set_columnwise_scale_inv(self, t, dtype=None):
        return self.set_tensor(_lib.kColumnwiseScaleInv, t, dtype)

    def set_with_gemm_swizzled_scales(self, flag: bool):
        b = C.c_uint8(1 if flag else 0)
        api.nvte_set_tensor_param_v2(self.handle, _lib.kWithGEMMSwizzledScales, C.byref(b), 1)
        return self

    def clear_amax(self):
        self._set(_lib.kAmax, 0, DType.kFloat32, (0,))
        return self

    def set_shape_only(self, param: int, dtype: DType, shape: Sequence[int]):
        self._set(param, 0, dtype, shape)
        return self

    # -- getters -------------------------------------------------------------------
    def shape(self) -> Tuple[int, ...]:
        s = api.nvte_tensor_shape(self.handle)
        return tuple(s.data[i] for i in range(s.ndim))

    def get_param(self, param: int) -> _lib.NVTEBasicTensor:
        return api.nvte_get_tensor_param(self.handle, param)

    def destroy(self):
        if self.handle:
            _lib.raw_lib.nvte_destroy_tensor(self.handle)
            self.handle = None
            self._keep = []

    def __del__(self):  # pragma: no cover - best effort
        try:
            self.destroy()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.destroy()
        return False


class _CachedWrapper(TensorWrapper):
    """A handle owned by the tensor object it describes: callers' ``destroy()`` leaves it alone, it goes when
    the owner drops it (``__del__``) - a Float8Tensor is wrapped by every GEMM that reads it."""

    def destroy(self):
        pass

    def __del__(self):  # pragma: no cover - best effort
        try:
            TensorWrapper.destroy(self)
        except Exception:
            pass


_PLAIN = {}          # (data_ptr, dtype, shape) -> handle: a descriptor of that memory, whoever owns it now
_PLAIN_LOCK = threading.Lock()
_PLAIN_MAX = 512


def wrap_plain(t: torch.Tensor) -> TensorWrapper:
    """High-precision (or raw byte) tensor -> handle with only rowwise data.  The handle only says "this
    dtype and shape at this address", so it is kept and handed out again for the same triple (the caching
    allocator returns the same blocks step after step): three foreign calls per wrap otherwise, a dozen wraps
    per Linear step.  The cache itself holds no reference to the tensor (the object returned does, for as long
    as the caller keeps it), and callers never modify the handles they get here."""
    if not t.is_contiguous():
        raise ValueError("NVTETensor buffers must be contiguous")
    key = (t.data_ptr(), t.dtype, tuple(t.shape))
    w = _PLAIN.get(key)
    if w is None:
        w = _CachedWrapper()
        w._set(_lib.kRowwiseData, key[0], te_dtype(t.dtype), key[2])
        with _PLAIN_LOCK:      # the autograd thread wraps too: evict and insert under one lock
            while len(_PLAIN) >= _PLAIN_MAX:
                _PLAIN.pop(next(iter(_PLAIN)), None)
            _PLAIN[key] = w
    return _PlainRef(w, t)


class _PlainRef:
    """What wrap_plain hands out: the shared handle plus a reference to the tensor, so that a temporary
    (``wrap_plain(x.contiguous())``) lives as long as the caller holds this object, as it did when every wrap
    built its own handle."""
    __slots__ = ("handle", "_w", "_t")

    def __init__(self, w, t):
        self.handle, self._w, self._t = w.handle, w, t

    def destroy(self):
        self._t = None

    def __getattr__(self, name):          # getters of the underlying wrapper (shape(), get_param(), ...)
        # the handle is SHARED by everybody who wraps the same (address, dtype, shape): a setter through one
        # reference would change it under every other holder
        if name.startswith(("set_", "clear_", "_set")):
            raise AttributeError(f"wrap_plain() handles are shared and read-only: {name}() is not available; "
                                 "build a TensorWrapper of your own")
        return getattr(self._w, name)


class QuantizationConfig:
    """RAII NVTEQuantizationConfig."""

    def __init__(self, force_pow_2_scales: bool = False, amax_epsilon: float = 0.0,
                 noop: Optional[TensorWrapper] = None):
        self.handle = api.nvte_create_quantization_config()
        b = C.c_uint8(1 if force_pow_2_scales else 0)
        api.nvte_set_quantization_config_attribute(self.handle, _lib.kQForcePow2Scales, C.byref(b), 1)
        f = C.c_float(amax_epsilon)
        api.nvte_set_quantization_config_attribute(self.handle, _lib.kQAmaxEpsilon, C.byref(f), 4)
        self._noop = noop
        if noop is not None:
            p = C.c_void_p(noop.handle)
            api.nvte_set_quantization_config_attribute(self.handle, _lib.kQNoopTensor, C.byref(p),
                                                       C.sizeof(p))

    def __del__(self):  # pragma: no cover
        try:
            if self.handle:
                _lib.raw_lib.nvte_destroy_quantization_config(self.handle)
                self.handle = None
        except Exception:
            pass


# =====================================================================================
# quantized tensor storage
# =====================================================================================
class Float8Tensor:
    """Per-tensor-scaled FP8 tensor (attribute names as the reference's Float8TensorStorage,
    ref pytorch/csrc/type_converters.cpp:17-53)."""

    def __init__(self, *, shape, dtype, data, fp8_scale_inv, fp8_dtype, data_transpose=None,
                 quantizer=None):
        self._shape = tuple(shape)
        self.dtype = dtype                      # nominal high-precision dtype
        self._data = data                       # uint8 [...], may be None
        self._transpose = data_transpose        # uint8 [last, prod(leading)], may be None
        self._transpose_invalid = data_transpose is None
        self._scale_inv = fp8_scale_inv         # fp32 [1], shared by both layouts
        self._fp8_dtype = DType.cast(fp8_dtype)
        self._quantizer = quantizer

    @property
    def shape(self):
        return self._shape

    def size(self, dim=None):
        return self._shape if dim is None else self._shape[dim]

    def numel(self):
        n = 1
        for d in self._shape:
            n *= d
        return n

    @property
    def device(self):
        return (self._data if self._data is not None else self._transpose).device

    def wrap(self) -> TensorWrapper:
        with_t = self._transpose is not None and not self._transpose_invalid
        sig = (self._data.data_ptr() if self._data is not None else 0,
               self._transpose.data_ptr() if with_t else 0, self._scale_inv.data_ptr(), self._shape)
        cached = getattr(self, "_wrap_cache", None)
        if cached is not None and cached[0] == sig:
            return cached[1]
        w = _CachedWrapper(_lib.NVTE_DELAYED_TENSOR_SCALING)
        if self._data is not None:
            w.set_rowwise_data(self._data, self._fp8_dtype, self._shape)
            w.set_rowwise_scale_inv(self._scale_inv)
        if with_t:
            rows = self.numel() // self._shape[-1] if self._shape[-1] else 0
            w.set_columnwise_data(self._transpose, self._fp8_dtype, (self._shape[-1], rows))
            w.set_columnwise_scale_inv(self._scale_inv)
        self._wrap_cache = (sig, w)
        return w

    def dequantize(self, dtype: Optional[torch.dtype] = None) -> torch.Tensor:
        dtype = dtype or self.dtype
        if self._data is None:
            raise NotImplementedError("dequantize needs the rowwise payload")
        out = torch.empty(self._shape, dtype=dtype, device=self._data.device)
        w = TensorWrapper().set_rowwise_data(self._data, self._fp8_dtype, self._shape)
        w.set_rowwise_scale_inv(self._scale_inv)
        o = wrap_plain(out)
        api.nvte_dequantize(w.handle, o.handle, current_stream_ptr())
        w.destroy(); o.destroy()
        return out

    def _create_transpose(self):
        """Rebuild the transposed copy from the rowwise bytes (ref pytorch/distributed.py:1075
        -> nvte_transpose): used after an FP8 all-gather."""
        rows = self.numel() // self._shape[-1]
        cols = self._shape[-1]
        data2d = self._data.view(rows, cols)
        out = torch.empty((cols, rows), dtype=torch.uint8, device=self._data.device)
        wi = TensorWrapper().set_rowwise_data(data2d, self._fp8_dtype)
        wo = TensorWrapper().set_rowwise_data(out, self._fp8_dtype)
        api.nvte_transpose(wi.handle, wo.handle, current_stream_ptr())
        wi.destroy(); wo.destroy()
        self._transpose = out
        self._transpose_invalid = False

    def update_usage(self, rowwise_usage: Optional[bool] = None, columnwise_usage: Optional[bool] = None):
        if columnwise_usage and (self._transpose is None or self._transpose_invalid):
            self._create_transpose()
        if rowwise_usage is False:
            self._data = None
        if columnwise_usage is False:
            self._transpose = None
            self._transpose_invalid = True


class MXFP8Tensor:
    """MXFP8 tensor (attribute names as the reference's MXFP8TensorStorage,
    ref pytorch/csrc/type_converters.cpp:55-89)."""

    def __init__(self, *, shape, dtype, rowwise_data, rowwise_scale_inv, columnwise_data,
                 columnwise_scale_inv, fp8_dtype, quantizer=None, with_gemm_swizzled_scales=False,
                 columnwise_transposed=False):
        self._shape = tuple(shape)
        self.dtype = dtype
        self._rowwise_data = rowwise_data
        self._rowwise_scale_inv = rowwise_scale_inv
        self._columnwise_data = columnwise_data
        self._columnwise_scale_inv = columnwise_scale_inv
        self._fp8_dtype = DType.cast(fp8_dtype)
        self._quantizer = quantizer
        self._with_gemm_swizzled_scales = with_gemm_swizzled_scales
        # amd_ext.h extension: columnwise payload stored as [last, prod(leading)], scales
        # [pad128(last), pad4(rows/32)] - what the NN / NT GEMMs (dgrad / wgrad) read
        self._columnwise_transposed = columnwise_transposed

    @property
    def shape(self):
        return self._shape

    def numel(self):
        n = 1
        for d in self._shape:
            n *= d
        return n

    @property
    def device(self):
        return (self._rowwise_data if self._rowwise_data is not None else self._columnwise_data).device

    def wrap(self) -> TensorWrapper:
        w = TensorWrapper(_lib.NVTE_MXFP8_1D_SCALING)
        if self._rowwise_data is not None:
            w.set_rowwise_data(self._rowwise_data, self._fp8_dtype, self._shape)
            w.set_rowwise_scale_inv(self._rowwise_scale_inv, DType.kFloat8E8M0)
        if self._columnwise_data is not None:
            cshape = tuple(self._columnwise_data.shape) if self._columnwise_transposed else self._shape
            w.set_columnwise_data(self._columnwise_data, self._fp8_dtype, cshape)
            w.set_columnwise_scale_inv(self._columnwise_scale_inv, DType.kFloat8E8M0)
        w.set_with_gemm_swizzled_scales(self._with_gemm_swizzled_scales)
        if self._columnwise_transposed:
            api.nvte_amd_set_columnwise_transposed(w.handle, 1)
        return w

    def dequantize(self, dtype: Optional[torch.dtype] = None) -> torch.Tensor:
        dtype = dtype or self.dtype
        ref = self._rowwise_data if self._rowwise_data is not None else self._columnwise_data
        out = torch.empty(self._shape, dtype=dtype, device=ref.device)
        w = self.wrap()
        o = wrap_plain(out)
        api.nvte_dequantize(w.handle, o.handle, current_stream_ptr())
        w.destroy(); o.destroy()
        return out


# =====================================================================================
# quantizers
# =====================================================================================
class Quantizer:
    """Base class (ref transformer_engine/pytorch/quantized_tensor.py)."""

    def __init__(self, *, rowwise: bool = True, columnwise: bool = True):
        self.rowwise_usage = rowwise
        self.columnwise_usage = columnwise
        self.internal = False

    def set_usage(self, *, rowwise: Optional[bool] = None, columnwise: Optional[bool] = None):
        if rowwise is not None:
            self.rowwise_usage = rowwise
        if columnwise is not None:
            self.columnwise_usage = columnwise

    def __call__(self, tensor: torch.Tensor):
        return self.quantize(tensor)

    def quantize(self, tensor: torch.Tensor, *, out=None, noop_flag: Optional[torch.Tensor] = None):
        if out is None:
            out = self.make_empty(tensor.shape, dtype=tensor.dtype, device=tensor.device)
        self.update_quantized(tensor, out, noop_flag=noop_flag)
        return out


def _check_input(t: torch.Tensor) -> torch.Tensor:
    if not t.is_cuda:
        raise RuntimeError("quantization runs on the GPU only (no CPU fallback in this library)")
    return t if t.is_contiguous() else t.contiguous()


def _flat2d(shape) -> Tuple[int, int]:
    if len(shape) == 0:
        return 1, 1
    rows = 1
    for d in shape[:-1]:
        rows *= d
    return rows, shape[-1]


class Float8Quantizer(Quantizer):
    """Per-tensor DELAYED scaling: scale comes from the recipe state, amax is written back
    (ref pytorch/tensor/float8_tensor.py:41-183, csrc/quantizer.cpp:317-625)."""

    def __init__(self, scale: torch.Tensor, amax: torch.Tensor, fp8_dtype, *, rowwise: bool = True,
                 columnwise: bool = True):
        super().__init__(rowwise=rowwise, columnwise=columnwise)
        self.scale = scale
        self.amax = amax
        self.dtype = DType.cast(fp8_dtype)

    def copy(self):
        q = Float8Quantizer(self.scale, self.amax, self.dtype, rowwise=self.rowwise_usage,
                            columnwise=self.columnwise_usage)
        q.internal = self.internal
        return q

    def make_empty(self, shape, *, dtype=torch.float32, device=None) -> Float8Tensor:
        device = device or torch.device("cuda")
        rows, cols = _flat2d(tuple(shape))
        # ref csrc/quantizer.cpp:342-356: where non-TN FP8 GEMMs are supported the row-wise payload serves both
        # usages and no transposed copy is allocated (here: NVTE_AMD_NON_TN_FP8_GEMM=1, the GEMM reads it K-major);
        # otherwise - the default answer of the library - column-wise usage materialises the transpose
        non_tn = non_tn_fp8_gemm_supported()
        data = torch.empty(tuple(shape), dtype=torch.uint8, device=device) if (self.rowwise_usage or non_tn) else None
        tr = torch.empty((cols, rows), dtype=torch.uint8, device=device) if (self.columnwise_usage and not non_tn) else None
        sinv = torch.empty(1, dtype=torch.float32, device=device)
        return Float8Tensor(shape=shape, dtype=dtype, data=data, fp8_scale_inv=sinv,
                            fp8_dtype=self.dtype, data_transpose=tr, quantizer=self)

    def _wrap_out(self, out: Float8Tensor) -> TensorWrapper:
        w = TensorWrapper(_lib.NVTE_DELAYED_TENSOR_SCALING)
        rows, cols = _flat2d(out.shape)
        if out._data is not None:
            w.set_rowwise_data(out._data, self.dtype, out.shape)
            w.set_rowwise_scale_inv(out._scale_inv)
        if out._transpose is not None:
            w.set_columnwise_data(out._transpose, self.dtype, (cols, rows))
            w.set_columnwise_scale_inv(out._scale_inv)  # same tensor for both (quantizer.cpp:421-427)
        return w

    def update_quantized(self, src: torch.Tensor, dst: Float8Tensor, *, noop_flag=None):
        src = _check_input(src)
        if src.numel() == 0:
            return dst  # ref csrc/quantizer.cpp:615-617
        wi = wrap_plain(src)
        wo = self._wrap_out(dst)
        wo.set_scale(self.scale).set_amax(self.amax)
        cfg = None
        if noop_flag is not None:
            wn = wrap_plain(noop_flag)
            cfg = QuantizationConfig(noop=wn)
        api.nvte_quantize_v2(wi.handle, wo.handle, cfg.handle if cfg else None, current_stream_ptr())
        dst._transpose_invalid = dst._transpose is None
        wi.destroy(); wo.destroy()
        return dst


class Float8CurrentScalingQuantizer(Quantizer):
    """Per-tensor CURRENT scaling: amax scan -> (optional MAX all-reduce) -> scale -> cast
    (ref pytorch/tensor/float8_tensor.py:185-330, csrc/quantizer.cpp:960-1012)."""

    def __init__(self, fp8_dtype, device=None, *, rowwise: bool = True, columnwise: bool = True,
                 use_existing_amax: bool = False, with_amax_reduction: bool = False,
                 amax_reduction_group=None, force_pow_2_scales: bool = False,
                 amax_epsilon: float = 0.0, scale=None, amax=None):
        super().__init__(rowwise=rowwise, columnwise=columnwise)
        self.dtype = DType.cast(fp8_dtype)
        self.with_amax_reduction = with_amax_reduction
        self.amax_reduction_group = amax_reduction_group
        self.force_pow_2_scales = force_pow_2_scales
        self.amax_epsilon = amax_epsilon
        self._device = device

    def copy(self):
        return Float8CurrentScalingQuantizer(
            self.dtype, self._device, rowwise=self.rowwise_usage, columnwise=self.columnwise_usage,
            with_amax_reduction=self.with_amax_reduction,
            amax_reduction_group=self.amax_reduction_group,
            force_pow_2_scales=self.force_pow_2_scales, amax_epsilon=self.amax_epsilon)

    make_empty = Float8Quantizer.make_empty
    _wrap_out = Float8Quantizer._wrap_out

    def update_quantized(self, src: torch.Tensor, dst: Float8Tensor, *, noop_flag=None):
        src = _check_input(src)
        if src.numel() == 0:
            return dst
        stream = current_stream_ptr()
        scale = torch.empty(1, dtype=torch.float32, device=src.device)
        amax = torch.empty(1, dtype=torch.float32, device=src.device)
        dst._amax = amax
        dst._scale = scale
        wi = wrap_plain(src)
        wo = self._wrap_out(dst)
        wo.set_scale(scale).set_amax(amax)
        wn = wrap_plain(noop_flag) if noop_flag is not None else None
        cfg = QuantizationConfig(self.force_pow_2_scales, self.amax_epsilon, wn)
        api.nvte_compute_amax_with_config(wi.handle, wo.handle, cfg.handle, stream)
        if self.with_amax_reduction:
            # MAX all-reduce of one float over the group the tensor is sharded on (ref csrc/quantizer.cpp:994-999)
            from .distributed import all_reduce_max_
            all_reduce_max_(amax, self.amax_reduction_group)
        api.nvte_compute_scale_from_amax(wo.handle, cfg.handle, stream)
        wo.clear_amax()  # avoid the atomic amax update in the cast (quantizer.cpp:1006)
        api.nvte_quantize_v2(wi.handle, wo.handle, cfg.handle, stream)
        dst._transpose_invalid = dst._transpose is None
        wi.destroy(); wo.destroy()
        return dst


def _round_up(a: int, b: int) -> int:
    return (a + b - 1) // b * b


class MXFP8Quantizer(Quantizer):
    """MXFP8: one e8m0 scale per 32 consecutive elements, rowwise and/or columnwise
    (ref pytorch/tensor/mxfp8_tensor.py:29-160, csrc/quantizer.cpp:1700-1830)."""

    def __init__(self, fp8_dtype, *, rowwise: bool = True, columnwise: bool = True,
                 columnwise_transposed: bool = False):
        super().__init__(rowwise=rowwise, columnwise=columnwise)
        self.dtype = DType.cast(fp8_dtype)
        self.optimize_for_gemm = False
        # extension (include/transformer_engine/amd_ext.h): K-contiguous columnwise copy for the
        # NN / NT GEMMs; needs 128-multiple dims
        self.columnwise_transposed = columnwise_transposed

    def copy(self):
        q = MXFP8Quantizer(self.dtype, rowwise=self.rowwise_usage, columnwise=self.columnwise_usage,
                           columnwise_transposed=self.columnwise_transposed)
        q.optimize_for_gemm = self.optimize_for_gemm
        return q

    @staticmethod
    def scale_shapes(shape) -> Tuple[Tuple[int, int], Tuple[int, int]]:
        rows, cols = _flat2d(tuple(shape))
        return ((_round_up(rows, 128), _round_up((cols + 31) // 32, 4)),
                (_round_up((rows + 31) // 32, 4), _round_up(cols, 128)))

    def make_empty(self, shape, *, dtype=torch.float32, device=None) -> MXFP8Tensor:
        device = device or torch.device("cuda")
        rs, cs = self.scale_shapes(shape)
        rows, cols = _flat2d(tuple(shape))
        # the padding of the scale matrices reads as zero (ref mxfp8_tensor.py make_empty); whole 128 x 128 tiles
        # have none, every scale byte is then written by the quantize kernel and the two fill launches are saved
        scales = torch.empty if (rows % 128 == 0 and cols % 128 == 0) else torch.zeros
        rd = rsi = cd = csi = None
        if self.rowwise_usage:
            rd = torch.empty(tuple(shape), dtype=torch.uint8, device=device)
            rsi = scales(rs, dtype=torch.uint8, device=device)
        if self.columnwise_usage:
            if self.columnwise_transposed:
                cd = torch.empty((cols, rows), dtype=torch.uint8, device=device)
                csi = scales((cs[1], cs[0]), dtype=torch.uint8, device=device)
            else:
                cd = torch.empty(tuple(shape), dtype=torch.uint8, device=device)
                csi = scales(cs, dtype=torch.uint8, device=device)
        return MXFP8Tensor(shape=shape, dtype=dtype, rowwise_data=rd, rowwise_scale_inv=rsi,
                           columnwise_data=cd, columnwise_scale_inv=csi, fp8_dtype=self.dtype,
                           quantizer=self, with_gemm_swizzled_scales=self.optimize_for_gemm,
                           columnwise_transposed=self.columnwise_usage and self.columnwise_transposed)

    def update_quantized(self, src: torch.Tensor, dst: MXFP8Tensor, *, noop_flag=None):
        src = _check_input(src)
        if src.numel() == 0:
            return dst
        wi = wrap_plain(src)
        wo = dst.wrap()
        cfg = None
        if noop_flag is not None:
            cfg = QuantizationConfig(noop=wrap_plain(noop_flag))
        api.nvte_quantize_v2(wi.handle, wo.handle, cfg.handle if cfg else None, current_stream_ptr())
        wi.destroy(); wo.destroy()
        return dst
