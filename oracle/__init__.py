"""CPU oracle for the FP8 linear hot path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import this package.  It wraps ``oracle/te_oracle.c`` (a C restatement of the reference's
``tests/cpp`` host references; every C function cites the reference lines it follows)
with numpy-friendly helpers.  Nothing under ``transformer_engine_amd/`` imports it.
"""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "libte_oracle.so")

E4M3, E5M2 = 0, 1
FP8_MAX = {E4M3: 448.0, E5M2: 57344.0}
# NVTEDType values
DT_F32, DT_F16, DT_BF16 = 4, 5, 6


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "te_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or (
        os.path.exists(src) and os.path.getmtime(src) > os.path.getmtime(_LIB_PATH)
    ):
        subprocess.check_call(["make", "-C", _HERE], stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = ctypes.CDLL(_LIB_PATH)
        c = ctypes
        L.te_float_to_fp8.restype = c.c_uint8
        L.te_float_to_fp8.argtypes = [c.c_float, c.c_int]
        L.te_fp8_to_float.restype = c.c_float
        L.te_fp8_to_float.argtypes = [c.c_uint8, c.c_int]
        L.te_float_to_e8m0.restype = c.c_uint8
        L.te_float_to_e8m0.argtypes = [c.c_float]
        L.te_exp2f_rcp.restype = c.c_float
        L.te_exp2f_rcp.argtypes = [c.c_uint8]
        L.te_e8m0_to_float.restype = c.c_float
        L.te_e8m0_to_float.argtypes = [c.c_uint8]
        L.te_float_to_bf16.restype = c.c_uint16
        L.te_float_to_bf16.argtypes = [c.c_float]
        L.te_float_to_half.restype = c.c_uint16
        L.te_float_to_half.argtypes = [c.c_float]
        L.te_half_to_float.restype = c.c_float
        L.te_half_to_float.argtypes = [c.c_uint16]
        for name in ("te_cast_transpose_ref", "te_cast_transpose_ref_omp"):
            f = getattr(L, name)
            f.restype = None
            f.argtypes = [c.c_void_p, c.c_int, c.c_void_p, c.c_void_p, c.c_size_t, c.c_size_t,
                          c.c_float, c.c_int, c.c_void_p, c.c_void_p]
        L.te_amax_ref.restype = c.c_float
        L.te_amax_ref.argtypes = [c.c_void_p, c.c_int, c.c_size_t]
        L.te_scale_from_amax_ref.restype = c.c_float
        L.te_scale_from_amax_ref.argtypes = [c.c_float, c.c_float, c.c_int, c.c_float]
        L.te_mxfp8_ref.restype = None
        L.te_mxfp8_ref.argtypes = [c.c_void_p, c.c_int, c.c_size_t, c.c_size_t, c.c_int, c.c_void_p,
                                   c.c_void_p, c.c_size_t, c.c_void_p, c.c_void_p, c.c_size_t]
        L.te_swizzle_scales_ref.restype = None
        L.te_swizzle_scales_ref.argtypes = [c.c_void_p, c.c_void_p, c.c_size_t, c.c_size_t]
        L.te_swizzle_col_scales_ref.restype = None
        L.te_swizzle_col_scales_ref.argtypes = [c.c_void_p, c.c_void_p, c.c_size_t, c.c_size_t]
        L.te_delayed_scaling_ref.restype = None
        L.te_delayed_scaling_ref.argtypes = [c.c_void_p, c.c_void_p, c.c_void_p, c.c_void_p,
                                             c.c_size_t, c.c_size_t, c.c_int, c.c_float, c.c_float]
        L.te_delayed_scaling_bulk_ref.restype = None
        L.te_delayed_scaling_bulk_ref.argtypes = [c.c_void_p, c.c_void_p, c.c_void_p, c.c_size_t,
                                                  c.c_size_t, c.c_int, c.c_float, c.c_float]
        L.te_dequant_ref.restype = None
        L.te_dequant_ref.argtypes = [c.c_void_p, c.c_int, c.c_float, c.c_void_p, c.c_size_t]
        L.te_dequant_mxfp8_ref.restype = None
        L.te_dequant_mxfp8_ref.argtypes = [c.c_void_p, c.c_int, c.c_void_p, c.c_size_t, c.c_int,
                                           c.c_void_p, c.c_size_t, c.c_size_t]
        L.te_gemm_tn_f32_ref.restype = None
        L.te_gemm_tn_f32_ref.argtypes = [c.c_void_p, c.c_void_p, c.c_void_p, c.c_size_t, c.c_size_t,
                                         c.c_size_t, c.c_float, c.c_float]
        L.te_gemm_tn_f32_fast.restype = None
        L.te_gemm_tn_f32_fast.argtypes = [c.c_void_p, c.c_void_p, c.c_void_p, c.c_size_t,
                                          c.c_size_t, c.c_size_t]
        L.te_gelu_ref.restype = None
        L.te_gelu_ref.argtypes = [c.c_void_p, c.c_void_p, c.c_size_t, c.c_int]
        L.te_norm_stats_ref.restype = None
        L.te_norm_stats_ref.argtypes = [c.c_int, c.c_void_p, c.c_int, c.c_void_p, c.c_void_p, c.c_size_t,
                                        c.c_size_t, c.c_double]
        L.te_norm_output_ref.restype = None
        L.te_norm_output_ref.argtypes = [c.c_int, c.c_void_p, c.c_void_p, c.c_void_p, c.c_int, c.c_int, c.c_void_p,
                                         c.c_void_p, c.c_void_p, c.c_size_t, c.c_size_t, c.c_void_p, c.c_float,
                                         c.c_int, c.c_int]
        L.te_norm_backward_ref.restype = None
        L.te_norm_backward_ref.argtypes = [c.c_int, c.c_void_p, c.c_void_p, c.c_void_p, c.c_int, c.c_int,
                                           c.c_void_p, c.c_void_p, c.c_void_p, c.c_void_p, c.c_void_p, c.c_void_p,
                                           c.c_size_t, c.c_size_t, c.c_int]
        L.te_act_ref.restype = None
        L.te_act_ref.argtypes = [c.c_int, c.c_int, c.c_void_p, c.c_void_p, c.c_int, c.c_void_p, c.c_size_t,
                                 c.c_size_t]
        L.te_colsum_ref.restype = None
        L.te_colsum_ref.argtypes = [c.c_void_p, c.c_int, c.c_size_t, c.c_size_t, c.c_void_p]
        L.te_mxfp8_partial_amax_ref.restype = None
        L.te_mxfp8_partial_amax_ref.argtypes = [c.c_void_p, c.c_int, c.c_size_t, c.c_size_t, c.c_size_t, c.c_size_t,
                                                c.c_void_p, c.c_size_t, c.c_void_p, c.c_size_t]
        L.te_e8m0_from_amax_ref.restype = None
        L.te_e8m0_from_amax_ref.argtypes = [c.c_void_p, c.c_size_t, c.c_void_p]
        L.te_mxfp8_partial_cast_ref.restype = None
        L.te_mxfp8_partial_cast_ref.argtypes = [c.c_void_p, c.c_int, c.c_size_t, c.c_size_t, c.c_size_t, c.c_size_t,
                                                c.c_void_p, c.c_size_t, c.c_void_p, c.c_size_t, c.c_void_p,
                                                c.c_void_p]
        L.te_scale_and_scale_inv_ref.restype = None
        L.te_scale_and_scale_inv_ref.argtypes = [c.c_void_p, c.c_size_t, c.c_float, c.c_int, c.c_float, c.c_void_p,
                                                 c.c_void_p]
        L.te_adam_fp8_ref.restype = c.c_float
        L.te_adam_fp8_ref.argtypes = [c.c_void_p, c.c_int, c.c_size_t, c.c_void_p, c.c_void_p, c.c_void_p,
                                      c.c_void_p, c.c_float, c.c_int, c.c_float, c.c_float, c.c_float, c.c_float,
                                      c.c_int, c.c_int, c.c_int, c.c_float]
        L.te_clamped_swiglu_ref.restype = None
        L.te_clamped_swiglu_ref.argtypes = [c.c_int, c.c_void_p, c.c_void_p, c.c_int, c.c_void_p, c.c_size_t,
                                            c.c_size_t, c.c_float, c.c_float, c.c_float]
        L.te_oracle_num_threads.restype = c.c_int
        _lib = L
    return _lib


def _ptr(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def _in_dtype(a: np.ndarray) -> int:
    """bf16 travels as uint16 (numpy has no bfloat16); fp16 as float16; fp32 as float32."""
    if a.dtype == np.float32:
        return DT_F32
    if a.dtype == np.float16:
        return DT_F16
    if a.dtype == np.uint16:
        return DT_BF16
    raise TypeError(f"unsupported oracle input dtype {a.dtype}")


def float_to_fp8(x: np.ndarray, fmt: int) -> np.ndarray:
    x = np.ascontiguousarray(x, dtype=np.float32)
    out = np.empty(x.shape, dtype=np.uint8)
    flat_in, flat_out = x.reshape(-1), out.reshape(-1)
    # reuse the cast_transpose reference with scale 1 and no transpose: vectorised in C
    lib().te_cast_transpose_ref(_ptr(flat_in), DT_F32, _ptr(flat_out), None, 1, flat_in.size,
                                ctypes.c_float(1.0), fmt, None, None)
    return out


def fp8_to_float(q: np.ndarray, fmt: int) -> np.ndarray:
    q = np.ascontiguousarray(q, dtype=np.uint8)
    out = np.empty(q.shape, dtype=np.float32)
    lib().te_dequant_ref(_ptr(q), fmt, ctypes.c_float(1.0), _ptr(out), q.size)
    return out


def cast_transpose(x: np.ndarray, scale: float, fmt: int, transpose: bool = True, omp: bool = False):
    """Returns (out_c [N,H] u8, out_t [H,N] u8 or None, amax f32, scale_inv f32)."""
    x = np.ascontiguousarray(x)
    n = int(np.prod(x.shape[:-1])) if x.ndim > 1 else 1
    h = x.shape[-1] if x.ndim >= 1 else 1
    out_c = np.empty((n, h), dtype=np.uint8)
    out_t = np.empty((h, n), dtype=np.uint8) if transpose else None
    amax = np.zeros(1, dtype=np.float32)
    sinv = np.zeros(1, dtype=np.float32)
    fn = lib().te_cast_transpose_ref_omp if omp else lib().te_cast_transpose_ref
    fn(_ptr(x), _in_dtype(x), _ptr(out_c), _ptr(out_t), n, h, ctypes.c_float(scale), fmt,
       _ptr(amax), _ptr(sinv))
    return out_c, out_t, amax[0], sinv[0]


def amax(x: np.ndarray) -> np.float32:
    x = np.ascontiguousarray(x)
    return np.float32(lib().te_amax_ref(_ptr(x), _in_dtype(x), x.size))


def scale_from_amax(amax_v: float, fmt: int, force_pow_2: bool = False, epsilon: float = 0.0) -> np.float32:
    return np.float32(lib().te_scale_from_amax_ref(ctypes.c_float(amax_v), ctypes.c_float(FP8_MAX[fmt]),
                                                   int(force_pow_2), ctypes.c_float(epsilon)))


def round_up(a: int, b: int) -> int:
    return (a + b - 1) // b * b


def mxfp8_scale_shapes(rows: int, cols: int):
    """(rowwise, columnwise) padded scale shapes (ref common/transformer_engine.cpp:93-115)."""
    return ((round_up(rows, 128), round_up((cols + 31) // 32, 4)),
            (round_up((rows + 31) // 32, 4), round_up(cols, 128)))


def mxfp8(x: np.ndarray, fmt: int, rowwise: bool = True, colwise: bool = True):
    """Returns dict with out_row, sc_row, out_col, sc_col (padding entries of scales = 0)."""
    x = np.ascontiguousarray(x)
    rows = int(np.prod(x.shape[:-1]))
    cols = x.shape[-1]
    rs, cs = mxfp8_scale_shapes(rows, cols)
    res = {}
    out_row = np.zeros((rows, cols), np.uint8) if rowwise else None
    sc_row = np.zeros(rs, np.uint8) if rowwise else None
    out_col = np.zeros((rows, cols), np.uint8) if colwise else None
    sc_col = np.zeros(cs, np.uint8) if colwise else None
    lib().te_mxfp8_ref(_ptr(x), _in_dtype(x), rows, cols, fmt, _ptr(out_row), _ptr(sc_row), rs[1],
                       _ptr(out_col), _ptr(sc_col), cs[1])
    res.update(out_row=out_row, sc_row=sc_row, out_col=out_col, sc_col=sc_col)
    return res


def swizzle_scales(sc: np.ndarray) -> np.ndarray:
    sc = np.ascontiguousarray(sc, dtype=np.uint8)
    out = np.zeros(sc.size, np.uint8)
    lib().te_swizzle_scales_ref(_ptr(sc), _ptr(out), sc.shape[0], sc.shape[1])
    return out


def swizzle_col_scales(sc: np.ndarray) -> np.ndarray:
    sc = np.ascontiguousarray(sc, dtype=np.uint8)
    out = np.zeros(sc.size, np.uint8)
    lib().te_swizzle_col_scales_ref(_ptr(sc), _ptr(out), sc.shape[0], sc.shape[1])
    return out


def delayed_scaling(hist: np.ndarray, scale: np.ndarray, algo: str, fmt: int, margin: float = 0.0):
    hist = np.ascontiguousarray(hist, np.float32)
    scale = np.ascontiguousarray(scale, np.float32)
    nh, ns = np.empty_like(hist), np.empty_like(scale)
    lib().te_delayed_scaling_ref(_ptr(hist), _ptr(scale), _ptr(nh), _ptr(ns), hist.shape[0],
                                 hist.shape[1], 1 if algo == "max" else 0,
                                 ctypes.c_float(FP8_MAX[fmt]), ctypes.c_float(margin))
    return nh, ns


def delayed_scaling_bulk(reduced, hist: np.ndarray, scale: np.ndarray, algo: str, fmt: int,
                         margin: float = 0.0):
    hist = np.array(hist, np.float32, copy=True, order="C")
    scale = np.array(scale, np.float32, copy=True, order="C")
    red = None if reduced is None else np.ascontiguousarray(reduced, np.float32)
    lib().te_delayed_scaling_bulk_ref(_ptr(red), _ptr(hist), _ptr(scale), hist.shape[0],
                                      hist.shape[1], 1 if algo == "max" else 0,
                                      ctypes.c_float(FP8_MAX[fmt]), ctypes.c_float(margin))
    return hist, scale


def dequant(q: np.ndarray, fmt: int, scale_inv: float) -> np.ndarray:
    q = np.ascontiguousarray(q, np.uint8)
    out = np.empty(q.shape, np.float32)
    lib().te_dequant_ref(_ptr(q), fmt, ctypes.c_float(scale_inv), _ptr(out), q.size)
    return out


def dequant_mxfp8(q: np.ndarray, fmt: int, sc: np.ndarray, rowwise: bool) -> np.ndarray:
    q = np.ascontiguousarray(q, np.uint8)
    sc = np.ascontiguousarray(sc, np.uint8)
    out = np.empty(q.shape, np.float32)
    lib().te_dequant_mxfp8_ref(_ptr(q), fmt, _ptr(sc), sc.shape[1], int(rowwise), _ptr(out),
                               q.shape[0], q.shape[1])
    return out


def gemm_tn(P: np.ndarray, Q: np.ndarray, alpha: float = 1.0, beta: float = 0.0, D=None,
            fast: bool = False) -> np.ndarray:
    """D[i, j] = alpha * sum_k P[i, k] Q[j, k] + beta * D."""
    P = np.ascontiguousarray(P, np.float32)
    Q = np.ascontiguousarray(Q, np.float32)
    n, k = P.shape
    m = Q.shape[0]
    out = np.zeros((n, m), np.float32) if D is None else np.array(D, np.float32, copy=True, order="C")
    if fast:
        lib().te_gemm_tn_f32_fast(_ptr(P), _ptr(Q), _ptr(out), n, m, k)
    else:
        lib().te_gemm_tn_f32_ref(_ptr(P), _ptr(Q), _ptr(out), n, m, k, ctypes.c_float(alpha),
                                 ctypes.c_float(beta))
    return out


def gelu(x: np.ndarray) -> np.ndarray:
    x = np.ascontiguousarray(x, np.float32)
    out = np.empty_like(x)
    lib().te_gelu_ref(_ptr(x), _ptr(out), x.size, 0)
    return out


def dgelu(x: np.ndarray) -> np.ndarray:
    x = np.ascontiguousarray(x, np.float32)
    out = np.empty_like(x)
    lib().te_gelu_ref(_ptr(x), _ptr(out), x.size, 1)
    return out


def norm_forward(x: np.ndarray, gamma: np.ndarray, beta, eps: float, *, scale: float = 1.0,
                 zero_centered_gamma: bool = False, out_dtype: int = DT_F32):
    """LayerNorm (beta given) / RMSNorm (beta None).  Returns (pre-conversion output * scale [N, H] fp32,
    mu or None, rsigma, amax) - ref tests/cpp/operator/test_normalization.h:33-135."""
    x = np.ascontiguousarray(x)
    gamma = np.ascontiguousarray(gamma)
    N, H = x.shape
    is_ln = beta is not None
    beta_c = np.ascontiguousarray(beta) if is_ln else None
    mu = np.zeros(N, np.float32)
    rs = np.zeros(N, np.float32)
    lib().te_norm_stats_ref(int(is_ln), _ptr(x), _in_dtype(x), _ptr(mu), _ptr(rs), N, H, float(eps))
    out = np.empty((N, H), np.float32)
    amax = np.zeros(1, np.float32)
    lib().te_norm_output_ref(int(is_ln), _ptr(x), _ptr(gamma), _ptr(beta_c) if is_ln else None, _in_dtype(x),
                             _in_dtype(gamma), _ptr(out), _ptr(mu), _ptr(rs), N, H, _ptr(amax),
                             ctypes.c_float(scale), int(zero_centered_gamma), out_dtype)
    return out, (mu if is_ln else None), rs, amax[0]


def norm_backward(dz: np.ndarray, x: np.ndarray, mu, rsigma: np.ndarray, gamma: np.ndarray, *, add=None,
                  zero_centered_gamma: bool = False):
    """Returns (dx, dgamma, dbeta or None) in fp32 - ref test_normalization.h:138-190."""
    dz, x, gamma = np.ascontiguousarray(dz), np.ascontiguousarray(x), np.ascontiguousarray(gamma)
    N, H = x.shape
    is_ln = mu is not None
    dx = np.empty((N, H), np.float32)
    dg = np.empty(H, np.float32)
    db = np.empty(H, np.float32) if is_ln else None
    add_c = np.ascontiguousarray(add) if add is not None else None
    mu_c = np.ascontiguousarray(mu, np.float32) if is_ln else None
    rs_c = np.ascontiguousarray(rsigma, np.float32)
    lib().te_norm_backward_ref(int(is_ln), _ptr(dz), _ptr(add_c) if add_c is not None else None, _ptr(x),
                               _in_dtype(x), _in_dtype(gamma), _ptr(mu_c) if is_ln else None, _ptr(rs_c), _ptr(gamma),
                               _ptr(dx), _ptr(dg), _ptr(db) if is_ln else None, N, H, int(zero_centered_gamma))
    return dx, dg, db


ACT_IDS = {"none": 0, "gelu": 1, "silu": 2, "relu": 3, "qgelu": 4, "srelu": 5, "sigmoid": 6}
ACT_MODES = {"unary": 0, "dact": 1, "gated": 2, "dgated": 3}


def activation(mode: str, act: str, x: np.ndarray, grad=None) -> np.ndarray:
    """fp32 result of act / grad*act' / gated forward / gated backward (ref common/util/math.h,
    util/vectorized_pointwise.h:410-560); x is [R, H] or, for the gated modes, [R, 2H]."""
    x = np.ascontiguousarray(x)
    R = x.shape[0]
    H = x.shape[1] // 2 if mode in ("gated", "dgated") else x.shape[1]
    out = np.empty((R, 2 * H if mode == "dgated" else H), np.float32)
    g = np.ascontiguousarray(grad) if grad is not None else None
    lib().te_act_ref(ACT_MODES[mode], ACT_IDS[act], _ptr(x), _ptr(g) if g is not None else None, _in_dtype(x),
                     _ptr(out), R, H)
    return out


def clamped_swiglu(x: np.ndarray, grad=None, *, limit: float, alpha: float, offset: float = 1.0) -> np.ndarray:
    """x [R, 2H] (activation half first); grad None -> forward [R, H], else backward [R, 2H] (fp32)."""
    x = np.ascontiguousarray(x)
    R, H = x.shape[0], x.shape[1] // 2
    out = np.empty((R, H if grad is None else 2 * H), np.float32)
    g = None if grad is None else np.ascontiguousarray(grad)
    f = ctypes.c_float
    lib().te_clamped_swiglu_ref(0 if grad is None else 1, _ptr(x), _ptr(g), _in_dtype(x), _ptr(out), R, H, f(limit),
                                f(alpha), f(offset))
    return out


def scaled_activation(kind: str, x: np.ndarray, scales: np.ndarray, grad=None, *, interleave: int = 0,
                      limit: float = 0.5, alpha: float = 1.702, offset: float = 0.5):
    """Activation times a per-row scale and its backward, fp32 numpy restatement of the reference's own host
    oracle tests/cpp/operator/test_scaled_activation.cu:46-133 (glu_indices, gated_grads, compute_reference;
    activation forms tests/cpp/test_common.h:522-531).  kind: "swiglu" | "clamped_swiglu" | "srelu";
    x [R, 2H] (gated: [act | gate], or blocks of `interleave` act / gate columns) or [R, H]; scales [R];
    grad [R, H] or None.  Returns (out [R, H], grad_input like x or None, grad_scales [R] or None), all fp32."""
    f = np.float32
    x = np.ascontiguousarray(x, f)
    sc = np.ascontiguousarray(scales, f).reshape(-1, 1)
    sig = lambda v: (f(1) / (f(1) + np.exp(-v, dtype=f))).astype(f)
    gated = kind != "srelu"
    R = x.shape[0]
    if gated:
        H = x.shape[1] // 2
        if interleave > 0:                                  # test_scaled_activation.cu:46-59
            blk = x.reshape(R, H // interleave, 2, interleave)
            a, g = blk[:, :, 0, :].reshape(R, H), blk[:, :, 1, :].reshape(R, H)
        else:
            a, g = x[:, :H], x[:, H:]
    else:
        H = x.shape[1]
        a, g = x, None
    if kind == "swiglu":                                    # :64-70
        s = sig(a)
        act = a * s
        unscaled = act * g
        dact = (a * (s * (f(1) - s)) + s) * g
        dlin = act
    elif kind == "clamped_swiglu":                          # :71-82
        lim, al = f(limit), f(alpha)
        c = np.minimum(lim, a)
        s = sig(al * c)
        act = c * s
        base = np.where(a <= lim, al * c * (s * (f(1) - s)) + s, f(0)).astype(f)
        lin = np.minimum(np.maximum(-lim, g), lim) + f(offset)
        unscaled = act * lin
        dact = base * lin
        dlin = np.where((g <= lim) & (g >= -lim), act, f(0)).astype(f)
    else:                                                   # :83-88, test_common.h:530-531
        unscaled = np.where(a > 0, a * a, f(0)).astype(f)
        dact = np.maximum(f(0), f(2) * a)
        dlin = None
    out = (unscaled * sc).astype(f)
    if grad is None:
        return out, None, None
    dy = np.ascontiguousarray(grad, f)
    sg = dy * sc                                            # :116-118,122-123
    da = sg * dact
    if gated:
        dg = sg * dlin
        if interleave > 0:
            dx = np.stack([da.reshape(R, H // interleave, interleave), dg.reshape(R, H // interleave, interleave)],
                          axis=2).reshape(R, 2 * H)
        else:
            dx = np.concatenate([da, dg], axis=1)
    else:
        dx = da
    dscales = np.sum((dy * unscaled).astype(np.float64), axis=1).astype(f)   # :127 (serial fp32 sum there)
    return out, np.ascontiguousarray(dx, f), dscales


def colsum(x: np.ndarray) -> np.ndarray:
    x = np.ascontiguousarray(x)
    rows = int(np.prod(x.shape[:-1]))
    out = np.zeros(x.shape[-1], np.float32)
    lib().te_colsum_ref(_ptr(x), _in_dtype(x), rows, x.shape[-1], _ptr(out))
    return out


# ---------------------------------------------------------------- FP8 weight-update path
def mxfp8_partial_amax(shard: np.ndarray, rows: int, cols: int, start: int):
    """Block amaxes of the shard [start, start+len) of a [rows, cols] matrix, in padded fp32 arrays
    of the MXFP8 scale shapes (blocks the shard does not touch, and the padding, are 0)."""
    shard = np.ascontiguousarray(shard).reshape(-1)
    rs, cs = mxfp8_scale_shapes(rows, cols)
    ar, ac = np.zeros(rs, np.float32), np.zeros(cs, np.float32)
    lib().te_mxfp8_partial_amax_ref(_ptr(shard), _in_dtype(shard), shard.size, rows, cols, start, _ptr(ar),
                                    rs[1], _ptr(ac), cs[1])
    return ar, ac


def e8m0_from_amax(amax: np.ndarray) -> np.ndarray:
    amax = np.ascontiguousarray(amax, np.float32)
    out = np.empty(amax.shape, np.uint8)
    lib().te_e8m0_from_amax_ref(_ptr(amax), amax.size, _ptr(out))
    return out


def mxfp8_partial_cast(shard: np.ndarray, sc_row: np.ndarray, sc_col: np.ndarray, rows: int, cols: int,
                       start: int):
    shard = np.ascontiguousarray(shard).reshape(-1)
    sc_row, sc_col = np.ascontiguousarray(sc_row, np.uint8), np.ascontiguousarray(sc_col, np.uint8)
    out_r, out_c = np.empty(shard.size, np.uint8), np.empty(shard.size, np.uint8)
    lib().te_mxfp8_partial_cast_ref(_ptr(shard), _in_dtype(shard), shard.size, rows, cols, start, _ptr(sc_row),
                                    sc_row.shape[1], _ptr(sc_col), sc_col.shape[1], _ptr(out_r), _ptr(out_c))
    return out_r, out_c


def scale_and_scale_inv(amax: np.ndarray, fmt: int, force_pow_2: bool = False, epsilon: float = 0.0):
    amax = np.ascontiguousarray(amax, np.float32)
    scale, sinv = np.empty_like(amax), np.empty_like(amax)
    lib().te_scale_and_scale_inv_ref(_ptr(amax), amax.size, ctypes.c_float(FP8_MAX[fmt]), int(force_pow_2),
                                     ctypes.c_float(epsilon), _ptr(scale), _ptr(sinv))
    return scale, sinv


def adam_fp8(g: np.ndarray, p_master: np.ndarray, m: np.ndarray, v: np.ndarray, scale: float, fmt: int, *,
             lr: float, beta1: float, beta2: float, eps: float, step: int, mode: int, bias_correction: int,
             weight_decay: float):
    """Returns (p_fp8 bytes, p_master, m, v, amax) after one Adam step on copies of the inputs."""
    g = np.ascontiguousarray(g).reshape(-1)
    pm, m, v = (np.array(a, np.float32, copy=True, order="C").reshape(-1) for a in (p_master, m, v))
    p8 = np.empty(g.size, np.uint8)
    f = ctypes.c_float
    amax = lib().te_adam_fp8_ref(_ptr(g), _in_dtype(g), g.size, _ptr(pm), _ptr(m), _ptr(v), _ptr(p8), f(scale),
                                 fmt, f(lr), f(beta1), f(beta2), f(eps), step, mode, bias_correction,
                                 f(weight_decay))
    return p8, pm, m, v, np.float32(amax)


def float_to_bf16_bits(x: np.ndarray) -> np.ndarray:
    """fp32 -> bf16 bit patterns (uint16), round-to-nearest-even."""
    u = np.ascontiguousarray(x, np.float32).view(np.uint32).astype(np.uint64)
    nan = (u & 0x7FFFFFFF) > 0x7F800000
    r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16).astype(np.uint16)
    r[nan] = ((u[nan] >> 16) | 0x40).astype(np.uint16)
    return r


def bf16_bits_to_float(b: np.ndarray) -> np.ndarray:
    return (np.ascontiguousarray(b, np.uint16).astype(np.uint32) << 16).view(np.float32)


def num_threads() -> int:
    return int(lib().te_oracle_num_threads())
