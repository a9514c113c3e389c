"""CPU-only checks of the C-ABI boundary: the library loads, exports every symbol that
include/transformer_engine/*.h declares, and its host-side logic (handle arena, descriptor
getters/setters, config objects, validators, error modes) behaves like the reference's
common/transformer_engine.cpp.  No compute call is made (no GPU here)."""
import ctypes as C
import glob
import os
import re
import subprocess
import sys
import threading

import pytest

from transformer_engine_amd import _lib
from transformer_engine_amd._lib import NVTEError, api

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    names = set()
    for h in glob.glob(os.path.join(ROOT, "include", "transformer_engine", "*.h")):
        txt = open(h).read()
        txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
        txt = re.sub(r"//[^\n]*", "", txt)
        for m in re.finditer(r"\b(nvte_[a-z0-9_]+)\s*\(", txt):
            names.add(m.group(1))
    return names


def test_every_declared_symbol_is_exported():
    declared = _declared_symbols()
    assert len(declared) >= 60
    out = subprocess.check_output(["nm", "-D", "--defined-only", _lib.LIB_PATH], text=True)
    exported = set()
    for line in out.splitlines():
        sym = line.split()[-1]
        exported.add(sym)
    mangled_ok = any("nvte_delayed_scaling_recipe_amax_and_scale_update_after_reduction" in s and s.startswith("_Z")
                     for s in exported)
    assert mangled_ok, "the C++-mangled after_reduction entry point (ref recipe.h:71) is missing"
    missing = [n for n in declared
               if n not in exported and n != "nvte_delayed_scaling_recipe_amax_and_scale_update_after_reduction"]
    assert not missing, f"declared but not exported: {missing}"
    # nothing but the nvte_* surface and the transformer_engine:: C++ helpers of runtime.h (the symbols the
    # reference's version script lists, ref common/libtransformer_engine.version:5-17) leaves the shared object
    leaked = [s for s in exported if not s.startswith("nvte_") and "nvte_delayed_scaling" not in s
              and not s.startswith("_ZN18transformer_engine")]
    assert not leaked, leaked[:5]


def test_ctypes_table_covers_the_headers():
    declared = _declared_symbols() - {"nvte_delayed_scaling_recipe_amax_and_scale_update_after_reduction"}
    assert declared <= set(_lib.exported_symbols()), declared - set(_lib.exported_symbols())


def test_enum_values_are_the_reference_abi():
    # ref include/transformer_engine/transformer_engine.h:25-38, 66-94, 99-119
    assert (_lib.kByte, _lib.kFloat32, _lib.kFloat16, _lib.kBFloat16, _lib.kFloat8E4M3, _lib.kFloat8E5M2,
            _lib.kFloat8E8M0) == (0, 4, 5, 6, 7, 8, 9)
    assert (_lib.kRowwiseData, _lib.kColumnwiseData, _lib.kScale, _lib.kAmax, _lib.kRowwiseScaleInv,
            _lib.kColumnwiseScaleInv, _lib.kColumnwiseAmax, _lib.kWithGEMMSwizzledScales) == tuple(range(8))
    assert C.sizeof(_lib.NVTEShape) == 16 * 8
    assert C.sizeof(_lib.NVTEBasicTensor) == 8 + 8 + 16 * 8


def _basic(ptr, dtype, dims):
    return _lib.NVTEBasicTensor(C.c_void_p(ptr), dtype, _lib.make_shape(dims))


def _set(h, param, ptr, dtype, dims):
    bt = _basic(ptr, dtype, dims)
    api.nvte_set_tensor_param_v2(h, param, C.byref(bt), C.sizeof(bt))


def test_handle_lifecycle_and_defaults():
    h = api.nvte_create_tensor(_lib.NVTE_DELAYED_TENSOR_SCALING)
    assert h
    assert api.nvte_tensor_type(h) == _lib.kFloat32
    s = api.nvte_tensor_shape(h)
    assert s.ndim == 1 and s.data[0] == 0          # "absent" = {nullptr, shape {0}} (ref common.h:251)
    assert api.nvte_tensor_numel(h) == 0
    assert api.nvte_tensor_data(h) is None
    assert api.nvte_tensor_scaling_mode(h) == 0
    api.nvte_destroy_tensor(h)
    api.nvte_destroy_tensor(h)                      # double destroy is harmless
    api.nvte_destroy_tensor(None)                   # null handle is ignored
    # null-handle getters return the documented defaults (ref transformer_engine.cpp:602-726)
    assert api.nvte_tensor_type(None) == _lib.kFloat32
    assert api.nvte_tensor_element_size(None) == 4
    assert api.nvte_tensor_size_bytes(None) == 0
    assert api.nvte_tensor_scale_inv_shape(None).ndim == 1
    with pytest.raises(NVTEError, match="Invalid tensor"):
        api.nvte_tensor_shape(None)


def test_batch_create_destroy_reuses_slots():
    n = 3000                                         # crosses an arena chunk boundary (1024)
    arr = (C.c_void_p * n)()
    api.nvte_create_tensors(_lib.NVTE_MXFP8_1D_SCALING, arr, n)
    vals = [arr[i] for i in range(n)]
    assert len(set(vals)) == n and all(vals)
    assert api.nvte_tensor_scaling_mode(arr[n - 1]) == _lib.NVTE_MXFP8_1D_SCALING
    api.nvte_destroy_tensors(arr, n)
    h = api.nvte_create_tensor(0)
    assert h in vals                                 # freed slot handed out again
    assert api.nvte_tensor_scaling_mode(h) == 0      # and fully reset
    api.nvte_destroy_tensor(h)


def test_set_get_params_and_logical_shape():
    h = api.nvte_create_tensor(0)
    _set(h, _lib.kRowwiseData, 0x1000, _lib.kFloat8E4M3, [4, 6, 32])
    _set(h, _lib.kRowwiseScaleInv, 0x2000, _lib.kFloat32, [1])
    assert api.nvte_tensor_type(h) == _lib.kFloat8E4M3
    assert api.nvte_tensor_ndims(h) == 3 and api.nvte_tensor_size(h, 2) == 32
    assert api.nvte_tensor_numel(h) == 4 * 6 * 32 and api.nvte_tensor_size_bytes(h) == 4 * 6 * 32
    assert api.nvte_tensor_data(h) == 0x1000 and api.nvte_tensor_scale_inv(h) == 0x2000
    bt = api.nvte_get_tensor_param(h, _lib.kRowwiseData)
    assert bt.data_ptr == 0x1000 and bt.dtype == _lib.kFloat8E4M3 and bt.shape.ndim == 3
    # columnwise-only per-tensor-scaled tensor reports the UN-transposed shape (ref common.h:386-406)
    h2 = api.nvte_create_tensor(0)
    _set(h2, _lib.kColumnwiseData, 0x3000, _lib.kFloat8E5M2, [32, 4, 6])
    s = api.nvte_tensor_shape(h2)
    assert [s.data[i] for i in range(s.ndim)] == [4, 6, 32]
    assert api.nvte_tensor_type(h2) == _lib.kFloat8E5M2
    cs = api.nvte_tensor_columnwise_shape(h2)
    assert [cs.data[i] for i in range(cs.ndim)] == [32, 4, 6]
    # MXFP8: columnwise data keeps the logical shape
    h3 = api.nvte_create_tensor(_lib.NVTE_MXFP8_1D_SCALING)
    _set(h3, _lib.kColumnwiseData, 0x3000, _lib.kFloat8E4M3, [64, 96])
    s = api.nvte_tensor_shape(h3)
    assert [s.data[i] for i in range(s.ndim)] == [64, 96]
    # bool flag through the v2 interface
    flag = C.c_uint8(1)
    api.nvte_set_tensor_param_v2(h3, _lib.kWithGEMMSwizzledScales, C.byref(flag), 1)
    got, n = C.c_uint8(0), C.c_size_t(0)
    api.nvte_get_tensor_param_v2(h3, _lib.kWithGEMMSwizzledScales, C.byref(got), 1, C.byref(n))
    assert got.value == 1 and n.value == 1
    api.nvte_get_tensor_param_v2(h3, _lib.kRowwiseData, None, 0, C.byref(n))       # size query
    assert n.value == C.sizeof(_lib.NVTEBasicTensor)
    with pytest.raises(NVTEError, match="Buffer is too small"):
        api.nvte_set_tensor_param_v2(h3, _lib.kRowwiseData, C.byref(flag), 1)
    with pytest.raises(NVTEError, match="Invalid NVTETensorParam"):
        api.nvte_set_tensor_param_v2(h3, 99, C.byref(flag), 1)
    for x in (h, h2, h3):
        api.nvte_destroy_tensor(x)


def test_quantization_and_matmul_config_objects():
    cfg = api.nvte_create_quantization_config()
    b, f, n = C.c_uint8(1), C.c_float(0.25), C.c_size_t(0)
    api.nvte_set_quantization_config_attribute(cfg, _lib.kQForcePow2Scales, C.byref(b), 1)
    api.nvte_set_quantization_config_attribute(cfg, _lib.kQAmaxEpsilon, C.byref(f), 4)
    gb, gf = C.c_uint8(0), C.c_float(0)
    api.nvte_get_quantization_config_attribute(cfg, _lib.kQForcePow2Scales, C.byref(gb), 1, C.byref(n))
    api.nvte_get_quantization_config_attribute(cfg, _lib.kQAmaxEpsilon, C.byref(gf), 4, C.byref(n))
    assert gb.value == 1 and gf.value == 0.25 and n.value == 4
    with pytest.raises(NVTEError, match="too small"):
        api.nvte_set_quantization_config_attribute(cfg, _lib.kQAmaxEpsilon, C.byref(f), 2)
    with pytest.raises(NVTEError, match="Unsupported"):
        api.nvte_set_quantization_config_attribute(cfg, 77, C.byref(f), 4)
    api.nvte_destroy_quantization_config(cfg)
    m = api.nvte_create_matmul_config()
    i = C.c_int(64)
    api.nvte_set_matmul_config_attribute(m, _lib.kMSMCount, C.byref(i), 4)
    gi = C.c_int(0)
    api.nvte_get_matmul_config_attribute(m, _lib.kMSMCount, C.byref(gi), 4, C.byref(n))
    assert gi.value == 64
    api.nvte_destroy_matmul_config(m)


def test_validators_fire_before_any_launch():
    """Shape / dtype contract violations are reported as errors (text as the reference's)."""
    hin = api.nvte_create_tensor(0)
    hout = api.nvte_create_tensor(0)
    _set(hin, _lib.kRowwiseData, 0x1000, _lib.kBFloat16, [128, 256])
    with pytest.raises(NVTEError, match="Either rowwise or columnwise output data need to be allocated"):
        api.nvte_quantize(hin, hout, None)
    _set(hout, _lib.kRowwiseData, 0x2000, _lib.kFloat8E4M3, [128, 256])
    with pytest.raises(NVTEError, match="_scale_inverse must be allocated"):
        api.nvte_quantize(hin, hout, None)
    _set(hout, _lib.kRowwiseScaleInv, 0x3000, _lib.kFloat32, [2])
    with pytest.raises(NVTEError, match="invalid scale_inv shape"):
        api.nvte_quantize(hin, hout, None)
    _set(hout, _lib.kRowwiseScaleInv, 0x3000, _lib.kFloat32, [1])
    _set(hout, _lib.kColumnwiseData, 0x4000, _lib.kFloat8E4M3, [256, 128])
    _set(hout, _lib.kColumnwiseScaleInv, 0x5000, _lib.kFloat32, [1])
    with pytest.raises(NVTEError, match="need to share scale-inverse"):
        api.nvte_quantize(hin, hout, None)
    # columnwise-only per-tensor quantize is rejected (ref dispatch/quantize.cuh:69-70)
    hc = api.nvte_create_tensor(0)
    _set(hc, _lib.kColumnwiseData, 0x4000, _lib.kFloat8E4M3, [256, 128])
    _set(hc, _lib.kColumnwiseScaleInv, 0x5000, _lib.kFloat32, [1])
    with pytest.raises(NVTEError, match="only the columnwise direction not supported"):
        api.nvte_quantize(hin, hc, None)
    # MXFP8 scale shapes must be padded to (128, 4) (ref transformer_engine.cpp:93-115)
    hm = api.nvte_create_tensor(_lib.NVTE_MXFP8_1D_SCALING)
    _set(hm, _lib.kRowwiseData, 0x2000, _lib.kFloat8E4M3, [128, 256])
    _set(hm, _lib.kRowwiseScaleInv, 0x3000, _lib.kFloat8E8M0, [128, 7])
    with pytest.raises(NVTEError, match=r"expected \(128,8\)"):
        api.nvte_quantize(hin, hm, None)
    # GEMM argument checks
    ha, hb, hd = (api.nvte_create_tensor(0) for _ in range(3))
    _set(ha, _lib.kRowwiseData, 0x1000, _lib.kFloat8E4M3, [64, 32]); _set(ha, _lib.kRowwiseScaleInv, 0x10, _lib.kFloat32, [1])
    _set(hb, _lib.kRowwiseData, 0x2000, _lib.kFloat8E4M3, [16, 48]); _set(hb, _lib.kRowwiseScaleInv, 0x20, _lib.kFloat32, [1])
    _set(hd, _lib.kRowwiseData, 0x3000, _lib.kBFloat16, [16, 64])
    al, be = C.c_float(1), C.c_float(0)
    with pytest.raises(NVTEError, match="incompatible dimensions"):
        api.nvte_cublas_gemm_v2(1, 0, C.byref(al), ha, hb, C.byref(be), hd, hd, None, None, None)
    with pytest.raises(NVTEError, match="amax_compute_algo|Unsupported amax compute algorithm"):
        h1 = api.nvte_create_tensor(0); _set(h1, _lib.kRowwiseData, 0x1, _lib.kFloat32, [4, 2])
        h2 = api.nvte_create_tensor(0); _set(h2, _lib.kRowwiseData, 0x2, _lib.kFloat32, [2])
        api.nvte_delayed_scaling_recipe_amax_and_scale_update(h1, h2, h1, h2, b"median", _lib.kFloat8E4M3, 0.0, None)
    for x in (hin, hout, hc, hm, ha, hb, hd):
        api.nvte_destroy_tensor(x)


def test_error_modes():
    raw = _lib.raw_lib
    assert raw.nvte_amd_get_error_mode() == 1           # the ctypes binding records instead of throwing
    raw.nvte_tensor_shape(None)
    assert b"Invalid tensor" in raw.nvte_amd_last_error()
    raw.nvte_amd_clear_error()
    assert raw.nvte_amd_last_error() is None
    h = raw.nvte_create_tensor(0)                        # a successful call leaves no error behind
    assert raw.nvte_amd_last_error() is None
    raw.nvte_destroy_tensor(h)
    # errors are per thread
    seen = {}

    def worker():
        raw.nvte_tensor_shape(None)
        seen["t"] = raw.nvte_amd_last_error()

    t = threading.Thread(target=worker); t.start(); t.join()
    assert seen["t"] is not None and raw.nvte_amd_last_error() is None


def test_arena_is_thread_safe():
    errs = []

    def worker():
        try:
            for _ in range(2000):
                h = api.nvte_create_tensor(0)
                _set(h, _lib.kRowwiseData, 0x1000, _lib.kBFloat16, [8, 8])
                assert api.nvte_tensor_numel(h) == 64
                api.nvte_destroy_tensor(h)
        except Exception as e:  # pragma: no cover
            errs.append(e)

    ts = [threading.Thread(target=worker) for _ in range(8)]
    [t.start() for t in ts]; [t.join() for t in ts]
    assert not errs


def test_misc_entry_points():
    assert api.nvte_is_non_tn_fp8_gemm_supported() == (1 if os.environ.get("NVTE_AMD_NON_TN_FP8_GEMM", "0") not in ("", "0") else 0)
    # the opt-in answer, in a process of its own (the library reads the variable once)
    code = ("import sys; sys.path.insert(0, %r); from transformer_engine_amd import _lib; "
            "print(_lib.api.nvte_is_non_tn_fp8_gemm_supported())" % ROOT)
    out = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, NVTE_AMD_NON_TN_FP8_GEMM="1"),
                         capture_output=True, text=True, timeout=120)
    assert out.stdout.strip().endswith("1"), (out.stdout, out.stderr)
    assert api.nvte_get_num_compute_streams() == 4
    s = api.nvte_make_shape((C.c_size_t * 3)(2, 3, 4), 3)
    assert s.ndim == 3 and s.data[2] == 4
    with pytest.raises(NVTEError, match="Too many dims"):
        api.nvte_make_shape(None, 16)
    pack = _lib.NVTETensorPack()
    api.nvte_tensor_pack_create(C.byref(pack))
    assert all(pack.tensors[i] for i in range(10))
    api.nvte_tensor_pack_destroy(C.byref(pack))
    assert "gfx950" in _lib.version()


def test_grouped_tensor_handles_and_config_host_only():
    """Grouped-tensor descriptor plumbing needs no GPU (ref transformer_engine.cpp:1147-1365)."""
    import ctypes as C
    from transformer_engine_amd import _lib
    api = _lib.api
    h = api.nvte_create_grouped_tensor(_lib.NVTE_DELAYED_TENSOR_SCALING, 3, _lib.make_shape([96, 64]))
    assert h
    assert api.nvte_grouped_tensor_num_tensors(h) == 3
    assert api.nvte_grouped_tensor_scaling_mode(h) == _lib.NVTE_DELAYED_TENSOR_SCALING
    s = api.nvte_get_grouped_tensor_logical_shape(h)
    assert (s.ndim, s.data[0], s.data[1]) == (2, 96, 64)
    bt = _lib.NVTEBasicTensor(C.c_void_p(0x1000), _lib.kFloat8E4M3, _lib.make_shape([96 * 64]))
    api.nvte_set_grouped_tensor_param(h, 0, C.byref(bt), C.sizeof(bt))
    assert api.nvte_grouped_tensor_type(h) == _lib.kFloat8E4M3
    got = _lib.NVTEBasicTensor()
    written = C.c_size_t(0)
    api.nvte_get_grouped_tensor_param(h, 0, C.byref(got), C.sizeof(got), C.byref(written))
    assert written.value == C.sizeof(got) and got.data_ptr == 0x1000 and got.shape.data[0] == 96 * 64
    flag = C.c_uint8(1)
    api.nvte_set_grouped_tensor_param(h, 10, C.byref(flag), 1)
    back = C.c_uint8(0)
    api.nvte_get_grouped_tensor_param(h, 10, C.byref(back), 1, None)
    assert back.value == 1
    with pytest.raises(_lib.NVTEError, match="Buffer is too small"):
        api.nvte_set_grouped_tensor_param(h, 0, C.byref(bt), 4)
    api.nvte_destroy_grouped_tensor(h)
    assert api.nvte_grouped_tensor_num_tensors(h) == 0          # dead handle reads as empty
    with pytest.raises(_lib.NVTEError, match="Number of tensors must be greater than 0"):
        api.nvte_create_grouped_tensor(0, 0, _lib.make_shape([1, 1]))
    # grouped matmul config attributes round-trip; setup workspace grows with the member count
    cfg = api.nvte_create_grouped_matmul_config()
    v = C.c_int64(4096)
    api.nvte_set_grouped_matmul_config_attribute(cfg, 0, C.byref(v), 8)
    r = C.c_int64(0)
    api.nvte_get_grouped_matmul_config_attribute(cfg, 0, C.byref(r), 8, None)
    assert r.value == 4096
    api.nvte_destroy_grouped_matmul_config(cfg)
    assert api.nvte_get_grouped_gemm_setup_workspace_size(64) > api.nvte_get_grouped_gemm_setup_workspace_size(8) > 0
