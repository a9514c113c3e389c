"""Bandwidth of every quantize-side kernel of section 8(a) on the configs[1]/[2] tensor sizes
(C-ABI calls, outputs preallocated, back-to-back launches: upper bounds incl. Infinity-Cache help
for tensors < 256 MiB; dY-sized tensors are HBM-bound)."""
import sys; sys.path.insert(0, '.')
import time, torch
from transformer_engine_amd import api
from transformer_engine_amd.tensor import *
from transformer_engine_amd.tensor import wrap_plain, TensorWrapper, current_stream_ptr, QuantizationConfig
def timeit(fn, it=30):
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / it
st = current_stream_ptr()
def report(name, f, nbytes):
    f(); t0 = time.time()
    while time.time() - t0 < 0.3: f()
    t = sorted(timeit(f) for _ in range(3))[1]
    print(f"{name:70s} {t*1e3:8.1f} us  {nbytes/t/1e6:7.0f} GB/s  ({nbytes/1e6:.0f} MB algorithmic)")
for (R, C, tag) in [(8192, 14336, "dY"), (8192, 4096, "X"), (8192, 8192, "C3 operand")]:
    x = torch.randn(R, C, device="cuda", dtype=torch.bfloat16); wx = wrap_plain(x)
    n = R * C
    # Q2 cast_transpose (delayed)
    q = Float8Quantizer(torch.full((1,), 4.0, device="cuda"), torch.zeros(1, device="cuda"), DType.kFloat8E5M2)
    o = q.make_empty((R, C), dtype=x.dtype, device="cuda"); wo = q._wrap_out(o); wo.set_scale(q.scale).set_amax(q.amax)
    report(f"[{tag} {R}x{C}] Q2 cast_transpose bf16->fp8 row+T", lambda: api.nvte_quantize(wx.handle, wo.handle, st), n * 4)
    # Q3 rowwise-only cast
    q3 = Float8Quantizer(torch.full((1,), 4.0, device="cuda"), torch.zeros(1, device="cuda"), DType.kFloat8E4M3, columnwise=False)
    o3 = q3.make_empty((R, C), dtype=x.dtype, device="cuda"); wo3 = q3._wrap_out(o3); wo3.set_scale(q3.scale).set_amax(q3.amax)
    report(f"[{tag}] Q3 rowwise cast bf16->fp8", lambda: api.nvte_quantize(wx.handle, wo3.handle, st), n * 3)
    # Q4 amax
    cfg = QuantizationConfig(False, 0.0, None)
    report(f"[{tag}] Q4 compute_amax", lambda: api.nvte_compute_amax_with_config(wx.handle, wo.handle, cfg.handle, st), n * 2)
    # Q8 transpose of fp8 bytes
    wt_in = TensorWrapper().set_rowwise_data(o._data, DType.kFloat8E5M2); tb = torch.empty(C, R, dtype=torch.uint8, device="cuda")
    wt_out = TensorWrapper().set_rowwise_data(tb, DType.kFloat8E5M2)
    report(f"[{tag}] Q8 transpose fp8", lambda: api.nvte_transpose(wt_in.handle, wt_out.handle, st), n * 2)
    # Q10 dequantize
    d = torch.empty(R, C, device="cuda", dtype=torch.bfloat16); wd = wrap_plain(d)
    wdq = TensorWrapper().set_rowwise_data(o._data, DType.kFloat8E5M2); wdq.set_rowwise_scale_inv(o._scale_inv)
    report(f"[{tag}] Q10 dequantize fp8->bf16", lambda: api.nvte_dequantize(wdq.handle, wd.handle, st), n * 3)
    # Q6 MXFP8 row+col, rowwise only, row + transposed-columnwise
    for name, kw, bpe in (("row+col (reference layout)", dict(), 4.0625), ("rowwise only", dict(columnwise=False), 3.03125),
                          ("row + transposed columnwise (amd_ext)", dict(columnwise_transposed=True), 4.0625)):
        qm = MXFP8Quantizer(DType.kFloat8E4M3, **kw)
        om = qm.make_empty((R, C), dtype=x.dtype, device="cuda"); wm = om.wrap()
        report(f"[{tag}] Q6 mxfp8 {name}", lambda: api.nvte_quantize(wx.handle, wm.handle, st), n * bpe)
    qm = MXFP8Quantizer(DType.kFloat8E4M3, columnwise=False); qm.optimize_for_gemm = True
    om = qm.make_empty((R, C), dtype=x.dtype, device="cuda"); wm = om.wrap()
    report(f"[{tag}] Q6+Q7 mxfp8 rowwise, GEMM-swizzled scales emitted directly", lambda: api.nvte_quantize(wx.handle, wm.handle, st), n * 3.03125)
