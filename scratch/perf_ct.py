import sys; sys.path.insert(0,'.')
import torch, time
from transformer_engine_amd.tensor import *
from transformer_engine_amd._lib import api
def timeit(fn, it=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    s,e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e)/it
for shape in [(8192,14336),(8192,4096),(14336,4096),(16384,16384)]:
    x = torch.randn(shape, device="cuda", dtype=torch.bfloat16)
    n = x.numel()
    scale = torch.ones(1, device="cuda"); amax = torch.zeros(1, device="cuda")
    q = Float8Quantizer(scale, amax, DType.kFloat8E4M3)
    out = q.make_empty(x.shape, dtype=x.dtype, device=x.device)
    stream = current_stream_ptr()
    wi = wrap_plain(x); wo = q._wrap_out(out); wo.set_scale(scale).set_amax(amax)
    t1 = timeit(lambda: api.nvte_quantize(wi.handle, wo.handle, stream))
    wo2 = q._wrap_out(out); wo2.set_scale(scale)
    t2 = timeit(lambda: api.nvte_quantize(wi.handle, wo2.handle, stream))
    # rowwise only
    q3 = Float8Quantizer(scale, amax, DType.kFloat8E4M3, columnwise=False); out3 = q3.make_empty(x.shape, dtype=x.dtype, device=x.device)
    wo3 = q3._wrap_out(out3); wo3.set_scale(scale).set_amax(amax)
    t3 = timeit(lambda: api.nvte_quantize(wi.handle, wo3.handle, stream))
    # amax only
    wo4 = q._wrap_out(out); wo4.set_scale(scale).set_amax(amax)
    t4 = timeit(lambda: api.nvte_compute_amax(wi.handle, wo4.handle, stream))
    # transpose of bytes
    b = out._data; bt = torch.empty((shape[1], shape[0]), dtype=torch.uint8, device="cuda")
    wb, wbt = wrap_plain(b), wrap_plain(bt)
    t5 = timeit(lambda: api.nvte_transpose(wb.handle, wbt.handle, stream))
    # mxfp8
    qm = MXFP8Quantizer(DType.kFloat8E4M3); om = qm.make_empty(x.shape, dtype=x.dtype, device=x.device); wm = om.wrap()
    t6 = timeit(lambda: api.nvte_quantize(wi.handle, wm.handle, stream))
    y = torch.empty_like(x)
    t0 = timeit(lambda: y.copy_(x))
    print(f"{shape}: copy {n*4/t0/1e6:.0f} GB/s | ct+amax {n*4/t1/1e6:.0f} | ct noamax {n*4/t2/1e6:.0f} | rowwise {n*3/t3/1e6:.0f} | amax {n*2/t4/1e6:.0f} | transpose_u8 {n*2/t5/1e6:.0f} | mxfp8 both {n*4.0625/t6/1e6:.0f} GB/s")
