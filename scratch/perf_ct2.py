"""cast_transpose (persistent tile kernel): isolated (back-to-back) and cold (a 640 MB write between launches,
the cache state a real step leaves) timings for the configs[1] tensors.  NVTE_AMD_CT_WGS_PER_CU varies the grid."""
import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from transformer_engine_amd.tensor import *
from transformer_engine_amd._lib import api
def timeit(fn, it=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / it
def cold(fn, flush, it=12, read_only=False):
    evs = []
    for i in range(it):
        if read_only:
            flush.view(torch.int32).sum()      # leaves CLEAN lines of another buffer in the caches
        else:
            flush.fill_(i)                     # leaves DIRTY lines: their write-back lands in the timed kernel
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); evs.append((a, b))
    torch.cuda.synchronize()
    v = sorted(a.elapsed_time(b) for a, b in evs[2:])
    return v[len(v) // 2]
flush = torch.empty(640 * 1024 * 1024, dtype=torch.uint8, device="cuda")
res = []
for name, shape, fmt in (("dY", (8192, 14336), DType.kFloat8E5M2), ("X", (8192, 4096), DType.kFloat8E4M3), ("W", (14336, 4096), DType.kFloat8E4M3),
                         ("ragged", (8192 + 48, 4096 + 72), DType.kFloat8E4M3)):
    x = torch.randn(shape, device="cuda", dtype=torch.bfloat16)
    n = x.numel()
    scale = torch.ones(1, device="cuda"); amax = torch.zeros(1, device="cuda")
    q = Float8Quantizer(scale, amax, fmt)
    out = q.make_empty(x.shape, dtype=x.dtype, device=x.device)
    stream = current_stream_ptr()
    wi = wrap_plain(x); wo = q._wrap_out(out); wo.set_scale(scale).set_amax(amax)
    fn = lambda: api.nvte_quantize(wi.handle, wo.handle, stream)
    t1 = timeit(fn); t2 = cold(fn, flush); t3 = cold(fn, flush, read_only=True)
    res.append(f"{name} {shape}: isolated {t1*1000:.1f} us = {n*4/t1/1e6:.0f} GB/s | after 640 MB write {t2*1000:.1f} us = {n*4/t2/1e6:.0f} GB/s | after 640 MB read {t3*1000:.1f} us = {n*4/t3/1e6:.0f} GB/s")
print("WGS_PER_CU", os.environ.get("NVTE_AMD_CT_WGS_PER_CU", "default"), " || ".join(res), flush=True)
