import sys; sys.path.insert(0, '.')
import time, torch
from transformer_engine_amd import api
from transformer_engine_amd.tensor import *
from transformer_engine_amd.tensor import wrap_plain, TensorWrapper, current_stream_ptr
def timeit(fn, it=30):
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / it
N, H = 8192, 14336
st = current_stream_ptr()
def fp8_out(shape, fmt=DType.kFloat8E4M3):
    q = Float8Quantizer(torch.full((1,), 4.0, device="cuda"), torch.zeros(1, device="cuda"), fmt)
    o = q.make_empty(shape, dtype=torch.bfloat16, device="cuda")
    w = q._wrap_out(o); w.set_scale(q.scale).set_amax(q.amax)
    return o, w
x2 = torch.randn(N, 2 * H, device="cuda", dtype=torch.bfloat16); wx2 = wrap_plain(x2)
x1 = torch.randn(N, H, device="cuda", dtype=torch.bfloat16); wx1 = wrap_plain(x1)
g = torch.randn(N, H, device="cuda", dtype=torch.bfloat16); wg = wrap_plain(g)
o1, wo1 = fp8_out((N, H)); o2, wo2 = fp8_out((N, 2 * H), DType.kFloat8E5M2); o3, wo3 = fp8_out((N, H), DType.kFloat8E5M2)
db = torch.empty(H, device="cuda", dtype=torch.bfloat16); wdb = wrap_plain(db)
ws = wrap_plain(torch.empty(N // 128, H, device="cuda", dtype=torch.float32))
cases = [
  ("swiglu [8192,2x14336] -> fp8 + transpose", lambda: api.nvte_swiglu(wx2.handle, wo1.handle, st), N * H * (4 + 2)),
  ("dswiglu -> fp8 [8192,2x14336] + transpose", lambda: api.nvte_dswiglu(wg.handle, wx2.handle, wo2.handle, st), N * H * (6 + 4)),
  ("gelu [8192,14336] -> fp8 + transpose", lambda: api.nvte_gelu(wx1.handle, wo1.handle, st), N * H * 4),
  ("quantize_dbias_dgelu -> fp8 + transpose + dbias", lambda: api.nvte_quantize_dbias_dgelu(wg.handle, wx1.handle, wo3.handle, wdb.handle, ws.handle, st), N * H * 6),
  ("quantize_dbias (fused) -> fp8 + transpose + dbias", lambda: api.nvte_quantize_dbias(wg.handle, wo3.handle, wdb.handle, ws.handle, st), N * H * 4),
  ("cast_transpose (plain, for reference)", lambda: api.nvte_quantize(wg.handle, wo3.handle, st), N * H * 4),
]
t0 = time.time()
while time.time() - t0 < 0.5: cases[0][1]()
for name, f, nbytes in cases:
    t = sorted(timeit(f) for _ in range(3))[1]
    print(f"{name}: {t*1e3:.1f} us  {nbytes/t/1e6:.0f} GB/s algorithmic")
