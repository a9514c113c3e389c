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
def fp8_out(shape, fmt=DType.kFloat8E5M2, columnwise=True):
    q = Float8Quantizer(torch.full((1,), 4.0, device="cuda"), torch.zeros(1, device="cuda"), fmt, columnwise=columnwise)
    o = q.make_empty(shape, dtype=torch.bfloat16, device="cuda")
    w = q._wrap_out(o); w.set_scale(q.scale).set_amax(q.amax)
    return o, w
x1 = torch.randn(N, H, device="cuda", dtype=torch.bfloat16); wx1 = wrap_plain(x1)
g = torch.randn(N, H, device="cuda", dtype=torch.bfloat16); wg = wrap_plain(g)
o3, wo3 = fp8_out((N, H)); o4, wo4 = fp8_out((N, H), columnwise=False)
hp = torch.empty(N, H, device="cuda", dtype=torch.bfloat16); whp = wrap_plain(hp)
db = torch.empty(H, device="cuda", dtype=torch.bfloat16); wdb = wrap_plain(db)
ws = wrap_plain(torch.empty(N // 128, H, device="cuda", dtype=torch.float32))
cases = [
  ("cast_transpose plain", lambda: api.nvte_quantize(wg.handle, wo3.handle, st), N * H * 4),
  ("relu -> fp8 + T (fused kernel, no dbias)", lambda: api.nvte_relu(wx1.handle, wo3.handle, st), N * H * 4),
  ("relu -> fp8 rowwise only", lambda: api.nvte_relu(wx1.handle, wo4.handle, st), N * H * 3),
  ("relu -> bf16", lambda: api.nvte_relu(wx1.handle, whp.handle, st), N * H * 4),
  ("quantize_dbias fused (+T +dbias)", lambda: api.nvte_quantize_dbias(wg.handle, wo3.handle, wdb.handle, ws.handle, st), N * H * 4),
  ("drelu -> fp8 + T", lambda: api.nvte_drelu(wg.handle, wx1.handle, wo3.handle, st), N * H * 6),
]
t0 = time.time()
while time.time() - t0 < 0.5: cases[0][1]()
for name, f, nbytes in cases:
    t = sorted(timeit(f) for _ in range(3))[1]
    print(f"{name}: {t*1e3:.1f} us  {nbytes/t/1e6:.0f} GB/s algorithmic")
