"""Bandwidth of the per-row-scaled activations on an expert-MLP sized tensor (C-ABI calls, outputs preallocated)."""
import sys; sys.path.insert(0, '.')
import ctypes as C, time, torch
from transformer_engine_amd import api
from transformer_engine_amd.tensor import wrap_plain, current_stream_ptr
def timeit(fn, it=20):
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / it
st = current_stream_ptr()
R, H = 16384, 14336
x = torch.randn(R, 2 * H, device="cuda", dtype=torch.bfloat16); sc = torch.rand(R, device="cuda")
dy = torch.randn(R, H, device="cuda", dtype=torch.bfloat16)
out = torch.empty(R, H, device="cuda", dtype=torch.bfloat16); dx = torch.empty_like(x); dsc = torch.empty(R, device="cuda")
wx, ws, wdy, wo, wdx, wds = (wrap_plain(t) for t in (x, sc, dy, out, dx, dsc))
xs = x[:, :H].contiguous(); dxs = torch.empty_like(xs); wxs, wdxs = wrap_plain(xs), wrap_plain(dxs)
cases = [
 ("scaled_swiglu fwd", lambda: api.nvte_scaled_swiglu(wx.handle, ws.handle, wo.handle, C.c_int64(0), st), R * H * 6),
 ("scaled_swiglu fwd interleave 32", lambda: api.nvte_scaled_swiglu(wx.handle, ws.handle, wo.handle, C.c_int64(32), st), R * H * 6),
 ("scaled_dswiglu bwd + scale grad", lambda: api.nvte_scaled_dswiglu(wdy.handle, wx.handle, ws.handle, wdx.handle, wds.handle, C.c_int64(0), st), R * H * 10),
 ("scaled_dswiglu bwd, no scale grad", lambda: api.nvte_scaled_dswiglu(wdy.handle, wx.handle, ws.handle, wdx.handle, None, C.c_int64(0), st), R * H * 10),
 ("scaled_clamped_dswiglu bwd + scale grad", lambda: api.nvte_scaled_clamped_dswiglu(wdy.handle, wx.handle, ws.handle, wdx.handle, wds.handle, C.c_float(7.0), C.c_float(1.702), C.c_float(1.0), C.c_int64(32), st), R * H * 10),
 ("scaled_srelu fwd", lambda: api.nvte_scaled_srelu(wxs.handle, ws.handle, wo.handle, st), R * H * 4),
 ("scaled_dsrelu bwd + scale grad", lambda: api.nvte_scaled_dsrelu(wdy.handle, wxs.handle, ws.handle, wdxs.handle, wds.handle, st), R * H * 6),
]
for name, f, nbytes in cases:
    f(); t0 = time.time()
    while time.time() - t0 < 0.2: f()
    t = sorted(timeit(f) for _ in range(3))[1]
    print(f"[{R}x{H}] {name:42s} {t*1e3:8.1f} us  {nbytes/t/1e6:7.0f} GB/s")
y = torch.empty_like(x)
t = sorted(timeit(lambda: y.copy_(x)) for _ in range(3))[1]
print(f"torch copy of the same input ({x.numel()*2/1e6:.0f} MB read + written): {t*1e3:.1f} us  {x.numel()*4/t/1e6:.0f} GB/s")
z = torch.empty_like(dy)
t = sorted(timeit(lambda: torch.mul(dy, 2.0, out=z)) for _ in range(3))[1]
print(f"torch mul on dy ({dy.numel()*2/1e6:.0f} MB read + written): {t*1e3:.1f} us  {dy.numel()*4/t/1e6:.0f} GB/s")
