import sys; sys.path.insert(0, '.')
import time, torch
from transformer_engine_amd.tensor import *
from transformer_engine_amd.normalization import *
from transformer_engine_amd import api
from transformer_engine_amd.tensor import wrap_plain, TensorWrapper, current_stream_ptr
def timeit(fn, it=30):
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / it
for (N, H) in [(8192, 4096), (8192, 8192), (16384, 8192), (8192, 12288)]:
    x = torch.randn(N, H, device="cuda", dtype=torch.bfloat16)
    g = torch.ones(H, device="cuda", dtype=torch.bfloat16)
    b = torch.zeros(H, device="cuda", dtype=torch.bfloat16)
    # direct C-ABI calls with preallocated outputs (no host allocation inside the timed region)
    q = Float8Quantizer(torch.full((1,), 8.0, device="cuda"), torch.zeros(1, device="cuda"), DType.kFloat8E4M3, columnwise=False)
    out = q.make_empty((N, H), dtype=x.dtype, device="cuda")
    rs = torch.empty(N, device="cuda"); mu = torch.empty(N, device="cuda")
    wx, wg, wb, wrs, wmu = wrap_plain(x), wrap_plain(g), wrap_plain(b), wrap_plain(rs), wrap_plain(mu)
    wz = q._wrap_out(out); wz.set_scale(q.scale).set_amax(q.amax)
    ws = wrap_plain(torch.empty(16, dtype=torch.uint8, device="cuda"))
    st = current_stream_ptr()
    f_rms = lambda: api.nvte_rmsnorm_fwd(wx.handle, wg.handle, 1e-5, wz.handle, wrs.handle, ws.handle, 0, False, st)
    f_ln = lambda: api.nvte_layernorm_fwd(wx.handle, wg.handle, wb.handle, 1e-5, wz.handle, wmu.handle, wrs.handle, ws.handle, 0, False, st)
    hp = torch.empty_like(x); wh = wrap_plain(hp)
    f_rms_bf16 = lambda: api.nvte_rmsnorm_fwd(wx.handle, wg.handle, 1e-5, wh.handle, wrs.handle, ws.handle, 0, False, st)
    t0 = time.time()
    while time.time() - t0 < 0.3: f_rms()
    for name, f, bpe in (("rmsnorm->fp8", f_rms, 3), ("layernorm->fp8", f_ln, 3), ("rmsnorm->bf16", f_rms_bf16, 4)):
        t = sorted(timeit(f) for _ in range(3))[1]
        print(f"[{N}x{H}] {name}: {t*1e3:.1f} us  {N*H*bpe/t/1e6:.0f} GB/s ({bpe} B/elem)")
    dz = torch.randn_like(x); dx = torch.empty_like(x); dg = torch.empty_like(g)
    wdz, wdx, wdg = wrap_plain(dz), wrap_plain(dx), wrap_plain(dg)
    wsq = TensorWrapper()
    api.nvte_rmsnorm_bwd(wdz.handle, wx.handle, wrs.handle, wg.handle, wdx.handle, wdg.handle, wsq.handle, 0, False, st)
    bt = wsq.get_param(0); nb = bt.shape.data[0]
    wsb = wrap_plain(torch.empty(nb, dtype=torch.uint8, device="cuda"))
    f_bwd = lambda: api.nvte_rmsnorm_bwd(wdz.handle, wx.handle, wrs.handle, wg.handle, wdx.handle, wdg.handle, wsb.handle, 0, False, st)
    db = torch.empty_like(g); wdb = wrap_plain(db)
    f_lnb = lambda: api.nvte_layernorm_bwd(wdz.handle, wx.handle, wmu.handle, wrs.handle, wg.handle, wdx.handle, wdg.handle, wdb.handle, wsb.handle, 0, False, st)
    if H <= 12288:
        for name, f in (("rmsnorm bwd", f_bwd), ("layernorm bwd", f_lnb)):
            t = sorted(timeit(f) for _ in range(3))[1]
            print(f"[{N}x{H}] {name}: {t*1e3:.1f} us  {N*H*6/t/1e6:.0f} GB/s (6 B/elem)")
