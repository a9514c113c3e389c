"""GEMM microbench for the three C2 shapes (+ an MXFP8 fprop).  NVTE_AMD_GEMM_V1=1 selects the old kernel."""
import sys; sys.path.insert(0, '.')
import os, torch
from transformer_engine_amd.tensor import *
from transformer_engine_amd.gemm import general_gemm


def timeit(fn, it=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / it


def q8(x, dt=DType.kFloat8E4M3):
    q = Float8Quantizer(torch.ones(1, device="cuda"), torch.zeros(1, device="cuda"), dt)
    return q(x)


tok, hid, ffn = 8192, 4096, 14336
x = torch.randn(tok, hid, device="cuda", dtype=torch.bfloat16)
w = torch.randn(ffn, hid, device="cuda", dtype=torch.bfloat16) * 0.02
dy = torch.randn(tok, ffn, device="cuda", dtype=torch.bfloat16)
xq, wq, dyq = q8(x), q8(w), q8(dy, DType.kFloat8E5M2)
tag = "v1" if os.environ.get("NVTE_AMD_GEMM_V1") == "1" else "v2"
res = []
for name, fn, fl in [
    ("fprop  TN", lambda: general_gemm(wq, xq, out_dtype=torch.bfloat16, layout="TN"), 2 * tok * hid * ffn),
    ("dgrad  NN", lambda: general_gemm(wq, dyq, out_dtype=torch.bfloat16, layout="NN"), 2 * tok * hid * ffn),
    ("wgrad  NT", lambda: general_gemm(xq, dyq, out_dtype=torch.bfloat16, layout="NT"), 2 * tok * hid * ffn),
    ("wgrad  NT fp32 acc", None, 2 * tok * hid * ffn),
]:
    if fn is None:
        out = torch.zeros(ffn, hid, device="cuda", dtype=torch.float32)
        fn = lambda: general_gemm(xq, dyq, out_dtype=torch.float32, layout="NT", out=out, accumulate=True)
    t = timeit(fn)
    res.append(f"{name}: {t*1e3:.0f} us {fl/t/1e9:.0f} TF")
qm = MXFP8Quantizer(DType.kFloat8E4M3)
xm, wm = qm(x), qm(w)
t = timeit(lambda: general_gemm(wm, xm, out_dtype=torch.bfloat16, layout="TN"))
res.append(f"mxfp8 fprop: {t*1e3:.0f} us {2*tok*hid*ffn/t/1e9:.0f} TF")
# square sanity
a = q8(torch.randn(8192, 8192, device="cuda", dtype=torch.bfloat16)); b = q8(torch.randn(8192, 8192, device="cuda", dtype=torch.bfloat16))
t = timeit(lambda: general_gemm(a, b, out_dtype=torch.bfloat16, layout="TN"))
res.append(f"8192^3: {t*1e3:.0f} us {2*8192**3/t/1e9:.0f} TF")
print(tag, " | ".join(res))
