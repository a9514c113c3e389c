import sys; sys.path.insert(0, '.')
import os, torch
from transformer_engine_amd.tensor import *
from transformer_engine_amd.gemm import general_gemm
def timeit(fn, it=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / it
def q8(x, dt=DType.kFloat8E4M3):
    q = Float8Quantizer(torch.full((1,), 64.0, device="cuda"), torch.zeros(1, device="cuda"), dt)
    return q(x)
res = []
n, m = 8192, 14336
for k in [2048, 2176, 3072, 4096, 4224, 5120, 6144, 8192, 10240, 12288, 14336, 16384, 16512]:
    b = q8(torch.randn(n, k, device="cuda", dtype=torch.bfloat16))
    a = q8(torch.randn(m, k, device="cuda", dtype=torch.bfloat16))
    t = timeit(lambda: general_gemm(a, b, out_dtype=torch.bfloat16, layout="TN"))
    res.append(f"K={k}: {2*n*m*k/t/1e9:.0f}")
print("DBG", os.environ.get("NVTE_AMD_GEMM_DBG", "0"), "TF |", " | ".join(res))
