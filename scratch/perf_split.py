import sys; sys.path.insert(0, '.')
import os, torch, time
from transformer_engine_amd.tensor import *
from transformer_engine_amd.gemm import general_gemm
def q8(x, dt=DType.kFloat8E4M3):
    q = Float8Quantizer(torch.full((1,), 64.0, device="cuda"), torch.zeros(1, device="cuda"), dt)
    return q(x)
for (n, m, k) in [(1024, 1024, 2048), (8192, 9216, 1152), (14336, 4096, 1024), (14336, 4096, 2048), (14336, 4096, 8192)]:
    b = q8(torch.randn(n, k, device="cuda", dtype=torch.bfloat16)); a = q8(torch.randn(m, k, device="cuda", dtype=torch.bfloat16))
    out = torch.empty(n, m, device="cuda", dtype=torch.bfloat16)
    for _ in range(3): general_gemm(a, b, out=out, layout="TN")
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); general_gemm(a, b, out=out, layout="TN"); e.record(); torch.cuda.synchronize()
        ts.append(s.elapsed_time(e) * 1e3)
    print((n, m, k), "us per call:", [round(t) for t in ts], f"{2*n*m*k/min(ts)/1e6:.0f} TF")
