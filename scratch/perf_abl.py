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
x = q8(torch.randn(tok, hid, device="cuda", dtype=torch.bfloat16))
w = q8(torch.randn(ffn, hid, device="cuda", dtype=torch.bfloat16))
dy = q8(torch.randn(tok, ffn, device="cuda", dtype=torch.bfloat16))
fl = 2 * tok * hid * ffn
t1 = timeit(lambda: general_gemm(w, x, out_dtype=torch.bfloat16, layout="TN"))
t2 = timeit(lambda: general_gemm(w, dy, out_dtype=torch.bfloat16, layout="NN"))
print("ABL", os.environ.get("NVTE_AMD_GEMM_ABLATE", "0"), f"fprop(K=4096) {t1*1e3:.0f} us {fl/t1/1e9:.0f} TF | dgrad(K=14336) {t2*1e3:.0f} us {fl/t2/1e9:.0f} TF")
