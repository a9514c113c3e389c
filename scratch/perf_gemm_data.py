"""Same GEMM launches on operands of different bit statistics: is the kernel bound by switching power?"""
import sys; sys.path.insert(0, '.')
import torch
from transformer_engine_amd.gemm import general_gemm
from transformer_engine_amd.tensor import DType, Float8CurrentScalingQuantizer
E4, E5 = DType.kFloat8E4M3, DType.kFloat8E5M2
q = lambda t, f: Float8CurrentScalingQuantizer(f, t.device)(t)
tok, kin, nout = 8192, 4096, 14336
def timeit(fn, it=30):
    for _ in range(10): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / it * 1e3
def make(kind):
    g = torch.Generator(device="cuda").manual_seed(1)
    if kind == "randn":
        x = torch.randn(tok, kin, device="cuda", generator=g); w = torch.randn(nout, kin, device="cuda", generator=g) * 0.02
    elif kind == "zeros":
        x = torch.zeros(tok, kin, device="cuda"); w = torch.zeros(nout, kin, device="cuda")
    elif kind == "ones":
        x = torch.ones(tok, kin, device="cuda"); w = torch.ones(nout, kin, device="cuda")
    elif kind == "sparse90":
        x = torch.randn(tok, kin, device="cuda", generator=g); x = x * (torch.rand(tok, kin, device="cuda", generator=g) > 0.9)
        w = torch.randn(nout, kin, device="cuda", generator=g) * 0.02
    elif kind == "pm1":
        x = torch.sign(torch.randn(tok, kin, device="cuda", generator=g)); w = torch.sign(torch.randn(nout, kin, device="cuda", generator=g))
    return q(x.to(torch.bfloat16), E4), q(w.to(torch.bfloat16), E4)
y = torch.empty(tok, nout, device="cuda", dtype=torch.bfloat16)
for kind in ("randn", "zeros", "ones", "pm1", "sparse90", "randn"):
    x8, w8 = make(kind)
    t = timeit(lambda: general_gemm(w8, x8, layout="TN", out=y))
    print(f"fprop on {kind:9s}: {t:7.1f} us  {2*tok*kin*nout/t/1e6:6.0f} TFLOP/s")
