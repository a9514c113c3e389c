import sys; sys.path.insert(0, '.')
import os, torch, time
from transformer_engine_amd.tensor import *
from transformer_engine_amd.gemm import general_gemm
def timeit(fn, it=20):
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / it
cfgs = [(8192, 8192, 8192, False), (8192, 8192, 8192, True), (8192, 14336, 4096, False), (8192, 14336, 4096, True)]
ops = []
for (n, m, k, swz) in cfgs:
    q = MXFP8Quantizer(DType.kFloat8E4M3, columnwise=False)
    q.optimize_for_gemm = swz
    b = q(torch.randn(n, k, device="cuda", dtype=torch.bfloat16)); a = q(torch.randn(m, k, device="cuda", dtype=torch.bfloat16))
    out = torch.empty(n, m, device="cuda", dtype=torch.bfloat16)
    ops.append((a, b, out))
t0 = time.time()
while time.time() - t0 < 1.0:
    for a, b, out in ops: general_gemm(a, b, out=out, layout="TN")
torch.cuda.synchronize()
rows = {c: [] for c in cfgs}
for rnd in range(5):
    for c, (a, b, out) in zip(cfgs, ops):
        t = timeit(lambda: general_gemm(a, b, out=out, layout="TN"))
        rows[c].append(2 * c[0] * c[1] * c[2] / t / 1e9)
for c in cfgs:
    v = sorted(rows[c]); print(f"  MX {c}: median {v[len(v)//2]:.0f} TF  min {v[0]:.0f} max {v[-1]:.0f}")
