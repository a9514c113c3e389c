"""Same FLOPs (rounds x K-tiles = 224 per workgroup), different number of output tiles per workgroup:
the slope over `rounds` is the per-output-tile overhead of the persistent GEMM."""
import torch
from transformer_engine_amd.gemm import general_gemm
from transformer_engine_amd.tensor import DType, Float8Quantizer

dev = "cuda"


def q8(rows, cols, fmt=DType.kFloat8E4M3):
    x = torch.randn(rows, cols, device=dev, dtype=torch.bfloat16)
    q = Float8Quantizer(torch.ones(1, device=dev), torch.zeros(1, device=dev), fmt, columnwise=False)
    return q(x)


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


for n, m, k in [(4096, 4096, 28672), (8192, 4096, 14336), (8192, 8192, 7168), (8192, 14336, 4096), (16384, 14336, 2048),
                (16384, 28672, 1024)]:
    A = q8(m, k)   # weight-like operand [m, k]
    B = q8(n, k)   # activation-like operand [n, k]
    out = torch.empty(n, m, device=dev, dtype=torch.bfloat16)
    tiles = (n // 256) * (m // 256)
    timeit(lambda: general_gemm(A, B, out=out, layout="TN"), iters=5)   # the first shape measured is cold otherwise
    t = timeit(lambda: general_gemm(A, B, out=out, layout="TN"))
    print(f"n={n:6d} m={m:6d} k={k:6d}: tiles/WG {tiles/256:5.1f}  K-tiles {k//128:4d}  {t:7.1f} us  {2*n*m*k/t/1e6:7.0f} TFLOP/s")
