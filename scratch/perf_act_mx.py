"""Fused activation -> MXFP8 against the unfused sequence (activation -> bf16, then nvte_quantize)."""
import torch
from transformer_engine_amd.activation import activation, dactivation, dbias_dact
from transformer_engine_amd.tensor import DType, MXFP8Quantizer

N, H = 8192, 14336


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


x1 = torch.randn(N, H, device="cuda", dtype=torch.bfloat16)
x2 = torch.randn(N, 2 * H, device="cuda", dtype=torch.bfloat16)
g = torch.randn(N, H, device="cuda", dtype=torch.bfloat16)
for layout in ("compact", "tcol"):
    q = MXFP8Quantizer(DType.kFloat8E4M3, columnwise_transposed=(layout == "tcol"))
    for tag, fused, unfused, by in (
        ("gelu", lambda: activation("gelu", x1, q), lambda: q(activation("gelu", x1)), N * H * (2 + 2 + 2 / 32)),
        ("swiglu", lambda: activation("swiglu", x2, q), lambda: q(activation("swiglu", x2)), N * H * (4 + 2 + 2 / 32)),
        ("dgelu+dbias", lambda: dbias_dact("gelu", g, x1, q), None, N * H * (4 + 2 + 2 / 32)),
        ("dswiglu", lambda: dactivation("swiglu", g, x2, q), lambda: q(dactivation("swiglu", g, x2)), N * H * (6 + 4 + 4 / 32)),
    ):
        tf = timeit(fused)
        tu = timeit(unfused) if unfused else float("nan")
        print(f"{layout:8s} {tag:12s}: fused {tf:7.1f} us = {by/tf/1e3:6.0f} GB/s   unfused {tu:7.1f} us")
