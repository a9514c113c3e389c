"""MXFP8 quantize of the configs[2] tensors, isolated: rowwise only, rowwise + columnwise (reference layout), rowwise +
transposed columnwise; compact and GEMM-swizzled scales."""
import sys; sys.path.insert(0, '.')
import torch, bench
from transformer_engine_amd.tensor import DType, MXFP8Quantizer
dev = torch.device("cuda")
for shape in ((8192, 8192), (8192, 14336)):
    t = torch.randn(*shape, device=dev, dtype=torch.bfloat16)
    for name, kw, opt in (("row", dict(columnwise=False), True), ("row+col reference", dict(), True), ("row+col reference compact", dict(), False),
                          ("row+tcol", dict(columnwise_transposed=True), True), ("col only reference", dict(rowwise=False), True)):
        q = MXFP8Quantizer(DType.kFloat8E4M3, **kw); q.optimize_for_gemm = opt
        o = q.make_empty(t.shape, dtype=t.dtype, device=dev)
        ms = bench.time_kernel(lambda: q.update_quantized(t, o), 30)
        n = t.numel(); outs = (1 if kw.get("rowwise", True) else 0) + (1 if kw.get("columnwise", True) else 0)
        b = n * (2 + outs * (1 + 1 / 32))
        print(f"{shape} {name}: {ms*1e3:.1f} us = {b/ms/1e9:.2f} TB/s", flush=True)
