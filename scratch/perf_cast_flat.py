"""Row-wise cast (cast_flat_kernel) on the configs[1] tensors: isolated, back to back."""
import sys; sys.path.insert(0, '.')
import torch, bench
from transformer_engine_amd.tensor import DType, Float8Quantizer
dev = torch.device("cuda")
for name, shape, dt in (("X", (8192, 4096), DType.kFloat8E4M3), ("dY", (8192, 14336), DType.kFloat8E5M2), ("W", (14336, 4096), DType.kFloat8E4M3)):
    t = torch.randn(*shape, device=dev, dtype=torch.bfloat16)
    for col in (False, True):
        q = Float8Quantizer(torch.ones(1, device=dev), torch.zeros(1, device=dev), dt, columnwise=col)
        o = q.make_empty(t.shape, dtype=t.dtype, device=dev)
        ms = bench.time_kernel(lambda: q.update_quantized(t, o), 50)
        b = t.numel() * (4 if col else 3)
        print(f"{name} {'cast_transpose' if col else 'cast'}: {ms*1e3:.1f} us = {b/ms/1e9:.2f} TB/s", flush=True)
