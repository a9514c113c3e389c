"""Grouped activation + MXFP8 cast (nvte_group_gelu / nvte_group_dgelu) against the single-tensor fused kernel on
the same [16384, 14336] operand (8 uniform members through first_dims)."""
import sys; sys.path.insert(0, '.')
import torch, bench
from transformer_engine_amd.activation import activation, dactivation
from transformer_engine_amd.grouped import group_activation
from transformer_engine_amd.tensor import DType, MXFP8Quantizer
dev = torch.device("cuda:0")
M, N, G = 16384, 14336, 8
x = torch.randn(M, N, device=dev, dtype=torch.bfloat16)
g = torch.randn(M, N, device=dev, dtype=torch.bfloat16)
fd = torch.tensor([M // G] * G, dtype=torch.int64, device=dev)
for tcol in (False, True):
    q = MXFP8Quantizer(DType.kFloat8E4M3, columnwise_transposed=tcol)
    for rep in range(2):
        t0 = bench.time_kernel(lambda: activation("gelu", x, q), 10)
        t1 = bench.time_kernel(lambda: group_activation("gelu", x, q, G, fd), 10)
        t2 = bench.time_kernel(lambda: dactivation("gelu", g, x, q), 10)
        t3 = bench.time_kernel(lambda: group_activation("gelu", x, q, G, fd, grad=g), 10)
    print(f"tcol={tcol}: gelu dense {t0*1e3:.1f} us, grouped {t1*1e3:.1f}; dgelu dense {t2*1e3:.1f}, grouped {t3*1e3:.1f}", flush=True)
