"""Target program for the rocprofv3 passes of profiles/README.md: launches each hot kernel of the
configs[1] step a few times on resident data (no graph, no timing) so that --kernel-trace --stats
and the --pmc passes see plain dispatches.  Usage: python3 profiles/pmc_target.py [reps]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from transformer_engine_amd import general_gemm  # noqa: E402
from transformer_engine_amd.tensor import DType, Float8Quantizer  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
dev = torch.device("cuda", 0)
M, K, N = 8192, 4096, 14336
torch.manual_seed(0)
x = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
w = (torch.randn(N, K, device=dev, dtype=torch.float32) * 0.02).to(torch.bfloat16)
dy = (torch.randn(M, N, device=dev, dtype=torch.float32) * 1e-2).to(torch.bfloat16)


def quantizer(scale, dt, columnwise=True):
    return Float8Quantizer(torch.full((1,), scale, device=dev), torch.zeros(1, device=dev), dt, columnwise=columnwise)


qx, qw, qg = quantizer(64.0, DType.kFloat8E4M3), quantizer(2048.0, DType.kFloat8E4M3), quantizer(65536.0, DType.kFloat8E5M2)
x8, w8, dy8 = qx(x), qw(w), qg(dy)
ox = qx.make_empty(x.shape, dtype=x.dtype, device=dev)
ody = qg.make_empty(dy.shape, dtype=dy.dtype, device=dev)
y = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
dx = torch.empty(M, K, device=dev, dtype=torch.bfloat16)
dw = torch.empty(N, K, device=dev, dtype=torch.bfloat16)
# the step without FP8 transposes (LinearFP8State fp8_transposes="none", the library default): row-wise casts,
# dgrad / wgrad read the row-wise payloads K-major
rx, rw, rg = (quantizer(64.0, DType.kFloat8E4M3, False), quantizer(2048.0, DType.kFloat8E4M3, False),
              quantizer(65536.0, DType.kFloat8E5M2, False))
x8r, w8r, dy8r = rx(x), rw(w), rg(dy)
oxr = rx.make_empty(x.shape, dtype=x.dtype, device=dev)
odyr = rg.make_empty(dy.shape, dtype=dy.dtype, device=dev)
torch.cuda.synchronize()
for _ in range(reps):
    qx.update_quantized(x, ox)                                   # cast_transpose X   (e4m3)
    general_gemm(w8, x8, layout="TN", out=y)                     # fprop
    qg.update_quantized(dy, ody)                                 # cast_transpose dY  (e5m2)
    general_gemm(w8, dy8, layout="NN", out=dx, grad=True)        # dgrad
    general_gemm(x8, dy8, layout="NT", out=dw, grad=True)        # wgrad
torch.cuda.synchronize()
for _ in range(reps):
    rx.update_quantized(x, oxr)                                  # cast X (row-wise only)
    general_gemm(w8r, x8r, layout="TN", out=y)                   # fprop (the same kernel as above)
    rg.update_quantized(dy, odyr)                                # cast dY
    general_gemm(w8r, dy8r, layout="NN", out=dx, grad=True)      # dgrad, W read K-major     (kernel variant KM = 2)
    general_gemm(x8r, dy8r, layout="NT", out=dw, grad=True)      # wgrad, X and dY K-major   (KM = 3)
torch.cuda.synchronize()
print("pmc_target done", reps)
