"""cast_transpose on shapes that are not whole 128x128 tiles (guarded vector path) next to the whole-tile shape."""
import sys; sys.path.insert(0, '.')
import torch, bench
from transformer_engine_amd.tensor import DType, Float8Quantizer
dev = torch.device("cuda:0")
for shape in 2 * ((2000, 4104), (8192, 4096), (8176, 4104), (8192, 14336), (8176, 14344), (2000, 4104), (2048, 4096)):
    x = torch.randn(*shape, device=dev, dtype=torch.bfloat16)
    q = Float8Quantizer(torch.ones(1, device=dev), torch.zeros(1, device=dev), DType.kFloat8E4M3)
    out = q.make_empty(x.shape, dtype=x.dtype, device=dev)
    t = bench.time_kernel(lambda: q.update_quantized(x, out), 30)
    print(f"{shape}: {t*1e3:.1f} us {x.numel()*4/t/1e6:.0f} GB/s", flush=True)
