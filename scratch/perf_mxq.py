"""MXFP8 quantize rates (row+col reference layout, row only, row + transposed columnwise, swizzled) on the
configs[1]/[2] operand shapes; NVTE_AMD_MX_WGS_PER_CU picks the persistent grid."""
import sys, os; sys.path.insert(0, '.')
import torch, bench
from transformer_engine_amd.tensor import DType, MXFP8Quantizer
dev = torch.device("cuda:0")
print("per_cu", os.environ.get("NVTE_AMD_MX_WGS_PER_CU", "default"), flush=True)
for shape in ((8192, 14336), (8192, 4096), (8192, 8192), (16768, 14336)):
    x = torch.randn(*shape, device=dev, dtype=torch.bfloat16)
    n = x.numel()
    for name, kw, swz, bpe in (("row+col", {}, False, 4.0625), ("row", {"columnwise": False}, False, 3.03125),
                               ("row+tcol", {"columnwise_transposed": True}, False, 4.0625),
                               ("row+tcol swz", {"columnwise_transposed": True}, True, 4.0625),
                               ("row swz", {"columnwise": False}, True, 3.03125)):
        q = MXFP8Quantizer(DType.kFloat8E4M3, **kw); q.optimize_for_gemm = swz
        out = q.make_empty(x.shape, dtype=x.dtype, device=dev)
        t = bench.time_kernel(lambda: q.update_quantized(x, out), 30)
        print(f"{shape} {name:13s} {t*1e3:7.1f} us {n*bpe/t/1e6:6.0f} GB/s", flush=True)
