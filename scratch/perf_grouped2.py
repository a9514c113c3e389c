import sys; sys.path.insert(0, '.')
import time, numpy as np, torch
from transformer_engine_amd.tensor import *
from transformer_engine_amd.grouped import *
def timeit(fn, it=20):
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / it
K, N = 4096, 14336
def run(m_splits, tag):
    G = len(m_splits); M = sum(m_splits)
    x = torch.randn(M, K, device="cuda", dtype=torch.bfloat16)
    w = torch.randn(G * N, K, device="cuda", dtype=torch.bfloat16) * 0.02
    fd = torch.tensor(m_splits, dtype=torch.int64, device="cuda")
    gx = group_quantize(x, Float8CurrentScalingQuantizer(DType.kFloat8E4M3, torch.device("cuda")), G, first_dims=fd)
    gw = group_quantize(w, Float8CurrentScalingQuantizer(DType.kFloat8E4M3, torch.device("cuda")), G)
    out = GroupedTensor(num_tensors=G, logical_shape=(M, N), dtype=torch.bfloat16,
                        rowwise_data=torch.empty(M * N, dtype=torch.bfloat16, device="cuda"), first_dims=fd,
                        tensor_offsets=splits_to_offsets(fd, N))
    f = lambda: general_grouped_gemm_for_grouped_tensor(gw, gx, out, layout="TN")
    t0 = time.time()
    while time.time() - t0 < 0.7: f()
    ts = sorted(timeit(f) for _ in range(5))
    print(f"{tag}: {ts[2]*1e3:.1f} us  {2*M*N*K/ts[2]/1e9:.0f} TF")
run([8192], "G=1 m=8192")
run([16384], "G=1 m=16384 (one weight, the plain-GEMM equivalent of the C5 size)")
run([2048] * 8, "G=8 m=2048 each")
run([2080, 1936, 2048, 2048, 2160, 2112, 2032, 2032], "G=8 ragged")
run([2048] * 4, "G=4 m=2048 each")
