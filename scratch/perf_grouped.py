import sys; sys.path.insert(0, '.')
import time, numpy as np, torch
from transformer_engine_amd.tensor import *
from transformer_engine_amd.grouped import *
from transformer_engine_amd.gemm import general_gemm
def timeit(fn, it=20):
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / it
G, K, N, tokens = 8, 4096, 14336, 16384
g = np.random.default_rng(1234)
counts = np.bincount(g.integers(0, G, size=tokens), minlength=G)
m_splits = [int((c + 15) // 16 * 16) for c in counts]
M = sum(m_splits)
print("m_splits", m_splits)
x = torch.randn(M, K, device="cuda", dtype=torch.bfloat16)
w = torch.randn(G * N, K, device="cuda", dtype=torch.bfloat16) * 0.02
fd = torch.tensor(m_splits, dtype=torch.int64, device="cuda")
qx = Float8CurrentScalingQuantizer(DType.kFloat8E4M3, torch.device("cuda"))
gx = group_quantize(x, qx, G, first_dims=fd)
gw = group_quantize(w, Float8CurrentScalingQuantizer(DType.kFloat8E4M3, torch.device("cuda")), G)
numel = M * N
out = GroupedTensor(num_tensors=G, logical_shape=(M, N), dtype=torch.bfloat16,
                    rowwise_data=torch.empty(numel, dtype=torch.bfloat16, device="cuda"), first_dims=fd,
                    tensor_offsets=splits_to_offsets(fd, N))
t0 = time.time()
while time.time() - t0 < 1.0:
    general_grouped_gemm_for_grouped_tensor(gw, gx, out, layout="TN")
torch.cuda.synchronize()
flops = 2 * M * N * K
for rnd in range(3):
    t = timeit(lambda: general_grouped_gemm_for_grouped_tensor(gw, gx, out, layout="TN"))
    print(f"grouped GEMM fprop: {t*1e3:.1f} us  {flops / t / 1e9:.0f} TF")
    tq = timeit(lambda: group_quantize(x, qx, G, first_dims=fd))
    print(f"group_quantize X (amax+scale+cast_transpose, current scaling): {tq*1e3:.1f} us  {M*K*4/tq/1e6:.0f} GB/s (4 B/elem algorithmic)")
# per-expert loop for comparison (host-known splits)
ws = []
r0 = 0
xs = []
q1 = Float8CurrentScalingQuantizer(DType.kFloat8E4M3, torch.device("cuda"))
for i, m in enumerate(m_splits):
    xs.append(q1(x[r0:r0 + m].contiguous())); ws.append(q1(w[i * N:(i + 1) * N].contiguous())); r0 += m
outs = [torch.empty(m, N, device="cuda", dtype=torch.bfloat16) for m in m_splits]
def loop():
    for i in range(G): general_gemm(ws[i], xs[i], out=outs[i], layout="TN")
for rnd in range(3):
    t = timeit(loop)
    print(f"per-expert loop of 8 GEMMs: {t*1e3:.1f} us  {flops / t / 1e9:.0f} TF")
