"""C5-shaped GroupedLinear step with the MXFP8 recipe (members padded to 128 rows) next to the current-scaling
one, and its parts (group quantize of X / dY, the three grouped GEMMs)."""
import sys; sys.path.insert(0, '.')
import numpy as np, torch
import bench
from transformer_engine_amd.grouped_linear import GroupedLinearFP8State, grouped_linear_backward, grouped_linear_forward
from transformer_engine_amd.grouped import GroupedTensor, general_grouped_gemm_for_grouped_tensor, group_quantize, splits_to_offsets
from transformer_engine_amd.recipe import Float8CurrentScaling, MXFP8BlockScaling
from transformer_engine_amd.tensor import DType, MXFP8Quantizer
dev = torch.device("cuda:0")
G, K, N, routed = 8, 4096, 14336, 16384
counts = np.bincount(np.random.default_rng(1234).integers(0, G, size=routed), minlength=G)
m_splits = [int((c + 127) // 128 * 128) for c in counts]
M = sum(m_splits)
x = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
w = torch.randn(G * N, K, device=dev, dtype=torch.bfloat16) * 0.02
dy = torch.randn(M, N, device=dev, dtype=torch.bfloat16)
fd = torch.tensor(m_splits, dtype=torch.int64, device=dev)
fl = 2.0 * M * N * K
print("splits", m_splits, "M", M, flush=True)
for name, rec in (("current", Float8CurrentScaling()), ("mxfp8", MXFP8BlockScaling())):
    st = GroupedLinearFP8State(rec, G, dev)
    grads = [torch.zeros(N, K, device=dev, dtype=torch.bfloat16) for _ in range(G)]
    grouped_linear_forward(x, w, fd, st, is_first_microbatch=True)
    def step():
        y, saved = grouped_linear_forward(x, w, fd, st, is_first_microbatch=False)
        grouped_linear_backward(dy, saved, fd, st, wgrad_out=grads, accumulate_wgrad=False)
    ms = bench.time_kernel(step, 20)
    print(f"{name}: step {ms:.3f} ms = {3*fl/ms/1e9:.0f} TF", flush=True)

E4, E5 = DType.kFloat8E4M3, DType.kFloat8E5M2
mk = lambda dt: MXFP8Quantizer(dt, rowwise=True, columnwise=True, columnwise_transposed=True)
qx, qw, qdy = mk(E4), mk(E4), mk(E5)
for q in (qx, qw, qdy): q.optimize_for_gemm = True
t = bench.time_kernel(lambda: group_quantize(x, qx, G, first_dims=fd), 20)
print(f"group_quantize X  mx: {t*1e3:.1f} us = {x.numel()*4.0625/t/1e6:.0f} GB/s", flush=True)
t = bench.time_kernel(lambda: group_quantize(dy, qdy, G, first_dims=fd), 20)
print(f"group_quantize dY mx: {t*1e3:.1f} us = {dy.numel()*4.0625/t/1e6:.0f} GB/s", flush=True)
x8, w8, dy8 = group_quantize(x, qx, G, first_dims=fd), group_quantize(w, qw, G), group_quantize(dy, qdy, G, first_dims=fd)
def hp(rows, cols):
    return GroupedTensor(num_tensors=G, logical_shape=(rows, cols), dtype=torch.bfloat16,
                         rowwise_data=torch.empty(rows * cols, dtype=torch.bfloat16, device=dev),
                         first_dims=fd, tensor_offsets=splits_to_offsets(fd, cols))
y, dx = hp(M, N), hp(M, K)
dws = [torch.empty(N, K, dtype=torch.bfloat16, device=dev) for _ in range(G)]
for nm, fn in (("fprop", lambda: general_grouped_gemm_for_grouped_tensor(w8, x8, y, layout="TN")),
               ("dgrad", lambda: general_grouped_gemm_for_grouped_tensor(w8, dy8, dx, layout="NN")),
               ("wgrad", lambda: general_grouped_gemm_for_grouped_tensor(x8, dy8, dws, layout="NT"))):
    t = bench.time_kernel(fn, 20)
    print(f"grouped mx {nm}: {t*1e3:.1f} us = {fl/t/1e9:.0f} TF", flush=True)
