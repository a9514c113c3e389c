"""configs[2]-shaped MXFP8 Linear step (8192 tokens, 8192 -> 8192), fwd+bwd, weights cached; with the module
default (GEMM-swizzled scales) and with compact scales."""
import sys; sys.path.insert(0, '.')
import torch, bench
from transformer_engine_amd import linear as L
from transformer_engine_amd.recipe import MXFP8BlockScaling, DelayedScaling
dev = torch.device("cuda:0")
M = K = N = 8192
x = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
w = (torch.randn(N, K, device=dev) * 0.02).to(torch.bfloat16)
dy = (torch.randn(M, N, device=dev) * 0.01).to(torch.bfloat16)
for name, st in 2 * (("mxfp8 swizzled, transposed columnwise (amd_ext)", L.LinearFP8State(MXFP8BlockScaling(), device=dev, mx_columnwise="transposed")),
                 ("mxfp8 swizzled, reference columnwise (K-major GEMM reads)", L.LinearFP8State(MXFP8BlockScaling(), device=dev, mx_columnwise="reference")),
                 ("mxfp8 swizzled, mixed (input + weight reference, grad_output transposed)", L.LinearFP8State(MXFP8BlockScaling(), device=dev, mx_columnwise="mixed")),
                 ("mxfp8 compact, reference columnwise", L.LinearFP8State(MXFP8BlockScaling(), device=dev, mx_columnwise="reference", compact_mx_scales=("input", "weight", "grad_output"))),
                 ("mxfp8 compact, transposed columnwise", L.LinearFP8State(MXFP8BlockScaling(), device=dev, mx_columnwise="transposed", compact_mx_scales=("input", "weight", "grad_output"))),
                 ("delayed", L.LinearFP8State(DelayedScaling(), device=dev))):
    L.linear_forward(x, w, None, st, is_first_microbatch=True)
    def step():
        y, saved = L.linear_forward(x, w, None, st, is_first_microbatch=False)
        st.update_fwd()
        L.linear_backward(dy, saved, st)
        st.update_bwd()
    ms = bench.time_kernel(step, 20)
    print(f"{name}: {ms:.3f} ms/step = {6.0*M*N*K/ms/1e9:.0f} TF", flush=True)
