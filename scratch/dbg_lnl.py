import sys; sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import torch
from transformer_engine_amd.linear import LinearFP8State, layernorm_linear_backward, layernorm_linear_forward
from transformer_engine_amd.recipe import DelayedScaling
M, K, N = 512, 1024, 768
torch.manual_seed(7)
x = torch.randn(M, K, device="cuda", dtype=torch.bfloat16)
g = (torch.randn(K, device="cuda") * 0.2 + 1.0).to(torch.bfloat16)
b = (torch.randn(K, device="cuda") * 0.1).to(torch.bfloat16)
w = (torch.randn(N, K, device="cuda") * 0.05).to(torch.bfloat16)
dy = torch.randn(M, N, device="cuda", dtype=torch.bfloat16)
for norm in ("LayerNorm", "RMSNorm"):
    st = LinearFP8State(DelayedScaling())
    for it in range(3):
        y, saved = layernorm_linear_forward(x, g, b if norm == "LayerNorm" else None, w, None, st, normalization=norm)
        st.update_fwd()
        dx, dgamma, dbeta, dw, _ = layernorm_linear_backward(dy, saved, g, st, normalization=norm)
        st.update_bwd()
        torch.cuda.synchronize()
        print(norm, it, "fwd scale", st.fwd_state.scale.tolist(), "amax0", st.fwd_state.amax_history[0].tolist(), "bwd scale", st.bwd_state.scale.tolist())
    xf = x.float(); gf = g.float(); wf = w.float()
    ln = torch.nn.functional.layer_norm(xf, (K,), gf, b.float(), 1e-5) if norm == "LayerNorm" else xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + 1e-5) * gf
    dwr = dy.float().t() @ ln
    err = (dw.float() - dwr).abs()
    print(norm, "dw err max", err.max().item(), "mean", err.mean().item(), "ref absmax", dwr.abs().max().item())
    i = err.argmax().item(); print("  worst at", divmod(i, K), dw.view(-1)[i].item(), dwr.view(-1)[i].item())
    ln8 = saved[0]
    print("  ln8 transpose ok:", torch.equal(ln8._transpose, ln8._data.t().contiguous()), " deq err", (ln8.dequantize(torch.float32) - ln).abs().max().item())
