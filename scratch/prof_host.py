"""Host-side cost of one eager Linear fwd+bwd step at configs[0] size (768 -> 3072, 2048 tokens)."""
import sys; sys.path.insert(0, '.')
import cProfile, pstats, time, torch
from transformer_engine_amd import linear as L
from transformer_engine_amd.recipe import DelayedScaling
dev = torch.device("cuda:0")
M, K, N = 2048, 768, 3072
x = torch.randn(M, K, device=dev, dtype=torch.bfloat16); w = (torch.randn(N, K, device=dev) * 0.02).to(torch.bfloat16)
dy = (torch.randn(M, N, device=dev) * 0.01).to(torch.bfloat16)
st = L.LinearFP8State(DelayedScaling(), device=dev)
def step(first=False):
    y, saved = L.linear_forward(x, w, None, st, is_first_microbatch=first)
    st.update_fwd()
    dx, dw, _ = L.linear_backward(dy, saved, st)
    st.update_bwd()
step(True)
for _ in range(20): step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(200): step()
t_host = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print(f"host enqueue {t_host/200*1e6:.1f} us/step, wall {t_all/200*1e6:.1f} us/step")
pr = cProfile.Profile(); pr.enable()
for _ in range(200): step()
pr.disable(); torch.cuda.synchronize()
ps = pstats.Stats(pr); ps.sort_stats("cumulative"); ps.print_stats(28)
