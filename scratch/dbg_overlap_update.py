"""Serial vs side-stream order of the recipe updates around the two step graphs: same recipe state, bit for bit."""
import sys; sys.path.insert(0, '.')
import torch
from transformer_engine_amd import linear as L
from transformer_engine_amd.recipe import DelayedScaling
dev = torch.device("cuda:0")
M, K, N = 2048, 1024, 3072
def run(overlap, steps=12):
    torch.manual_seed(5)
    x = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
    w = (torch.randn(N, K, device=dev) * 0.02).to(torch.bfloat16)
    dys = [(torch.randn(M, N, device=dev) * (0.01 * (i + 1))).to(torch.bfloat16) for i in range(steps)]
    xs = [x * (1.0 + 0.3 * i) for i in range(steps)]
    st = L.LinearFP8State(DelayedScaling(), device=dev)
    xin, dyin = xs[0].clone(), dys[0].clone()
    h = {}
    def fwd(): h["y"], h["saved"] = L.linear_forward(xin, w, None, st, is_first_microbatch=False)
    def bwd(): h["dx"], h["dw"], _ = L.linear_backward(dyin, h["saved"], st)
    L.linear_forward(xin, w, None, st, is_first_microbatch=True); st.update_fwd()
    side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        fwd(); st.update_fwd(); bwd(); st.update_bwd()
    torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
    gf, gb = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
    with torch.cuda.graph(gf): fwd()
    with torch.cuda.graph(gb, pool=gf.pool()): bwd()
    comm = torch.cuda.Stream(); ev = {k: torch.cuda.Event() for k in ("f", "fd", "b", "bd")}
    main = torch.cuda.current_stream()
    outs = []
    for i in range(steps):
        xin.copy_(xs[i]); dyin.copy_(dys[i])
        if not overlap:
            gf.replay(); st.update_fwd(); gb.replay(); st.update_bwd()
        else:
            main.wait_event(ev["fd"]); gf.replay(); ev["f"].record(main)
            with torch.cuda.stream(comm):
                comm.wait_event(ev["f"]); st.update_fwd(); ev["fd"].record(comm)
            main.wait_event(ev["bd"]); gb.replay(); ev["b"].record(main)
            with torch.cuda.stream(comm):
                comm.wait_event(ev["b"]); st.update_bwd(); ev["bd"].record(comm)
        outs.append((h["y"].clone(), h["dx"].clone(), h["dw"].clone()))
    torch.cuda.synchronize()
    meta = [q.scale.clone() for q in st.fwd_q] + [q.scale.clone() for q in st.bwd_q]
    return outs, meta
a, ma = run(False); b, mb = run(True)
print("outputs equal:", all(torch.equal(p, q) for s, t in zip(a, b) for p, q in zip(s, t)),
      "scales equal:", all(torch.equal(p, q) for p, q in zip(ma, mb)), [float(m) for m in ma])
