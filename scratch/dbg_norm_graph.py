"""First-ever norm launch of the process INSIDE a stream capture (the occupancy query runs on first use)."""
import sys; sys.path.insert(0, '.')
import torch
from transformer_engine_amd.normalization import rmsnorm_fwd, rmsnorm_bwd, layernorm_fwd
from transformer_engine_amd.tensor import Float8Quantizer, DType
x = torch.randn(512, 4096, device="cuda", dtype=torch.bfloat16)
g = torch.ones(4096, device="cuda", dtype=torch.bfloat16)
dz = torch.randn_like(x)
q = Float8Quantizer(torch.full((1,), 8.0, device="cuda"), torch.zeros(1, device="cuda"), DType.kFloat8E4M3)
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        out, _, rs = rmsnorm_fwd(x, g, 1e-5, quantizer=q)
        dx, dg = rmsnorm_bwd(dz, x, rs, g)
    gr.replay(); gr.replay()
torch.cuda.synchronize()
ref, _, rs2 = rmsnorm_fwd(x, g, 1e-5, quantizer=q)
dx2, dg2 = rmsnorm_bwd(dz, x, rs2, g)
torch.cuda.synchronize()
print("fwd equal:", torch.equal(out._data, ref._data), "bwd equal:", torch.equal(dx, dx2), torch.equal(dg, dg2))
