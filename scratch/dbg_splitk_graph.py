import torch
from transformer_engine_amd import gemm as G, general_gemm
from transformer_engine_amd.tensor import DType, Float8Tensor
E4 = DType.kFloat8E4M3
def mk(shape, seed):
    g = torch.Generator().manual_seed(seed)
    v = torch.randint(-3, 4, shape, generator=g).float()
    t = Float8Tensor(shape=shape, dtype=torch.float32, data=v.to(torch.float8_e4m3fn).view(torch.uint8).cuda(),
                     fp8_scale_inv=torch.ones(1, device="cuda"), fp8_dtype=E4)
    return t, v.cuda()
M, N, K = 4352, 4352, 1024
W, w = mk((N, K), 21); X, x = mk((M, K), 22)
ref = (x.double() @ w.double().T).float()
out = torch.zeros(M, N, device="cuda")
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    for i in range(6):
        out.zero_(); general_gemm(W, X, out=out, layout="TN"); torch.cuda.synchronize()
        print("eager", i, "bad", int((out != ref).sum()), end="; ")
    print()
    for i in range(6):   # eager back to back, no sync between
        general_gemm(W, X, out=out, layout="TN")
    torch.cuda.synchronize(); print("eager back-to-back bad", int((out != ref).sum()))
    ws = G.get_workspace(torch.device("cuda", 0))
    rem = 33
    flags = ws[rem * 256 * 256 * 4: rem * 256 * 256 * 4 + rem * 8].view(torch.int32)
    print("flags after eager", flags[:8].tolist())
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=s):
        general_gemm(W, X, out=out, layout="TN")
torch.cuda.synchronize()
for i in range(5):
    out.zero_()
    torch.cuda.synchronize()
    graph.replay()
    torch.cuda.synchronize()
    bad = (out != ref)
    print("replay", i, "bad", int(bad.sum()), "flags", flags[:8].tolist())
