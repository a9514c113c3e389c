import sys; sys.path.insert(0, '.')
import torch
from transformer_engine_amd.gemm import general_gemm, multi_gemm, get_workspace
from transformer_engine_amd.tensor import DType, Float8CurrentScalingQuantizer
E4, E5 = DType.kFloat8E4M3, DType.kFloat8E5M2
q = lambda t, f: Float8CurrentScalingQuantizer(f, t.device)(t)
tok, kin, nout = 8192, 4096, 14336
x8 = q(torch.randn(tok, kin, device="cuda", dtype=torch.bfloat16), E4)
w8 = q((torch.randn(nout, kin, device="cuda") * 0.02).to(torch.bfloat16), E4)
dy8 = q((torch.randn(tok, nout, device="cuda") * 0.01).to(torch.bfloat16), E5)
dx = torch.empty(tok, kin, device="cuda", dtype=torch.bfloat16); dw = torch.empty(nout, kin, device="cuda", dtype=torch.bfloat16)
def timeit(fn, it=20):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / it * 1e3
sep = lambda: (general_gemm(w8, dy8, layout="NN", grad=True, out=dx), general_gemm(x8, dy8, layout="NT", grad=True, out=dw))
fused = lambda: multi_gemm([(w8, dy8, "NN", dx, False), (x8, dy8, "NT", dw, False)], grad=True)
one_d = lambda: multi_gemm([(w8, dy8, "NN", dx, False), (w8, dy8, "NN", dx, False)], grad=True)
one_w = lambda: multi_gemm([(x8, dy8, "NT", dw, False), (x8, dy8, "NT", dw, False)], grad=True)
print("separate dgrad+wgrad: %.1f us" % timeit(sep))
print("fused pair          : %.1f us" % timeit(fused))
print("2 x dgrad separate  : %.1f us" % timeit(lambda: (general_gemm(w8, dy8, layout="NN", grad=True, out=dx), general_gemm(w8, dy8, layout="NN", grad=True, out=dx))))
print("2 x dgrad fused     : %.1f us" % timeit(one_d))
print("2 x wgrad separate  : %.1f us" % timeit(lambda: (general_gemm(x8, dy8, layout="NT", grad=True, out=dw), general_gemm(x8, dy8, layout="NT", grad=True, out=dw))))
print("2 x wgrad fused     : %.1f us" % timeit(one_w))
fused(); torch.cuda.synchronize()
ws = get_workspace(torch.device("cuda:0"))
base = (ws.data_ptr() + 255) // 256 * 256 - ws.data_ptr()
pre = ws[base + 256: base + 256 + 48].view(torch.int32).cpu().tolist()
print("tile_start + xrun:", pre)
