"""Tile-order experiments on the stamps build: TE_GEMM_DBG = GROUP_I (4, 8, 16, 32) per launch, interleaved rounds."""
import os, sys, ctypes as C, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("TE_AMD_LIBRARY", os.path.join(ROOT, "scratch", "lib_stamps", "libtransformer_engine.so"))
sys.path.insert(0, ROOT)
import numpy as np, torch
from transformer_engine_amd import _lib, general_gemm
from transformer_engine_amd.tensor import DType, Float8CurrentScalingQuantizer
tok, hid, ffn = 8192, 4096, 14336
g = torch.Generator(device="cuda").manual_seed(1234)
x = torch.randn(tok, hid, device="cuda", dtype=torch.bfloat16, generator=g)
w = (torch.randn(ffn, hid, device="cuda", dtype=torch.float32, generator=g) * 0.02).to(torch.bfloat16)
dy = (torch.randn(tok, ffn, device="cuda", dtype=torch.float32, generator=g) * 0.01).to(torch.bfloat16)
E4, E5 = DType.kFloat8E4M3, DType.kFloat8E5M2
x8, w8, dy8 = Float8CurrentScalingQuantizer(E4, "cuda")(x), Float8CurrentScalingQuantizer(E4, "cuda")(w), Float8CurrentScalingQuantizer(E5, "cuda")(dy)
roles = {"fprop": (w8, x8, "TN", (tok, ffn)), "dgrad": (w8, dy8, "NN", (tok, hid)), "wgrad": (x8, dy8, "NT", (ffn, hid))}
variants = [int(v) for v in (sys.argv[1].split(",") if len(sys.argv) > 1 else "8,4,16,32".split(","))]
t0 = time.time()
A, B, layout, oshape = roles["fprop"]
out = torch.empty(oshape, dtype=torch.bfloat16, device="cuda")
while time.time() - t0 < 1.0:
    for _ in range(20): general_gemm(A, B, layout=layout, out=out)
    torch.cuda.synchronize()
for role, (A, B, layout, oshape) in roles.items():
    out = torch.empty(oshape, dtype=torch.bfloat16, device="cuda")
    res = {v: [] for v in variants}
    for rnd in range(7):
        for v in variants:
            os.environ["TE_GEMM_DBG"] = str(v)
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            general_gemm(A, B, layout=layout, out=out)
            s.record()
            for _ in range(10): general_gemm(A, B, layout=layout, out=out)
            e.record(); torch.cuda.synchronize()
            res[v].append(s.elapsed_time(e) / 10 * 1000)
    print(role, {v: (round(float(np.median(r)), 1), round(float(np.min(r)), 1)) for v, r in res.items()}, flush=True)
