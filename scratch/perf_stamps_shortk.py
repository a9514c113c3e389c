"""Stamps build: timeline of one workgroup across output tiles for a short-K GEMM (an expert's wgrad: 14336 x 4096, K = 2048)."""
import os, sys, ctypes as C, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("TE_AMD_LIBRARY", os.path.join(ROOT, "scratch", "lib_stamps", "libtransformer_engine.so"))
sys.path.insert(0, ROOT)
import numpy as np, torch
from transformer_engine_amd import _lib, general_gemm
from transformer_engine_amd.tensor import DType, Float8CurrentScalingQuantizer
n, m, k = 14336, 4096, int(sys.argv[1]) if len(sys.argv) > 1 else 2048
g = torch.Generator(device="cuda").manual_seed(1234)
a = torch.randn(m, k, device="cuda", dtype=torch.bfloat16, generator=g)
b = torch.randn(n, k, device="cuda", dtype=torch.bfloat16, generator=g)
E4 = DType.kFloat8E4M3
a8, b8 = Float8CurrentScalingQuantizer(E4, "cuda")(a), Float8CurrentScalingQuantizer(E4, "cuda")(b)
out = torch.empty((n, m), dtype=torch.bfloat16, device="cuda")
fn = lambda: general_gemm(a8, b8, layout="TN", out=out)
lib = _lib.raw_lib
lib.nvte_amd_debug_set_gemm_stamps.argtypes = [C.c_void_p]
SL = 512
stamps = torch.zeros(256 * SL, dtype=torch.int64, device="cuda")
for _ in range(50): fn()
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(20): fn()
e.record(); torch.cuda.synchronize()
print(f"kernel {s.elapsed_time(e)/20*1000:.1f} us, {2*n*m*k/(s.elapsed_time(e)/20)/1e9:.0f} TF, K tiles per tile {k//128}")
lib.nvte_amd_debug_set_gemm_stamps(C.c_void_p(stamps.data_ptr()))
fn(); fn(); torch.cuda.synchronize()
lib.nvte_amd_debug_set_gemm_stamps(None)
st = stamps.cpu().numpy().reshape(256, SL)
cnt = int(st[0, SL - 4]); t = st[:, :cnt].astype(np.float64) * 10.0 / 1000.0   # us
nk = k // 128
per = nk + 2      # stamps per tile: one per K tile + before / after the epilogue
for wg in (0, 37, 200):
    iv = np.diff(t[wg])
    print(f"WG {wg}: {cnt} stamps; first 3 tiles' intervals (us):")
    for tl in range(3):
        seg = iv[1 + tl * per: 1 + (tl + 1) * per]
        print("   ", " ".join(f"{v:.2f}" for v in seg))
iv = np.diff(t, axis=1)
tiles = (cnt - 1) // per
ktile = np.concatenate([iv[:, 1 + c * per + 3: 1 + c * per + nk - 1].ravel() for c in range(1, tiles - 1)])
print(f"steady K-tile interval median {np.median(ktile):.3f} us; tiles per WG {tiles}")
for name, off in (("last K tile -> epilogue start", nk - 1), ("epilogue", nk), ("epilogue end -> first barrier of next tile", nk + 1), ("2nd K tile of next", nk + 2), ("3rd", nk + 3)):
    v = np.array([iv[:, 1 + c * per + off] for c in range(1, tiles - 2)]).ravel()
    print(f"  {name:45s} median {np.median(v):.2f} us  mean {v.mean():.2f}")
