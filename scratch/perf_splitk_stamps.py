"""Timeline of the split-K tail of the configs[1] wgrad (stamps build)."""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("TE_AMD_LIBRARY", os.path.join(ROOT, "scratch", "lib_stamps", "libtransformer_engine.so"))
sys.path.insert(0, ROOT)
import numpy as np, torch, time
from transformer_engine_amd import _lib, general_gemm
from transformer_engine_amd.tensor import DType, Float8CurrentScalingQuantizer
tok, hid, ffn = 8192, 4096, 14336
g = torch.Generator(device="cuda").manual_seed(1234)
x = torch.randn(tok, hid, device="cuda", dtype=torch.bfloat16, generator=g)
dy = (torch.randn(tok, ffn, device="cuda", dtype=torch.float32, generator=g) * 0.01).to(torch.bfloat16)
E4, E5 = DType.kFloat8E4M3, DType.kFloat8E5M2
x8, dy8 = Float8CurrentScalingQuantizer(E4, "cuda")(x), Float8CurrentScalingQuantizer(E5, "cuda")(dy)
out = torch.empty((ffn, hid), dtype=torch.bfloat16, device="cuda")
fn = lambda: general_gemm(x8, dy8, layout="NT", out=out)
lib = _lib.raw_lib
lib.nvte_amd_debug_set_gemm_stamps.argtypes = [C.c_void_p]
SL = 512
stamps = torch.zeros(256 * SL, dtype=torch.int64, device="cuda")
t0 = time.time()
while time.time() - t0 < 1.0:
    for _ in range(20): fn()
    torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(20): fn()
e.record(); torch.cuda.synchronize()
print("avg ms (stamps off):", s.elapsed_time(e) / 20)
lib.nvte_amd_debug_set_gemm_stamps(C.c_void_p(stamps.data_ptr()))
for _ in range(3): fn()
torch.cuda.synchronize()
lib.nvte_amd_debug_set_gemm_stamps(None)
st = stamps.cpu().numpy().reshape(256, SL)
n = st[:, SL - 4]
print("stamps per WG: min %d max %d" % (n.min(), n.max()))
t = st.astype(np.float64) * 0.01   # us
t0all = min(t[b, 0] for b in range(256))
nn = int(n[0])
full = (nn - 1 - 34) // 66 if (nn - 1 - 34) % 66 == 0 else None
print("full tiles per WG:", full)
if full is not None:
    base = 1 + full * 66
    rows = []
    for b in range(256):
        tb = t[b] - t0all
        start_last = tb[base - 1]          # end of previous epilogue
        kt = np.diff(tb[base - 1: base + 32])     # 32 K tile intervals (first includes first_fragments etc.)
        epi_b, epi_e = tb[base + 32], tb[base + 33]
        rows.append((start_last, kt[0], np.median(kt[1:]), kt[1:].max(), tb[base + 31], epi_b, epi_e))
    r = np.array(rows)
    names = ["last-round start", "first K interval", "median K interval", "max K interval", "last barrier", "epi begin", "epi end"]
    for i, nm in enumerate(names):
        print(f"{nm:20s}: first halves (WG<128) p50 {np.median(r[:128, i]):8.2f} max {r[:128, i].max():8.2f} | second halves p50 {np.median(r[128:, i]):8.2f} max {r[128:, i].max():8.2f}")
    d = r[:, 6] - r[:, 5]
    print("epilogue duration: p10 %.2f p50 %.2f p90 %.2f max %.2f" % tuple(np.percentile(d, [10, 50, 90, 100])))
    # earlier full tiles for comparison
    kt_all = np.diff(t[:, 1:1 + 64], axis=1)
    print("full tile 0 median K interval %.3f ; epilogue %.2f" % (np.median(kt_all[:, 2:]), np.median(t[:, 66] - t[:, 65])))
    print("kernel span %.1f" % (max(t[b, int(n[b]) - 1] for b in range(256)) - t0all))
