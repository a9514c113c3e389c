"""Where the waves of gemm_fp8_kernel are parked at the K-tile hand-over (diagnostic build with TE_GEMM_STAMPS):
per wave, shader cycles spent in `s_waitcnt vmcnt(0)` (its own staged loads not landed yet) and in `s_barrier`
(waiting for the other waves), against the cycles of the whole K loop; plus the epilogue cycles.
   make -C transformer_engine_amd/csrc stamps ; python scratch/perf_wait_stamps.py [fprop|dgrad|wgrad] [TE_GEMM_DBG]"""
import os, sys, ctypes as C, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("TE_AMD_LIBRARY", os.path.join(ROOT, "scratch", "lib_stamps", "libtransformer_engine.so"))
if len(sys.argv) > 2:
    os.environ["TE_GEMM_DBG"] = sys.argv[2]
sys.path.insert(0, ROOT)
import numpy as np, torch
from transformer_engine_amd import _lib, general_gemm
from transformer_engine_amd.tensor import DType, Float8CurrentScalingQuantizer
role = sys.argv[1] if len(sys.argv) > 1 else "fprop"
tok, hid, ffn = 8192, 4096, 14336
g = torch.Generator(device="cuda").manual_seed(1234)
x = torch.randn(tok, hid, device="cuda", dtype=torch.bfloat16, generator=g)
w = (torch.randn(ffn, hid, device="cuda", dtype=torch.float32, generator=g) * 0.02).to(torch.bfloat16)
dy = (torch.randn(tok, ffn, device="cuda", dtype=torch.float32, generator=g) * 0.01).to(torch.bfloat16)
E4, E5 = DType.kFloat8E4M3, DType.kFloat8E5M2
x8, w8, dy8 = Float8CurrentScalingQuantizer(E4, "cuda")(x), Float8CurrentScalingQuantizer(E4, "cuda")(w), Float8CurrentScalingQuantizer(E5, "cuda")(dy)
A, B, layout, oshape, K = {"fprop": (w8, x8, "TN", (tok, ffn), hid), "dgrad": (w8, dy8, "NN", (tok, hid), ffn),
                           "wgrad": (x8, dy8, "NT", (ffn, hid), tok)}[role]
out = torch.empty(oshape, dtype=torch.bfloat16, device="cuda")
fn = lambda: general_gemm(A, B, layout=layout, out=out)
lib = _lib.raw_lib
lib.nvte_amd_debug_set_gemm_wait_stamps.argtypes = [C.c_void_p]
ws = torch.zeros(256 * 8 * 8, dtype=torch.int64, device="cuda")
t0 = time.time()
while time.time() - t0 < 1.0:
    for _ in range(20): fn()
    torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(20): fn()
e.record(); torch.cuda.synchronize()
plain = s.elapsed_time(e) / 20
lib.nvte_amd_debug_set_gemm_wait_stamps(C.c_void_p(ws.data_ptr()))
for _ in range(3): fn()
s.record(); fn(); e.record(); torch.cuda.synchronize()
stamped = s.elapsed_time(e)
lib.nvte_amd_debug_set_gemm_wait_stamps(None)
d = ws.cpu().numpy().reshape(256, 8, 8).astype(np.float64)
vm, bar, loop, tiles, epi, yseg, bar2 = d[..., 0], d[..., 1], d[..., 2], d[..., 3], d[..., 4], d[..., 5], d[..., 6]
np.set_printoptions(linewidth=200, precision=1, suppress=True)
print(f"{role} dbg={os.environ.get('TE_GEMM_DBG', '0')}: launch {plain * 1e3:.1f} us without stamps, {stamped * 1e3:.1f} us with; K tiles per wave {tiles.mean():.0f}")
print("cycles per K tile, mean over workgroups, by wave 0..7:")
print("  K-loop span / K tile      ", (loop / tiles).mean(axis=0))
print("  parked in vmcnt(0) wait   ", (vm / tiles).mean(axis=0))
print("  parked in s_barrier       ", (bar / tiles).mean(axis=0))
print("  epilogue cycles per K tile", (epi / tiles).mean(axis=0))
print("  post-barrier segment (6 MFMAs + 8 LDS-DMA + 12 ds_read)", (yseg / tiles).mean(axis=0))
print("  parked in the Y-segment barrier", (bar2 / tiles).mean(axis=0))
print("  X segment (10 MFMAs + 12 ds_read)                      ", ((loop - vm - bar - epi - yseg - bar2) / tiles).mean(axis=0))
print(f"shares of the K-loop span: vmcnt wait {vm.sum() / loop.sum():.3f}, barrier {bar.sum() / loop.sum():.3f}; epilogue / (loop span) {epi.sum() / loop.sum():.3f}")
print("ideal MFMA cycles per K tile: 2048")
