"""In-kernel timeline of gemm_fp8_kernel (diagnostic build with TE_GEMM_STAMPS):
   make -C transformer_engine_amd/csrc stamps ; TE_AMD_LIBRARY=scratch/lib_stamps/libtransformer_engine.so python scratch/perf_stamps.py [fprop|dgrad|wgrad]"""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("TE_AMD_LIBRARY", os.path.join(ROOT, "scratch", "lib_stamps", "libtransformer_engine.so"))
sys.path.insert(0, ROOT)
import numpy as np, torch
from transformer_engine_amd import _lib, general_gemm
from transformer_engine_amd.tensor import DType, Float8CurrentScalingQuantizer
role = sys.argv[1] if len(sys.argv) > 1 else "fprop"
tok, hid, ffn = 8192, int(os.environ.get("HID", "4096")), int(os.environ.get("FFN", "14336"))
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
lib.nvte_amd_debug_set_gemm_stamps.argtypes = [C.c_void_p]
SL = 512
stamps = torch.zeros(256 * SL, dtype=torch.int64, device="cuda")
# warm up under load (~1 s)
import time
t0 = time.time()
while time.time() - t0 < 1.0:
    for _ in range(20): fn()
    torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(20): fn()
e.record(); torch.cuda.synchronize()
print(role, "unstamped-path avg ms (stamp build, stamps off):", s.elapsed_time(e) / 20)
lib.nvte_amd_debug_set_gemm_stamps(C.c_void_p(stamps.data_ptr()))
for _ in range(3): fn()
s.record(); fn(); e.record()
torch.cuda.synchronize()
print("stamped launch ms:", s.elapsed_time(e))
lib.nvte_amd_debug_set_gemm_stamps(None)
st = stamps.cpu().numpy().reshape(256, SL)
n = int(st[0, SL - 4])
nk = (K + 127) // 128
per_tile = nk + 2
ntile = (n - 1) // per_tile
t = st[:, :n].astype(np.float64) * 10.0   # ns (100 MHz)
t0all = t[:, 0].min()
dur = (t[:, n - 1].max() - t0all) / 1000
clk = (st[:, SL - 2] - st[:, SL - 1]).astype(np.float64) / ((t[:, n - 1] - t[:, 0]) * 1e-9) / 1e9
print(f"stamps per WG {n}, tiles per WG {ntile}, K tiles {nk}; kernel span {dur:.1f} us; in-kernel clock GHz median {np.median(clk):.3f} min {clk.min():.3f} max {clk.max():.3f}")
print("start skew us: p50 %.2f max %.2f ; end skew: min %.1f max %.1f" % (np.median(t[:, 0] - t0all) / 1000, (t[:, 0] - t0all).max() / 1000,
      (t[:, n - 1].min() - t0all) / 1000, (t[:, n - 1].max() - t0all) / 1000))
# cycle profile: intervals inside one tile cycle, averaged over WGs and over the middle tiles
iv = np.diff(t, axis=1)            # [256, n-1]; iv[:, j] = t[j+1]-t[j]
# stamp layout: 0 = start; tile c: stamps 1+c*per_tile .. : nk barrier stamps, epi begin, epi end
rows = []
for c in range(ntile):
    base = 1 + c * per_tile
    seg = t[:, base - 1: base + per_tile]            # previous stamp + this tile's stamps
    rows.append(np.diff(seg, axis=1))                # [256, per_tile]: interval ENDING at each stamp of the tile
prof = np.stack(rows)                                # [ntile, 256, per_tile]
np.set_printoptions(linewidth=200, precision=2, suppress=True)
print("interval (us) ending at stamp j of a tile [j < nk: barrier of K tile j; nk: epilogue begin; nk+1: epilogue end], mean over WGs:")
for c in range(ntile):
    m = prof[c].mean(axis=0) / 1000
    print(f" tile {c}: first8 {m[:8]}  mid-mean {m[8:nk].mean():.3f}  epi-begin {m[nk]:.2f} epi {m[nk+1]:.2f}  tile total {prof[c].sum(axis=1).mean()/1000:.2f}")
steady = prof[1:, :, 8:nk].mean() / 1000
print(f"steady K-tile interval {steady:.3f} us -> main loop rate {2*256*256*128/steady/1e6*256/1e6:.0f} TF if all WGs; ideal 5PF = {2*256*256*128*256/5e15*1e6:.3f} us")
# spread across WGs of the tile-end times (lockstep?)
for c in range(ntile):
    endc = t[:, 1 + c * per_tile + nk + 1]
    print(f" tile {c} epilogue-end time over WGs: min {(endc.min()-t0all)/1000:.1f} p50 {(np.median(endc)-t0all)/1000:.1f} max {(endc.max()-t0all)/1000:.1f}")
xcc = st[:, SL - 3]
print("xcc of blocks 0..15:", xcc[:16])
np.save(os.path.join(ROOT, "gpurun_out", f"stamps_{role}.npy"), st[:, :n])
tot = (t[:, n - 1] - t[:, 0]) / 1000
if ntile * per_tile + 1 == n:
    print("per-WG total us by xcd (rows) x local (cols):")
    print(tot.reshape(32, 8).T)
