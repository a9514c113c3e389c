"""Same-box A/B of two builds of the library on the configs[1] GEMMs (+ MXFP8 8192^3): one child process per
build and round, interleaved, >= 1 s of warm-up under load, median of the rounds.
usage: python scratch/perf_gemm_ab.py <libA.so> <libB.so> [<libC.so> ...] [rounds]"""
import os, subprocess, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys, time, json
sys.path.insert(0, %r)
import torch
from transformer_engine_amd.tensor import DType, Float8Quantizer, MXFP8Quantizer
from transformer_engine_amd.gemm import general_gemm
def q8(x, dt=DType.kFloat8E4M3, s=1.0, columnwise=True):
    return Float8Quantizer(torch.full((1,), s, device="cuda"), torch.zeros(1, device="cuda"), dt, columnwise=columnwise)(x)
tok, hid, ffn = 8192, 4096, 14336
g = torch.Generator(device="cuda").manual_seed(1)
x = torch.randn(tok, hid, device="cuda", dtype=torch.bfloat16, generator=g)
w = torch.randn(ffn, hid, device="cuda", dtype=torch.bfloat16, generator=g) * 0.02
dy = torch.randn(tok, ffn, device="cuda", dtype=torch.bfloat16, generator=g) * 0.01
xq, wq, dyq = q8(x, s=64.0), q8(w, s=2048.0), q8(dy, DType.kFloat8E5M2, 65536.0)
y = torch.empty(tok, ffn, device="cuda", dtype=torch.bfloat16); dx = torch.empty(tok, hid, device="cuda", dtype=torch.bfloat16); dw = torch.empty(ffn, hid, device="cuda", dtype=torch.bfloat16)
qm = MXFP8Quantizer(DType.kFloat8E4M3); qm.optimize_for_gemm = True
a8 = torch.randn(8192, 8192, device="cuda", dtype=torch.bfloat16, generator=g); b8 = torch.randn(8192, 8192, device="cuda", dtype=torch.bfloat16, generator=g)
am, bm = qm(a8), qm(b8)
o8 = torch.empty(8192, 8192, device="cuda", dtype=torch.bfloat16)
xs = q8(torch.randn(8192, 256, device="cuda", dtype=torch.bfloat16, generator=g)); wsm = q8(torch.randn(8192, 256, device="cuda", dtype=torch.bfloat16, generator=g))
qms = MXFP8Quantizer(DType.kFloat8E4M3); qms.optimize_for_gemm = True
xsm, wsmm = qms(torch.randn(8192, 256, device="cuda", dtype=torch.bfloat16, generator=g)), qms(torch.randn(8192, 256, device="cuda", dtype=torch.bfloat16, generator=g))
xr, wr, dyr = q8(x, s=64.0, columnwise=False), q8(w, s=2048.0, columnwise=False), q8(dy, DType.kFloat8E5M2, 65536.0, columnwise=False)
qmr = MXFP8Quantizer(DType.kFloat8E4M3, columnwise=True); qmr.optimize_for_gemm = True
amr, bmr = qmr(a8), qmr(b8)    # reference layout: columnwise payload un-transposed
fns = {"dgrad_km": lambda: general_gemm(wr, dyr, layout="NN", out=dx), "wgrad_km": lambda: general_gemm(xr, dyr, layout="NT", out=dw),
       "mxNN": lambda: general_gemm(amr, bmr, layout="NN", out=o8), "mxNT": lambda: general_gemm(amr, bmr, layout="NT", out=o8),
       "k256": lambda: general_gemm(wsm, xs, layout="TN", out=o8), "mxk256": lambda: general_gemm(wsmm, xsm, layout="TN", out=o8), "fprop": lambda: general_gemm(wq, xq, layout="TN", out=y), "dgrad": lambda: general_gemm(wq, dyq, layout="NN", out=dx),
       "wgrad": lambda: general_gemm(xq, dyq, layout="NT", out=dw), "mx8192": lambda: general_gemm(am, bm, layout="TN", out=o8)}
t0 = time.time()
while time.time() - t0 < 1.0:
    for f in fns.values(): f()
    torch.cuda.synchronize()
res = {}
for name, f in fns.items():
    for _ in range(5): f()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(30): f()
    e.record(); torch.cuda.synchronize()
    res[name] = s.elapsed_time(e) / 30 * 1e3
print("RES", json.dumps(res))
''' % ROOT
libs = [a for a in sys.argv[1:] if not a.isdigit()]
rounds = int(sys.argv[-1]) if sys.argv[-1].isdigit() else 3
acc = {l: {} for l in libs}
for r in range(rounds):
    for l in libs:
        lib, *kv = l.split("@")
        env = dict(os.environ, TE_AMD_LIBRARY=os.path.abspath(lib), **dict(x.split("=") for x in kv))
        out = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True, timeout=300)
        line = [x for x in out.stdout.splitlines() if x.startswith("RES")]
        if not line:
            print(out.stdout[-2000:], out.stderr[-2000:]); sys.exit(1)
        for k, v in json.loads(line[0][4:]).items():
            acc[l].setdefault(k, []).append(v)
med = lambda a: sorted(a)[len(a) // 2]
for l in libs:
    print(l, {k: round(med(v), 1) for k, v in acc[l].items()}, "us;  GEMM mean",
          round(sum(med(acc[l][k]) for k in ("fprop", "dgrad", "wgrad")) / 3, 1), "us =",
          round(2 * 8192 * 4096 * 14336 / (sum(med(acc[l][k]) for k in ("fprop", "dgrad", "wgrad")) / 3) / 1e6 / 5000, 4), "of 5 PF")
