"""Where do the 4.6-6.9 ms of the bf16-gradient Adam step (DESIGN 3.10 observation) go?  Per call: host
time inside the call, GPU time between an event pair around the call, and the wall time of the loop."""
import time, torch
from transformer_engine_amd import weight_update as WU
from transformer_engine_amd.tensor import DType
dev = "cuda"
n = 14336 * 4096
keep = []
def measure(fn, iters=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    host = []
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    s.record()
    for a, b in ev:
        a.record()
        h0 = time.perf_counter()
        fn()
        host.append(time.perf_counter() - h0)
        b.record()
    e.record()
    t_issue = time.perf_counter() - t0
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    per = [a.elapsed_time(b) for a, b in ev]
    return dict(loop_ms=s.elapsed_time(e) / iters, call_gpu_ms=sorted(per)[len(per) // 2], call_gpu_max=max(per),
                host_call_ms=1e3 * sorted(host)[len(host) // 2], host_call_max=1e3 * max(host),
                issue_ms=1e3 * t_issue / iters, wall_ms=1e3 * wall / iters)
for tag, gd, gscale, hold in (("bf16 unit-scale alone", torch.bfloat16, 1.0, False),
                              ("fp32 0.01 (kept alive)", torch.float32, 0.01, True),
                              ("bf16 0.01 after kept fp32", torch.bfloat16, 0.01, False),
                              ("bf16 unit-scale again", torch.bfloat16, 1.0, False),
                              ("bf16 0.01 again", torch.bfloat16, 0.01, False)):
    k = 4
    g = [(torch.randn(n, device=dev) * gscale).to(gd) for _ in range(k)]
    pm = [torch.randn(n, device=dev) for _ in range(k)]
    m = [torch.zeros(n, device=dev) for _ in range(k)]
    v = [torch.zeros(n, device=dev) for _ in range(k)]
    p8 = [torch.zeros(n, dtype=torch.uint8, device=dev) for _ in range(k)]
    sc = [torch.ones(1, device=dev) for _ in range(k)]
    am = [torch.zeros(1, device=dev) for _ in range(k)]
    si = [torch.zeros(1, device=dev) for _ in range(k)]
    flag = torch.zeros(1, dtype=torch.int32, device=dev)
    fn = lambda: WU.multi_tensor_applier(WU.multi_tensor_adam_fp8, flag, [g, p8, m, v, pm, sc, am, si], 1e-3, 0.9,
                                         0.999, 1e-8, 3, 1, 1, 0.01, DType.kFloat8E4M3)
    r = measure(fn)
    # the same state, pointer tables prebuilt once: is it the host side of the call?
    print(f"{tag:28s}: " + "  ".join(f"{a}={b:7.3f}" for a, b in r.items()), flush=True)
    print(f"{'':28s}  mem allocated {torch.cuda.memory_allocated()/2**30:.2f} GiB reserved {torch.cuda.memory_reserved()/2**30:.2f} GiB", flush=True)
    if hold:
        keep.append((g, pm, m, v, p8))
    del g, pm, m, v, p8, fn
