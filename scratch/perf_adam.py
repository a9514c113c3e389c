import sys, torch
from transformer_engine_amd import weight_update as WU
from transformer_engine_amd.tensor import DType
dev = "cuda"
n = 14336 * 4096
def timeit(fn, iters=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3
for gd, gsz in ((torch.bfloat16, 2), (torch.float32, 4), (torch.float16, 2), (torch.bfloat16, 2)):
    k = 4
    g = [(torch.randn(n, device=dev) * 0.01).to(gd) for _ in range(k)]
    pm = [torch.randn(n, device=dev) for _ in range(k)]
    m = [torch.zeros(n, device=dev) for _ in range(k)]
    v = [torch.zeros(n, device=dev) for _ in range(k)]
    p8 = [torch.zeros(n, dtype=torch.uint8, device=dev) for _ in range(k)]
    sc = [torch.ones(1, device=dev) for _ in range(k)]
    am = [torch.zeros(1, device=dev) for _ in range(k)]
    si = [torch.zeros(1, device=dev) for _ in range(k)]
    flag = torch.zeros(1, dtype=torch.int32, device=dev)
    t = timeit(lambda: WU.multi_tensor_applier(WU.multi_tensor_adam_fp8, flag, [g, p8, m, v, pm, sc, am, si], 1e-3, 0.9,
                                               0.999, 1e-8, 3, 1, 1, 0.01, DType.kFloat8E4M3))
    by = k * n * (gsz + 12 + 12 + 1)
    print(f"adam_fp8 g={str(gd)[6:]:9s} {k} x {n} elems: {t*1e6:8.1f} us  {by/t/1e9:7.0f} GB/s", flush=True)
    del g, pm, m, v, p8
