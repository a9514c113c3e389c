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
keep = []
for tag, gd, gsz, gscale, hold in (("bf16 unit-scale alone", torch.bfloat16, 2, 1.0, False),
                                   ("fp32 0.01 (kept alive)", torch.float32, 4, 0.01, True),
                                   ("bf16 0.01 after kept fp32", torch.bfloat16, 2, 0.01, False),
                                   ("bf16 unit-scale again", torch.bfloat16, 2, 1.0, False)):
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
    t = timeit(lambda: WU.multi_tensor_applier(WU.multi_tensor_adam_fp8, flag, [g, p8, m, v, pm, sc, am, si], 1e-3, 0.9,
                                               0.999, 1e-8, 3, 1, 1, 0.01, DType.kFloat8E4M3))
    by = k * n * (gsz + 12 + 12 + 1)
    print(f"{tag:28s}: {t*1e6:8.1f} us  {by/t/1e9:7.0f} GB/s  ptrs g={g[0].data_ptr():#x} pm={pm[0].data_ptr():#x} m={m[0].data_ptr():#x} v={v[0].data_ptr():#x} p8={p8[0].data_ptr():#x}", flush=True)
    if hold:
        keep.append((g, pm, m, v, p8))
    del g, pm, m, v, p8
