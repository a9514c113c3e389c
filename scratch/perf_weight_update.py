"""Bandwidth of the weight-update kernels on a Llama-70B-like FFN weight (14336 x 4096)."""
import torch
from transformer_engine_amd import weight_update as WU
from transformer_engine_amd.tensor import DType

dev = "cuda"
h, w = 14336, 4096
n = h * w


def timeit(fn, iters=20):
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


def pad(x, p):
    return (x + p - 1) // p * p


rs, cs = (pad(h, 128), pad(w // 32, 4)), (pad(h // 32, 4), pad(w, 128))
for dtype, esz in ((torch.bfloat16, 2), (torch.float32, 4)):
    for start, frac in ((0, 1.0), (0, 0.125), (8 * 1001, 0.125), (8 * 1001 + 3, 0.125)):
        ln = int(n * frac)
        inp = torch.randn(ln + 8, device=dev).to(dtype)[:ln]
        ar = torch.zeros(rs, dtype=dtype, device=dev)
        ac = torch.zeros(cs, dtype=dtype, device=dev)
        t = timeit(lambda: WU.mxfp8_scaling_compute_partial_amax(inp, ar, ac, h, w, start))
        by = ln * esz + (ar.numel() + ac.numel()) * esz
        print(f"partial_amax {str(dtype)[6:]:9s} start={start:6d} frac={frac:5.3f}: {t*1e6:8.1f} us  {by/t/1e9:7.0f} GB/s")
        if dtype == torch.bfloat16:
            sr = torch.empty(rs, dtype=torch.uint8, device=dev)
            sc = torch.empty(cs, dtype=torch.uint8, device=dev)
            t = timeit(lambda: WU.multi_tensor_applier(WU.multi_tensor_compute_scale_inv_e8m0, None, [[ar, ac], [sr, sc]]))
            print(f"  e8m0 {ar.numel()+ac.numel()} elems: {t*1e6:8.1f} us  {(ar.numel()+ac.numel())*3/t/1e9:7.0f} GB/s")
        else:
            sr = torch.full(rs, 127, dtype=torch.uint8, device=dev)
            sc = torch.full(cs, 127, dtype=torch.uint8, device=dev)
        o1 = torch.empty(ln, dtype=torch.uint8, device=dev)
        o2 = torch.empty(ln, dtype=torch.uint8, device=dev)
        t = timeit(lambda: WU.mxfp8_scaling_partial_cast(inp, o1, o2, sr, sc, h, w, start))
        by = ln * (esz + 2) + ln / 32 * 2
        print(f"partial_cast {str(dtype)[6:]:9s} start={start:6d} frac={frac:5.3f}: {t*1e6:8.1f} us  {by/t/1e9:7.0f} GB/s")

# Adam with FP8 parameters: 8 weights of 14336 x 4096
for gd, gsz in ((torch.float32, 4), (torch.bfloat16, 2)):
    k = 4
    g = [torch.randn(n, device=dev).to(gd) for _ in range(k)]
    pm = [torch.randn(n, device=dev) for _ in range(k)]
    m = [torch.zeros(n, device=dev) for _ in range(k)]
    v = [torch.zeros(n, device=dev) for _ in range(k)]
    p8 = [torch.zeros(n, dtype=torch.uint8, device=dev) for _ in range(k)]
    sc = [torch.ones(1, device=dev) for _ in range(k)]
    am = [torch.zeros(1, device=dev) for _ in range(k)]
    si = [torch.zeros(1, device=dev) for _ in range(k)]
    flag = torch.zeros(1, dtype=torch.int32, device=dev)
    t = timeit(lambda: WU.multi_tensor_applier(WU.multi_tensor_adam_fp8, flag, [g, p8, m, v, pm, sc, am, si], 1e-3, 0.9,
                                               0.999, 1e-8, 3, 1, 1, 0.01, DType.kFloat8E4M3), iters=10)
    by = k * n * (gsz + 12 + 12 + 1)
    print(f"adam_fp8 g={str(gd)[6:]:9s} {k} x {n} elems: {t*1e6:8.1f} us  {by/t/1e9:7.0f} GB/s")

# scale_and_scale_inv on 200 one-element tensors (one optimizer step of a 200-weight model)
k = 200
am = [torch.rand(1, device=dev) for _ in range(k)]
sc = [torch.empty(1, device=dev) for _ in range(k)]
si = [torch.empty(1, device=dev) for _ in range(k)]
flag = torch.zeros(1, dtype=torch.int32, device=dev)
t = timeit(lambda: WU.multi_tensor_applier(WU.multi_tensor_compute_scale_and_scale_inv, flag, [am, sc, si], 448.0, False, 0.0))
print(f"scale_and_scale_inv 200 tensors: {t*1e6:8.1f} us (host wrapping included)")
