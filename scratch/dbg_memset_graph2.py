"""hipMemsetAsync node followed by a kernel node with a large argument block (the FP8 GEMM, ~250 B of
arguments) in one captured graph: does the memset still write its value on the 2nd, 3rd replay?"""
import ctypes, torch
from transformer_engine_amd import general_gemm
from transformer_engine_amd._lib import api
from transformer_engine_amd.tensor import DType, Float8Tensor, current_stream_ptr
hip = ctypes.CDLL("libamdhip64.so")
hip.hipMemsetAsync.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t, ctypes.c_void_p]
E4 = DType.kFloat8E4M3
mk = lambda r, c: Float8Tensor(shape=(r, c), dtype=torch.float32, data=torch.randint(0, 100, (r, c), dtype=torch.uint8, device="cuda"),
                               fp8_scale_inv=torch.ones(1, device="cuda"), fp8_dtype=E4)
W, X = mk(512, 256), mk(512, 256)
out = torch.zeros(512, 512, device="cuda")
for how in ("hipMemsetAsync", "nvte_memset"):
    t = torch.full((264,), 5, dtype=torch.uint8, device="cuda")
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        general_gemm(W, X, out=out, layout="TN"); torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            if how == "hipMemsetAsync":
                hip.hipMemsetAsync(t.data_ptr(), 0, 264, torch.cuda.current_stream().cuda_stream)
            else:
                api.nvte_memset(t.data_ptr(), 0, 264, current_stream_ptr())
            general_gemm(W, X, out=out, layout="TN")
    torch.cuda.synchronize()
    for i in range(4):
        t.fill_(9); torch.cuda.synchronize()
        g.replay(); torch.cuda.synchronize()
        print(how, "replay", i, "nonzero bytes", int((t != 0).sum()), "first 16:", t[:16].tolist())
