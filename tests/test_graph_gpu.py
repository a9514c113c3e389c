"""HIP-graph replay of launches that zero device memory first.

ROCm 7.2 / gfx950: a hipMemsetAsync captured as a memset node and followed by a kernel node with a large
argument block (the FP8 GEMM) writes a 16-byte pattern of foreign bytes instead of its value from the
second replay on (scratch/dbg_memset_graph2.py reproduces it with the bare runtime call).  The library
therefore zeroes with a kernel of its own (fill_async, tensor_api.cpp) everywhere a launch can be
captured; these tests pin that: the reference relies on cudaMemsetAsync in the same places
(common/gemm/cublaslt_gemm.cu amax of an FP8 output, nvte_memset / nvte_zero_tensor
common/transformer_engine.cpp)."""
import pytest
import torch

from transformer_engine_amd import general_gemm
from transformer_engine_amd._lib import api
from transformer_engine_amd.tensor import DType, Float8Quantizer, Float8Tensor, current_stream_ptr

pytestmark = pytest.mark.gpu
E4 = DType.kFloat8E4M3


def _fp8(rows, cols, seed, lo=-3, hi=4):
    g = torch.Generator().manual_seed(seed)
    v = torch.randint(lo, hi, (rows, cols), generator=g).float()
    t = Float8Tensor(shape=(rows, cols), dtype=torch.float32, data=v.to(torch.float8_e4m3fn).view(torch.uint8).cuda(),
                     fp8_scale_inv=torch.ones(1, device="cuda"), fp8_dtype=E4)
    return t, v.cuda()


def test_nvte_memset_then_gemm_in_a_graph_stays_correct_on_every_replay():
    W, w = _fp8(512, 256, 1)
    X, x = _fp8(512, 256, 2)
    out = torch.zeros(512, 512, device="cuda")
    buf = torch.full((264,), 5, dtype=torch.uint8, device="cuda")
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        general_gemm(W, X, out=out, layout="TN")
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=s):
            api.nvte_memset(buf.data_ptr(), 0, buf.numel(), current_stream_ptr())
            general_gemm(W, X, out=out, layout="TN")
    torch.cuda.synchronize()
    ref = x @ w.T
    for _ in range(4):
        buf.fill_(9)
        out.zero_()
        torch.cuda.synchronize()
        graph.replay()
        torch.cuda.synchronize()
        assert int((buf != 0).sum()) == 0
        assert torch.equal(out, ref)


def test_fp8_output_gemm_amax_is_rearmed_on_every_replay():
    """D in FP8 with delayed scaling: the GEMM zeroes amax, then takes the atomic max of |D| (ref
    cublaslt_gemm.cu:619-634).  Replayed with smaller inputs the amax must go DOWN again."""
    W, w = _fp8(256, 256, 3)
    X, x = _fp8(256, 256, 4)
    scale, amax = torch.ones(1, device="cuda"), torch.zeros(1, device="cuda")
    q = Float8Quantizer(scale, amax, E4, columnwise=False)
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        out = general_gemm(W, X, layout="TN", quantization_params=q, out_dtype=torch.bfloat16)[0]
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=s):
            general_gemm(W, X, layout="TN", quantization_params=q, out=out)
    torch.cuda.synchronize()
    big = float((x @ w.T).abs().max())
    saved = X._data.clone()
    for i in range(4):
        if i % 2 == 1:
            X._data.zero_()            # all-zero operand: |D| max = 0
        else:
            X._data.copy_(saved)
        torch.cuda.synchronize()
        graph.replay()
        torch.cuda.synchronize()
        assert float(amax) == (0.0 if i % 2 == 1 else big), (i, float(amax), big)
