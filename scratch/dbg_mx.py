import sys; sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np, torch, oracle
from transformer_engine_amd.tensor import *
from transformer_engine_amd.gemm import general_gemm
E4, E5 = DType.kFloat8E4M3, DType.kFloat8E5M2
bits = lambda t: oracle.float_to_bf16_bits(t.float().cpu().numpy())
for (M, K, N) in [(384, 512, 256), (256, 384, 512), (256, 512, 384), (256, 256, 256), (256, 384, 256), (256, 256, 384), (128, 128, 128)]:
    g = np.random.default_rng(7)
    x = torch.from_numpy(g.standard_normal((M, K), dtype=np.float32)).cuda().to(torch.bfloat16)
    w = torch.from_numpy(g.standard_normal((N, K), dtype=np.float32) * 0.05).cuda().to(torch.bfloat16)
    dy = torch.from_numpy(g.standard_normal((M, N), dtype=np.float32) * 0.01).cuda().to(torch.bfloat16)
    mk = lambda dt: MXFP8Quantizer(dt, rowwise=True, columnwise=True, columnwise_transposed=True)
    x8, w8, dy8 = mk(E4)(x), mk(E4)(w), mk(E5)(dy)
    dx, *_ = general_gemm(w8, dy8, out_dtype=torch.float32, layout="NN", grad=True)
    dw, *_ = general_gemm(x8, dy8, out_dtype=torch.float32, layout="NT", grad=True)
    torch.cuda.synchronize()
    rx, rw, rdy = oracle.mxfp8(bits(x), oracle.E4M3), oracle.mxfp8(bits(w), oracle.E4M3), oracle.mxfp8(bits(dy), oracle.E5M2)
    fdy_r = oracle.dequant_mxfp8(rdy["out_row"], oracle.E5M2, rdy["sc_row"], True)
    fw_c = oracle.dequant_mxfp8(rw["out_col"], oracle.E4M3, rw["sc_col"], False)
    fdy_c = oracle.dequant_mxfp8(rdy["out_col"], oracle.E5M2, rdy["sc_col"], False)
    fx_c = oracle.dequant_mxfp8(rx["out_col"], oracle.E4M3, rx["sc_col"], False)
    ref_dx = fdy_r.astype(np.float64) @ fw_c.astype(np.float64)
    ref_dw = fdy_c.astype(np.float64).T @ fx_c.astype(np.float64)
    for name, got, ref in (("dx", dx, ref_dx), ("dw", dw, ref_dw)):
        e = np.abs(got.cpu().numpy() - ref)
        bad = e > 1e-3 * np.abs(ref) + 1e-3 * np.abs(ref).max()
        rows_bad = np.where(bad.any(1))[0]; cols_bad = np.where(bad.any(0))[0]
        print((M, K, N), name, got.shape, "max err", e.max(), "absmax", np.abs(ref).max(), "bad", bad.sum(),
              "rows", (rows_bad.min(), rows_bad.max()) if len(rows_bad) else None, "cols", (cols_bad.min(), cols_bad.max()) if len(cols_bad) else None)
