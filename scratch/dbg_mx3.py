import sys; sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np, torch, oracle
from transformer_engine_amd.tensor import *
bits = lambda t: oracle.float_to_bf16_bits(t.float().cpu().numpy())
M, N = 256, 512
g = torch.Generator(device="cuda").manual_seed(5)
_ = torch.randn(M, 384, device="cuda", dtype=torch.bfloat16, generator=g)
_ = torch.randn(N, 384, device="cuda", dtype=torch.float32, generator=g)
dy = (torch.randn(M, N, device="cuda", dtype=torch.float32, generator=g) * 0.02).to(torch.bfloat16)
for tr in (False, True):
  for fmt, ofmt in ((DType.kFloat8E5M2, oracle.E5M2), (DType.kFloat8E4M3, oracle.E4M3)):
    q = MXFP8Quantizer(fmt, rowwise=True, columnwise=True, columnwise_transposed=tr)
    o = q(dy); torch.cuda.synchronize()
    r = oracle.mxfp8(bits(dy), ofmt)
    sr = o._rowwise_scale_inv.cpu().numpy()[:M, :N//32]
    dr = o._rowwise_data.cpu().numpy()
    print("transposed", tr, fmt, "row scale mismatches", (sr != r["sc_row"][:M, :N//32]).sum(), "row data mismatches", (dr != r["out_row"]).sum())
    if (sr != r["sc_row"][:M, :N//32]).any():
        i, j = np.argwhere(sr != r["sc_row"][:M, :N//32])[0]
        blk = dy[i, j*32:(j+1)*32].float().cpu().numpy()
        print("  first scale mismatch at", i, j, "gpu", sr[i, j], "oracle", r["sc_row"][i, j], "block amax", np.abs(blk).max())
