import sys; sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import numpy as np, torch, oracle
from transformer_engine_amd.tensor import *
x = torch.zeros(128, 128, dtype=torch.float32, device="cuda")
x[0:32, 32:64] = 1e-38
x[32:64, :] = torch.linspace(-3e38, 3e38, 128, device="cuda")
x[64:96, 0:32] = 2.0 ** -130
torch.manual_seed(0)
x[96:, 64:] = torch.randn(32, 64, device="cuda") * 1e-3
for fp8,fmt in ((DType.kFloat8E5M2,1),(DType.kFloat8E4M3,0)):
    out = MXFP8Quantizer(fp8)(x); torch.cuda.synchronize()
    ref = oracle.mxfp8(x.cpu().numpy(), fmt)
    for name, got, r in (("row", out._rowwise_data, ref["out_row"]), ("col", out._columnwise_data, ref["out_col"])):
        g = got.cpu().numpy()
        bad = np.argwhere(g != r)
        print(fp8, name, len(bad))
        for (i,j) in bad[:8]:
            sc = ref["sc_row"][i, j//32] if name=="row" else ref["sc_col"][i//32, j]
            print(i,j, repr(float(x[i,j])), hex(g[i,j]), hex(r[i,j]), "scale", sc, "xr", float(x[i,j])*float(oracle.lib().te_exp2f_rcp(int(sc))))
