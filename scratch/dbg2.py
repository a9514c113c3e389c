import sys; sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import numpy as np, torch, oracle
from helpers import *
from transformer_engine_amd import general_gemm
from transformer_engine_amd.tensor import *
E4, E5 = DType.kFloat8E4M3, DType.kFloat8E5M2
M,N,K=128,96,256
x = randn_input((M, K), torch.bfloat16, 11); w = randn_input((N, K), torch.bfloat16, 12, scale=0.05)
qx = Float8CurrentScalingQuantizer(E4, None)(x); qw = Float8CurrentScalingQuantizer(E4, None)(w)
y,*_ = general_gemm(qw, qx, out_dtype=torch.float32, layout="TN"); torch.cuda.synchronize()
fx = oracle.dequant(u8(qx._data), 0, qx._scale_inv.item()); fw = oracle.dequant(u8(qw._data), 0, qw._scale_inv.item())
ref = oracle.gemm_tn(fx, fw); g = y.cpu().numpy()
err = np.abs(g-ref); print("nonMX max abs err", err.max(), "ref absmax", np.abs(ref).max(), "max rel", (err/np.maximum(np.abs(ref),1e-3*np.abs(ref).max())).max())
# fp32 torch matmul for comparison of accumulate-order noise
t = (torch.from_numpy(fx) @ torch.from_numpy(fw).T).numpy(); print("torch fp32 vs ref", np.abs(t-ref).max())

# MX decode
def mx_tensor(rows, K, scale_fn):
    data = torch.full((rows, K), 0x38, dtype=torch.uint8, device="cuda")  # 1.0 in e4m3
    rs = ((rows+127)//128*128, ((K//32)+3)//4*4)
    sc = np.full(rs, 127, np.uint8)
    for r in range(rows):
        for kb in range(K//32):
            sc[r,kb] = scale_fn(r,kb)
    return MXFP8Tensor(shape=(rows,K), dtype=torch.float32, rowwise_data=data, rowwise_scale_inv=torch.from_numpy(sc).cuda(), columnwise_data=None, columnwise_scale_inv=None, fp8_dtype=E4), sc
for name, fnP, fnQ in [("P kb-dependent", lambda r,kb: 127+kb, lambda r,kb: 127),
                       ("Q kb-dependent", lambda r,kb: 127, lambda r,kb: 127+kb),
                       ("P row-dependent", lambda r,kb: 127+(r%8), lambda r,kb: 127),
                       ("Q row-dependent", lambda r,kb: 127, lambda r,kb: 127+(r%8))]:
    M,N,K = 64,64,128
    P, sp = mx_tensor(M,K,fnP); Q, sq = mx_tensor(N,K,fnQ)
    y,*_ = general_gemm(Q, P, out_dtype=torch.float32, layout="TN"); torch.cuda.synchronize()
    g = y.cpu().numpy()
    exp = np.zeros((M,N))
    for kb in range(K//32):
        exp += 32.0 * np.outer(2.0**(sp[:M,kb].astype(int)-127), 2.0**(sq[:N,kb].astype(int)-127))
    print(name, "match" if np.array_equal(g,exp) else "MISMATCH")
    if not np.array_equal(g,exp):
        print(" got row0[:8]", g[0,:8], " exp", exp[0,:8]); print(" got col0[:8]", g[:8,0], " exp", exp[:8,0]); print(" got[32:36,0]", g[32:36,0], g[0,32:36])
