import sys; sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import numpy as np, torch, oracle
from transformer_engine_amd import general_gemm
from transformer_engine_amd.tensor import *
E4 = DType.kFloat8E4M3
def mx(rows, K, data, sc):
    rs = ((rows+127)//128*128, ((K//32)+3)//4*4)
    s = np.full(rs, 127, np.uint8); s[:rows,:K//32] = sc
    return MXFP8Tensor(shape=(rows,K), dtype=torch.float32, rowwise_data=torch.from_numpy(data).cuda(), rowwise_scale_inv=torch.from_numpy(s).cuda(), columnwise_data=None, columnwise_scale_inv=None, fp8_dtype=E4)
M=N=32; K=128
for quarter in range(8):
    d = np.zeros((M,K), np.uint8); d[:, quarter*16:(quarter+1)*16] = 0x38
    sc = np.tile(np.array([128,130,132,134],np.uint8),(M,1))   # kb0:2, kb1:8, kb2:32, kb3:128
    P = mx(M,K,d,sc); Q = mx(N,K,np.full((N,K),0x38,np.uint8), np.full((N,K//32),127,np.uint8))
    y,*_ = general_gemm(Q,P,out_dtype=torch.float32, layout="TN"); torch.cuda.synchronize()
    print("P quarter", quarter, "-> per-element scale", y[0,0].item()/16)
for quarter in range(8):
    d = np.zeros((M,K), np.uint8); d[:, quarter*16:(quarter+1)*16] = 0x38
    sc = np.tile(np.array([128,130,132,134],np.uint8),(M,1))
    Q = mx(M,K,d,sc); P = mx(N,K,np.full((N,K),0x38,np.uint8), np.full((N,K//32),127,np.uint8))
    y,*_ = general_gemm(Q,P,out_dtype=torch.float32, layout="TN"); torch.cuda.synchronize()
    print("Q quarter", quarter, "-> per-element scale", y[0,0].item()/16)
