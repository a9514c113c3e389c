import sys; sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np, torch, oracle
from transformer_engine_amd import linear as L
from transformer_engine_amd.recipe import MXFP8BlockScaling
from transformer_engine_amd.gemm import general_gemm
bits = lambda t: oracle.float_to_bf16_bits(t.float().cpu().numpy())
M, K, N = 256, 384, 512
g = torch.Generator(device="cuda").manual_seed(5)
x = torch.randn(M, K, device="cuda", dtype=torch.bfloat16, generator=g)
w = (torch.randn(N, K, device="cuda", dtype=torch.float32, generator=g) * 0.05).to(torch.bfloat16)
dy = (torch.randn(M, N, device="cuda", dtype=torch.float32, generator=g) * 0.02).to(torch.bfloat16)
st = L.LinearFP8State(MXFP8BlockScaling(), device=torch.device("cuda"))
y, saved = L.linear_forward(x, w, None, st)
dx, dw, dbias = L.linear_backward(dy, saved, st)
torch.cuda.synchronize()
x8, w8 = saved
print("x8 col", None if x8._columnwise_data is None else x8._columnwise_data.shape, x8._columnwise_transposed, "w8 col", w8._columnwise_data.shape, w8._columnwise_transposed)
rx, rw, rdy = oracle.mxfp8(bits(x), oracle.E4M3), oracle.mxfp8(bits(w), oracle.E4M3), oracle.mxfp8(bits(dy), oracle.E5M2)
print("w8 colT bytes equal:", np.array_equal(w8._columnwise_data.cpu().numpy(), rw["out_col"].T), " scales:", np.array_equal(w8._columnwise_scale_inv.cpu().numpy()[:K, :N//32], rw["sc_col"][:N//32, :K].T))
print("x8 colT bytes equal:", np.array_equal(x8._columnwise_data.cpu().numpy(), rx["out_col"].T))
d = oracle.dequant_mxfp8
fdy_r = d(rdy["out_row"], oracle.E5M2, rdy["sc_row"], True); fw_c = d(rw["out_col"], oracle.E4M3, rw["sc_col"], False)
ref_dx = fdy_r.astype(np.float64) @ fw_c.astype(np.float64)
e = np.abs(dx.float().cpu().numpy() - ref_dx); print("dx err", e.max(), "absmax", np.abs(ref_dx).max(), dx.dtype)
bad = e > 1.6e-2 * np.abs(ref_dx) + 1e-3 * np.abs(ref_dx).max(); print("bad", bad.sum(), np.where(bad.any(1))[0][:10], np.where(bad.any(0))[0][:10])
i, j = np.unravel_index(e.argmax(), e.shape); print(i, j, float(dx[i, j]), ref_dx[i, j])
from transformer_engine_amd.tensor import *
qg = st.bwd_q[L.BWD_GRAD_OUTPUT]
dy8 = qg(dy)
print("dy8 row bytes equal:", np.array_equal(dy8._rowwise_data.cpu().numpy(), rdy["out_row"]), "scales", np.array_equal(dy8._rowwise_scale_inv.cpu().numpy()[:M, :N//32], rdy["sc_row"][:M, :N//32]))
for od in (torch.float32, torch.bfloat16):
    o, *_ = general_gemm(w8, dy8, out_dtype=od, layout="NN", grad=True)
    torch.cuda.synchronize()
    e = np.abs(o.float().cpu().numpy() - ref_dx); print(od, "dx err", e.max())
o, *_ = general_gemm(w8, dy8, out_dtype=torch.float32, layout="NN", grad=True)
o2, *_ = general_gemm(w8, dy8, out_dtype=torch.float32, layout="NN", grad=True)
torch.cuda.synchronize()
print("repeatable:", torch.equal(o, o2))
e = np.abs(o.cpu().numpy() - ref_dx)
print("err by 64-col blocks:", [float(e[:, c:c+64].max()) for c in range(0, 384, 64)])
print("err by 64-row blocks:", [float(e[r:r+64, :].max()) for r in range(0, 256, 64)])
