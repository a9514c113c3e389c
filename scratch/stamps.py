import sys; sys.path.insert(0, '.')
import os, torch, time
buf = torch.zeros(256 * 64, dtype=torch.int64, device="cuda")
os.environ["NVTE_AMD_GEMM_STAMPS"] = hex(buf.data_ptr())
from transformer_engine_amd.tensor import *
from transformer_engine_amd.gemm import general_gemm
def q8(x, dt=DType.kFloat8E4M3):
    q = Float8Quantizer(torch.full((1,), 64.0, device="cuda"), torch.zeros(1, device="cuda"), dt)
    return q(x)
n, m, k = 8192, 14336, 4096
b = q8(torch.randn(n, k, device="cuda", dtype=torch.bfloat16)); a = q8(torch.randn(m, k, device="cuda", dtype=torch.bfloat16))
out = torch.empty(n, m, device="cuda", dtype=torch.bfloat16)
t0 = time.time()
while time.time() - t0 < 0.5: general_gemm(a, b, out=out, layout="TN")
torch.cuda.synchronize()
general_gemm(a, b, out=out, layout="TN"); torch.cuda.synchronize()
s = buf.cpu().view(256, 64).numpy()
import numpy as np
# per WG: stamp0 start; per tile: [after kt0, after kt1, before epilogue, after epilogue]
for wg in (0, 1, 8, 100, 255):
    r = s[wg]; base = r[0]
    print("wg", wg, [(int(x - base)) * 10 for x in r[1:29]])  # ns
d = s[:, 1:29].astype(np.int64) - s[:, :1]
t = d.reshape(256, 7, 4) * 10.0
print("mean per tile (ns): first ktile", np.diff(np.concatenate([np.zeros((256,1)), t[:, :, 0]], 1)).mean())
print("epilogue dur mean us", ((t[:, :, 3] - t[:, :, 2]).mean()) / 1e3)
print("ktile0 dur (after epilogue) mean us", ((t[:, 1:, 0] - t[:, :-1, 3]).mean()) / 1e3, " ktile1 dur", ((t[:, :, 1] - t[:, :, 0]).mean()) / 1e3)
print("main loop rest (kt2..end) mean us", ((t[:, :, 2] - t[:, :, 1]).mean()) / 1e3, "=> per ktile", ((t[:, :, 2] - t[:, :, 1]).mean()) / 1e3 / 30)
print("total us", d.max() * 10 / 1e3)
