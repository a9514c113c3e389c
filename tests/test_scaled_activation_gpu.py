"""GPU parity of the per-row-scaled activations (nvte_scaled_swiglu / _clamped_swiglu / _srelu and their
backward forms) against the restated host oracle of the reference's own test.

Axes, shapes, clamp constants and tolerances are tests/cpp/operator/test_scaled_activation.cu:30-32,190-201,
264-291: shapes {17,64} (vector path, interleaved), {13,100} (element-by-element fallback), {1024,2048};
interleave {0, 32}; data dtype x scale dtype; grad_act_scales on / off (its dtype = the data dtype);
getTolerances(dtype), fp32 widened to 5e-5.  The reference has no golden vectors for this path: the oracle is
its compute_reference, restated in oracle.scaled_activation (parity pinned by tests/test_oracle_pinning.py's
element-loop check of that restatement)."""
import numpy as np
import pytest
import torch

import oracle
from helpers import randn_input
from transformer_engine_amd import NVTEError
from transformer_engine_amd.activation import scaled_activation, scaled_dactivation

pytestmark = pytest.mark.gpu

LIMIT, ALPHA, OFFSET = 0.5, 1.702, 0.5                      # test_scaled_activation.cu:30-32
TOL = {torch.float32: (5e-5, 5e-5), torch.bfloat16: (1e-5, 8e-3), torch.float16: (1e-5, 1e-3)}
KINDS = ["swiglu", "clamped_swiglu", "srelu"]


def _uniform(shape, dtype, seed):
    g = torch.Generator().manual_seed(seed)              # fillUniform: [-2, 1)
    return (torch.rand(shape, generator=g) * 3.0 - 2.0).to(dtype).cuda()


def _close(name, got, ref, dtype):
    atol, rtol = TOL[dtype]
    ref = torch.from_numpy(ref).to(dtype).float().numpy()   # the reference compares values of the output dtype
    got = got.float().cpu().numpy()
    bad = np.abs(got - ref) > atol + rtol * np.abs(ref)
    assert not bad.any(), f"{name}: {bad.sum()} of {bad.size} outside tolerance, max err {np.abs(got - ref).max()}"


def _run(kind, dtype, sdtype, rows, hidden, interleave, with_scale_grad):
    gated = kind != "srelu"
    x = _uniform((rows, hidden * (2 if gated else 1)), dtype, 1)
    sc = _uniform((rows,), sdtype, 2)
    dy = _uniform((rows, hidden), dtype, 3)
    kw = dict(glu_interleave_size=interleave, limit=LIMIT, alpha=ALPHA, glu_linear_offset=OFFSET)
    out = scaled_activation(kind, x, sc, **kw)
    dx, dsc = scaled_dactivation(kind, dy, x, sc, with_scale_grad=with_scale_grad, **kw)
    torch.cuda.synchronize()
    ro, rdx, rds = oracle.scaled_activation(kind, x.float().cpu().numpy(), sc.float().cpu().numpy(),
                                            dy.float().cpu().numpy(), interleave=interleave, limit=LIMIT,
                                            alpha=ALPHA, offset=OFFSET)
    assert out.dtype == dtype and out.shape == (rows, hidden) and dx.shape == x.shape
    _close("output", out, ro, dtype)
    _close("grad_input", dx, rdx, dtype)
    if with_scale_grad:
        # a sum of `hidden` products: the reference's serial fp32 sum and any other summation order differ by
        # rounding of the partial sums, relative to the sum of magnitudes rather than to the (cancelling) result
        atol, rtol = TOL[dtype]
        got = dsc.float().cpu().numpy()
        ref = torch.from_numpy(rds).to(dtype).float().numpy()
        mag = np.abs(dy.float().cpu().numpy() * ro / np.where(sc.float().cpu().numpy() == 0, 1, sc.float().cpu().numpy())[:, None]).sum(1)
        assert (np.abs(got - ref) <= atol + rtol * np.abs(ref) + 2e-6 * mag).all()
    else:
        assert dsc is None


@pytest.mark.parametrize("kind", KINDS)
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("sdtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("rows,hidden", [(17, 64), (13, 100), (1024, 2048)])
@pytest.mark.parametrize("interleave", [0, 32])
@pytest.mark.parametrize("with_scale_grad", [False, True])
def test_scaled_activation(kind, dtype, sdtype, rows, hidden, interleave, with_scale_grad):
    if kind == "srelu" and interleave != 0:
        pytest.skip("Interleave has no meaning for SReLU.")
    if interleave and hidden % interleave:
        pytest.skip("Hidden size must be divisible by GLU interleave.")
    _run(kind, dtype, sdtype, rows, hidden, interleave, with_scale_grad)


@pytest.mark.parametrize("kind", KINDS)
@pytest.mark.parametrize("sdtype", [torch.float32, torch.float16])
@pytest.mark.parametrize("rows,hidden", [(17, 64), (13, 100)])
@pytest.mark.parametrize("interleave", [0, 32])
@pytest.mark.parametrize("with_scale_grad", [False, True])
def test_scaled_activation_fp16(kind, sdtype, rows, hidden, interleave, with_scale_grad):
    if kind == "srelu" and interleave != 0:
        pytest.skip("Interleave has no meaning for SReLU.")
    if interleave and hidden % interleave:
        pytest.skip("Hidden size must be divisible by GLU interleave.")
    _run(kind, torch.float16, sdtype, rows, hidden, interleave, with_scale_grad)


def test_scaled_activation_mixed_output_dtype_and_moe_shape():
    """bf16 in, fp32 out (element path) and an expert-MLP sized call (8192 routed tokens, 2 x 14336 wide)."""
    x = _uniform((64, 512), torch.bfloat16, 5)
    sc = _uniform((64,), torch.float32, 6)
    out = scaled_activation("swiglu", x, sc, out_dtype=torch.float32)
    ro, _, _ = oracle.scaled_activation("swiglu", x.float().cpu().numpy(), sc.cpu().numpy())
    _close("output", out, ro, torch.float32)
    x = _uniform((8192, 2 * 14336), torch.bfloat16, 7)
    sc = _uniform((8192,), torch.float32, 8)
    dy = _uniform((8192, 14336), torch.bfloat16, 9)
    out = scaled_activation("swiglu", x, sc, glu_interleave_size=32)
    dx, dsc = scaled_dactivation("swiglu", dy, x, sc, glu_interleave_size=32, scale_grad_dtype=torch.float32)
    torch.cuda.synchronize()
    rows = slice(4000, 4032)                                 # the oracle on a band of rows
    ro, rdx, rds = oracle.scaled_activation("swiglu", x[rows].float().cpu().numpy(), sc[rows].cpu().numpy(),
                                            dy[rows].float().cpu().numpy(), interleave=32)
    _close("output", out[rows], ro, torch.bfloat16)
    _close("grad_input", dx[rows], rdx, torch.bfloat16)
    assert np.allclose(dsc[rows].cpu().numpy(), rds, rtol=1e-3, atol=1e-2)
    # linearity in the scales (a size-independent property): doubling the scales doubles every output exactly
    out2 = scaled_activation("swiglu", x, sc * 2, glu_interleave_size=32)
    assert torch.equal(out2.float(), out.float() * 2)


def test_scaled_activation_argument_errors():
    x = _uniform((16, 128), torch.bfloat16, 1)
    sc = _uniform((16,), torch.float32, 2)
    with pytest.raises(NVTEError, match="multiple of 32"):
        scaled_activation("swiglu", x, sc, glu_interleave_size=16)
    with pytest.raises(NVTEError, match="divisible by glu_interleave_size"):
        scaled_activation("swiglu", _uniform((16, 192), torch.bfloat16, 1), sc, glu_interleave_size=64)
    with pytest.raises(NVTEError, match="one value per row"):
        scaled_activation("srelu", x, sc[:8].contiguous())
    # empty input: nothing launched, nothing raised
    e = scaled_activation("srelu", torch.empty((0, 64), dtype=torch.bfloat16, device="cuda"),
                          torch.empty((0,), dtype=torch.float32, device="cuda"))
    assert e.shape == (0, 64)
