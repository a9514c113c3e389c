"""Module-level (torch.nn.Module + autograd) look-alikes of te.Linear / te.LayerNormLinear /
te.GroupedLinear against the plain-PyTorch fp32 composition, in the style of the reference's
tests/pytorch/test_numerics.py (same module, torch reference, tolerances per dtype/recipe).
FP8 tolerances: rtol 0.125 / atol 0.0675 x absmax (ref tests/pytorch/utils.py:103-106) on 99.9 % of
the elements, twice that bound on all, relative Frobenius error < 8 % (E5M2 gradients: see
tests/test_linear_gpu.py)."""
import pytest
import torch

from transformer_engine_amd.grouped_linear import GroupedLinear
from transformer_engine_amd.linear import LayerNormLinear, Linear
from transformer_engine_amd.recipe import DelayedScaling, Float8CurrentScaling, MXFP8BlockScaling

pytestmark = pytest.mark.gpu
RECIPES = {"delayed": DelayedScaling, "current": Float8CurrentScaling, "mxfp8": MXFP8BlockScaling}


def close(got, ref):
    got, ref = got.float(), ref.float()
    err = (got - ref).abs()
    bound = 0.125 * ref.abs() + 0.0675 * ref.abs().max()
    assert (err <= 2 * bound).all()
    assert (err <= bound).float().mean().item() >= 0.999
    assert (got - ref).norm().item() <= 0.08 * ref.norm().item()


@pytest.mark.parametrize("recipe", list(RECIPES))
@pytest.mark.parametrize("bias", [False, True])
def test_linear_module_autograd(recipe, bias):
    torch.manual_seed(1)
    M, K, N = 256, 512, 384
    mod = Linear(K, N, bias=bias, recipe=RECIPES[recipe]())
    if bias:
        torch.nn.init.normal_(mod.bias, std=0.5)
    x = torch.randn(2, M // 2, K, device="cuda", dtype=torch.bfloat16, requires_grad=True)
    dy = torch.randn(2, M // 2, N, device="cuda", dtype=torch.bfloat16)
    for _ in range(3 if recipe == "delayed" else 1):     # delayed scaling: let the scales settle
        mod.zero_grad(); x.grad = None
        y = mod(x)
        y.backward(dy)
    xr = x.detach().float().requires_grad_(True)
    wr = mod.weight.detach().float().requires_grad_(True)
    br = mod.bias.detach().float().requires_grad_(True) if bias else None
    yr = torch.nn.functional.linear(xr, wr, br)
    yr.backward(dy.float())
    close(y, yr)
    close(x.grad, xr.grad)
    close(mod.weight.grad, wr.grad)
    if bias:
        close(mod.bias.grad, br.grad)


@pytest.mark.parametrize("recipe", list(RECIPES))
@pytest.mark.parametrize("normalization", ["LayerNorm", "RMSNorm"])
def test_layernorm_linear_module_autograd(recipe, normalization):
    torch.manual_seed(2)
    M, K, N = 256, 1024, 384
    mod = LayerNormLinear(K, N, bias=True, normalization=normalization, recipe=RECIPES[recipe]())
    with torch.no_grad():
        mod.layer_norm_weight.normal_(1.0, 0.2)
        if mod.layer_norm_bias is not None:
            mod.layer_norm_bias.normal_(0.0, 0.1)
        mod.weight.normal_(0.0, 0.05)
    x = torch.randn(M, K, device="cuda", dtype=torch.bfloat16, requires_grad=True)
    dy = torch.randn(M, N, device="cuda", dtype=torch.bfloat16)
    for _ in range(3 if recipe == "delayed" else 1):
        mod.zero_grad(); x.grad = None
        y = mod(x)
        y.backward(dy)
    xr = x.detach().float().requires_grad_(True)
    g = mod.layer_norm_weight.detach().float().requires_grad_(True)
    w = mod.weight.detach().float().requires_grad_(True)
    b = mod.bias.detach().float().requires_grad_(True)
    if normalization == "LayerNorm":
        beta = mod.layer_norm_bias.detach().float().requires_grad_(True)
        ln = torch.nn.functional.layer_norm(xr, (K,), g, beta, mod.eps)
    else:
        ln = xr * torch.rsqrt(xr.pow(2).mean(-1, keepdim=True) + mod.eps) * g
    yr = torch.nn.functional.linear(ln, w, b)
    yr.backward(dy.float())
    close(y, yr)
    close(x.grad, xr.grad)
    close(mod.weight.grad, w.grad)
    close(mod.layer_norm_weight.grad, g.grad)
    close(mod.bias.grad, b.grad)
    if normalization == "LayerNorm":
        close(mod.layer_norm_bias.grad, beta.grad)


def test_grouped_linear_module_autograd():
    torch.manual_seed(3)
    G, K, N = 4, 256, 384
    splits = [96, 0, 160, 64]
    mod = GroupedLinear(G, K, N)
    m_splits = torch.tensor(splits, device="cuda", dtype=torch.int64)
    x = torch.randn(sum(splits), K, device="cuda", dtype=torch.bfloat16, requires_grad=True)
    dy = torch.randn(sum(splits), N, device="cuda", dtype=torch.bfloat16)
    y = mod(x, m_splits)
    y.backward(dy)
    xr = x.detach().float().requires_grad_(True)
    wr = mod.weight.detach().float().requires_grad_(True)
    outs, o = [], 0
    for i, m in enumerate(splits):
        outs.append(xr[o:o + m] @ wr[i * N:(i + 1) * N].t())
        o += m
    yr = torch.cat(outs)
    yr.backward(dy.float())
    close(y, yr)
    close(x.grad, xr.grad)
    # the expert without tokens has a zero weight gradient
    assert (mod.weight.grad[N:2 * N] == 0).all()
    close(mod.weight.grad, wr.grad)
    with pytest.raises(TypeError):
        mod(x, splits)
