"""Pins the weight-update part of the CPU oracle (oracle/te_oracle.c: te_mxfp8_partial_*,
te_e8m0_from_amax_ref, te_scale_and_scale_inv_ref, te_adam_fp8_ref) against what the reference's
own tests hold for this path, restated here with numpy / CPU torch:

  * tests/pytorch/test_partial_cast.py:18-53  (compute_partial_amax_reference,
    partial_cast_reference) on the test's own shard cases (:131-137, the 16384x6144 one scaled down),
  * tests/pytorch/test_multi_tensor.py:398-406 (the e8m0 exponent formula), on EVERY finite
    non-negative bf16 value,
  * tests/pytorch/test_multi_tensor.py:343-374 (scale, scale_inv exact) through the reference's
    pure-torch scale function when the reference tree is present, else the pinned scalar oracle,
  * tests/pytorch/test_fused_optimizer.py:486-489 (FP8 Adam: master weights assert_close to
    torch.optim.Adam(W) with the fp32 defaults; the FP8 model weights are the saturating E4M3 cast
    of master * scale, i.e. within half an E4M3 ulp, 2^-4 relative, after dequantisation).
No GPU needed.
"""
import importlib.util
import os

import numpy as np
import pytest
import torch

import oracle

# (shard length, rows, cols, start_offset): ref tests/pytorch/test_partial_cast.py:131-137
SHARD_CASES = [(3, 32, 64, 31), (64 * 64 - 2, 64, 64, 1), (512 * 384, 512, 384, 0), (32768, 256, 128, 0),
               (131072, 768, 256, 0), (65536, 768, 256, 131072), (98304, 128, 768, 0),
               (1000, 96, 160, 777), (0, 32, 32, 5)]


def _bf16(x: torch.Tensor) -> np.ndarray:
    return x.to(torch.bfloat16).view(torch.int16).numpy().view(np.uint16)


def _shard(n, seed):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(n, generator=g).to(torch.bfloat16)


@pytest.mark.parametrize("n,h,w,start", SHARD_CASES)
def test_partial_amax_matches_reference_test_formula(n, h, w, start):
    inp = _shard(n, 11)
    # ref test_partial_cast.py:18-29
    full = torch.zeros(h * w, dtype=inp.dtype)
    full[start:start + n].copy_(inp)
    full = torch.abs(full).float()
    ref_row = full.view(h, w // 32, 32).max(dim=2).values.numpy()
    ref_col = full.view(h // 32, 32, w).max(dim=1).values.numpy()
    ar, ac = oracle.mxfp8_partial_amax(_bf16(inp), h, w, start)
    assert np.array_equal(ar[:h, :w // 32], ref_row)
    assert np.array_equal(ac[:h // 32, :w], ref_col)
    # padding untouched (the reference compares the whole padded arrays, :79-80)
    assert not ar[h:].any() and not ar[:, w // 32:].any()
    assert not ac[h // 32:].any() and not ac[:, w:].any()


def test_e8m0_from_amax_on_every_bf16_value():
    bits = np.arange(0, 0x7F80, dtype=np.uint16)  # +0 .. largest finite bf16
    amax = oracle.bf16_bits_to_float(bits)
    got = oracle.e8m0_from_amax(amax)
    # ref test_multi_tensor.py:398-406
    u = (torch.from_numpy(amax) / 448.0).view(torch.int32)
    exponent = u // 2 ** 23
    mantissa = u & 0x7FFFFF
    exponent += (((mantissa > 0) & (exponent != 0xFE)) & ~((exponent == 0) & (mantissa <= 0x400000))).to(torch.int32)
    assert np.array_equal(got, exponent.to(torch.uint8).numpy())


@pytest.mark.parametrize("n,h,w,start", SHARD_CASES)
def test_partial_cast_matches_reference_test_formula(n, h, w, start):
    inp = _shard(n, 12)
    full_bits = np.zeros(h * w, np.uint16)
    g = torch.Generator().manual_seed(13)
    full = torch.randn(h * w, generator=g).to(torch.bfloat16)
    full[start:start + n].copy_(inp)  # scales come from the WHOLE matrix, as after the all-reduce
    full_bits[:] = _bf16(full)
    ar, ac = oracle.mxfp8_partial_amax(full_bits, h, w, 0)
    sr, sc = oracle.e8m0_from_amax(ar), oracle.e8m0_from_amax(ac)
    got_r, got_c = oracle.mxfp8_partial_cast(_bf16(inp), sr, sc, h, w, start)
    # ref test_partial_cast.py:32-53
    row_scale = ((254 - torch.from_numpy(sr).int()) * 2 ** 23).view(torch.float32)[:h, :w // 32].contiguous()
    col_scale = ((254 - torch.from_numpy(sc).int()) * 2 ** 23).view(torch.float32)[:h // 32, :w].contiguous()
    f = full.float()
    scaled = (f.view(-1, 32) * row_scale.view(-1, 1)).view(-1)
    ref_r = scaled[start:start + n].to(torch.float8_e4m3fn).view(torch.uint8).numpy()
    scaled = (f.view(h // 32, 32, w) * col_scale.view(h // 32, 1, w)).view(-1)
    ref_c = scaled[start:start + n].to(torch.float8_e4m3fn).view(torch.uint8).numpy()
    assert np.array_equal(got_r, ref_r)
    assert np.array_equal(got_c, ref_c)


@pytest.mark.parametrize("fmt", [oracle.E4M3, oracle.E5M2])
@pytest.mark.parametrize("pow2", [False, True])
@pytest.mark.parametrize("eps", [0.0, 100.0])
def test_scale_and_scale_inv(fmt, pow2, eps):
    rng = np.random.default_rng(5)
    amax = np.abs(rng.standard_normal(777)).astype(np.float32)
    amax[:6] = [0.0, 1e-30, 3e38, np.inf, np.nan, 448.0]
    scale, sinv = oracle.scale_and_scale_inv(amax, fmt, pow2, eps)
    for a, s, si in zip(amax, scale, sinv):
        want = oracle.scale_from_amax(float(a), fmt, pow2, eps)  # pinned in test_oracle_pinning.py
        assert s.view(np.uint32) == want.view(np.uint32)
        assert si.view(np.uint32) == (np.float32(1.0) / want).view(np.uint32)
    ref = "/root/reference/tests/pytorch/references/quantize_scale_calc.py"
    if os.path.exists(ref):  # not on the GPU box
        spec = importlib.util.spec_from_file_location("quantize_scale_calc", ref)
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        qd = torch.float8_e4m3fn if fmt == oracle.E4M3 else torch.float8_e5m2
        fin = np.isfinite(amax)
        s_ref, si_ref, _ = mod.scale_from_amax_tensor(torch.float32, torch.from_numpy(amax[fin]), qd, eps=eps,
                                                      pow_2_scales=pow2)
        assert np.array_equal(scale[fin], s_ref.numpy())
        assert np.array_equal(sinv[fin], si_ref.numpy())


@pytest.mark.parametrize("mode", [0, 1])
def test_adam_fp8_against_torch_optim(mode):
    torch.manual_seed(3)
    n, steps = 4099, 6
    p0 = torch.randn(n)
    kw = dict(lr=1e-2, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.05)
    ref_p = torch.nn.Parameter(p0.clone())
    opt = (torch.optim.Adam if mode == 0 else torch.optim.AdamW)([ref_p], **kw)
    pm, m, v = p0.numpy().copy(), np.zeros(n, np.float32), np.zeros(n, np.float32)
    scale = 448.0 / float(p0.abs().max()) / 2
    for step in range(1, steps + 1):
        g = torch.randn(n)
        ref_p.grad = g.clone()
        opt.step()
        p8, pm, m, v, amax = oracle.adam_fp8(g.numpy(), pm, m, v, scale, oracle.E4M3, lr=kw["lr"], beta1=0.9,
                                             beta2=0.999, eps=kw["eps"], step=step, mode=mode, bias_correction=1,
                                             weight_decay=kw["weight_decay"])
        # ref test_fused_optimizer.py:486 (assert_close defaults for fp32)
        torch.testing.assert_close(torch.from_numpy(pm), ref_p.detach())
        assert amax == np.abs(pm).max()
        # ref :488-489, with the E4M3 half-ulp written out
        deq = oracle.fp8_to_float(p8, oracle.E4M3) / np.float32(scale)
        torch.testing.assert_close(torch.from_numpy(deq), ref_p.detach(), rtol=1e-3 + 2 ** -4, atol=1e-3)
        assert np.array_equal(p8, oracle.float_to_fp8(pm * np.float32(scale), oracle.E4M3))
