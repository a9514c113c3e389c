"""Shared helpers for the parity tests (torch <-> numpy views, seeded inputs)."""
import numpy as np
import torch

import oracle
from transformer_engine_amd.tensor import DType

FMT = {DType.kFloat8E4M3: oracle.E4M3, DType.kFloat8E5M2: oracle.E5M2}


def to_oracle_input(t: torch.Tensor) -> np.ndarray:
    """torch (cpu or cuda) high-precision tensor -> numpy array the oracle accepts."""
    t = t.detach().cpu().contiguous()
    if t.dtype == torch.bfloat16:
        return t.view(torch.int16).numpy().view(np.uint16)
    if t.dtype == torch.float16:
        return t.numpy()
    return t.float().numpy()


def uniform_input(shape, dtype, seed, lo=-2.0, hi=1.0, device="cuda"):
    """The reference's C++ tests fill inputs with U[-2, 1) (ref tests/cpp/test_common.cu:874-897)."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    x = torch.rand(shape, generator=g, dtype=torch.float32) * (hi - lo) + lo
    return x.to(dtype).to(device)


def randn_input(shape, dtype, seed, device="cuda", scale=1.0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return (torch.randn(shape, generator=g, dtype=torch.float32) * scale).to(dtype).to(device)


def u8(t: torch.Tensor) -> np.ndarray:
    return t.detach().cpu().contiguous().numpy()
