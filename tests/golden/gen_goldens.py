#!/usr/bin/env python
"""Generates the committed golden vectors under tests/golden/ FROM THE REFERENCE.

Run in the authoring container only (needs /root/reference and torch CPU):
    python tests/golden/gen_goldens.py

1. scale_from_amax.json  - outputs of the reference's OWN pure-torch function
   tests/pytorch/references/quantize_scale_calc.py::scale_from_amax_tensor, imported by file
   path (it needs nothing but torch).  Pins oracle.scale_from_amax (and through it
   nvte_compute_scale_from_amax).
2. current_scaling_128x256.npz - the byte-exact recipe of the reference's
   tests/pytorch/test_float8_current_scaling_exact.py:786-809 (input randn(128,256) bf16, seed
   123): amax -> scale_from_amax_tensor -> (x.float() * scale).clamp(+-fp8_max) -> torch fp8 cast,
   i.e. CurrentScalingQuantizerRef's math (custom_recipes/quantization_ref_current_scaling.py:
   155-260) evaluated with torch CPU.  Holds inputs + expected bytes + scale_inv.
3. fp8_decode_tables.npz - torch's decode of all 256 E4M3 / E5M2 codes.
"""
import importlib.util
import json
import os

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/tests/pytorch/references/quantize_scale_calc.py"


def load_ref():
    spec = importlib.util.spec_from_file_location("quantize_scale_calc", REF)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.scale_from_amax_tensor


def main():
    fn = load_ref()
    amaxes = [0.0, 1e-30, 1e-12, 6.1e-5, 1e-3, 0.3, 1.0, 1.7, 2.0, 3.999, 448.0, 449.0, 1000.0, 57344.0,
              65504.0, 1e10, 3.0e38, 7.5e-37, 1.2e-38, 1e-45]
    rows = []
    for qd, name in ((torch.float8_e4m3fn, "e4m3"), (torch.float8_e5m2, "e5m2")):
        for pow2 in (False, True):
            for eps in (0.0, 1e-4):
                for a in amaxes:
                    amax = torch.tensor([a], dtype=torch.float32)
                    scale, scale_inv, amax_out = fn(torch.float32, amax, qd, eps=eps, pow_2_scales=pow2)
                    rows.append({"fp8": name, "pow2": pow2, "eps": eps,
                                 "amax_bits": int(amax.view(torch.int32).item()) & 0xFFFFFFFF,
                                 "scale_bits": int(scale.view(torch.int32).item()) & 0xFFFFFFFF,
                                 "scale_inv_bits": int(scale_inv.view(torch.int32).item()) & 0xFFFFFFFF})
    with open(os.path.join(HERE, "scale_from_amax.json"), "w") as f:
        json.dump({"source": "reference tests/pytorch/references/quantize_scale_calc.py::scale_from_amax_tensor "
                             "(x_dtype=float32)", "rows": rows}, f, indent=0)

    torch.manual_seed(123)
    x = torch.randn(128, 256, dtype=torch.float32).to(torch.bfloat16)
    out = {"x_bf16_bits": x.view(torch.int16).numpy().view(np.uint16)}
    for qd, name in ((torch.float8_e4m3fn, "e4m3"), (torch.float8_e5m2, "e5m2")):
        for pow2 in (False, True):
            amax = x.float().abs().max().reshape(1)
            scale, scale_inv, _ = fn(torch.float32, amax, qd, eps=0.0, pow_2_scales=pow2)
            fmax = torch.finfo(qd).max
            q = (x.float() * scale).clamp(-fmax, fmax).to(qd)
            tag = f"{name}_{'pow2' if pow2 else 'std'}"
            out[f"q_{tag}"] = q.view(torch.uint8).numpy()
            out[f"qt_{tag}"] = q.t().contiguous().view(torch.uint8).numpy()
            out[f"scale_{tag}"] = scale.numpy()
            out[f"scale_inv_{tag}"] = scale_inv.numpy()
            out[f"amax_{tag}"] = amax.numpy()
    np.savez_compressed(os.path.join(HERE, "current_scaling_128x256.npz"), **out)

    codes = torch.arange(256, dtype=torch.int32).to(torch.uint8)
    np.savez_compressed(os.path.join(HERE, "fp8_decode_tables.npz"),
                        e4m3=codes.view(torch.float8_e4m3fn).float().numpy(),
                        e5m2=codes.view(torch.float8_e5m2).float().numpy())
    print("wrote goldens to", HERE)


if __name__ == "__main__":
    main()
