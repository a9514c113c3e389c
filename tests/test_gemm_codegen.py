"""Codegen guards for gemm_fp8_kernel (CPU only: hipcc cross-compiles ONE instantiation to assembly, ~7 s each).

What they pin (all three were met as silent 2-7 % or correctness hazards):
  * the steady-state K loop holds exactly ONE vector-memory wait, the kernel's own `vmcnt(0)` at the K-tile hand-over - no
    compiler-inserted `s_waitcnt vmcnt` in front of an MFMA or an LDS access (a conditional global load left pending,
    or the compiler's LDS-DMA aliasing rule, both put one there);
  * no scratch traffic in that loop and no spilled vector registers in the dense kernels;
  * where the K loop issues LDS-DMA from inline assembly WITHOUT padding for VALU-written SGPR operands (dense per-tensor
    K-major variants), the loop contains no VALU instruction that writes an SGPR at all."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "transformer_engine_amd", "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")

pytestmark = pytest.mark.skipif(shutil.which(HIPCC) is None and not os.path.exists(HIPCC), reason="hipcc not available")


def _asm_of(variant: str, tmp_path) -> str:
    out = os.path.join(str(tmp_path), "k.s")
    cmd = [HIPCC, "-std=c++17", "-O3", "-fPIC", "--offload-arch=gfx950", "-I../../include", "-I.", "-fno-gpu-rdc",
           "-DTE_GEMM_PART=0", f"-DTE_GEMM_DEV_ONE={variant}", "--cuda-device-only", "-S", "gemm_fp8.hip", "-o", out]
    subprocess.run(cmd, cwd=CSRC, check=True, capture_output=True, timeout=600)
    with open(out) as fh:
        return fh.read()


def _k_loop(asm: str) -> str:
    """The basic block with the most MFMAs = one K tile of the steady-state loop."""
    blocks = re.split(r"\n(?=\.LBB[0-9_]+:)", asm)
    return max(blocks, key=lambda b: b.count("v_mfma"))


def _instructions(block: str):
    return [l.strip() for l in block.split("\n") if l.strip() and not l.strip().startswith((".", ";"))]


# (MX, EPI, GROUPED, KM) as the kernel's template arguments
VARIANTS = {
    "dense per-tensor": "false,false,false,0",
    "dense MXFP8": "true,false,false,0",
    "per-tensor, Q K-major (dgrad)": "false,false,false,2",
    "per-tensor, P and Q K-major (wgrad)": "false,false,false,3",
    "MXFP8, P and Q K-major, compact scales": "true,false,false,15",
}


@pytest.mark.parametrize("name", list(VARIANTS))
def test_k_loop_has_no_compiler_vmem_wait_and_no_scratch(name, tmp_path):
    asm = _asm_of(VARIANTS[name], tmp_path)
    loop = _k_loop(asm)
    assert loop.count("v_mfma") == 16, "the K-tile loop block was not found"
    waits = [i for i in _instructions(loop) if i.startswith("s_waitcnt") and "vmcnt" in i]
    assert waits == ["s_waitcnt vmcnt(0) lgkmcnt(0)"], f"{name}: vector-memory waits in the K loop: {waits}"
    assert "scratch_" not in loop, f"{name}: scratch traffic in the K loop"
    spills = int(re.search(r"\.vgpr_spill_count:\s+(\d+)", asm).group(1))
    assert spills == 0, f"{name}: {spills} spilled VGPRs"


@pytest.mark.parametrize("name", ["per-tensor, Q K-major (dgrad)", "per-tensor, P and Q K-major (wgrad)"])
def test_bare_asm_dma_in_k_loop_sees_no_valu_written_sgpr(name, tmp_path):
    loop = _k_loop(_asm_of(VARIANTS[name], tmp_path))
    ins = _instructions(loop)
    assert sum(1 for i in ins if i.startswith("buffer_load_dwordx4") and i.endswith("lds")) == 8
    valu_sgpr = [i for i in ins if i.startswith(("v_readlane", "v_readfirstlane")) or re.match(r"v_cmpx?_\w+\s+s\[", i)
                 or re.match(r"v_(add|sub|addc|subb)\w*co\w*\s", i)]
    assert not valu_sgpr, (f"{name}: a VALU instruction writes an SGPR inside the K loop, where the LDS-DMA statements carry "
                           f"no wait states for it (glds_b128_pair<BARE>): {valu_sgpr}")
