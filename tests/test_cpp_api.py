"""The header-only C++ layer (include/transformer_engine/cxx_wrappers.h, gemm.h wrappers, runtime.h):
a translation unit written against the reference's C++ API (ref transformer_engine.h:670-1515,
gemm.h:490-640, common/util/cuda_runtime.h) is compiled with plain g++ against include/ and linked with
libtransformer_engine.so - tests/cpp/wrapper_smoke.cpp.  CPU: compile + host-side checks.  GPU: the
prebuilt binary quantizes / dequantizes / multiplies through the wrappers and checks the results exactly."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CPP = os.path.join(ROOT, "tests", "cpp")
BIN = os.path.join(CPP, "wrapper_smoke")


def test_cpp_caller_compiles_links_and_runs_host_side():
    subprocess.check_call(["make", "-C", CPP])           # g++ only, a few seconds, no GPU
    out = subprocess.run([BIN], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    assert out.stdout.strip() in ("HOST-ONLY OK", "GPU OK")


def test_version_script_exports_the_cxx_helpers():
    lib = os.path.join(ROOT, "transformer_engine_amd", "lib", "libtransformer_engine.so")
    syms = subprocess.run(["nm", "-DC", "--defined-only", lib], capture_output=True, text=True).stdout
    for name in ("transformer_engine::cuda::sm_count(int)", "transformer_engine::cuda::sm_arch(int)",
                 "transformer_engine::cuda::current_device()", "transformer_engine::cuda::supports_multicast(int)",
                 "transformer_engine::cuda::stream_priority_range(int*, int*, int)",
                 "transformer_engine::cuda::cublas_version()",
                 "transformer_engine::typeToSize(transformer_engine::DType)",
                 "transformer_engine::typeToNumBits(transformer_engine::DType)",
                 "nvte_delayed_scaling_recipe_amax_and_scale_update_after_reduction("):
        assert name in syms, name
    # nothing of the library's own namespace leaks
    assert " te::" not in syms


@pytest.mark.gpu
def test_cpp_caller_runs_on_the_gpu():
    assert os.path.exists(BIN), "tests/cpp/wrapper_smoke is built by __graft_entry__.build()"
    out = subprocess.run([BIN], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    assert out.stdout.strip() == "GPU OK"
