"""CPU-only checks of the Python binding's handle caches (transformer_engine_amd/tensor.py): a plain tensor's
NVTETensor handle is shared per (address, dtype, shape), what wrap_plain returns keeps a temporary alive, the
cache is bounded, and the MXFP8 module states choose the scale layout the way the reference's modules do
(ref pytorch/module/linear.py:2067-2078, module/base.py:1211-1229).  No compute call is made."""
import gc
import weakref

import torch

from transformer_engine_amd import tensor as T
from transformer_engine_amd._lib import api


def _shape_of(handle):
    s = api.nvte_tensor_shape(handle)
    return tuple(s.data[i] for i in range(s.ndim))


def test_wrap_plain_shares_one_handle_per_address_dtype_shape():
    a = torch.zeros(4, 6)
    w1, w2 = T.wrap_plain(a), T.wrap_plain(a)
    assert w1.handle == w2.handle
    assert _shape_of(w1.handle) == (4, 6)
    assert api.nvte_tensor_data(w1.handle) == a.data_ptr()
    w3 = T.wrap_plain(a.view(6, 4))                 # same memory, another shape: its own handle
    assert w3.handle != w1.handle and _shape_of(w3.handle) == (6, 4)
    w1.destroy()                                    # callers' destroy() leaves the shared handle usable
    assert _shape_of(w2.handle) == (4, 6)


def test_wrap_plain_keeps_a_temporary_alive():
    t = torch.ones(3, 5)
    ref = weakref.ref(t)
    w = T.wrap_plain(t)
    del t
    gc.collect()
    assert ref() is not None                        # what the handle points at is still allocated
    w.destroy()
    del w
    gc.collect()
    assert ref() is None


def test_wrap_plain_cache_is_bounded():
    keep = [torch.zeros(2, 2 + i) for i in range(T._PLAIN_MAX + 40)]
    for k in keep:
        T.wrap_plain(k)
    assert len(T._PLAIN) <= T._PLAIN_MAX
    w = T.wrap_plain(keep[0])                       # evicted entries are rebuilt on demand
    assert _shape_of(w.handle) == (2, 2)


def test_mxfp8_states_pick_the_scale_layout_like_the_reference_modules():
    from transformer_engine_amd import linear as L
    from transformer_engine_amd.grouped_linear import GroupedLinearFP8State
    from transformer_engine_amd.recipe import MXFP8BlockScaling
    cpu = torch.device("cpu")
    st = L.LinearFP8State(MXFP8BlockScaling(), device=cpu)
    assert st.fwd_q[L.FWD_INPUT].optimize_for_gemm and st.fwd_q[L.FWD_WEIGHT].optimize_for_gemm
    assert st.bwd_q[L.BWD_GRAD_OUTPUT].optimize_for_gemm
    st = L.LinearFP8State(MXFP8BlockScaling(), device=cpu, compact_mx_scales=("input",))     # column + SP, norm
    assert not st.fwd_q[L.FWD_INPUT].optimize_for_gemm and st.fwd_q[L.FWD_WEIGHT].optimize_for_gemm
    st = L.LinearFP8State(MXFP8BlockScaling(), device=cpu, compact_mx_scales=("grad_output",))   # row + SP
    assert not st.bwd_q[L.BWD_GRAD_OUTPUT].optimize_for_gemm and st.fwd_q[L.FWD_INPUT].optimize_for_gemm
    g = GroupedLinearFP8State(MXFP8BlockScaling(), 4, cpu)
    assert g.q_input.optimize_for_gemm and g.q_weight.optimize_for_gemm and g.q_grad.optimize_for_gemm


def test_wrap_plain_handles_are_read_only():
    import pytest
    """wrap_plain hands every caller the SAME cached handle for one (address, dtype, shape): a setter through
    one reference would silently change it under every other holder (ADVICE r2), so setters are refused."""
    from transformer_engine_amd import tensor as T
    import inspect
    src = inspect.getsource(T._PlainRef.__getattr__)
    assert "set_" in src and "clear_" in src

    class _W:
        handle = 1
        def set_amax(self, *a): raise AssertionError("reached the shared handle")
        def shape(self): return (2, 3)
    ref = T._PlainRef(_W(), None)
    assert ref.shape() == (2, 3)
    with pytest.raises(AttributeError, match="read-only"):
        ref.set_amax(None)
    with pytest.raises(AttributeError, match="read-only"):
        ref.clear_amax()
