"""A descriptor with a shape but no data pointer must end as an NVTE error in the launcher, never as a
kernel dereferencing null (the reference's has_data() is `dptr != nullptr`, common/common.h:269)."""
import pytest
import torch

from transformer_engine_amd._lib import NVTEError, api
from transformer_engine_amd.tensor import DType, Float8Quantizer, TensorWrapper, current_stream_ptr, wrap_plain

pytestmark = pytest.mark.gpu


def test_shape_only_descriptors_are_rejected_before_launch():
    x = torch.randn(128, 256, device="cuda", dtype=torch.bfloat16)
    q = Float8Quantizer(torch.ones(1, device="cuda"), torch.zeros(1, device="cuda"), DType.kFloat8E4M3)
    out = q.make_empty(x.shape, dtype=x.dtype, device=x.device)
    stream = current_stream_ptr()
    # input without data
    wi = TensorWrapper().set_shape_only(0, DType.kBFloat16, (128, 256))
    wo = q._wrap_out(out)
    wo.set_scale(q.scale).set_amax(q.amax)
    with pytest.raises(NVTEError, match="no data pointer"):
        api.nvte_quantize(wi.handle, wo.handle, stream)
    # rowwise-only cast, output without data
    wi2 = wrap_plain(x)
    wo2 = TensorWrapper().set_shape_only(0, DType.kFloat8E4M3, (128, 256))
    wo2.set_rowwise_scale_inv(out._scale_inv).set_scale(q.scale)
    with pytest.raises(NVTEError, match="no data pointer"):
        api.nvte_quantize(wi2.handle, wo2.handle, stream)
    # activation on a shape-only input
    wo3 = wrap_plain(torch.empty(128, 256, device="cuda", dtype=torch.bfloat16))
    with pytest.raises(NVTEError, match="no data pointer"):
        api.nvte_gelu(wi.handle, wo3.handle, stream)
    torch.cuda.synchronize()   # nothing was launched, nothing faulted
    for w in (wi, wo, wi2, wo2, wo3):
        w.destroy()
