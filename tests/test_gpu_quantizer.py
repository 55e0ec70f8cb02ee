"""Pins v_cvt_pk_u8_f32 (the one-instruction quantiser) to round-half-even + saturate."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def run(vali, gpu, fn, values):
    import torch

    x = torch.from_numpy(values).to("cuda")
    y = torch.zeros(values.size, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    assert fn(x.data_ptr(), y.data_ptr(), values.size, 0) == 0
    torch.cuda.synchronize()
    return y.cpu().numpy()


def test_quantizer_matches_oracle(vali, gpu, oracle):
    from vali_amd._native import shim

    halves = np.arange(-4, 520, dtype=np.float32) * 0.5            # every tie in range
    dense = np.linspace(-3.0, 259.0, 200001, dtype=np.float32)
    near = np.nextafter(halves, np.float32(np.inf)), np.nextafter(halves, np.float32(-np.inf))
    special = np.array([-1e30, 1e30, -0.0, 0.0, 254.5, 255.5, 1e-30, np.inf, -np.inf], np.float32)
    values = np.concatenate([halves, dense, near[0], near[1], special]).astype(np.float32)
    want = np.clip(np.rint(values.astype(np.float64)), 0, 255).astype(np.uint8)
    assert np.array_equal(oracle.q_u8(values[:2000]), want[:2000])  # oracle == numpy model
    got_hw = run(vali, gpu, shim.debug_quantize_u8, values)
    got_portable = run(vali, gpu, shim.debug_quantize_u8_portable, values)
    assert np.array_equal(got_portable, want)
    assert np.array_equal(got_hw, want), "v_cvt_pk_u8_f32 is not RNE+saturate on this chip"
