"""The float stage of the fused pre-processing definition, pinned against torch itself.

PySurfacePreprocessor is defined as the chain of the reference's samples
(tests/test_TorchSegmentation.py:228-240):  x = tensor / 255 (nppiScale_8u32f_C3R) ;
torch.divide(x, 255.0) ; torchvision Normalize = tensor.sub_(mean[:, None, None]).div_(std[:, None, None]).
The GPU tests restate that stage with float32 numpy; here the numpy restatement is checked
bit for bit against the torch operators (CPU), for every u8 value and channel -- which is also
exactly the 3 x 256 table the kernel builds."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")

MEAN, STD = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)


@pytest.mark.parametrize("div", [255.0, 1.0, 2.0])
def test_numpy_float_stage_equals_torch(div):
    q = np.arange(256, dtype=np.uint8)
    x = np.broadcast_to(q[None, :, None], (3, 256, 4)).astype(np.float32) / np.float32(255.0)
    want_np = ((x / np.float32(div)) - np.asarray(MEAN, np.float32)[:, None, None]) / np.asarray(STD, np.float32)[:, None, None]
    t = torch.from_numpy(np.ascontiguousarray(x)).clone()
    t = torch.divide(t, div)
    mean = torch.as_tensor(MEAN, dtype=t.dtype).view(-1, 1, 1)
    std = torch.as_tensor(STD, dtype=t.dtype).view(-1, 1, 1)
    t = t.sub_(mean).div_(std)                       # torchvision.transforms.functional.normalize
    assert np.array_equal(want_np.astype(np.float32).view(np.uint32), t.numpy().view(np.uint32))


def test_u8_to_float_is_division_by_255():
    """nppiScale_8u32f_C3R(0, 1): v / 255 -- and torch agrees with numpy on that division."""
    q = torch.arange(256, dtype=torch.uint8)
    a = (q.to(torch.float32) / 255.0).numpy()
    b = np.arange(256, dtype=np.float32) / np.float32(255.0)
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
