#!/usr/bin/env python3
"""NV12 frames -> network input, the MI355X way: ONE launch per batch instead of the chain of the
reference's samples (reference tests/test_TorchSegmentation.py:176-240: NV12 -> RGB -> RGB_32F ->
RGB_32F_PLANAR, then torch.divide and torchvision Normalize).

    python examples/preprocess_for_inference.py [raw_nv12_file width height]

Without arguments it runs on synthetic 1080p frames.  The result is handed to torch through
DLPack without a copy (the Surface keeps owning the memory, as in the reference)."""
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import python_vali as vali  # noqa: E402  (the reference's module name; same objects as vali_amd)

import torch  # noqa: E402


def main():
    gpu_id, batch, net_w, net_h = 0, 8, 640, 384
    if len(sys.argv) == 4:
        dec = vali.PyDecoder(input=sys.argv[1], opts={"video_size": f"{sys.argv[2]}x{sys.argv[3]}", "pixel_format": "nv12"},
                             gpu_id=gpu_id)
        w, h = dec.Width, dec.Height
        frames = [vali.Surface.Make(vali.PixelFormat.NV12, w, h, gpu_id) for _ in range(batch)]
        n = 0
        for f in frames:
            ok, _ = dec.DecodeSingleSurface(f)
            if not ok:
                break
            n += 1
        frames = frames[:n]
    else:
        w, h = 1920, 1080
        rng = np.random.default_rng(0)
        up = vali.PyFrameUploader(gpu_id)
        frames = [vali.Surface.Make(vali.PixelFormat.NV12, w, h, gpu_id) for _ in range(batch)]
        for f in frames:
            assert up.Run(rng.integers(16, 236, w * h * 3 // 2, dtype=np.uint8), f)[0]

    inputs = [vali.Surface.Make(vali.PixelFormat.RGB_32F_PLANAR, net_w, net_h, gpu_id) for _ in frames]
    pre = vali.PySurfacePreprocessor(gpu_id, mean=(0.485, 0.456, 0.406), std=(0.229, 0.224, 0.225), div=1.0)
    cc = vali.ColorspaceConversionContext(vali.ColorSpace.BT_709, vali.ColorRange.MPEG)
    ok, info = pre.RunBatch(frames, inputs, cc)          # resize + colour conversion + /255 + normalise
    assert ok, info
    tensors = [torch.from_dlpack(s) for s in inputs]     # zero-copy views, shape (3 * H, W) each
    batch_tensor = torch.stack([t.view(3, net_h, net_w) for t in tensors])
    print("network input:", tuple(batch_tensor.shape), batch_tensor.dtype, batch_tensor.device,
          "mean %.4f std %.4f" % (batch_tensor.mean().item(), batch_tensor.std().item()))


if __name__ == "__main__":
    main()
