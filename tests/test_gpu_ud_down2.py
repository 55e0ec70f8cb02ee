"""UD with an exact 2x horizontal downscale (source width == 2 x output width) and an 8-bit output
runs on its own kernels (k_ud_lean / k_ud_half -- round 1: k_ud_down2 --, and k_ud_down2_t / k_ud_half_t for the 90/270 degree outputs: 16-byte loads,
no coordinate divisions; BASELINE config 4's "2x downsample"; float outputs stay on the general
kernel).  It must be bit-identical to the oracle -- which restates the general texture-filter
arithmetic -- and to the general kernel, for every output format, any height, ragged widths, the
clamped first column, foreign unaligned memory, batches and the rotated outputs."""
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import make_nv12
from test_gpu_edge_geometry import download, foreign_nv12

pytestmark = pytest.mark.gpu

DSTS = ["YUV444", "RGB", "RGB_PLANAR", "RGB_32F", "RGB_32F_PLANAR"]
GEOMS = [(3840, 2160, 1920, 1080),   # config 4
         (1280, 720, 640, 360),      # the reference's golden geometry (tests/test_PySurfaceUD.py)
         (848, 464, 424, 232), (848, 464, 424, 301), (848, 464, 424, 1000),   # any height
         (1002, 500, 501, 250), (1006, 38, 503, 19), (518, 64, 259, 77),      # ragged: 501 = 4 k + 1, ...
         (8, 4, 4, 2), (4, 4, 2, 2), (2, 2, 1, 1), (10, 6, 5, 3),             # tiny (sw < 16: byte path)
         (2048, 16, 1024, 5), (512, 2, 256, 1), (1024, 64, 512, 32), (1040, 20, 520, 10),
         (2064, 40, 1032, 33), (1030, 24, 515, 12)]                           # one full wave + a ragged second one


@pytest.mark.parametrize("dst", DSTS)
@pytest.mark.parametrize("geom", GEOMS)
def test_down2_bit_exact(vali, gpu, oracle, dst, geom):
    sw, sh, dw, dh = geom
    assert sw == 2 * dw
    nv = make_nv12(sw, sh, 31)
    src = vali.Surface.Make(vali.NV12, sw, sh, gpu)
    assert vali.PyFrameUploader(gpu).Run(nv.reshape(-1), src)[0]
    out = vali.Surface.Make(vali.PixelFormat[dst], dw, dh, gpu)
    assert vali.PySurfaceUD(gpu).Run(src, out) == (True, vali.TaskExecInfo.SUCCESS)
    dt = np.float32 if "32F" in dst else np.uint8
    want = oracle.ud_nv12(nv, sw, sh, "NV12", dw, dh, dst).reshape(-1)
    got = download(vali, gpu, out, dt)
    assert np.array_equal(got.view(np.uint8), want.view(np.uint8))


def test_down2_random_frames_and_extreme_values(vali, gpu, oracle):
    rng = np.random.default_rng(77)
    for sw, sh, dw, dh in ((640, 360, 320, 180), (258, 130, 129, 97)):
        for fill in ("random", 0, 255):
            nv = (rng.integers(0, 256, (sh * 3 // 2, sw), dtype=np.uint8) if fill == "random"
                  else np.full((sh * 3 // 2, sw), fill, np.uint8))
            src = vali.Surface.Make(vali.NV12, sw, sh, gpu)
            assert vali.PyFrameUploader(gpu).Run(nv.reshape(-1), src)[0]
            for dst in ("RGB", "YUV444", "RGB_32F"):
                out = vali.Surface.Make(vali.PixelFormat[dst], dw, dh, gpu)
                assert vali.PySurfaceUD(gpu).Run(src, out)[0]
                dt = np.float32 if "32F" in dst else np.uint8
                want = oracle.ud_nv12(nv, sw, sh, "NV12", dw, dh, dst).reshape(-1)
                assert np.array_equal(download(vali, gpu, out, dt).view(np.uint8), want.view(np.uint8))


@pytest.mark.parametrize("pad,skew", [(3, 0), (16, 5), (8, 0), (24, 8), (14, 0), (30, 16)])
def test_down2_foreign_memory(vali, gpu, oracle, pad, skew):
    """rows / bases that are not 16-byte aligned take the kernel's byte path; (16, 0) stays on the
    vector path with a foreign pitch."""
    sw, sh, dw, dh = 322, 180, 161, 120
    nv = make_nv12(sw, sh, 5)
    src, keep = foreign_nv12(vali, sw, sh, nv, pad, skew)
    for dst in ("RGB", "RGB_PLANAR"):
        out = vali.Surface.Make(vali.PixelFormat[dst], dw, dh, gpu)
        assert vali.PySurfaceUD(gpu).Run(src, out)[0]
        assert np.array_equal(download(vali, gpu, out), oracle.ud_nv12(nv, sw, sh, "NV12", dw, dh, dst).reshape(-1))
    del keep


@pytest.mark.parametrize("angle", [90.0, 180.0, 270.0])
@pytest.mark.parametrize("geom", [(3840, 2160, 1920, 1080), (1002, 500, 501, 333), (70, 48, 35, 5), (8, 8, 4, 4),
                                  (1040, 200, 520, 100), (144, 24, 72, 12), (2064, 520, 1032, 260)])   # exactly 2:1 both ways: the 64 x 64 tile form
def test_down2_rotated(vali, gpu, oracle, angle, geom):
    sw, sh, uw, uh = geom
    nv = make_nv12(sw, sh, 23)
    src = vali.Surface.Make(vali.NV12, sw, sh, gpu)
    assert vali.PyFrameUploader(gpu).Run(nv.reshape(-1), src)[0]
    k = int(round(angle / 90.0)) % 4
    dw, dh = (uh, uw) if k & 1 else (uw, uh)
    fused = vali.Surface.Make(vali.RGB, dw, dh, gpu)
    assert vali.PySurfaceUD(gpu).RunRotated(src, fused, angle) == (True, vali.TaskExecInfo.SUCCESS)
    want = np.rot90(oracle.ud_nv12(nv, sw, sh, "NV12", uw, uh, "RGB").reshape(uh, uw, 3), k=k)
    assert np.array_equal(download(vali, gpu, fused).reshape(dh, dw, 3), want)


def test_down2_batch(vali, gpu, oracle):
    sw, sh, dw, dh, n = 1920, 1080, 960, 540, 6
    nvs = [make_nv12(sw, sh, 40 + i) for i in range(n)]
    srcs = [vali.Surface.Make(vali.NV12, sw, sh, gpu) for _ in range(n)]
    for s_, nv in zip(srcs, nvs):
        assert vali.PyFrameUploader(gpu).Run(nv.reshape(-1), s_)[0]
    dsts = [vali.Surface.Make(vali.RGB, dw, dh, gpu) for _ in range(n)]
    assert vali.PySurfaceUD(gpu).RunBatch(srcs, dsts) == (True, vali.TaskExecInfo.SUCCESS)
    for d, nv in zip(dsts, nvs):
        assert np.array_equal(download(vali, gpu, d), oracle.ud_nv12(nv, sw, sh, "NV12", dw, dh, "RGB").reshape(-1))


def test_down2_equals_the_general_kernel():
    """A/B in a child process with VALI_UD_DOWN2=0 (the switch is read once per process): the two
    kernels agree byte for byte on a frame, plain and rotated."""
    code = r"""
import sys, numpy as np
sys.path.insert(0, %r)
import torch, vali_amd as vali
rng = np.random.default_rng(3)
sw, sh, dw, dh = 1280, 720, 640, 407
nv = rng.integers(0, 256, sw * sh * 3 // 2, dtype=np.uint8)
src = vali.Surface.Make(vali.NV12, sw, sh, 0)
assert vali.PyFrameUploader(0).Run(nv, src)[0]
import hashlib
h = hashlib.sha256()
for fmt in (vali.RGB, vali.YUV444, vali.RGB_32F_PLANAR):
    out = vali.Surface.Make(fmt, dw, dh, 0)
    assert vali.PySurfaceUD(0).Run(src, out)[0]
    buf = np.zeros(out.HostSize, np.uint8)
    assert vali.PySurfaceDownloader(0).Run(out, buf)[0]
    h.update(buf.tobytes())
rot = vali.Surface.Make(vali.RGB, dh, dw, 0)
assert vali.PySurfaceUD(0).RunRotated(src, rot, 90.0)[0]
buf = np.zeros(rot.HostSize, np.uint8)
assert vali.PySurfaceDownloader(0).Run(rot, buf)[0]
h.update(buf.tobytes())
print("DIGEST", h.hexdigest())
""" % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    digests = []
    for flag in ("1", "0"):
        env = dict(os.environ, VALI_UD_DOWN2=flag)
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        digests.append([l for l in r.stdout.splitlines() if l.startswith("DIGEST")][0])
    assert digests[0] == digests[1]


@pytest.mark.parametrize("pad,skew", [(5, 0), (16, 3), (16, 0), (4, 8)])
@pytest.mark.parametrize("angle", [0.0, 180.0])
def test_down2_foreign_destination(vali, gpu, oracle, angle, pad, skew):
    """packed RGB into borrowed memory whose rows are not 16-byte aligned: the kernel's LDS-strip
    stores give way to the general per-lane store, row by row."""
    import torch

    sw, sh, dw, dh = 1040, 66, 520, 33
    nv = make_nv12(sw, sh, 9)
    src = vali.Surface.Make(vali.NV12, sw, sh, gpu)
    assert vali.PyFrameUploader(gpu).Run(nv.reshape(-1), src)[0]
    pitch = dw * 3 + pad
    raw = torch.full((dh * pitch + 64,), 7, dtype=torch.uint8, device="cuda")
    view = raw[skew: skew + dh * pitch].view(dh, pitch)[:, : dw * 3]
    dst = vali.Surface.from_dlpack(torch.utils.dlpack.to_dlpack(view), vali.RGB)
    assert dst.Width == dw and dst.Height == dh and dst.Pitch == pitch
    ud = vali.PySurfaceUD(gpu)
    ok = ud.Run(src, dst) if angle == 0.0 else ud.RunRotated(src, dst, angle)
    assert ok == (True, vali.TaskExecInfo.SUCCESS)
    torch.cuda.synchronize()
    want = oracle.ud_nv12(nv, sw, sh, "NV12", dw, dh, "RGB").reshape(dh, dw, 3)
    if angle == 180.0:
        want = np.rot90(want, k=2)
    got = view.cpu().numpy().reshape(dh, dw, 3)
    assert np.array_equal(got, want)
    # nothing outside the rows was touched
    full = raw.cpu().numpy()
    pad_bytes = full[skew: skew + dh * pitch].reshape(dh, pitch)[:, dw * 3:]
    assert (pad_bytes == 7).all() and (full[:skew] == 7).all() and (full[skew + dh * pitch:] == 7).all()


def test_down2_random_geometries(vali, gpu, oracle):
    """60 random (width, height, output height, output, quarter turn) cases around the tile and
    wave boundaries of both kernels (512-column waves of k_ud_lean, 256 x 32 tiles of k_ud_down2_t)."""
    rng = np.random.default_rng(2024)
    ud = vali.PySurfaceUD(gpu)
    for case in range(60):
        dw = int(rng.choice([rng.integers(1, 40), rng.integers(250, 262), rng.integers(505, 520),
                             rng.integers(760, 775), rng.integers(1020, 1030), rng.integers(1, 1100)]))
        sw = 2 * dw
        sh = 2 * int(rng.integers(1, 70))
        dh = int(rng.choice([sh // 2, rng.integers(1, 80), rng.integers(30, 36)]))
        k = int(rng.integers(0, 4))
        dst = "RGB" if k else str(rng.choice(["RGB", "RGB_PLANAR", "YUV444"]))
        nv = rng.integers(0, 256, (sh * 3 // 2, sw), dtype=np.uint8)
        src = vali.Surface.Make(vali.NV12, sw, sh, gpu)
        assert vali.PyFrameUploader(gpu).Run(nv.reshape(-1), src)[0]
        ow, oh = (dh, dw) if k & 1 else (dw, dh)
        out = vali.Surface.Make(vali.PixelFormat[dst], ow, oh, gpu)
        ok = ud.RunRotated(src, out, 90.0 * k) if k else ud.Run(src, out)
        assert ok == (True, vali.TaskExecInfo.SUCCESS), (case, sw, sh, dw, dh, dst, k)
        want = oracle.ud_nv12(nv, sw, sh, "NV12", dw, dh, dst)
        if k:
            want = np.rot90(want.reshape(dh, dw, 3), k=k)
        got = download(vali, gpu, out)
        assert np.array_equal(got, np.ascontiguousarray(want).reshape(-1)), (case, sw, sh, dw, dh, dst, k)


# ---- the same kernel at a 1:1 width ratio (colour conversion with chroma interpolation) -------------
SAME_GEOMS = [(1920, 1080, 1080), (1280, 720, 720), (848, 464, 232), (848, 464, 1000), (1002, 500, 333),
              (518, 64, 77), (8, 4, 4), (6, 4, 2), (2, 2, 2), (10, 6, 3), (1024, 16, 5), (520, 20, 20), (1030, 24, 12)]


@pytest.mark.parametrize("dst", ["YUV444", "RGB", "RGB_PLANAR", "RGB_32F_PLANAR"])
@pytest.mark.parametrize("geom", SAME_GEOMS)
def test_same_width_bit_exact(vali, gpu, oracle, dst, geom):
    sw, sh, dh = geom
    nv = make_nv12(sw, sh, 37)
    src = vali.Surface.Make(vali.NV12, sw, sh, gpu)
    assert vali.PyFrameUploader(gpu).Run(nv.reshape(-1), src)[0]
    out = vali.Surface.Make(vali.PixelFormat[dst], sw, dh, gpu)
    assert vali.PySurfaceUD(gpu).Run(src, out) == (True, vali.TaskExecInfo.SUCCESS)
    dt = np.float32 if "32F" in dst else np.uint8
    want = oracle.ud_nv12(nv, sw, sh, "NV12", sw, dh, dst).reshape(-1)
    assert np.array_equal(download(vali, gpu, out, dt).view(np.uint8), want.view(np.uint8))


def test_same_width_random_geometries_rotation_and_foreign_memory(vali, gpu, oracle):
    rng = np.random.default_rng(4048)
    ud = vali.PySurfaceUD(gpu)
    for case in range(40):
        sw = 2 * int(rng.choice([rng.integers(1, 20), rng.integers(126, 132), rng.integers(254, 262),
                                 rng.integers(510, 518), rng.integers(1, 600)]))
        sh = 2 * int(rng.integers(1, 50))
        dh = int(rng.choice([sh, rng.integers(1, 60)]))
        k = int(rng.choice([0, 0, 2]))
        dst = "RGB" if k else str(rng.choice(["RGB", "RGB_PLANAR", "YUV444"]))
        nv = rng.integers(0, 256, (sh * 3 // 2, sw), dtype=np.uint8)
        if case % 5 == 4:
            src, keep = foreign_nv12(vali, sw, sh, nv, pad=int(rng.choice([3, 8, 13, 16])), skew=int(rng.choice([0, 5, 8])))
        else:
            src, keep = vali.Surface.Make(vali.NV12, sw, sh, gpu), None
            assert vali.PyFrameUploader(gpu).Run(nv.reshape(-1), src)[0]
        out = vali.Surface.Make(vali.PixelFormat[dst], sw, dh, gpu)
        ok = ud.RunRotated(src, out, 180.0) if k else ud.Run(src, out)
        assert ok == (True, vali.TaskExecInfo.SUCCESS), (case, sw, sh, dh, dst, k)
        want = oracle.ud_nv12(nv, sw, sh, "NV12", sw, dh, dst)
        if k:
            want = np.rot90(want.reshape(dh, sw, 3), k=2)
        assert np.array_equal(download(vali, gpu, out), np.ascontiguousarray(want).reshape(-1)), (case, sw, sh, dh, dst, k)
        del keep


# ---- exactly 3:2 both ways (1080p -> 720p ...): the dot-product kernel k_ud_32 ------------------------------------------
@pytest.mark.parametrize("dst", ["RGB", "RGB_PLANAR", "YUV444"])
@pytest.mark.parametrize("geom", [(1920, 1080, 1280, 720), (96, 48, 64, 32), (1536, 24, 1024, 16), (12, 6, 8, 4),
                                  (3840, 2160, 2560, 1440), (1548, 54, 1032, 36)])     # one full wave + a partial second one
def test_three_to_two_bit_exact(vali, gpu, oracle, dst, geom):
    sw, sh, dw, dh = geom
    assert 2 * sw == 3 * dw and 2 * sh == 3 * dh and dw % 8 == 0 and dh % 4 == 0
    for seed in (41, 42):
        nv = make_nv12(sw, sh, seed)
        if seed == 42:
            nv[:] = np.where(np.random.default_rng(1).random(nv.shape) < 0.5, 0, 255)     # extreme values: saturation both ways
        src = vali.Surface.Make(vali.NV12, sw, sh, gpu)
        assert vali.PyFrameUploader(gpu).Run(nv.reshape(-1), src)[0]
        out = vali.Surface.Make(vali.PixelFormat[dst], dw, dh, gpu)
        want = oracle.ud_nv12(nv, sw, sh, "NV12", dw, dh, dst).reshape(-1)
        assert vali.PySurfaceUD(gpu).Run(src, out) == (True, vali.TaskExecInfo.SUCCESS)
        assert np.array_equal(download(vali, gpu, out), want)
        with vali.tuning.Override(UD_DOWN2=0):                 # the general kernel on the same geometry
            assert vali.PySurfaceUD(gpu).Run(src, out)[0] and np.array_equal(download(vali, gpu, out), want)
        for rows in (2, 4):                                    # short waves (what single frames pick)
            with vali.tuning.Override(ROWS_PER_WAVE=rows):
                assert vali.PySurfaceUD(gpu).Run(src, out)[0] and np.array_equal(download(vali, gpu, out), want)


def test_three_to_two_batch_and_foreign_memory(vali, gpu, oracle):
    import torch
    sw, sh, dw, dh, n = 960, 540, 640, 360, 5
    frames = [make_nv12(sw, sh, 50 + i) for i in range(2)]
    srcs = [vali.Surface.Make(vali.NV12, sw, sh, gpu) for _ in range(n)]
    dsts = [vali.Surface.Make(vali.RGB, dw, dh, gpu) for _ in range(n)]
    for i, s in enumerate(srcs):
        assert vali.PyFrameUploader(gpu).Run(frames[i % 2].reshape(-1), s)[0]
    assert vali.PySurfaceUD(gpu).RunBatch(srcs, dsts) == (True, vali.TaskExecInfo.SUCCESS)
    wants = [oracle.ud_nv12(f, sw, sh, "NV12", dw, dh, "RGB").reshape(-1) for f in frames]
    for i, d in enumerate(dsts):
        assert np.array_equal(download(vali, gpu, d), wants[i % 2])
    # source and destination borrowed from torch tensors at odd byte offsets and foreign pitches
    t_src = torch.zeros((sh * 3 // 2, sw + 37), dtype=torch.uint8, device=f"cuda:{gpu}")
    t_dst = torch.zeros((dh, dw * 3 + 11), dtype=torch.uint8, device=f"cuda:{gpu}")
    t_src[:, 5:5 + sw] = torch.from_numpy(frames[0]).to(t_src.device)
    src = vali.Surface.from_dlpack(t_src[:, 5:5 + sw], vali.NV12)
    dst = vali.Surface.from_dlpack(t_dst[:, 7:7 + dw * 3], vali.RGB)
    assert vali.PySurfaceUD(gpu).Run(src, dst)[0]
    assert np.array_equal(t_dst[:, 7:7 + dw * 3].cpu().numpy().reshape(-1), wants[0])
    assert int(t_dst[:, :7].sum()) == 0 and int(t_dst[:, 7 + dw * 3:].sum()) == 0      # nothing outside the rows
