"""The LDS-staged rotation by any angle (k_rotate_affine_lds, round 6) against the oracle, bit for bit, through every tile shape
the library keeps (VALI_TUNE_ROTATE_AFFINE) and at the places where its box logic can go wrong: source edges inside a tile (the
box slides left / repeats the last row), planes just wider and just narrower than a staged row (the gather form takes over),
quarter-turn angles with non-canonical shifts, pure translations (a = b = 0 on every pixel: the column right of the last one
and the row below the last one are weighted with exactly 0), destinations wholly outside the source.
Specification: oracle/vali_oracle.c vali_oracle_rotate_plane (RotateSurface.cpp:73-89, nppiRotate with NPPI_INTER_LINEAR)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

FORMATS = {"RGB": (np.uint8, 3), "BGR": (np.uint8, 3), "Y": (np.uint8, 1), "YUV444": (np.uint8, 1), "YUV420": (np.uint8, 1),
           "YUV444_10bit": (np.uint16, 1), "RGB_32F": (np.float32, 3)}
FORMS = (0, 2, 3, 4, 5, 6, 1)      # 1 = the gather form (round 2), the others stage in LDS (5: 64 x 128 tiles, one-channel 8-bit planes)


def _run(vali, gpu, oracle, fmt, sw, sh, dw, dh, angle, sx, sy, forms=FORMS, seed=0, batch=1):
    dt, ch = FORMATS[fmt]
    pf = vali.PixelFormat[fmt]
    rng = np.random.default_rng(seed + sw * 7 + dh)
    src = vali.Surface.Make(pf, sw, sh, gpu)
    n_el = src.HostSize // np.dtype(dt).itemsize
    host = (rng.random(n_el) * (255 if dt == np.uint8 else 1023 if dt == np.uint16 else 1.0)).astype(dt)
    assert vali.PyFrameUploader(gpu).Run(host.view(np.uint8), src)[0]
    want, off = [], 0
    probe = vali.Surface.Make(pf, dw, dh, gpu)
    for sp, dp in zip(src.Planes, probe.Planes):
        pw, ph = sp.Width // ch, sp.Height
        plane = np.ascontiguousarray(host[off: off + sp.Width * sp.Height].reshape(ph, sp.Width))
        want.append(oracle.rotate_plane(plane, ch, dp.Width // ch, dp.Height, angle, sx, sy, fill=9).reshape(-1))   # the same shifts for every plane (RotateSurface.cpp:132-159)
        off += sp.Width * sp.Height
    want = np.concatenate(want).view(np.uint8)
    rot = vali.PySurfaceRotator(gpu)
    for form in forms:
        with vali.tuning.Override(ROTATE_AFFINE=form):
            dsts = [vali.Surface.Make(pf, dw, dh, gpu) for _ in range(batch)]
            for d in dsts:
                nd = d.HostSize // np.dtype(dt).itemsize
                assert vali.PyFrameUploader(gpu).Run(np.full(nd, 9, dt).view(np.uint8), d)[0]
            if batch == 1:
                ok, info = rot.Run(src, dsts[0], angle, sx, sy)
            else:
                ok, info = rot.RunBatch([src] * batch, dsts, angle, sx, sy)
            assert ok, (info, form)
            for d in dsts:
                got = np.zeros(d.HostSize, np.uint8)
                assert vali.PySurfaceDownloader(gpu).Run(d, got)[0]
                bad = np.flatnonzero(got != want)
                assert bad.size == 0, (fmt, form, sw, sh, dw, dh, angle, sx, sy, bad[:5], got[bad[:5]], want[bad[:5]])


@pytest.mark.parametrize("fmt", list(FORMATS))
@pytest.mark.parametrize("angle,sx,sy", [(30.0, 0.0, 0.0), (-47.3, 120.0, 260.0), (10.0, 0.0, 60.0), (163.0, 400.0, 310.0), (271.5, -30.0, 350.0)])
def test_every_tile_shape_is_bit_exact(vali, gpu, oracle, fmt, angle, sx, sy):
    _run(vali, gpu, oracle, fmt, 400, 300, 420, 340, angle, sx, sy)


@pytest.mark.parametrize("fmt", ["RGB", "Y", "RGB_32F", "YUV444_10bit"])
@pytest.mark.parametrize("angle,sx,sy", [(0.0, 0.0, 0.0),          # identity: every sample sits ON a pixel (a = b = 0), also on the last row / column
                                         (0.0, 7.0, -5.0),         # whole-pixel translation
                                         (0.0, 3.25, 11.5),        # fractional translation: the right / bottom edges cut through tiles
                                         (90.0, 5.0, 100.0), (180.0, 250.0, 170.0), (270.0, 3.0, 0.0),   # quarter turns that are NOT the canonical permutations
                                         (360.0, 0.5, 0.5), (1e-3, 0.0, 0.0)])
def test_translations_and_non_canonical_quarter_turns(vali, gpu, oracle, fmt, angle, sx, sy):
    _run(vali, gpu, oracle, fmt, 331, 203, 350, 260, angle, sx, sy, forms=(0, 4, 6, 1))


@pytest.mark.parametrize("sw", [20, 47, 48, 49, 63, 64, 91, 92, 95, 96, 100, 129])
@pytest.mark.parametrize("fmt", ["RGB", "Y", "YUV420", "RGB_32F"])
def test_planes_around_the_width_of_a_staged_row(vali, gpu, oracle, fmt, sw):
    """A plane narrower than the widest staged row of a tile shape goes to the next smaller shape or to the gather form: the same
    bytes either way.  (YUV420: the chroma planes are half as wide as the luma plane of the same launch.)"""
    if fmt == "YUV420":
        sw += sw & 1
    _run(vali, gpu, oracle, fmt, sw, 150, 200, 180, 38.0, 60.0, 20.0, forms=(0, 3, 4, 5, 6))
    _run(vali, gpu, oracle, fmt, 150, max(2, sw // 2 * 2), 200, 180, -51.0, 10.0, 120.0, forms=(0, 4, 5))


def test_destination_outside_the_source_stays_untouched(vali, gpu, oracle):
    _run(vali, gpu, oracle, "RGB", 300, 200, 320, 240, 30.0, 5000.0, 0.0, forms=(0, 4, 6))
    _run(vali, gpu, oracle, "Y", 300, 200, 320, 240, 200.0, -4000.0, -4000.0, forms=(0, 4))
    _run(vali, gpu, oracle, "RGB", 300, 200, 320, 240, 45.0, 1e9, -1e9, forms=(0, 4))


@pytest.mark.parametrize("fmt,batch", [("RGB", 3), ("YUV420", 2), ("YUV444_10bit", 2)])
def test_batches(vali, gpu, oracle, fmt, batch):
    _run(vali, gpu, oracle, fmt, 640, 360, 640, 360, 30.0, 0.0, 0.0, forms=(0, 4, 6), batch=batch)


def test_random_geometries(vali, gpu, oracle):
    rng = np.random.default_rng(11)
    for k in range(40):
        fmt = ["RGB", "Y", "YUV420", "YUV444_10bit", "RGB_32F"][k % 5]
        sw, sh, dw, dh = (int(v) for v in rng.integers(50, 500, 4))
        if fmt == "YUV420":
            sw, sh, dw, dh = (v // 2 * 2 for v in (sw, sh, dw, dh))
        angle = float(rng.uniform(-360, 360))
        sx, sy = float(rng.uniform(-100, dw)), float(rng.uniform(-100, dh))
        _run(vali, gpu, oracle, fmt, sw, sh, dw, dh, angle, sx, sy, forms=(0, int(rng.choice([2, 3, 4, 5, 6]))), seed=k)


@pytest.mark.parametrize("fmt,dt,ch", [("RGB", np.uint8, 3), ("Y", np.uint8, 1), ("RGB_32F", np.float32, 3)])
@pytest.mark.parametrize("skew", [0, 1, 3])
def test_borrowed_surfaces_with_tight_pitch_and_skewed_base(vali, gpu, oracle, fmt, dt, ch, skew):
    """Foreign memory (DLPack): the source's pitch is its row (+0 / +5 elements), its base is skewed by `skew` elements and its last
    row ends where the buffer ends -- the staging loads of the box (16 bytes of a row per lane, a box at the right edge slid left)
    must never leave the rows; the destination's padding and the bytes in front of it stay untouched."""
    import torch

    sw, sh, dw, dh = 201, 117, 230, 150
    pf = vali.PixelFormat[fmt]
    tdt = torch.uint8 if dt == np.uint8 else torch.float32
    for extra in (0, 5):
        rng = np.random.default_rng(sw + skew + extra)
        host = (rng.random(sw * sh * ch) * (255 if dt == np.uint8 else 1.0)).astype(dt)
        sp, dpad = sw * ch + extra, dw * ch + 7
        sraw = torch.zeros(skew + (sh - 1) * sp + sw * ch, dtype=tdt, device="cuda")      # ends with the last row
        sview = torch.as_strided(sraw, (sh, sw * ch), (sp, 1), skew)
        sview.copy_(torch.from_numpy(host.reshape(sh, sw * ch)))
        fill = 90 if dt == np.uint8 else 0.25
        draw = torch.full((skew + dh * dpad,), fill, dtype=tdt, device="cuda")
        dview = torch.as_strided(draw, (dh, dw * ch), (dpad, 1), skew)
        torch.cuda.synchronize()
        src = vali.Surface.from_dlpack(torch.utils.dlpack.to_dlpack(sview), pf)
        dst = vali.Surface.from_dlpack(torch.utils.dlpack.to_dlpack(dview), pf)
        for form in (0, 4, 6):
            draw.fill_(fill)
            torch.cuda.synchronize()
            with vali.tuning.Override(ROTATE_AFFINE=form):
                assert vali.PySurfaceRotator(gpu).Run(src, dst, 23.0, 40.0, -12.0) == (True, vali.TaskExecInfo.SUCCESS)
            out = draw.cpu().numpy()
            got = np.lib.stride_tricks.as_strided(out[skew:], (dh, dw * ch), (dpad * out.itemsize, out.itemsize))
            want = oracle.rotate_plane(host.reshape(sh, sw * ch), ch, dw, dh, 23.0, 40.0, -12.0, fill=fill).reshape(dh, dw * ch)
            assert np.array_equal(np.ascontiguousarray(got).view(np.uint8), want.view(np.uint8)), (extra, form)    # (bits, also of the floats)
            pad = np.lib.stride_tricks.as_strided(out[skew + dw * ch:], (dh - 1, dpad - dw * ch), (dpad * out.itemsize, out.itemsize))
            assert np.all(pad == dt(fill)) and np.all(out[:skew] == dt(fill))
