"""Pins the rotation restatement against the reference's etalons.

reference tests/test_PySurfaceRotator.py:101-137 rotates data/frame_0.jpg by 90/180/270 and
requires PSNR >= 42 dB against data/frame_0_<angle>_deg.jpg (PIL-decoded).  The same four
JPEGs are committed under tests/golden/ (data fixtures of the reference's test-suite)."""
import numpy as np
import pytest

from conftest import GOLDEN

PIL = pytest.importorskip("PIL.Image")


def psnr(a, b):
    mse = np.mean((a.astype(np.float64) - b.astype(np.float64)) ** 2)
    return 10 * np.log10(255.0 ** 2 / mse)


@pytest.mark.parametrize("angle", [90, 180, 270])
def test_quarter_turns_match_reference_etalons(oracle, angle):
    img = np.asarray(PIL.open(GOLDEN / "frame_0.jpg"))
    h, w, _ = img.shape
    a, sx, sy = oracle.canonical_shifts(angle, w, h)
    dw, dh = (w, h) if angle == 180 else (h, w)
    out = oracle.rotate_plane(np.ascontiguousarray(img.reshape(h, w * 3)), 3, dw, dh, a, sx, sy)
    out = out.reshape(dh, dw, 3)
    etalon = np.asarray(PIL.open(GOLDEN / f"frame_0_{angle}_deg.jpg"))
    assert etalon.shape == out.shape
    assert psnr(out, etalon) >= 42.0                       # the reference's own threshold
    assert np.array_equal(out, np.rot90(img, k=angle // 90))   # exact permutation
    for wrong in {1, 2, 3} - {angle // 90}:                # direction really is pinned
        cand = np.rot90(img, k=wrong)
        if cand.shape == etalon.shape:
            assert psnr(cand, etalon) < 20.0


def test_identity_and_outside_pixels_untouched(oracle):
    rng = np.random.default_rng(0)
    src = rng.integers(0, 256, (20, 30), dtype=np.uint8)
    assert np.array_equal(oracle.rotate_plane(src, 1, 30, 20, 0.0), src)
    out = oracle.rotate_plane(src, 1, 40, 40, 30.0, 5.0, 12.0, fill=7)
    assert (out == 7).any() and (out != 7).any()           # NPP leaves unmapped dst pixels alone
    shifted = oracle.rotate_plane(src, 1, 30, 20, 0.0, 3.0, 2.0, fill=9)
    assert np.array_equal(shifted[2:, 3:], src[:-2, :-3]) and (shifted[:2] == 9).all()


def test_u16_and_f32_quarter_turn(oracle):
    rng = np.random.default_rng(1)
    for dt in (np.uint16, np.float32):
        src = (rng.random((12, 9 * 3)) * 1000).astype(dt)
        a, sx, sy = oracle.canonical_shifts(270, 9, 12)
        out = oracle.rotate_plane(src, 3, 12, 9, a, sx, sy)
        assert np.array_equal(out.reshape(9, 12, 3), np.rot90(src.reshape(12, 9, 3), k=3))
