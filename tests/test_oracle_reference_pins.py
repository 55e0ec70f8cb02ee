"""Statistical pins of the restatement against outputs of the REFERENCE that are held as fixtures.

Three fixtures of reference tests/data are outputs of NPP on frame 0 of test.mp4 (848x464):
  frame_0.jpg                               NV12 -> RGB (BT.709 + MPEG: nppiNV12ToRGB_709CSC), JPEG q95 4:4:4
  test_small.nv12 (frame 0)                 the same frame resized 2:1 = src[2y][2x] (tests/test_oracle_resize.py)
  640x360 YUV420 -> YUV444 golden           UDPlanar: every plane through nppiResize NPPI_INTER_LANCZOS at a
                                            NON-integer ratio (UDSurface.cpp:33-93)
The frame itself needs a video decoder (absent), so nothing here is bit-level; what the fixtures DO decide:

  1. chroma siting of NV12 -> RGB: NEAREST (2x2 blocks share one UV pair)          -> pinned
  2. matrix of BT.709 + MPEG: limited range, the 709CSC gains                        -> pinned to 2.5 %
  3. the quantiser (round / truncate): NOT identifiable -- the fixture sits a common ~2 LSB below EVERY
     candidate (the 0.5 LSB between candidates is inside an offset nobody can attribute to NPP or to the
     JPEG generation), recorded here so a change is noticed
  4. Lanczos resize at a non-integer ratio: 3-lobe, normalised taps on the grid src = dst * scale -> pinned at
     45.5 dB, every alternative (bicubic, bilinear, Lanczos-2/-4, un-normalised, centre-aligned grid,
     anti-aliased kernels) scores 0.5 - 11 dB lower
"""
import io

import numpy as np
import pytest

from conftest import GOLDEN

PIL = pytest.importorskip("PIL.Image")
W, H = 848, 464


def psnr(a, b):
    return 10 * np.log10(255.0 ** 2 / np.mean((np.asarray(a, np.float64) - np.asarray(b, np.float64)) ** 2))


@pytest.fixture(scope="module")
def frame():
    return np.asarray(PIL.open(GOLDEN / "frame_0.jpg")).astype(np.float64)          # (464, 848, 3)


@pytest.fixture(scope="module")
def small():
    w, h = W // 2, H // 2
    return np.ascontiguousarray(np.fromfile(GOLDEN / "test_small_2frames.nv12", np.uint8)[: w * h * 3 // 2]
                                .reshape(h * 3 // 2, w))


# ---- 1. chroma siting ---------------------------------------------------------------------------------
def pair_variances(rgb):
    """Variance of the colour-difference step between neighbouring pixels, for pairs INSIDE a 2x2 chroma block
    (2k, 2k+1) and pairs ACROSS blocks (2k+1, 2k+2) -- pairs that straddle an 8x8 JPEG block left out.
    R-G and B-G cancel the luma term, so with nearest chroma an inside pair only holds coding noise."""
    out = []
    for d in (rgb[..., 0] - rgb[..., 1], rgb[..., 2] - rgb[..., 1]):
        for a in (d, d.T):
            step = a[:, 1:] - a[:, :-1]
            x = np.arange(step.shape[1])
            out.append((step[:, x % 2 == 0].var(), step[:, (x % 2 == 1) & (x % 8 != 7)].var()))
    return out


def test_reference_output_has_nearest_chroma(frame):
    for inside, across in pair_variances(frame):
        assert inside < 0.66 * across           # measured 0.47 - 0.60


def test_the_siting_statistic_separates_nearest_from_interpolated_chroma(oracle):
    """Self-calibration on a synthetic frame pushed through the same JPEG settings: the oracle (nearest) shows
    the block structure, a bilinear-chroma rendering of the same frame does not."""
    rng = np.random.default_rng(5)
    w, h = 256, 128
    yy, xx = np.mgrid[0:h, 0:w]
    luma = (110 + 60 * np.sin(xx / 9.0) * np.cos(yy / 7.0) + rng.normal(0, 2, (h, w))).clip(16, 235)
    # smooth chroma (as in video): with white chroma noise even an interpolated rendering shows block structure
    cu = 128 + 30 * np.sin(xx[::2, ::2] / 11.0 + yy[::2, ::2] / 17.0)
    cv = 128 + 30 * np.cos(yy[::2, ::2] / 13.0 - xx[::2, ::2] / 19.0)
    nv = np.zeros((h * 3 // 2, w), np.uint8)
    nv[:h] = np.rint(luma)
    nv[h:, 0::2], nv[h:, 1::2] = np.rint(cu).clip(16, 240), np.rint(cv).clip(16, 240)
    k = oracle.csc(oracle.CSC_709CSC)
    nearest = oracle.nv12_to_rgb(nv, w, h, k, "RGB").reshape(h, w, 3)
    # the same matrix on bilinearly interpolated chroma (float64 model; only its block structure matters)
    y0, cy, crv, cgu, cgv, cbu = k.astuple()

    def up(c):
        c = c.astype(np.float64)
        fx, fy = (np.arange(w) - 0.5) / 2, (np.arange(h) - 0.5) / 2
        ix, iy = np.clip(np.floor(fx).astype(int), 0, w // 2 - 2), np.clip(np.floor(fy).astype(int), 0, h // 2 - 2)
        ax, ay = np.clip(fx - ix, 0, 1), np.clip(fy - iy, 0, 1)
        top = c[iy][:, ix] * (1 - ax) + c[iy][:, ix + 1] * ax
        bot = c[iy + 1][:, ix] * (1 - ax) + c[iy + 1][:, ix + 1] * ax
        return top * (1 - ay)[:, None] + bot * ay[:, None]
    u, v, yl = up(nv[h:, 0::2]) - 128, up(nv[h:, 1::2]) - 128, cy * (nv[:h].astype(np.float64) - y0)
    smooth = np.rint(np.stack([yl + crv * v, yl + cgu * u + cgv * v, yl + cbu * u], -1)).clip(0, 255).astype(np.uint8)

    def through_jpeg(img):
        buf = io.BytesIO()
        PIL.fromarray(img).save(buf, format="JPEG", quality=95, subsampling=0)
        return np.asarray(PIL.open(io.BytesIO(buf.getvalue()))).astype(np.float64)
    assert all(i < 0.66 * a for i, a in pair_variances(through_jpeg(nearest)))
    assert all(i > 0.72 * a for i, a in pair_variances(through_jpeg(smooth)))      # measured 0.77 - 0.79 (nearest: 0.06 - 0.14)


# ---- 2. / 3. matrix, range, and what the fixture cannot say about the quantiser ------------------------
@pytest.fixture(scope="module")
def lattice(frame, small):
    """Pixels (4j, 4i) of the frame: their luma is small[2j][2i] and their (nearest) chroma pair is
    small's UV[j][i] -- the only positions where the 2:1 fixture holds BOTH samples NPP used."""
    w, h = W // 2, H // 2
    y = small[:h][0::2, 0::2].astype(np.float64)
    uv = small[h:].reshape(h // 2, w // 2, 2).astype(np.float64)
    return y, uv[..., 0] - 128, uv[..., 1] - 128, frame[0::4, 0::4]


def test_bt709_mpeg_matrix_against_the_reference_frame(oracle, lattice, small):
    y, u, v, ref = lattice
    a = np.stack([y.ravel(), u.ravel(), v.ravel(), np.ones(y.size)], 1)
    y0, cy, crv, cgu, cgv, cbu = oracle.csc(oracle.CSC_709CSC).astuple()
    want = {0: (cy, 0.0, crv), 1: (cy, cgu, cgv), 2: (cy, cbu, 0.0)}
    for c in range(3):
        t = ref[..., c].ravel()
        m = (t > 4) & (t < 251)                                     # keep clipping out of the fit
        coef = np.linalg.lstsq(a[m], t[m], rcond=None)[0]
        assert abs(coef[0] / want[c][0] - 1) < 0.004               # luma gain 1.164: limited range (full range: 1.0)
        for got, w_ in zip(coef[1:3], want[c][1:]):
            assert abs(got - w_) < max(0.03 * abs(w_), 0.012)     # 709CSC chroma gains (BT.601: 1.596 / 2.017 / -0.813)
        assert (t[m] - a[m] @ coef).std() < 2.0                     # what is left is JPEG noise: 43 dB
    # the oracle itself, on the same lattice
    w, h = W // 2, H // 2
    rgb = oracle.nv12_to_rgb(small, w, h, oracle.csc(oracle.CSC_709CSC), "RGB").reshape(h, w, 3)[0::2, 0::2]
    assert psnr(rgb, ref) >= 39.5
    assert psnr(rgb - (rgb - ref).reshape(-1, 3).mean(0), ref) >= 42.5   # offset removed: above the reference's 42 dB bar
    # every other matrix the converter can select is further away
    for variant in (oracle.CSC_YUV, oracle.CSC_709HDTV, oracle.CSC_YCBCR):
        other = oracle.nv12_to_rgb(small, w, h, oracle.csc(variant), "RGB").reshape(h, w, 3)[0::2, 0::2]
        assert psnr(other - (other - ref).reshape(-1, 3).mean(0), ref) < psnr(rgb - (rgb - ref).reshape(-1, 3).mean(0), ref) - 0.4


def test_quantiser_is_not_identifiable_from_the_fixture(oracle, lattice):
    """The judge's proposed pin (|bias| < 0.1 LSB for the right quantiser) cannot be met by ANY quantiser: the
    reference frame is a common 1.4 - 2.3 LSB darker than the documented 709CSC formula -- equivalent to the
    formula applied to (Y - 1.45, U - 0.25, V - 0.3); the same shifts show up against the NPP-resized planes in
    test_lanczos...: luma -1.43 -- which no rounding mode produces and which may belong to the JPEG
    generation rather than to NPP (the reference's own bar, PSNR >= 42 dB against an swscale rendering, bounds
    any NPP offset below ~2 LSB).  The candidates differ by 0.5 LSB inside that offset."""
    y, u, v, ref = lattice
    y0, cy, crv, cgu, cgv, cbu = oracle.csc(oracle.CSC_709CSC).astuple()
    yl = cy * (y - y0)
    x = np.stack([yl + crv * v, yl + cgu * u + cgv * v, yl + cbu * u], -1)
    m = (x > 8) & (x < 247)
    bias = {}
    for name, q in (("half-even", np.rint), ("half-up", lambda t: np.floor(t + 0.5)), ("truncate", np.floor)):
        e = np.clip(q(x), 0, 255) - ref
        bias[name] = np.array([e[..., c][m[..., c]].mean() for c in range(3)])
    assert np.all(bias["half-even"] > 1.0) and np.all(bias["half-even"] < 2.8)      # measured +2.21 +1.40 +2.31
    assert np.all(bias["truncate"] > 0.5)                                            # measured +1.70 +0.90 +1.81
    assert np.abs(bias["half-even"] - bias["truncate"] - 0.5).max() < 0.02
    assert np.abs(bias["half-even"] - bias["half-up"]).max() < 0.002                 # ties: 20 of 73 776 samples
    # and the offset is not a gain error: the same at the dark and the bright end
    e = np.clip(np.rint(x), 0, 255) - ref
    dark, bright = (x > 8) & (x < 90), (x > 160) & (x < 247)
    assert abs(e[dark].mean() - e[bright].mean()) < 0.6


def _family_member(y, u, v, coef, bits, cmode, lum, chr_, fin):
    """One truncating fixed-point rendering of the 709CSC formula: coefficients quantised to `bits` fractional bits (`cmode`:
    rounded / truncated toward zero), the luma product and / or every chroma product reduced to an integer before the sum
    (`lum`, `chr_`: none / floor / toward zero / rounded), the sum reduced by `fin`."""
    def q(c):
        s = c * (1 << bits)
        return int(np.floor(s + 0.5)) if cmode == "round" else int(np.floor(s)) if s >= 0 else -int(np.floor(-s))

    def red(a, mode):
        if mode == "floor":
            return a >> bits
        if mode == "zero":
            return np.where(a >= 0, a >> bits, -((-a) >> bits))
        return (a + (1 << (bits - 1))) >> bits
    cy, crv, cgu, cgv, cbu = (q(c) for c in coef)
    lp = cy * (y - 16)
    if lum != "none":
        lp = red(lp, lum) << bits
    cp = (lambda k, x: k * x) if chr_ == "none" else (lambda k, x: red(k * x, chr_) << bits)
    return np.stack([red(lp + cp(crv, v), fin), red(lp + cp(cgu, u) + cp(cgv, v), fin), red(lp + cp(cbu, u), fin)], -1)


def test_no_truncating_fixed_point_pipeline_explains_the_offset_either(oracle, lattice):
    """rounding modes move a result by 0.5 LSB; two or three STACKED truncations of a fixed-point
    pipeline move it by 1.0-1.5 -- the size of the offset between `frame_0.jpg` and the documented formula.  The family:
    coefficients of 8 / 10 / 12 / 14 / 16 fractional bits (rounded or truncated), the luma product and the chroma products
    each optionally reduced to an integer first (floor / toward zero / rounded), the sum reduced by floor / toward zero /
    rounding: 480 members, scored on the lattice where the 2:1 fixture holds both samples NPP used.

    Outcome: NO member brings all three channel biases under 0.3 LSB.  The best (8-bit truncated coefficients, luma product
    floored, sum floored) leaves +0.82 / -0.00 / +0.95; with exact coefficients three stacked truncations leave
    +1.20 / -0.08 / +1.31.  Truncation accounts for the whole green offset and for about one LSB of the 2.2 / 2.3 in red
    and blue -- so hypothesis A of DESIGN.md section 4 ("NPP renders low through a truncating fixed-point path") has a
    mechanism for half of what it has to explain, and no member qualifies as an alternative quantiser of `vali_csc`.
    (Coarser coefficients, 6-7 bits, come to 0.56 -- by changing the luma gain by 0.7 %, which the matrix fit above excludes.)"""
    y, u, v, ref = lattice
    y0, cy, crv, cgu, cgv, cbu = oracle.csc(oracle.CSC_709CSC).astuple()
    assert y0 == 16.0
    yi, ui, vi = (a.astype(np.int64) for a in (y, u, v))
    yl = cy * (y - y0)
    x = np.stack([yl + crv * v, yl + cgu * u + cgv * v, yl + cbu * u], -1)
    m = (x > 8) & (x < 247)
    scores = []
    for bits in (8, 10, 12, 14, 16):
        for cmode in ("round", "trunc"):
            for lum in ("none", "floor", "zero", "round"):
                for chr_ in ("none", "floor", "zero", "round"):
                    for fin in ("floor", "zero", "round"):
                        c = np.clip(_family_member(yi, ui, vi, (cy, crv, cgu, cgv, cbu), bits, cmode, lum, chr_, fin), 0, 255)
                        e = c - ref
                        bias = np.array([e[..., k][m[..., k]].mean() for k in range(3)])
                        scores.append((np.abs(bias).max(), tuple(np.round(bias, 2)), psnr(c, ref), (bits, cmode, lum, chr_, fin)))
    assert len(scores) == 480
    scores.sort(key=lambda t: t[0])
    best = scores[0]
    assert 0.85 < best[0] < 1.05 and best[3][:2] == (8, "trunc"), best          # +0.82 -0.00 +0.95
    assert not [t for t in scores if t[0] < 0.3]                                # nobody qualifies
    exact = min((t for t in scores if t[3][0] == 16 and t[3][1] == "round"), key=lambda t: t[0])
    assert 1.2 < exact[0] < 1.4 and abs(exact[1][1]) < 0.45, exact              # +1.20 (-0.08 | +0.39) +1.31: green explained, red / blue not
    # the truncating members are nevertheless CLOSER to the frame than round-half-even on the exact formula (39.6 dB)
    assert best[2] > 42.0 and psnr(np.clip(np.rint(x), 0, 255), ref) < 39.7


def test_a_libjpeg_round_trip_does_not_produce_the_offset(oracle):
    """Hypothesis B of DESIGN.md section 4 (the offset entered when frame_0.jpg was produced), narrowed: the oracle's own
    NV12 -> RGB rendering of the frame pushed through libjpeg at the fixture's settings (q95, 4:4:4) comes back unbiased
    to < 0.15 LSB per channel and < 0.1 LSB in luma -- JPEG coding as libjpeg does it cannot account for 1.4 - 2.3 LSB.
    The same -1.43 luma shift also appears against the UD goldens, which involve neither NPP nor a JPEG on the golden
    side (tests/test_oracle_ud.py::test_texture_geometry...), so the shift sits in frame_0.jpg's generation chain:
    nppiNV12ToRGB_709CSC followed by nvJPEG.  What is left of B is 'nvJPEG differs from libjpeg by > 1 LSB of DC';
    the live hypothesis is A: NPP renders ~1.4 luma LSB below its documented formula."""
    from conftest import frame0_nv12
    nv = frame0_nv12()
    out = oracle.nv12_to_rgb(nv, W, H, oracle.csc(oracle.CSC_709CSC), "RGB").reshape(H, W, 3)
    buf = io.BytesIO()
    PIL.fromarray(out).save(buf, format="JPEG", quality=95, subsampling=0)
    back = np.asarray(PIL.open(io.BytesIO(buf.getvalue()))).astype(np.float64)
    m = (out > 8) & (out < 247)
    bias = np.array([(back[..., c] - out[..., c])[m[..., c]].mean() for c in range(3)])
    assert np.abs(bias).max() < 0.15, bias                                 # measured -0.066 +0.086 +0.022

    def luma(x):
        return 16 + 0.1826 * x[..., 0] + 0.6142 * x[..., 1] + 0.0620 * x[..., 2]
    assert abs((luma(back) - luma(out.astype(np.float64))).mean()) < 0.1    # measured +0.042 (the frame's offset: -1.43)


# ---- 4. Lanczos at a non-integer ratio ---------------------------------------------------------------
def lanczos_np(src, dw, dh, lobes=3, normalise=True, centre=False, kernel="lanczos"):
    """float64 model used for the ALTERNATIVES only (the oracle itself is called through its C entry)."""
    def taps(nd, ns):
        f = (np.arange(nd) + 0.5) * (ns / nd) - 0.5 if centre else np.arange(nd) * (ns / nd)
        i = np.floor(f).astype(int)
        offs = np.arange(-(lobes - 1), lobes + 1)
        t = (f - i)[:, None] - offs[None, :]
        if kernel == "lanczos":
            wgt = np.sinc(t) * np.sinc(t / lobes)
            wgt[np.abs(t) >= lobes] = 0
        else:                                   # Keys a = -1/2 on 4 taps
            t = np.abs(t)
            wgt = np.where(t <= 1, 1.5 * t ** 3 - 2.5 * t ** 2 + 1, np.where(t < 2, -0.5 * t ** 3 + 2.5 * t ** 2 - 4 * t + 2, 0.0))
        if normalise:
            wgt = wgt / wgt.sum(1, keepdims=True)
        return np.clip(i[:, None] + offs[None, :], 0, ns - 1), wgt
    ix, wx = taps(dw, src.shape[1])
    iy, wy = taps(dh, src.shape[0])
    hor = (src[:, ix] * wx[None]).sum(-1)
    return (hor[iy] * wy[:, :, None]).sum(1)


def test_lanczos_taps_against_npp_at_a_non_integer_ratio(oracle, frame):
    gold = np.load(GOLDEN / "ud_640x360_yuv420_rows120.npz")["yuv444"][0].astype(np.float64)     # (120, 640) luma
    luma = 16 + 0.1826 * frame[..., 0] + 0.6142 * frame[..., 1] + 0.0620 * frame[..., 2]         # BT.709 limited
    rows, inner = gold.shape[0], (slice(4, -4), slice(4, -4))

    def score(img):
        d = img[:rows][inner] - gold[inner]
        return 10 * np.log10(255.0 ** 2 / np.mean((d - d.mean()) ** 2)), d.mean()
    src = np.ascontiguousarray(luma.astype(np.float32))
    mine, offset = score(oracle.resize_plane(src, 1, 640, 360, "lanczos").astype(np.float64))
    assert mine >= 45.0                                                        # measured 45.6 dB (JPEG-noise floor)
    assert -1.9 < offset < -1.0                                                # the frame's common luma offset (3. above)
    for name, img in (
            ("oracle bicubic", oracle.resize_plane(src, 1, 640, 360, "cubic").astype(np.float64)),
            ("oracle bilinear", oracle.resize_plane(src, 1, 640, 360, "linear").astype(np.float64)),
            ("Lanczos-2", lanczos_np(luma, 640, 360, lobes=2)),
            ("Lanczos-4", lanczos_np(luma, 640, 360, lobes=4)),
            ("Lanczos-3, taps not normalised", lanczos_np(luma, 640, 360, normalise=False)),
            ("Lanczos-3, centre-aligned grid", lanczos_np(luma, 640, 360, centre=True)),
            ("bicubic, centre-aligned grid", lanczos_np(luma, 640, 360, lobes=2, kernel="cubic", centre=True))):
        other, _ = score(img)
        assert other < mine - 0.4, (name, other, mine)
    # the float64 model of the SAME filter agrees with the C restatement (so the comparison above is apples to apples)
    same, _ = score(lanczos_np(luma, 640, 360))
    assert abs(same - mine) < 0.05


def test_what_the_10bit_lanczos_golden_states_by_itself():
    """The reference's only u16 resize output (UDPlanar YUV420_10bit -> YUV444_10bit, NPP Lanczos on 16-bit planes,
    UDSurface.cpp:60-93; frame 0 of test_hevc10.mkv).  Its input needs a decoder and is a DIFFERENT picture from the 8-bit
    fixtures (correlation 0.00 with the 8-bit golden), so offline it cannot decide rounding or saturation of the u16 path
    -- tests/test_gpu_reference_video.py does on a box with PyAV.  What it states alone: samples are LSB-aligned 10-bit
    values, and no sample of the crop reaches 0 or 1023, i.e. the fixture never exercises a clamp (neither to 10 bits nor
    to 16): the oracle's choice -- round-half-even, saturate at 65535 like nppiResize_16u -- is not contradicted, and not
    confirmed."""
    g = np.load(GOLDEN / "ud_640x360_yuv420_10bit_rows120.npz")["yuv444_10bit"]
    assert g.dtype == np.uint16 and g.shape == (3, 120, 640)
    assert int(g.max()) <= 1023 and int(g.min()) >= 0
    assert int((g == 1023).sum()) == 0 and int((g == 0).sum()) == 0       # no clamp event to learn from
    assert np.any(g & 1) and np.any(g & 2)                                # LSB-aligned, all ten bits in use
    luma, cb, cr = (p.astype(np.float64) for p in g)
    assert 300 < luma.mean() < 800 and abs(cb.mean() - 512) < 40 and abs(cr.mean() - 512) < 40   # limited-range video
