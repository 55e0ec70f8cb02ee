#!/usr/bin/env python3
"""NV12 -> RGB on ragged widths / foreign alignment vs the aligned neighbour geometry.

For every ragged case one launch converts `n` frames resident in HBM (n chosen so src + dst ~ 6 GB, far beyond
the 256 MiB Infinity Cache); the figure of merit is GB/s of algorithmic bytes (4.5 B per pixel) relative to the
nearest 16-pixel-aligned width measured the same way.  `odd-stride` cases put the same pixels in torch tensors
whose row stride is W + 3 bytes (source) and 3 W + 5 bytes (destination): no row is 16-byte aligned.
Prints one JSON object per case and a markdown table at the end.
"""
import json
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402,F401
import vali_amd as vali  # noqa: E402
from bench_configs import DEV, timed  # noqa: E402

CASES = [((854, 480), (848, 480)), ((1366, 768), (1360, 768)), ((1918, 1078), (1920, 1080)), ((3838, 2158), (3840, 2160))]


def run(w, h, dst_fmt=None, odd_stride=False, target_bytes=6e9):
    dst_fmt = dst_fmt or vali.RGB
    n = int(max(64, min(4096, target_bytes // (w * h * 4.5))))
    cvt = vali.PySurfaceConverter(DEV)
    cc = vali.ColorspaceConversionContext(vali.ColorSpace.BT_709, vali.ColorRange.MPEG)
    rng = np.random.default_rng(1)
    seed = torch.from_numpy(rng.integers(16, 236, (h * 3 // 2, w), dtype=np.uint8)).cuda()
    keep = []
    if odd_stride:
        srcs, dsts = [], []
        for _ in range(n):
            sb = torch.empty((h * 3 // 2, w + 3), dtype=torch.uint8, device="cuda")
            sb[:, :w].copy_(seed)
            db = torch.empty((h, 3 * w + 5), dtype=torch.uint8, device="cuda")
            keep += [sb, db]
            srcs.append(vali.Surface.from_dlpack(torch.utils.dlpack.to_dlpack(sb[:, :w]), vali.NV12))
            dsts.append(vali.Surface.from_dlpack(torch.utils.dlpack.to_dlpack(db[:, :3 * w]), vali.RGB))
    else:
        srcs = [vali.Surface.Make(vali.NV12, w, h, DEV) for _ in range(n)]
        dsts = [vali.Surface.Make(dst_fmt, w, h, DEV) for _ in range(n)]
        for s in srcs:
            torch.from_dlpack(s)[:, :].copy_(seed)
    torch.cuda.synchronize()
    batch = cvt.PrepareBatch(srcs, dsts)
    ms, _ = timed(cvt.Stream, lambda: cvt.RunBatchAsync(batch, cc_ctx=cc), 10, 2)
    gbps = w * h * 4.5 * n / (ms * 1e-3) / 1e9
    return {"geometry": f"{w}x{h}", "dst": dst_fmt.name, "odd_stride": odd_stride, "frames": n, "ms_per_launch": round(ms, 4),
            "us_per_frame": round(ms * 1e3 / n, 3), "GBps": round(gbps, 1), "frac_of_8TBps": round(gbps / 8000, 4)}


if __name__ == "__main__":
    rows = []
    for ragged, aligned in CASES:
        a = run(*aligned)
        r = run(*ragged)
        o = run(*ragged, odd_stride=True)
        p = run(*ragged, dst_fmt=vali.RGB_PLANAR)
        pa = run(*aligned, dst_fmt=vali.RGB_PLANAR)
        for x in (a, r, o, pa, p):
            print(json.dumps(x), flush=True)
        rows.append((aligned, a, ragged, r, o, pa, p))
    print("\n| aligned | GB/s | ragged | GB/s | % | ragged + odd strides GB/s | % | planar aligned | planar ragged | % |")
    print("|---|---|---|---|---|---|---|---|---|---|")
    for aligned, a, ragged, r, o, pa, p in rows:
        print(f"| {a['geometry']} | {a['GBps']} | {r['geometry']} | {r['GBps']} | {100 * r['GBps'] / a['GBps']:.1f} | "
              f"{o['GBps']} | {100 * o['GBps'] / a['GBps']:.1f} | {pa['GBps']} | {p['GBps']} | {100 * p['GBps'] / pa['GBps']:.1f} |")
