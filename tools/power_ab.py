#!/usr/bin/env python3
"""Power / clock A/B of the headline conversion on ONE box: NV12 -> dst 3840x2160, `frames` per launch, every
variant run for ~`seconds` of steady load with the chip's state (sysfs: sclk, mclk, power, temperature) read between launches.
Prints one JSON line per variant: {"variant", "us_per_frame", "frac", "sclk_mhz": [min, max], "power_w": [min, max], ...}.

  python tools/power_ab.py [frames=256] [seconds=1.0]
Variants: the library's default (packed RGB through the LDS strip), the direct-store form (no strip), planar RGB (no strip, three
row stores), residency caps of 16 / 24 / 32 waves per CU, and BGR -- interleaved twice so that drift of the box shows."""
import json
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import vali_amd as vali  # noqa: E402
from vali_amd._native import shim  # noqa: E402
from vali_amd.pipeline import BatchedFramePipeline, gpu_clock_state  # noqa: E402

DEV = 0
frames = int(sys.argv[1]) if len(sys.argv) > 1 else 256
seconds = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
W, H = 3840, 2160
bus = shim.device_pci_bus_id(DEV)
cc = vali.ColorspaceConversionContext(vali.ColorSpace.BT_709, vali.ColorRange.MPEG)
VARIANTS = [("default RGB (LDS strip)", vali.RGB, {}), ("RGB direct store (no strip)", vali.RGB, {"NV12_DIRECT_STORE": 1}),
            ("RGB_PLANAR (no strip)", vali.RGB_PLANAR, {}), ("RGB, 16 waves per CU", vali.RGB, {"WAVES_PER_CU": 16}),
            ("RGB, 24 waves per CU", vali.RGB, {"WAVES_PER_CU": 24}), ("RGB, 32 waves per CU", vali.RGB, {"WAVES_PER_CU": 32}),
            ("BGR (LDS strip)", vali.BGR, {})]
pipes = {}
for fmt in {v[1] for v in VARIANTS}:
    p = BatchedFramePipeline(DEV, W, H, frames, fmt)
    p.set_coefficients(cc)
    pipes[fmt] = p
print(json.dumps({"box": bus, "frames": frames, "idle": gpu_clock_state(bus)}), flush=True)
for rep in range(2):
    for name, fmt, tune in VARIANTS:
        p = pipes[fmt]
        stream = p.Stream
        with vali.tuning.Override(**tune):
            for _ in range(5):
                p.run_async()
            shim.stream_sync(DEV, stream)
            seen, n, t0 = [], 0, time.perf_counter()
            a, b = shim.event_create(DEV), shim.event_create(DEV)
            ms = 0.0
            while time.perf_counter() - t0 < seconds:
                shim.event_record(DEV, a, stream)
                for _ in range(4):
                    p.run_async()
                shim.event_record(DEV, b, stream)
                shim.event_sync(DEV, b)
                ms += shim.event_elapsed_ms(a, b)
                n += 4
                seen.append(gpu_clock_state(bus))
            shim.event_destroy(DEV, a)
            shim.event_destroy(DEV, b)

        def span(k):
            v = [s[k] for s in seen if s.get(k) is not None]
            return [min(v), max(v), round(sum(v) / len(v), 1)] if v else None
        us = ms * 1e3 / n / frames
        print(json.dumps({"variant": name, "rep": rep, "us_per_frame": round(us, 4), "frac": round(p.bytes_per_frame / us / 1e6 / 8.0, 4),
                          "sclk_mhz": span("sclk_mhz"), "mclk_mhz": span("mclk_mhz"), "power_w": span("power_w"), "temp_c": span("temp_c"),
                          "launches": n}), flush=True)
