#!/usr/bin/env python3
"""Static instruction mix of a kernel's gfx950 ISA: python tools/isa_count.py vali_amd/csrc/x.hip <mangled-name-substring>
Prints VALU / SALU / LDS / VMEM / branch / wait counts for the whole kernel, for its prologue (up to the first loop
header) and for every top-level loop (header line to the last line that names it): what the 'instructions issued x 4
cycles' bound of DESIGN.md 5f is made of."""
import re
import subprocess
import sys
import tempfile
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent


def kind(op):
    if op.startswith(("s_cbranch", "s_branch")):
        return "branch"
    if op.startswith(("s_waitcnt", "s_nop")):
        return "wait"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    if op.startswith("v_"):
        return "valu"
    return "other"


def count(lines):
    c = dict.fromkeys(("valu", "salu", "lds", "vmem", "branch", "wait", "other"), 0)
    for l in lines:
        t = l.strip()
        if not t or t[0] in ";." or t.endswith(":"):
            continue
        c[kind(t.split()[0])] += 1
    c["total"] = sum(c.values())
    return c


def main():
    src, pat = sys.argv[1], sys.argv[2]
    with tempfile.TemporaryDirectory() as d:
        out = Path(d) / "k.s"
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-gpu-rdc",
                        "-S", "--cuda-device-only", f"-I{ROOT / 'include'}", src, "-o", str(out)], check=True,
                       stderr=subprocess.DEVNULL)
        asm = out.read_text().splitlines()
    name, body = None, []
    for l in asm:
        m = re.match(r"^(_Z\w+):", l)
        if m:
            if name and pat in name:
                break
            name, body = m.group(1), []
        elif name:
            body.append(l)
            if "s_endpgm" in l and pat in name:
                break
    print(name)
    print("  whole kernel   ", count(body))
    heads = [i for i, l in enumerate(body) if "Loop Header: Depth=1" in l]
    if heads:
        print("  before 1st loop", count(body[:heads[0]]))
    for h in heads:
        label = re.match(r"^\.(\w+):", body[h]).group(1)
        last = max((i for i, l in enumerate(body) if f"Header={label[1:]} " in l or f"Header={label[1:]}\t" in l or l.rstrip().endswith(f"Header={label[1:]} Depth=1")), default=h)
        print(f"  loop {label} ({last - h} lines)", count(body[h:last + 30]))
    if len(sys.argv) > 3:
        Path(sys.argv[3]).write_text("\n".join(body))


if __name__ == "__main__":
    main()
