#!/usr/bin/env python3
"""Per kernel: the ORDER of vector-memory instructions and vmcnt waits in the gfx950 ISA, compressed.

    python tools/isa_waits.py vali_amd/csrc/resize.hip [kernel-name-substring]

Compiles the file to assembly (hipcc -S, device only) and prints, for every kernel whose mangled name contains the
substring, one line of tokens:
    L / S      vector-memory load / store (global_*, buffer_*), `F` prefix for the flat_* forms
    wN         s_waitcnt vmcnt(N)
    |          a branch
    tok xK     K repetitions
What to look for (all four cost this repo 10-50 % until found):
  * `L... w0` inside a loop that is meant to keep rows in flight: the prefetch is drained every trip;
  * a prologue `Lx8` whose FIRST wait is w0/w1: the scheduler issued row 0 last (vmcnt retires in order);
  * any F token in a kernel that pipelines loads: a pending flat access turns every vmcnt wait into vmcnt(0);
  * w0 in front of an inner loop that only stores: the gfx9 "flush vmcnt in the preheader" heuristic.
"""
import re
import subprocess
import sys
import tempfile
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent


def tokens(asm, pat):
    cur, seq, out = None, [], []

    def flush():
        if cur and pat in cur and seq:
            comp = []
            for t in seq:
                if comp and comp[-1][0] == t:
                    comp[-1][1] += 1
                else:
                    comp.append([t, 1])
            out.append((cur, " ".join(f"{t}x{n}" if n > 1 else t for t, n in comp)))

    for line in asm.splitlines():
        m = re.match(r"^(_Z\w+):", line)
        if m:
            flush()
            cur, seq = m.group(1), []
            continue
        s = line.strip()
        if s.startswith(("global_load", "buffer_load")):
            seq.append("L")
        elif s.startswith(("global_store", "buffer_store")):
            seq.append("S")
        elif s.startswith("flat_load"):
            seq.append("FL")
        elif s.startswith("flat_store"):
            seq.append("FS")
        elif s.startswith("s_waitcnt") and "vmcnt" in s:
            seq.append("w" + re.search(r"vmcnt\((\d+)\)", s).group(1))
        elif s.startswith("s_cbranch"):
            seq.append("|")
        elif s.startswith("s_endpgm"):
            flush()
            cur, seq = None, []
    return out


def main():
    src = Path(sys.argv[1]).resolve()
    pat = sys.argv[2] if len(sys.argv) > 2 else ""
    with tempfile.TemporaryDirectory() as td:
        out = Path(td) / "k.s"
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-gpu-rdc",
                        "-S", "--cuda-device-only", str(src), "-o", str(out)], check=True, stderr=subprocess.DEVNULL, cwd=td)
        asm = out.read_text()
    try:
        names = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt"], input="\n".join(n for n, _ in tokens(asm, pat)), capture_output=True, text=True).stdout.splitlines()
    except OSError:
        names = []
    for i, (name, line) in enumerate(tokens(asm, pat)):
        print(names[i] if i < len(names) else name)
        print("   " + line)


if __name__ == "__main__":
    main()
