"""No kernel may use scratch (private) memory or spill registers.

Scratch is silent: results stay bit-exact and only the speed goes.  A `for (r = 0; r < (has_row1 ? 2 : 1); ++r)` over a
register array stayed rolled after an unrelated change, the array moved to scratch, and every converter with a packed
RGB / BGR destination ran at 2.5-4.3 TB/s instead of 5.9-6.2 for most of round 2 (tools/cliffs.py found it).  The build
keeps the compiler's per-kernel resource remarks next to the objects (vali_amd/build.py); this test reads them.
"""
import re
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
OBJ = ROOT / "vali_amd" / "csrc" / "_obj"

ALLOWED = set()   # (round 2 had one A/B instantiation that spilled; it also turned out to mis-render and was removed)


def kernels():
    out = {}
    for f in sorted(OBJ.glob("*.resources.txt")):
        name = None
        for line in f.read_text().splitlines():
            m = re.search(r"Function Name: (\S+)", line)
            if m:
                name = m.group(1)
                out[name] = {"file": f.name}
                continue
            m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (-?\d+)", line)
            if m and name:
                out[name][m.group(1).strip()] = int(m.group(2))
    return out


def test_resource_reports_exist_for_every_translation_unit():
    srcs = sorted((ROOT / "vali_amd" / "csrc").glob("*.hip"))
    assert srcs
    for s in srcs:
        assert (OBJ / (s.stem + ".resources.txt")).exists(), f"build did not leave a resource report for {s.name}"
    assert len(kernels()) > 100


def test_no_kernel_uses_scratch_or_spills():
    bad = []
    for name, r in kernels().items():
        if name in ALLOWED:
            continue
        if r.get("ScratchSize", 0) or r.get("VGPRs Spill", 0) or r.get("SGPRs Spill", 0) > 128:
            bad.append((name, r))
    assert not bad, "kernels with scratch / spills:\n" + "\n".join(f"{n}: {r}" for n, r in bad)
