"""Build the native parts of vali_amd in-tree (no network, no cmake needed).

  libvali_hip.so      hand-written gfx950 HIP kernels + the C ABI of include/vali_hip.h
  _vali_shim*.so      thin pybind11 module binding that C ABI 1:1 (plus DLPack capsules)
  oracle/libvali_oracle.so   CPU restatement used by tests / smoke / bench cpu_baseline only

hipcc cross-compiles for gfx950 without a GPU.  Everything is rebuilt only when a
source is newer than its product.  Usage: ``python -m vali_amd.build [--force]``.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
import sysconfig
from pathlib import Path

PKG = Path(__file__).resolve().parent
ROOT = PKG.parent
CSRC = PKG / "csrc"
OBJ = PKG / "csrc" / "_obj"
ORACLE = ROOT / "oracle"

ROCM = Path(os.environ.get("ROCM_PATH", "/opt/rocm"))
HIPCC = os.environ.get("HIPCC", str(ROCM / "bin" / "hipcc"))
ARCH = "gfx950"

# -ffp-contract=off: FMA is used only where the kernels say __builtin_fmaf, so GPU
# results are bit-identical to the C oracle (built with the same flag).
HIP_FLAGS = [
    f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
    "-fvisibility=hidden", "-Wall", "-Wno-unused-function", "-fno-gpu-rdc",
]


def _run(cmd):
    print("  $", " ".join(str(c) for c in cmd), file=sys.stderr, flush=True)  # stdout stays clean (bench.py prints ONE JSON line)
    subprocess.run([str(c) for c in cmd], check=True, stdout=sys.stderr)


def _compile_kernel(job):
    """One translation unit; the compiler's per-kernel resource remarks (registers, scratch, occupancy) go to
    <obj>.resources.txt -- tests/test_kernel_resources.py refuses scratch memory in any kernel: a rolled loop over a
    register array once moved a whole struct to scratch and cost every packed-destination converter half its speed
    without failing a single parity test."""
    cmd, report = job
    print("  $", " ".join(str(c) for c in cmd), file=sys.stderr, flush=True)
    r = subprocess.run([str(c) for c in cmd] + ["-Rpass-analysis=kernel-resource-usage", "-fno-caret-diagnostics"], stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, text=True)
    remarks = [l for l in r.stderr.splitlines() if "-Rpass-analysis=kernel-resource-usage" in l]
    other = [l for l in r.stderr.splitlines() if "-Rpass-analysis=kernel-resource-usage" not in l]
    if other:
        print("\n".join(other), file=sys.stderr, flush=True)
    if r.returncode != 0:
        raise subprocess.CalledProcessError(r.returncode, cmd)
    Path(report).write_text("\n".join(remarks) + "\n")


def _stale(product: Path, sources) -> bool:
    if not product.exists():
        return True
    t = product.stat().st_mtime
    return any(Path(s).stat().st_mtime > t for s in sources)


def lib_path() -> Path:
    return PKG / "libvali_hip.so"


def shim_path() -> Path:
    return PKG / ("_vali_shim" + sysconfig.get_config_var("EXT_SUFFIX"))


def oracle_path() -> Path:
    return ORACLE / "libvali_oracle.so"


def build_kernels(force=False):
    OBJ.mkdir(exist_ok=True)
    headers = list(CSRC.glob("*.hpp")) + [ROOT / "include" / "vali_hip.h"]
    objs, todo = [], []
    for src in sorted(CSRC.glob("*.hip")):
        obj = OBJ / (src.stem + ".o")
        report = OBJ / (src.stem + ".resources.txt")
        if force or _stale(obj, [src] + headers) or not report.exists():
            todo.append(([HIPCC, *HIP_FLAGS, "-c", src, "-o", obj], report))
        objs.append(obj)
    if todo:   # the translation units are independent: compile them side by side
        from concurrent.futures import ThreadPoolExecutor

        with ThreadPoolExecutor(max_workers=min(len(todo), os.cpu_count() or 1)) as pool:
            list(pool.map(_compile_kernel, todo))
    lib = lib_path()
    if force or _stale(lib, objs):
        _run([HIPCC, f"--offload-arch={ARCH}", "-shared", "-fPIC", *objs, "-o", lib,
              f"-Wl,-rpath,{ROCM / 'lib'}", "-Wl,-soname,libvali_hip.so"])
    return lib


def build_shim(force=False):
    import pybind11

    src = CSRC / "pyshim.cpp"
    out = shim_path()
    deps = [src, ROOT / "include" / "vali_hip.h", CSRC / "dlpack_min.h"]
    if force or _stale(out, deps):
        _run(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-fvisibility=hidden",
              f"-I{pybind11.get_include()}", f"-I{sysconfig.get_paths()['include']}",
              f"-I{ROOT / 'include'}", src, "-o", out,
              f"-L{PKG}", "-lvali_hip", "-Wl,-rpath,$ORIGIN"])
    return out


def build_oracle(force=False):
    srcs = sorted(ORACLE.glob("vali_oracle*.c"))
    out = oracle_path()
    if not srcs:
        return None
    if force or _stale(out, srcs + [ORACLE / "vali_oracle.h", ROOT / "include" / "vali_hip.h"]):
        # -mfma so fmaf() is the hardware instruction (same value either way);
        # -ffp-contract=off so nothing else gets fused.  AVX2+FMA exists on every
        # x86-64 host an MI355X ships in.
        _run(["gcc", "-O3", "-std=c11", "-D_GNU_SOURCE", "-fPIC", "-shared", "-mavx2", "-mfma",
              "-ffp-contract=off", "-fno-math-errno", "-fopenmp", *srcs, "-o", out, "-lm"])
    return out


def build_all(force=False):
    if shutil.which(HIPCC) is None and not Path(HIPCC).exists():
        raise RuntimeError(f"hipcc not found at {HIPCC}")
    build_kernels(force)
    build_shim(force)
    build_oracle(force)


if __name__ == "__main__":
    build_all(force="--force" in sys.argv)
    print("ok")
