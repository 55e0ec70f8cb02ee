"""The binding a reference maintainer would write (a LibNpp-style table of dlsym'd
entry points, reference src/TC/inc/LibNpp.hpp:35-198 + src/TC/inc/LibraryLoader.hpp:38-68) GENERATED from
include/vali_hip.h: every VALI_API symbol is looked up with dlsym and cast to decltype(&symbol) of the header's
own prototype, compiled with g++ and run without a GPU.  Also: every vali_* name INTEGRATION.md / README.md
mention exists in the header, so the documents cannot name stale symbols."""
import re
import shutil
import subprocess
from pathlib import Path

import pytest

from test_abi_symbols import declared_symbols

ROOT = Path(__file__).resolve().parent.parent


@pytest.mark.skipif(shutil.which("g++") is None, reason="g++ not available")
def test_generated_loader_table_resolves_every_symbol_with_the_headers_prototype(tmp_path):
    syms = declared_symbols()
    assert len(syms) == len(set(syms)) >= 40
    lines = ['#include <dlfcn.h>', '#include <cstdio>', '#include "vali_hip.h"', '',
             'template <typename F> static F load(void* lib, const char* name, int& missing) {',
             '  void* p = dlsym(lib, name);',
             '  if (!p) { std::fprintf(stderr, "missing %s\\n", name); ++missing; }',
             '  return reinterpret_cast<F>(p);',
             '}', '',
             'struct LibVali {']
    lines += [f'  decltype(&::{s}) {s}_;' for s in syms]
    lines += ['};', '', 'int main(int argc, char** argv) {',
              '  void* lib = dlopen(argc > 1 ? argv[1] : "libvali_hip.so", RTLD_NOW);',
              '  if (!lib) { std::fprintf(stderr, "%s\\n", dlerror()); return 2; }',
              '  int missing = 0;', '  LibVali t = {};']
    lines += [f'  t.{s}_ = load<decltype(&::{s})>(lib, "{s}", missing);' for s in syms]
    lines += ['  if (missing) return 1;',
              '  // a few calls that need no device: the table is usable as it stands',
              '  int v = -1;',
              '  if (t.vali_tuning_get_(VALI_TUNE_UD_DOWN2, &v) != VALI_OK || v != 1) return 3;',
              '  if (t.vali_nv12_to_rgb_(nullptr, nullptr, nullptr, nullptr) != VALI_ERR_INVALID_ARG) return 4;',
              '  if (t.vali_resize_(nullptr, nullptr, VALI_INTERP_LANCZOS, nullptr) != VALI_ERR_INVALID_ARG) return 5;',
              '  if (!t.vali_version_() || !t.vali_last_error_()) return 6;',
              f'  std::printf("%d symbols\\n", {len(syms)});', '  return 0;', '}']
    src = tmp_path / "libvali_table.cpp"
    src.write_text("\n".join(lines) + "\n")
    exe = tmp_path / "libvali_table"
    subprocess.run(["g++", "-std=c++17", "-Wall", "-Wextra", "-Werror", f"-I{ROOT / 'include'}", str(src), "-o", str(exe),
                    "-ldl"], check=True)
    lib = ROOT / "vali_amd" / "libvali_hip.so"
    r = subprocess.run([str(exe), str(lib)], capture_output=True, text=True, env={"LD_LIBRARY_PATH": "/opt/rocm/lib", "PATH": "/usr/bin:/bin"})
    assert r.returncode == 0, r.stdout + r.stderr
    assert f"{len(syms)} symbols" in r.stdout


@pytest.mark.parametrize("doc", ["INTEGRATION.md", "README.md", "DESIGN.md"])
def test_documents_only_name_symbols_the_header_declares(doc):
    declared = set(declared_symbols())
    text = (ROOT / doc).read_text()
    # vali_* C identifiers (types and enum prefixes of the header are fine too)
    header = (ROOT / "include" / "vali_hip.h").read_text()
    known = declared | set(re.findall(r"\b(vali_\w+)\b", header)) | {"vali_amd", "vali_hip", "vali_oracle"}
    named = set(re.findall(r"\b(vali_[a-z0-9_]+)\b", text))
    stale = sorted(n for n in named if n not in known and not n.startswith(("vali_amd", "vali_oracle", "vali_hip")))
    assert not stale, f"{doc} names symbols that include/vali_hip.h does not declare: {stale}"
