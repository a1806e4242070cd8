"""CPU-side checks of the drop-in boundary: the library builds/loads and exports every symbol that
include/zen_hip.h declares (no compute calls: there is no GPU here)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    hdr = open(os.path.join(ROOT, "include", "zen_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(zen_hip_[a-z0-9_]+)\s*\(", hdr)))


def test_header_symbols_are_bound_and_exported():
    from zen_amd import lib
    so = os.path.join(ROOT, "zen_amd", "libzen_hip.so")
    if not os.path.exists(so):
        from zen_amd import build
        build.build()
    L = ctypes.CDLL(so)
    names = declared_symbols()
    assert len(names) >= 40
    for n in names:
        assert hasattr(L, n), "libzen_hip.so does not export %s" % n
    bound = {s[0] for s in lib.SYMBOLS}
    assert set(names) == bound, (set(names) ^ bound)


def test_library_reports_version_without_gpu():
    import zen_amd
    L = zen_amd.load()
    assert b"gfx950" in L.zen_hip_version()


def test_header_cites_reference_interfaces():
    hdr = open(os.path.join(ROOT, "include", "zen_hip.h")).read()
    for cite in ("libzen/fftw.h:20-49", "libzen/mfilt.h:33-268", "libzen/box.h:30-215",
                 "libzen/libzen/io.h:16-81", "libzen/hps.h:152-322", "libzen/hps.cu:21-221"):
        assert cite in hdr


def test_header_compiles_as_c():
    import subprocess
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        src = os.path.join(d, "t.c")
        open(src, "w").write('#include "zen_hip.h"\nint main(void){return ZEN_HIP_OK;}\n')
        subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"),
                               "-c", src, "-o", os.path.join(d, "t.o")])
