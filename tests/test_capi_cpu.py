"""CPU-side checks of the drop-in boundary: the C-ABI library loads, exports every symbol the
header declares, fails loudly (no CPU fallback) when no GPU is present, and its design-time host
code agrees with the oracle.  No compute kernels are launched here."""
import re
import subprocess
from pathlib import Path

import numpy as np
import pytest

from conftest import ROOT, has_gpu

HEADER = ROOT / "include" / "yagi_hip.h"
LIB = ROOT / "yagi_amd" / "libyagi_hip.so"


def declared_symbols():
    """every function the header declares, macros expanded (the C preprocessor does the expansion)"""
    pre = subprocess.check_output(["gcc", "-E", "-P", "-x", "c", str(HEADER)], text=True)
    return set(re.findall(r"\b(yagi_hip_[a-z0-9_]+)\s*\(", pre))


def test_header_is_plain_c(tmp_path):
    """the boundary is a C ABI: the header must compile as C99 (and as C++) with nothing but itself"""
    src = tmp_path / "t.c"
    src.write_text('#include "yagi_hip.h"\nint main(void) { yagi_hip_fft p = 0; (void)p; return 0; }\n')
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", f"-I{HEADER.parent}", "-c",
                           str(src), "-o", str(tmp_path / "t.o")])
    subprocess.check_call(["g++", "-std=c++11", "-Wall", "-Werror", f"-I{HEADER.parent}", "-x", "c++", "-c", str(src),
                           "-o", str(tmp_path / "t2.o")])


def test_library_exports_every_declared_symbol():
    assert LIB.exists(), "run __graft_entry__.build() first"
    out = subprocess.check_output(["nm", "-D", "--defined-only", str(LIB)], text=True)
    exported = {l.split()[-1] for l in out.splitlines()}
    decl = declared_symbols()
    assert len(decl) > 330
    assert not (decl - exported), sorted(decl - exported)


def test_library_contains_gfx950_code_object():
    out = subprocess.run(["/opt/rocm/lib/llvm/bin/clang-offload-bundler", "--list", "--type=o",
                          f"--input={LIB}"], capture_output=True, text=True)
    blob = LIB.read_bytes()
    assert b"gfx950" in blob
    assert b"gfx942" not in blob and b"sm_" not in blob[:0]  # single target, no dual back-ends


def test_ctypes_binding_loads_and_reports_version():
    import yagi_amd
    assert b"gfx950" in yagi_amd.lib.yagi_hip_version()


def test_kaiser_design_matches_oracle(oracle):
    """host design code (kaiser.rs:16-51) == oracle restatement, incl. its error cases"""
    import yagi_amd
    for n, fc, as_, mu in [(63, 0.2, 60.0, 0.0), (256, 0.2, 60.0, 0.0), (51, 0.2, 60.0, 0.0),
                           (1025, 0.5 / 64, 60.0, 0.0), (2049, 1 / 256, 60.0, 0.0), (17, 0.1, 30.0, 0.25),
                           (9, 0.4, 15.0, -0.3)]:
        a = yagi_amd.fir_design_kaiser(n, fc, as_, mu)
        b = oracle.fir_design_kaiser(n, fc, as_, mu)
        np.testing.assert_allclose(a, b, rtol=2e-6, atol=1e-9)
    for bad in [(0, 0.2, 60.0, 0.0), (10, 0.0, 60.0, 0.0), (10, 0.6, 60.0, 0.0), (10, 0.2, 0.0, 0.0),
                (10, 0.2, 60.0, 0.6), (10, 0.2, 60.0, -0.5)]:
        with pytest.raises(yagi_amd.ConfigError):
            yagi_amd.fir_design_kaiser(*bad)


@pytest.mark.skipif(has_gpu(), reason="checks the no-GPU behaviour")
def test_no_cpu_fallback_without_gpu():
    """every object constructor and one-shot op must FAIL (DeviceError), never compute on the CPU"""
    import yagi_amd as ya
    assert ya.device_count() == 0
    with pytest.raises(ya.DeviceError):
        ya.FirFilter("crcf", np.ones(4, np.float32))
    with pytest.raises(ya.DeviceError):
        ya.FirDecimationFilter("rrrf", 2, np.ones(4, np.float32))
    with pytest.raises(ya.DeviceError):
        ya.FirPfbFilter("cccf", 2, np.ones(8, np.complex64))
    with pytest.raises(ya.DeviceError):
        ya.Fft(64, ya.Direction.Forward)
    with pytest.raises(ya.DeviceError):
        ya.dotprod(np.ones(3, np.float32), np.ones(3, np.float32))
    with pytest.raises(ya.DeviceError):
        ya.FirFftStream(np.ones(8, np.float32))
    with pytest.raises(ya.DeviceError):
        ya.FirPfbCh2.new_kaiser(8, 2, 60.0)


def test_product_does_not_touch_the_oracle():
    """yagi_amd/ must not import, link or call anything under oracle/"""
    for p in (ROOT / "yagi_amd").rglob("*"):
        if p.suffix in (".py", ".hip", ".cpp", ".hpp", ".h") or p.name == "Makefile":
            txt = p.read_text()
            assert "yo_" not in txt and "libyagi_oracle" not in txt, p
            assert not re.search(r"^\s*(from|import)\s+oracle", txt, re.M), p
    out = subprocess.check_output(["readelf", "-d", str(LIB)], text=True)
    assert "oracle" not in out
