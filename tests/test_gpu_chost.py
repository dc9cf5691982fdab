"""The boundary from the other side: a plain C program (tests/c/capi_host.c, compiled with gcc, linked against
libyagi_hip.so only) drives the headline stream and a FirFilter block; its results must equal what the Python
mirror computes through the same C ABI on the same generated input."""
import subprocess

import numpy as np
import pytest

from conftest import ROOT
from gpu_util import SEED

pytestmark = pytest.mark.gpu


def test_c_host_matches_python_mirror(tmp_path):
    import yagi_amd as ya
    exe = tmp_path / "capi_host"
    libdir = ROOT / "yagi_amd"
    subprocess.check_call(["gcc", "-std=c99", "-O1", "-Wall", "-Werror", str(ROOT / "tests" / "c" / "capi_host.c"),
                           f"-I{ROOT / 'include'}", f"-L{libdir}", "-lyagi_hip", f"-Wl,-rpath,{libdir}", "-lm", "-o", str(exe)])
    # the C host must not need Python's torch runtime: it resolves libamdhip64 through the library's RUNPATH
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr + out.stdout
    vals = {l.split()[0]: [float(v) for v in l.split()[1:]] for l in out.stdout.splitlines()}
    nfft, nframes = 4096, 64
    n = nfft * nframes
    h = ya.fir_design_kaiser(256, 0.2, 60.0)
    dx = ya.gen_complex_dev(SEED + 2, n)
    q = ya.FirFftStream(h)
    q.set_scale(0.4)
    dy = ya.DeviceArray(n, np.complex64)
    q.execute_dev(dx, nframes, dy)
    ya.synchronize()
    spec = dy.to_numpy()
    e = float(np.sum(np.abs(spec).astype(np.float64) ** 2))
    assert abs(vals["stream_energy"][0] / e - 1.0) <= 1e-6
    b = spec[33 * nfft + 100]
    assert abs(complex(*vals["stream_bin"]) - b) <= 1e-4 * (1 + abs(b))
    f = ya.FirFilter("crcf", h)
    f.set_scale(0.4)
    f.execute_block_dev(dx, n, dy)
    ya.synchronize()
    y7 = dy.to_numpy(8)[7]
    assert abs(complex(*vals["fir_y7"]) - y7) <= 1e-6 * (1 + abs(y7))
    assert vals["fir_len"] == [256.0] and vals["fft0_status"] == [2.0]
    # a per-sample execute_one() is a host-side sequential sum (256 taps), not a kernel launch
    assert vals["per_sample_ns"][0] < 1000.0, vals["per_sample_ns"]
