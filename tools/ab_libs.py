#!/usr/bin/env python3
"""A/B of builds and forms of the headline stream kernel in ONE process (GPU box): arms are interleaved round-robin
on the streamed workload (2^28 samples as 16 distinct 2^24-sample blocks), so clock drift and box-to-box spread cancel.
usage: python tools/ab_libs.py arm [arm ...]       arm = name[:lib=<so path or variant name>][:form=N][:fpw=N][:pipe=0|1]
Every arm is first checked against the f64 oracle on sampled frames."""
import ctypes as C
import os
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import numpy as np
import torch

import yagi_amd as ya
from oracle import oracle

ROUNDS = int(os.environ.get("AB_ROUNDS", "9"))
NB, NF = 16, 4096
n = NF * 4096
dev = torch.device("cuda")
st = torch.cuda.Stream()
torch.cuda.set_stream(st)
x = torch.empty(NB * n, dtype=torch.complex64, device=dev)
y = torch.empty(NB * n, dtype=torch.complex64, device=dev)
ya.gen_complex_dev(0x59414749 + 2, NB * n, out=x, stream=st.cuda_stream)
h = ya.fir_design_kaiser(256, 0.2, 60.0)
xp, yp = x.data_ptr(), y.data_ptr()
vp = C.c_void_p


def load(path):
    lib = C.CDLL(str(path))
    for name, args in (("create", (vp, C.c_size_t, C.c_size_t, C.POINTER(vp))), ("set_scale", (vp, C.c_float)),
                       ("set_variant", (vp, C.c_int)), ("set_stream", (vp, vp)), ("set_pipeline", (vp, C.c_int)),
                       ("execute_dev", (vp, vp, C.c_size_t, vp)), ("join", (vp,))):
        fn = getattr(lib, "yagi_hip_firfft_crcf_" + name)
        fn.argtypes, fn.restype = list(args), C.c_int
    return lib


class Arm:
    def __init__(self, spec):
        parts = spec.split(":")
        self.name = parts[0]
        kv = dict(p.split("=", 1) for p in parts[1:])
        lib = kv.get("lib", "")
        path = ROOT / "yagi_amd" / "libyagi_hip.so" if not lib else (
            Path(lib) if "/" in lib else ROOT / "yagi_amd" / "variants" / f"libyagi_{lib}.so")
        self.lib = load(path)
        os.environ["YAGI_FREQ_FORM"] = kv.get("form", "0")
        os.environ["YAGI_FREQ_FPW"] = kv.get("fpw", "1")
        self.q = vp()
        hh = np.ascontiguousarray(h, np.float32)
        assert self.lib.yagi_hip_firfft_crcf_create(hh.ctypes.data_as(vp), hh.size, 4096, C.byref(self.q)) == 0
        L = self.lib
        assert L.yagi_hip_firfft_crcf_set_scale(self.q, 0.4) == 0
        assert L.yagi_hip_firfft_crcf_set_variant(self.q, 4) == 0
        assert L.yagi_hip_firfft_crcf_set_stream(self.q, vp(st.cuda_stream)) == 0
        self.pipe = int(kv.get("pipe", "1"))
        assert L.yagi_hip_firfft_crcf_set_pipeline(self.q, self.pipe) == 0
        self.t = []

    def run(self):
        ex = self.lib.yagi_hip_firfft_crcf_execute_dev
        for b in range(NB):
            rc = ex(self.q, vp(xp + 8 * n * b), NF, vp(yp + 8 * n * b))
            assert rc == 0, rc
        self.lib.yagi_hip_firfft_crcf_join(self.q)

    def check(self):
        y.zero_()
        self.run()
        torch.cuda.synchronize()
        worst = 0.0
        for f in (0, 1, 4095, 4096, 4097, 9 * 4096 + 1234, NB * NF - 1):
            lo = max(0, f * 4096 - 255)
            xs = x[lo:(f + 1) * 4096].cpu().numpy()
            if f == 0:
                xs = np.concatenate([np.zeros(255, np.complex64), xs])
            truth = np.fft.fft(oracle.fir_block_f64("crcf", h, xs, scale=0.4)[-4096:])
            got = y[f * 4096:(f + 1) * 4096].cpu().numpy()
            worst = max(worst, float(np.linalg.norm(got - truth) / np.linalg.norm(truth)))
        return worst


arms = [Arm(s) for s in sys.argv[1:]]
for a in arms:
    e = a.check()
    print(f"{a.name:14s} rel L2 vs f64 oracle (7 frames) {e:.3e} {'OK' if e < 1e-5 else 'FAIL'}", flush=True)
for _ in range(3):
    for a in arms:
        a.run()
torch.cuda.synchronize()
for r in range(ROUNDS):
    for a in (arms if r % 2 == 0 else arms[::-1]):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        for _ in range(4):
            a.run()
        e1.record(st)
        torch.cuda.synchronize()
        a.t.append(e0.elapsed_time(e1) / 4 / NB * 1e3)
base = float(np.median(arms[0].t))
for a in arms:
    med = float(np.median(a.t))
    print(f"{a.name:14s} median {med:6.2f} us / 2^24 (min {min(a.t):6.2f})  {16 * n / med / 1e6 / 8:.3f} of HBM peak  "
          f"{(med / base - 1) * 100:+5.1f} % vs {arms[0].name}", flush=True)
