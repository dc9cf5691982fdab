#!/usr/bin/env python3
"""kernel micro-bench (GPU box): per-kernel time over interleaved rounds in ONE process.
usage: [YAGI_HIP_LIB=...] python tools/kbench.py [what ...]   what in fft fir1 fir2 fused chan1 chan2 decim"""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
import torch

import yagi_amd as ya

SUSTAIN = "--sustain" in sys.argv          # steady-state (DVFS-settled) timing: 300 launches, last 100 timed
what = [a for a in sys.argv[1:] if not a.startswith("--")] or ["fft", "fir2", "fused"]
n = 1 << 24
dev = torch.device("cuda")
x = torch.empty(n, dtype=torch.complex64, device=dev)
y = torch.empty(2 * n, dtype=torch.complex64, device=dev)
st = torch.cuda.current_stream()
ya.gen_complex_dev(0x59414749 + 2, n, out=x, stream=st.cuda_stream)
h = ya.fir_design_kaiser(256, 0.2, 60.0)
cases = {}
if "fft" in what:
    plan = ya.Fft(4096, ya.Direction.Forward)
    cases["fft4096 x4096"] = (lambda: plan.run_batch_dev(x, y, n // 4096, st.cuda_stream), 16 * n)
for name, k in (("fir1", 1), ("fir2", 2), ("fir3", 3), ("fir4", 4)):
    if name in what:
        q = ya.FirFilter("crcf", h)
        q.set_kernel(k)
        q.set_stream(st.cuda_stream)
        cases[f"firfilt_crcf256 kernel{k}"] = (lambda q=q: q.execute_block_dev(x, n, y), 16 * n)
for name, v in (("fused", 1), ("fused2", 2), ("fused3", 3), ("fused4", 4)):
    if name in what:
        f = ya.FirFftStream(h)
        f.set_stream(st.cuda_stream)
        f.set_variant(v)
        cases[f"fused fir256+fft4096 v{v}"] = (lambda f=f: f.execute_dev(x, n // 4096, y), 16 * n)
if "rrrf" in what:      # C1 filter at GPU block size: 63-tap real FIR over 2^24 real samples
    hr = ya.fir_design_kaiser(63, 0.2, 60.0)
    qr = ya.FirFilter("rrrf", hr)
    qr.set_stream(st.cuda_stream)
    xr = x.view(torch.float32)
    yr = y.view(torch.float32)
    cases["firfilt_rrrf 63-tap"] = (lambda: qr.execute_block_dev(xr, n, yr), 8 * n)
    qr2 = ya.FirFilter("rrrf", h)
    qr2.set_stream(st.cuda_stream)
    cases["firfilt_rrrf 256-tap"] = (lambda: qr2.execute_block_dev(xr, n, yr), 8 * n)
if "cccf" in what:
    hc = (h * np.exp(0.3j * np.arange(h.size))).astype(np.complex64)
    qc = ya.FirFilter("cccf", hc)
    qc.set_stream(st.cuda_stream)
    cases["firfilt_cccf 256-tap"] = (lambda: qc.execute_block_dev(x, n, y), 16 * n)
if "chan1" in what:
    c1 = ya.FirPfbCh.new_kaiser(64, 8, 60.0)
    c1.set_stream(st.cuda_stream)
    cases["firpfbch M=64 m=8"] = (lambda: c1.analyzer_execute_dev(x, n // 64, y), 16 * n)
if "chan2" in what:
    c2 = ya.FirPfbCh2.new_kaiser(256, 4, 60.0)
    c2.set_stream(st.cuda_stream)
    cases["firpfbch2 M=256 m=4"] = (lambda: c2.analyzer_execute_dev(x, n // 128, y), 24 * n)
if "decim" in what or "decim2" in what:
    d = ya.FirDecimationFilter.new_kaiser("crcf", 4, 8, 60.0)
    d.set_stream(st.cuda_stream)
    cases["firdecim_crcf M=4 L=65"] = (lambda: d.execute_block_dev(x, n // 4, y), 10 * n)
if "fft_big" in what:      # config C3: 65 536 transforms of 4096 points, 2 GiB in + 2 GiB out
    nb = 65536 * 4096
    xb = torch.empty(nb, dtype=torch.complex64, device=dev)
    yb = torch.empty(nb, dtype=torch.complex64, device=dev)
    ya.gen_complex_dev(0x59414749 + 3, nb, out=xb, stream=st.cuda_stream)
    planb = ya.Fft(4096, ya.Direction.Forward)
    cases["fft4096 x65536 (C3)"] = (lambda: planb.run_batch_dev(xb, yb, 65536, st.cuda_stream), 16 * nb)
    BIG = {"fft4096 x65536 (C3)": nb}
else:
    BIG = {}
for fn, _ in cases.values():
    fn()
torch.cuda.synchronize()
if SUSTAIN:
    for k, (fn, nbytes) in cases.items():
        for _ in range(200):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        for _ in range(100):
            fn()
        e1.record(st)
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 100
        nn = BIG.get(k, n)
        print(f"{k:28s} sustained {ms:8.4f} ms  {nn / ms / 1e6:9.1f} Gsamples/s  {nbytes / ms / 1e6:8.1f} GB/s algorithmic")
    sys.exit(0)
res = {k: [] for k in cases}
for rnd in range(5):
    for k, (fn, _) in cases.items():
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        for _ in range(10):
            fn()
        e1.record(st)
        torch.cuda.synchronize()
        res[k].append(e0.elapsed_time(e1) / 10)
for k, v in res.items():
    ms = float(np.median(v))
    nn = BIG.get(k, n)
    print(f"{k:28s} median {ms:8.4f} ms  min {min(v):8.4f}  {nn / ms / 1e6:9.1f} Gsamples/s  {cases[k][1] / ms / 1e6:8.1f} GB/s algorithmic")
