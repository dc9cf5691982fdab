#!/usr/bin/env python3
"""sustained timings of the section-8(f) objects on 2^24 complex samples (GPU box)"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
import torch
import yagi_amd as ya

n = 1 << 24
dev = torch.device("cuda")
x = torch.empty(n, dtype=torch.complex64, device=dev)
y = torch.empty(2 * n, dtype=torch.complex64, device=dev)
st = torch.cuda.current_stream()
ya.gen_complex_dev(9, n, out=x, stream=st.cuda_stream)
h = ya.fir_design_kaiser(256, 0.2, 60.0)
cases = {}
for nfft in (1024, 4096):
    sp = ya.Spgram(nfft, ya.WindowType.Hann, nfft, nfft // 2)
    cases[f"spgram nfft={nfft} hann delay=nfft/2"] = (lambda sp=sp: sp.write_dev(x, n), 8 * n)
ff = ya.FftFilt("crcf", h, 2048)
cases["fftfilt_crcf 256 taps n=2048"] = (lambda: ff.execute_blocks_dev(x, n // 2048, y), 16 * n)
rr = ya.Rresamp.new_kaiser("crcf", 3, 5, 15, -1.0, 60.0)
cases["rresamp_crcf 3/5 m=15"] = (lambda: rr.execute_block_dev(x, n // 5, y), 8 * n + 8 * (n // 5) * 3)
fi = ya.FirInterpolationFilter.new_kaiser("crcf", 4, 8, 60.0)
cases["firinterp_crcf x4 m=8"] = (lambda: fi.execute_block_dev(x, n // 4, y), 8 * (n // 4) + 8 * n)
md = ya.MsResamp2("crcf", ya.MsResamp2.DECIM, 3, 0.45, 0.0, 60.0)
cases["msresamp2_crcf decim /8 (3 stages)"] = (lambda: md.execute_block_dev(x, n // 8, y), 8 * n + n)
mi = ya.MsResamp2("crcf", ya.MsResamp2.INTERP, 3, 0.45, 0.0, 60.0)
cases["msresamp2_crcf interp x8 (3 stages), n/8 in"] = (lambda: mi.execute_block_dev(x, n // 8, y), n + 8 * n)
for k, (fn, nbytes) in cases.items():
    for _ in range(30):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(50):
        fn()
    e1.record(st)
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 50
    print(f"{k:36s} {ms:8.4f} ms  {n / ms / 1e6:8.1f} Gsamples/s  {nbytes / ms / 1e6:8.1f} GB/s algorithmic")
