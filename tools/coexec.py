#!/usr/bin/env python3
"""experiment: do the MFMA FIR kernel and the VALU sliding FIR kernel co-execute when launched on two
streams over two halves of the data?  (time vs each alone on half, and vs sum)"""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np, torch
import yagi_amd as ya
n = 1 << 24
dev = torch.device("cuda")
x = torch.empty(n, dtype=torch.complex64, device=dev)
y = torch.empty(n, dtype=torch.complex64, device=dev)
s0 = torch.cuda.current_stream()
ya.gen_complex_dev(1, n, out=x, stream=s0.cuda_stream)
h = ya.fir_design_kaiser(256, 0.2, 60.0)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
qa = ya.FirFilter("crcf", h); qa.set_kernel(3); qa.set_stream(s1.cuda_stream)
qb = ya.FirFilter("crcf", h); qb.set_kernel(2); qb.set_stream(s2.cuda_stream)
half = n // 2
xa, ya_ = x.data_ptr(), y.data_ptr()
xb, yb = xa + half * 8, ya_ + half * 8
def run(mode, reps=20):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        if mode in ("mfma", "both"): qa.execute_block_dev(xa, half, ya_)
        if mode in ("valu", "both"): qb.execute_block_dev(xb, half, yb)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3
for m in ("mfma", "valu", "both"): run(m, 3)
for rnd in range(3):
    print({m: round(run(m), 4) for m in ("mfma", "valu", "both")})
