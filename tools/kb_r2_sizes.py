#!/usr/bin/env python3
"""Resamp2 decimator and MsResamp2 /8 across block lengths (GPU box): where the long-block kernels take over (2^19 input
samples / 2^18 outputs) there must be no step up in the time per call."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np, torch
import yagi_amd as ya
dev = torch.device("cuda")
x = torch.empty(1 << 24, dtype=torch.complex64, device=dev)
y = torch.empty(1 << 23, dtype=torch.complex64, device=dev)
st = torch.cuda.current_stream()
ya.gen_complex_dev(5, 1 << 24, out=x, stream=st.cuda_stream)
r2 = ya.Resamp2.new("crcf", 12, 0.0, 60.0); r2.set_stream(st.cuda_stream)
ms3 = ya.MsResamp2("crcf", ya.MsResamp2.DECIM, 3, 0.4, 0.0, 60.0); ms3.set_stream(st.cuda_stream)
def timed(fn, reps=200):
    for _ in range(20): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for lg in (17, 18, 19, 20, 21, 22, 24):
    n = 1 << lg
    t = timed(lambda: r2.execute_block_dev(r2.DECIM, x, n, y))
    t3 = timed(lambda: ms3.execute_block_dev(x, n // 8, y))
    print(f"n=2^{lg}: resamp2 decim {t:8.2f} us ({t*1e3/n:.3f} ns/sample)   msresamp2/8 {t3:8.2f} us ({t3*1e3/n:.3f} ns/sample)")
