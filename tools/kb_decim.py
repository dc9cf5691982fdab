#!/usr/bin/env python3
"""decimator micro-bench (GPU box): python tools/kb_decim.py  -- a few (M, L) shapes over 2^24 input samples, crcf
and rrrf, sustained timing.  Run with YAGI_HIP_DECIM_WINDOW_MIN_STEPS=0 / =100000 to compare the two kernels."""
import os
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
import torch

import yagi_amd as ya

n = 1 << 24
dev = torch.device("cuda")
x = torch.empty(n, dtype=torch.complex64, device=dev)
y = torch.empty(n, dtype=torch.complex64, device=dev)
st = torch.cuda.current_stream()
ya.gen_complex_dev(0x59414749 + 2, n, out=x, stream=st.cuda_stream)
print("YAGI_HIP_DECIM_WINDOW_MIN_STEPS =", os.environ.get("YAGI_HIP_DECIM_WINDOW_MIN_STEPS"))
for kind in ("crcf", "rrrf"):
    for M, L in ((2, 33), (2, 129), (2, 513), (4, 65), (4, 129), (4, 257), (4, 1025), (8, 129), (8, 513), (8, 2049)):
        h = ya.fir_design_kaiser(L, 0.4 / M, 60.0)
        d = ya.FirDecimationFilter(kind, M, h)
        d.set_stream(st.cuda_stream)
        xin = x if kind == "crcf" else x.view(torch.float32)
        yo = y if kind == "crcf" else y.view(torch.float32)
        nin = n if kind == "crcf" else 2 * n
        fn = lambda: d.execute_block_dev(xin, nin // M, yo)
        for _ in range(60):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 50
        print(f"firdecim_{kind} M={M:<2d} L={L:<5d} {ms:8.4f} ms  {nin / ms / 1e6:8.1f} Gsamples/s in")
