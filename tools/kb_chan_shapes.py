#!/usr/bin/env python3
"""channelizer shape sweep (GPU box): firpfbch / firpfbch2 analyzers over 2^24 input samples"""
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
shapes = [(8, 4), (16, 4), (32, 4), (64, 8), (128, 4), (256, 4), (512, 4), (1024, 2), (48, 4), (100, 4)]
for M, m in shapes:
    for kind in ("ch", "ch2"):
        if kind == "ch":
            c = ya.FirPfbCh.new_kaiser(M, m, 60.0)
            fn = lambda c=c, M=M: c.analyzer_execute_dev(x, n // M, y)
            nbytes = 16 * n
        else:
            c = ya.FirPfbCh2.new_kaiser(M, m, 60.0)
            fn = lambda c=c, M=M: c.analyzer_execute_dev(x, n // (M // 2), y)
            nbytes = 24 * n
        c.set_stream(st.cuda_stream)
        for _ in range(10):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        for _ in range(20):
            fn()
        e1.record(st)
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        print(f"firpfb{kind:3s} M={M:5d} m={m}: {ms:8.4f} ms  {n / ms / 1e6:8.1f} Gsamples/s  {nbytes / ms / 1e6:8.1f} GB/s")
