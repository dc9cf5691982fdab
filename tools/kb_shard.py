#!/usr/bin/env python3
"""firpfbch2 256-channel analyzer, one rank's sub-band shard (config C5 per-GPU work) vs the unsharded kernel"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np, torch, yagi_amd as ya
n = 1 << 24
dev = torch.device("cuda")
x = torch.empty(n, dtype=torch.complex64, device=dev); y = torch.empty(2 * n, dtype=torch.complex64, device=dev)
st = torch.cuda.current_stream()
ya.gen_complex_dev(9, n, out=x, stream=st.cuda_stream)
M, m = 256, 4
ns = n // (M // 2)
for R in (1, 2, 4, 8):
    c = ya.FirPfbCh2.new_kaiser(M, m, 60.0); c.set_stream(st.cuda_stream)
    fn = (lambda: c.analyzer_execute_dev(x, ns, y)) if R == 1 else (lambda: c.analyzer_execute_shard_dev(x, ns, R - 1, R, y))
    for _ in range(10): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(20): fn()
    e1.record(st); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    nbytes = 8 * n + 16 * n / R
    print(f"firpfbch2 M=256 m=4 shard 1/{R}: {ms:8.4f} ms  {n / ms / 1e6:8.1f} Gsamples/s in  {nbytes / ms / 1e6:8.1f} GB/s algorithmic")
