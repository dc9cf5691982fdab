#!/usr/bin/env python3
"""firfilt_rrrf / crcf short filters over 2^24 samples (GPU box): the register-window kernel (C1-shaped filters)."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
import yagi_amd as ya

n = 1 << 24
dev = torch.device("cuda")
x = torch.empty(n, dtype=torch.complex64, device=dev)
y = torch.empty(n, dtype=torch.complex64, device=dev)
st = torch.cuda.current_stream()
ya.gen_complex_dev(9, n, out=x, stream=st.cuda_stream)
for kind, L in (("rrrf", 63), ("rrrf", 31), ("rrrf", 127), ("crcf", 63), ("crcf", 31)):
    q = ya.FirFilter(kind, ya.fir_design_kaiser(L, 0.2, 60.0))
    q.set_stream(st.cuda_stream)
    xin, yout, bps = (x.view(torch.float32), y.view(torch.float32), 8) if kind == "rrrf" else (x, y, 16)
    fn = lambda: q.execute_block_dev(xin, n, yout)
    for _ in range(20):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(50):
        fn()
    e1.record(st)
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 50
    fl = 2 * L * (1 if kind == "rrrf" else 2)
    print(f"firfilt_{kind} L={L:4d}: {ms:8.4f} ms  {n / ms / 1e6:8.1f} Gsamples/s  {bps * n / ms / 1e6:8.1f} GB/s  {fl * n / ms / 1e9:6.1f} TF")
