#!/usr/bin/env python3
"""Synthesis channelizers (GPU box): firpfbch (M channel samples -> M samples per frame, 16 B per sample) and firpfbch2
(M channel samples -> M/2 samples per step, 12 B per channel sample) over 2^24 channel samples."""
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


def timed(fn):
    for _ in range(10):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(20):
        fn()
    e1.record(st)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 20


for M, m in [(16, 4), (64, 8), (256, 4), (1024, 2), (100, 4)]:
    c = ya.FirPfbCh.new_kaiser(M, m, 60.0)
    c.set_stream(st.cuda_stream)
    ms = timed(lambda: c.synthesizer_execute_dev(x, n // M, y))
    print(f"firpfbch  synthesizer M={M:5d} m={m}: {ms:8.4f} ms  {n / ms / 1e6:8.1f} Gsamples/s  {16 * n / ms / 1e6:8.1f} GB/s")
for M, m in [(16, 4), (64, 2), (64, 3), (64, 4), (128, 4), (256, 2), (256, 4), (1024, 2), (100, 4)]:
    c = ya.FirPfbCh2.new_kaiser_synthesizer(M, m, 60.0)
    c.set_stream(st.cuda_stream)
    ms = timed(lambda: c.synthesizer_execute_dev(x, n // M, y))
    print(f"firpfbch2 synthesizer M={M:5d} m={m}: {ms:8.4f} ms  {n / ms / 1e6:8.1f} Gsamples/s  {12 * n / ms / 1e6:8.1f} GB/s")
