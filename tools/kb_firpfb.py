#!/usr/bin/env python3
"""FirPfbFilter block forms (GPU box): one branch over a block (execute_block), every branch per sample (execute_all),
a branch index per sample (execute_select) -- crcf, 2^22 .. 2^24 input samples."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
import torch
import yagi_amd as ya

dev = torch.device("cuda")
st = torch.cuda.current_stream()
n = 1 << 24
x = torch.empty(n, dtype=torch.complex64, device=dev)
y = torch.empty(4 * n, dtype=torch.complex64, device=dev)
ya.gen_complex_dev(5, n, out=x, stream=st.cuda_stream)


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


for nf, m in ((4, 8), (16, 4), (32, 4), (64, 6)):
    q = ya.FirPfbFilter.new_kaiser("crcf", nf, m, 0.5, 60.0)
    q.set_stream(st.cuda_stream)
    ms = timed(lambda: q.execute_block_dev(1, x, n, y))
    print(f"firpfb_crcf nf={nf:3d} m={m}: one branch      {ms:8.4f} ms  {n / ms / 1e6:8.1f} Gsamples/s  {16 * n / ms / 1e6:8.1f} GB/s")
    nin = n // nf if nf * n > 4 * n else n
    nin = min(n, (4 * n) // nf)
    ms = timed(lambda: q.execute_all_dev(x, nin, y))
    print(f"firpfb_crcf nf={nf:3d} m={m}: all branches     {ms:8.4f} ms  {nin / ms / 1e6:8.1f} Gsamples/s in  {8 * nin * (1 + nf) / ms / 1e6:8.1f} GB/s")
    idx = torch.randint(0, nf, (n,), dtype=torch.int32, device=dev)
    ms = timed(lambda: q.execute_select_dev(idx.data_ptr(), x, n, y))
    print(f"firpfb_crcf nf={nf:3d} m={m}: branch per sample {ms:8.4f} ms  {n / ms / 1e6:8.1f} Gsamples/s  {20 * n / ms / 1e6:8.1f} GB/s")
