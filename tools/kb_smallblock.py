#!/usr/bin/env python3
"""Small blocks and the per-sample calls (GPU box): what one FirFilter call costs below the HBM-bound region.
 * execute_block_dev over blocks of 2^6 .. 2^24 samples (crcf, 256 taps), device pointers, one stream: time per call
   and samples/s -- the launch-bound region is where the time per call stops falling;
 * push() + execute() per sample and execute_one() (host sample in, host sample out: the reference's per-sample API)."""
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
import torch

import yagi_amd as ya

dev = torch.device("cuda")
st = torch.cuda.current_stream()
h = ya.fir_design_kaiser(256, 0.2, 60.0)
q = ya.FirFilter("crcf", h)
q.set_stream(st.cuda_stream)
nmax = 1 << 24
x = torch.empty(nmax, dtype=torch.complex64, device=dev)
y = torch.empty(nmax, dtype=torch.complex64, device=dev)
ya.gen_complex_dev(11, nmax, out=x, stream=st.cuda_stream)
print("block      calls   us/call    Msamples/s")
for lg in (6, 8, 10, 12, 14, 16, 18, 20, 22, 24):
    n = 1 << lg
    calls = max(20, min(2000, (1 << 26) // n))
    for _ in range(10):
        q.execute_block_dev(x.data_ptr(), n, y.data_ptr())
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(calls):
        q.execute_block_dev(x.data_ptr(), n, y.data_ptr())
    torch.cuda.synchronize()
    us = (time.perf_counter() - t0) / calls * 1e6
    print(f"2^{lg:<2d}  {calls:8d}  {us:9.2f}  {n / us:12.1f}", flush=True)
# per-sample API
xs = (np.random.default_rng(1).standard_normal(2000) + 0j).astype(np.complex64)
t0 = time.perf_counter()
for v in xs:
    q.push(v)
    q.execute()
t1 = time.perf_counter()
for v in xs:
    q.execute_one(v)
t2 = time.perf_counter()
print(f"push()+execute(): {(t1 - t0) / len(xs) * 1e6:.1f} us per sample; execute_one(): {(t2 - t1) / len(xs) * 1e6:.1f} us per sample")
# host-array block (PCIe both ways) for scale
for lg in (10, 16, 22):
    xb = (np.random.default_rng(2).standard_normal(1 << lg) + 0j).astype(np.complex64)
    q.execute_block(xb)
    t0 = time.perf_counter()
    for _ in range(5):
        q.execute_block(xb)
    us = (time.perf_counter() - t0) / 5 * 1e6
    print(f"execute_block(host array 2^{lg}): {us:.1f} us per call, {(1 << lg) / us:.1f} Msamples/s")
