#!/usr/bin/env python3
"""The stream kernels on the STREAMED workload (GPU box): 2^28 samples as 16 distinct 2^24 blocks (nothing
Infinity-Cache resident), variants interleaved round-robin in one process; every variant is first checked against the
f64 oracle on sampled frames (incl. frame 0 and a block seam).  To A/B two builds of the library run it once per
build with YAGI_HIP_LIB=... .
usage: python tools/kb_freq.py [variant ...]      (FirFftStream.set_variant values; default: 4 = frequency-domain)"""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
import torch

import yagi_amd as ya
from oracle import oracle

forms = [int(a) for a in sys.argv[1:] if not a.startswith("--")] or [4]
ROUNDS = 7
NB, NF = 16, 4096
n = NF * 4096
dev = torch.device("cuda")
x = torch.empty(NB * n, dtype=torch.complex64, device=dev)
y = torch.empty(NB * n, dtype=torch.complex64, device=dev)
st = torch.cuda.Stream()
torch.cuda.set_stream(st)
ya.gen_complex_dev(0x59414749 + 2, NB * n, out=x, stream=st.cuda_stream)
h = ya.fir_design_kaiser(256, 0.2, 60.0)
xp, yp = x.data_ptr(), y.data_ptr()
blocks = [(xp + 8 * n * b, yp + 8 * n * b) for b in range(NB)]


def make(form):
    q = ya.FirFftStream(h)
    q.set_scale(0.4)
    q.set_variant(form % 100)
    q.set_stream(st.cuda_stream)
    if form >= 100:                 # 104 = variant 4 with pipelined block calls (two streams of the object)
        q.set_pipeline(True)
    return q


def run(q):
    for xb, yb in blocks:
        q.execute_dev(xb, NF, yb)
    q.join()


def check(form):
    q = make(form)
    y.zero_()
    run(q)
    torch.cuda.synchronize()
    worst = 0.0
    for f in (0, 1, 4095, 4096, 4097, 9 * 4096 + 1234, NB * NF - 1):
        lo = max(0, f * 4096 - 255)
        xs = x[lo:(f + 1) * 4096].cpu().numpy()
        if f == 0:
            xs = np.concatenate([np.zeros(255, np.complex64), xs])
        yref = oracle.fir_block_f64("crcf", h, xs, scale=0.4)[-4096:]
        truth = np.fft.fft(yref)
        got = y[f * 4096:(f + 1) * 4096].cpu().numpy()
        worst = max(worst, float(np.linalg.norm(got - truth) / np.linalg.norm(truth)))
    return worst


objs = {}
for fm in forms:
    err = check(fm)
    print(f"variant {fm:2d}: rel L2 vs f64 oracle (7 frames) {err:.3e} {'OK' if err < 1e-5 else 'FAIL'}", flush=True)
    objs[fm] = make(fm)
for q in objs.values():
    for _ in range(3):
        run(q)
torch.cuda.synchronize()
res = {fm: [] for fm in forms}
for r in range(ROUNDS):
    for fm, q in objs.items():
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        for _ in range(4):
            run(q)
        e1.record(st)
        torch.cuda.synchronize()
        res[fm].append(e0.elapsed_time(e1) / 4 / NB * 1e3)
for fm, v in res.items():
    med = float(np.median(v))
    print(f"variant {fm:2d}: median {med:7.2f} us / 2^24-sample block  (min {min(v):7.2f})  "
          f"{n / med / 1e3:7.1f} Gsamples/s  {16 * n / med / 1e6:6.2f} TB/s = {16 * n / med / 1e6 / 8:.3f} of HBM peak",
          flush=True)
