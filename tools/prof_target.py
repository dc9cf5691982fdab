#!/usr/bin/env python3
"""profiling target: launch each selected kernel a few times (for rocprofv3 --pmc / --kernel-trace).
usage: python3 tools/prof_target.py [fused fused2 fir2 fir3 fft fft_big chan1 chan2 rrrf63 rrrf256 cccf256
       decim4x65 decim4x257 decim8x513 interp4 syn2 resamp2 msresamp2 fft16384 fft65536 fft1048576] """
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
import torch
import yagi_amd as ya

what = sys.argv[1:] or ["fused"]
n = 1 << 24
dev = torch.device("cuda")
x = torch.empty(n, dtype=torch.complex64, device=dev)
y = torch.empty(2 * n, dtype=torch.complex64, device=dev)
st = torch.cuda.current_stream()
ya.gen_complex_dev(0x59414749 + 2, n, out=x, stream=st.cuda_stream)
h = ya.fir_design_kaiser(256, 0.2, 60.0)
keep = []
for w in what:
    if w == "stream16":      # the bench.py step: 16 distinct 2^24-sample blocks (nothing Infinity-Cache resident)
        xb = torch.empty(16 * n, dtype=torch.complex64, device=dev)
        yb = torch.empty(16 * n, dtype=torch.complex64, device=dev)
        ya.gen_complex_dev(0x59414749 + 2, 16 * n, out=xb, stream=st.cuda_stream)
        f = ya.FirFftStream(h); f.set_scale(0.4); f.set_stream(st.cuda_stream)
        def fn(f=f, xb=xb, yb=yb):
            for b in range(16):
                f.execute_dev(xb.data_ptr() + 8 * n * b, n // 4096, yb.data_ptr() + 8 * n * b)
    elif w in ("fused", "fused2", "fused3", "fused4"):
        f = ya.FirFftStream(h); f.set_stream(st.cuda_stream); f.set_variant({"fused": 1, "fused2": 2, "fused3": 3, "fused4": 4}[w])
        fn = lambda f=f: f.execute_dev(x, n // 4096, y)
    elif w in ("fir1", "fir2", "fir3", "fir4"):
        q = ya.FirFilter("crcf", h); q.set_kernel(int(w[3])); q.set_stream(st.cuda_stream)
        fn = lambda q=q: q.execute_block_dev(x, n, y)
    elif w == "fft":
        p = ya.Fft(4096, ya.Direction.Forward)
        fn = lambda p=p: p.run_batch_dev(x, y, n // 4096, st.cuda_stream)
    elif w == "chan1":
        c = ya.FirPfbCh.new_kaiser(64, 8, 60.0); c.set_stream(st.cuda_stream)
        fn = lambda c=c: c.analyzer_execute_dev(x, n // 64, y)
    elif w == "chan2":
        c = ya.FirPfbCh2.new_kaiser(256, 4, 60.0); c.set_stream(st.cuda_stream)
        fn = lambda c=c: c.analyzer_execute_dev(x, n // 128, y)
    elif w in ("rrrf63", "rrrf256", "cccf256"):
        kind = w[:4]
        L = int(w[4:])
        hh = ya.fir_design_kaiser(L, 0.2, 60.0)
        if kind == "cccf":
            hh = (hh * np.exp(0.3j * np.arange(L))).astype(np.complex64)
        q = ya.FirFilter(kind, hh); q.set_stream(st.cuda_stream)
        xin, yout, cnt = (x.view(torch.float32), y.view(torch.float32), n) if kind == "rrrf" else (x, y, n)
        fn = lambda q=q, xin=xin, yout=yout, cnt=cnt: q.execute_block_dev(xin, cnt, yout)
    elif w in ("decim4x65", "decim4x257", "decim8x513"):
        M, L = {"decim4x65": (4, 65), "decim4x257": (4, 257), "decim8x513": (8, 513)}[w]
        d = ya.FirDecimationFilter("crcf", M, ya.fir_design_kaiser(L, 0.4 / M, 60.0)); d.set_stream(st.cuda_stream)
        fn = lambda d=d, M=M: d.execute_block_dev(x, n // M, y)
    elif w == "interp4":
        fi = ya.FirInterpolationFilter.new_kaiser("crcf", 4, 8, 60.0); fi.set_stream(st.cuda_stream)
        fn = lambda fi=fi: fi.execute_block_dev(x, n // 4, y)
    elif w == "syn2":
        c = ya.FirPfbCh2.new_kaiser_synthesizer(256, 4, 60.0); c.set_stream(st.cuda_stream)
        fn = lambda c=c: c.synthesizer_execute_dev(x, n // 256, y)
    elif w == "resamp2":
        r2 = ya.Resamp2.new("crcf", 12, 0.0, 60.0)
        r2.set_stream(st.cuda_stream)
        fn = lambda r2=r2: r2.execute_block_dev(r2.DECIM, x, n, y)
    elif w == "msresamp2":
        ms = ya.MsResamp2("crcf", ya.MsResamp2.DECIM, 3, 0.4, 0.0, 60.0)
        ms.set_stream(st.cuda_stream)
        fn = lambda ms=ms: ms.execute_block_dev(x, n // 8, y)
    elif w in ("fft16384", "fft65536", "fft1048576"):
        N = int(w[3:])
        p = ya.Fft(N, ya.Direction.Forward)
        fn = lambda p=p, N=N: p.run_batch_dev(x, y, n // N, st.cuda_stream)
    else:
        raise SystemExit(f"unknown target {w}")
    keep.append(fn)
    for _ in range(4):
        fn()
torch.cuda.synchronize()
