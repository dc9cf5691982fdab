#!/usr/bin/env python3
"""A/B of library builds on any workload, in ONE process (GPU box): the package is imported once per build (each
import binds its own libyagi_hip .so), the arms run round-robin on the same device buffers, medians are compared.
usage: python tools/ab_pkg.py <workload> arm [arm ...]     arm = name[=variant name or .so path]  (no '=': the tree's build)
workloads: stream fft4096 c2 c2k3 c2k4 c4 c5 fftN:<n> msresamp2 resamp2 c4syn c5syn
env: AB_ROUNDS (9), AB_FORM / AB_FPW passed to the stream object as YAGI_FREQ_FORM / YAGI_FREQ_FPW"""
import importlib.util
import os
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import numpy as np
import torch

ROUNDS = int(os.environ.get("AB_ROUNDS", "9"))
workload = sys.argv[1]
specs = sys.argv[2:]


def import_build(tag, lib):
    if lib:
        path = Path(lib) if "/" in lib else ROOT / "yagi_amd" / "variants" / f"libyagi_{lib}.so"
        os.environ["YAGI_HIP_LIB"] = str(path)
    else:
        os.environ.pop("YAGI_HIP_LIB", None)
    name = f"yagi_amd_{tag}"
    spec = importlib.util.spec_from_file_location(name, ROOT / "yagi_amd" / "__init__.py",
                                                  submodule_search_locations=[str(ROOT / "yagi_amd")])
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


dev = torch.device("cuda")
st = torch.cuda.Stream()
torch.cuda.set_stream(st)
NTOT = 1 << 28
x = torch.empty(NTOT, dtype=torch.complex64, device=dev)
y = torch.empty(NTOT, dtype=torch.complex64, device=dev)
xp, yp = x.data_ptr(), y.data_ptr()
S = st.cuda_stream


def make(ya):
    """returns (callable, units, bytes_per_unit)"""
    n = 1 << 24
    h = ya.fir_design_kaiser(256, 0.2, 60.0)
    if workload == "stream":
        os.environ["YAGI_FREQ_FORM"] = os.environ.get("AB_FORM", "0")
        os.environ["YAGI_FREQ_FPW"] = os.environ.get("AB_FPW", "1")
        q = ya.FirFftStream(h); q.set_scale(0.4); q.set_stream(S); q.set_pipeline(True)
        def run():
            for b in range(16):
                q.execute_dev(xp + 8 * n * b, 4096, yp + 8 * n * b)
            q.join()
        return run, NTOT, 16, q
    if workload == "fft4096":
        plan = ya.Fft(4096, ya.Direction.Forward)
        return (lambda: plan.run_batch_dev(xp, yp, NTOT // 4096, S)), NTOT, 16, plan
    if workload.startswith("fftN:"):
        nn = int(workload.split(":")[1])
        plan = ya.Fft(nn, ya.Direction.Forward)
        nb = (1 << 27) // nn
        return (lambda: plan.run_batch_dev(xp, yp, nb, S)), nb * nn, 16, plan
    if workload in ("c2", "c2k3", "c2k4"):
        q = ya.FirFilter("crcf", h); q.set_scale(0.4); q.set_stream(S)
        q.set_kernel({"c2": 0, "c2k3": 3, "c2k4": 4}[workload])
        if os.environ.get("AB_PIPE") and hasattr(q, "set_pipeline"):
            q.set_pipeline(True)
        def run():
            for b in range(16):
                q.execute_block_dev(xp + 8 * n * b, n, yp + 8 * n * b)
            if hasattr(q, "join"):
                q.join()
        return run, NTOT, 16, q
    nc = 1 << 26
    if workload == "c4":
        q = ya.FirPfbCh.new_kaiser(64, 8, 60.0); q.set_stream(S)
        return (lambda: q.analyzer_execute_dev(xp, nc // 64, yp)), nc, 16, q
    if workload == "c5":
        q = ya.FirPfbCh2.new_kaiser(256, 4, 60.0); q.set_stream(S)
        return (lambda: q.analyzer_execute_dev(xp, nc // 128, yp)), nc, 24, q
    if workload == "c4syn":
        q = ya.FirPfbCh.new_kaiser(64, 8, 60.0); q.set_stream(S)
        return (lambda: q.synthesizer_execute_dev(xp, nc // 64, yp)), nc, 16, q
    if workload == "c5syn":
        q = ya.FirPfbCh2.new_kaiser(256, 4, 60.0); q.set_stream(S)
        return (lambda: q.synthesizer_execute_dev(xp, nc // 256, yp)), nc, 12, q
    if workload == "resamp2":
        q = ya.Resamp2.new("crcf", 12, 0.0, 60.0); q.set_stream(S)
        return (lambda: q.execute_block_dev(ya.Resamp2.DECIM, xp, nc, yp)), nc, 12, q
    if workload == "msresamp2":
        q = ya.MsResamp2("crcf", ya.MsResamp2.DECIM, 3, 0.4, 0.0, 60.0); q.set_stream(S)
        return (lambda: q.execute_block_dev(xp, nc // 8, yp)), nc, 9, q
    raise SystemExit(f"unknown workload {workload}")


arms = []
for i, sp in enumerate(specs):
    name, _, lib = sp.partition("=")
    ya = import_build(f"{i}", lib)
    if i == 0:
        if os.environ.get("AB_ZERO"):      # all-zero input: same instructions and traffic, no switching activity
            x.zero_()
        else:
            ya.gen_complex_dev(0x59414749 + 2, NTOT, out=x, stream=S)
        torch.cuda.synchronize()
    fn, units, bpu, keep = make(ya)
    arms.append({"name": name, "fn": fn, "units": units, "bpu": bpu, "keep": keep, "t": []})
# outputs of every arm against arm 0 (bitwise or to rounding)
ref = None
for a in arms:
    y.zero_()
    a["fn"]()
    torch.cuda.synchronize()
    out = y[: min(NTOT, 1 << 22)].clone()
    if ref is None:
        ref = out
    else:
        d = float(torch.linalg.vector_norm(out - ref) / (torch.linalg.vector_norm(ref) + 1e-30))
        print(f"{a['name']:12s} output vs {arms[0]['name']}: rel L2 {d:.2e}", flush=True)
for _ in range(3):
    for a in arms:
        a["fn"]()
torch.cuda.synchronize()
inner = 4
if os.environ.get("AB_SMI"):               # board power and shader clock while each arm loops (~3 s per arm)
    import subprocess, threading, time
    def smi():
        out = subprocess.run(["rocm-smi", "--showpower", "--showclocks"], capture_output=True, text=True, timeout=20).stdout
        pw = [l.split(":")[-1].strip() for l in out.splitlines() if "Power (W)" in l]
        ck = [l.split("(")[-1].rstrip(")") for l in out.splitlines() if "sclk" in l]
        return f"{pw[0] if pw else '?'} W, sclk {ck[0] if ck else '?'}"
    for a in arms:
        res = []
        th = threading.Thread(target=lambda: (time.sleep(1.2), res.append(smi()), res.append(smi())))
        th.start()
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < 3.0:
            for _ in range(8):
                a["fn"]()
            torch.cuda.synchronize()
        th.join()
        print(f"{a['name']:12s} under load: {' | '.join(res)}", flush=True)
for r in range(ROUNDS):
    for a in (arms if r % 2 == 0 else arms[::-1]):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        for _ in range(inner):
            a["fn"]()
        e1.record(st)
        torch.cuda.synchronize()
        a["t"].append(e0.elapsed_time(e1) / inner)
base = float(np.median(arms[0]["t"]))
for a in arms:
    med = float(np.median(a["t"]))
    gbs = a["bpu"] * a["units"] / med / 1e6
    print(f"{a['name']:12s} median {med * 1e3:9.2f} us (min {min(a['t']) * 1e3:9.2f})  {gbs / 1e3:5.2f} TB/s = "
          f"{gbs / 8000:.3f} of HBM peak  {(med / base - 1) * 100:+5.1f} % vs {arms[0]['name']}", flush=True)
