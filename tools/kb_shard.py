#!/usr/bin/env python3
"""firpfbch2 256-channel analyzer (config C5): what ONE rank of an R-GPU node computes, measured on one GPU.
Per R in 1, 2, 4, 8: the rank's shard kernel on the 2^26-sample block (sub-bands k = rank + R q) against its own
HBM floor (8 B read + 16/R B written per input sample), and the prediction this gives for the node:
    per-rank compute time   t_k(R)           (measured here)
    exchange                16/R B x 2^26 sent by every rank, received over R-1 xGMI links (~153 GB/s each, all links
                            busy in a direct all-gather): t_x(R) = (16/R x 2^26) / 153e9
    node time (chunked, exchange beside the next chunk's kernel)  ~ max(t_k, t_x) + one chunk of the other
Also: the chunked pipeline forced on one rank (kernel + 1-rank ncclAllGather + assemble): its fixed overhead.
usage: python tools/kb_shard.py > profiles/r02_c5_shard_table.txt"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np, torch, yagi_amd as ya
from yagi_amd.dist import Comm
n = 1 << 26
dev = torch.device("cuda")
M, m = 256, 4
ns = n // (M // 2)
x = torch.empty(n, dtype=torch.complex64, device=dev); y = torch.empty(2 * n, dtype=torch.complex64, device=dev)
st = torch.cuda.current_stream()
ya.gen_complex_dev(9, n, out=x, stream=st.cuda_stream)


def timed(fn, reps=20):
    for _ in range(5): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(reps): fn()
    e1.record(st); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


print(f"firpfbch2_crcf analyzer M={M} m={m}, block = 2^26 input samples ({ns} steps), one MI355X")
print(f"{'R':>2} {'shard kernel ms':>16} {'Gsamples/s':>11} {'alg. GB/s':>10} {'of 8 TB/s':>9} {'floor ms @6.29TB/s':>19} "
      f"{'xGMI ms (R-1 links)':>20} {'predicted node ms':>18} {'predicted Gsamples/s':>21}")
for R in (1, 2, 4, 8):
    c = ya.FirPfbCh2.new_kaiser(M, m, 60.0); c.set_stream(st.cuda_stream)
    fn = (lambda: c.analyzer_execute_dev(x, ns, y)) if R == 1 else (lambda: c.analyzer_execute_shard_dev(x, ns, R - 1, R, y))
    ms = timed(fn)
    nbytes = (8 + 16 / R) * n
    floor = nbytes / 6.29e9
    tx = 0.0 if R == 1 else (16 / R) * n / 153e6          # ms: every rank receives (R-1) slabs, one per link, in parallel
    node = max(ms, tx) + min(ms, tx) / 8                   # 8 chunks: the shorter leg's first/last chunk is exposed
    print(f"{R:>2} {ms:16.4f} {n / ms / 1e6:11.1f} {nbytes / ms / 1e6:10.1f} {nbytes / ms / 1e6 / 8000:9.3f} {floor:19.4f} "
          f"{tx:20.3f} {node:18.3f} {n / node / 1e6:21.1f}")
comm = Comm(Comm.unique_id(), 0, 1)
c = ya.FirPfbCh2.new_kaiser(M, m, 60.0); c.set_stream(st.cuda_stream)
base = timed(lambda: c.analyzer_execute_dev(x, ns, y))
for nch in (1, 4, 8, 16):
    ms = timed(lambda: c.analyzer_execute_sharded_dev(x, ns, comm, y, nchunks=-nch))
    print(f"forced pipeline on one rank, {nch:2d} chunk(s): {ms:.4f} ms (plain analyzer {base:.4f} ms): R=1 sharded kernel + "
          f"1-rank ncclAllGather (a 1 GiB device copy) + assemble")
