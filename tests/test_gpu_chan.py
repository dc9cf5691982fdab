"""firpfbch / firpfbch2 analyzers through the C ABI vs the oracle restatement + property tests.
PARITY UNPINNED by the reference (src/multichannel/mod.rs is empty): the conventions are
liquid-dsp's and are pinned here by (1) tone in channel k -> output k only, (2) channel-0 impulse
response = decimated prototype, (3) M-channel analyzer == M parallel firdecim branches,
(4) sharded sub-bands + assemble == unsharded."""
import numpy as np
import pytest

from gpu_util import SEED, rel_l2

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ya():
    import yagi_amd
    assert yagi_amd.device_count() > 0
    return yagi_amd


@pytest.mark.parametrize("M,m", [(4, 2), (8, 4), (64, 8), (6, 3), (10, 2), (256, 4), (1, 3)])
def test_firpfbch_vs_oracle(ya, oracle, M, m):
    h = oracle.fir_design_kaiser(2 * M * m + 1, 0.5 / M, 60.0) if M > 1 else np.float32([0.1, 0.5, 0.3, 0.2, 0.1, 0.05, 0.0])
    nfr = 150
    x = oracle.gen_complex(SEED + 4, nfr * M)
    ref = oracle.FirPfbCh(M, 2 * m, h)
    want = ref.analyzer_execute(x)
    q = ya.FirPfbCh(M, 2 * m, h)
    got = np.concatenate([q.analyzer_execute(x[: 37 * M]), q.analyzer_execute(x[37 * M:])])
    assert rel_l2(got, want) <= 2e-6
    q.reset()
    assert rel_l2(q.analyzer_execute(x[: 5 * M]), want[:5]) <= 2e-6


@pytest.mark.parametrize("M,m,nfr", [(64, 8, 4099), (64, 2, 1000), (128, 4, 777), (256, 8, 300), (256, 2, 64), (64, 4, 65),
                                     (512, 4, 333), (512, 2, 64), (1024, 2, 200), (1024, 4, 77), (512, 4, 20000),
                                     (8, 2, 5001), (16, 4, 3000), (32, 8, 2077), (8, 8, 70000),
                                     (64, 3, 1000), (256, 5, 300), (16, 1, 2000), (128, 6, 500), (32, 7, 640),
                                     (512, 3, 500), (1024, 1, 300), (512, 1, 200), (1024, 3, 130)])
def test_firpfbch_column_kernel_long_runs(ya, oracle, M, m, nfr):
    """the column-sliding kernels (M in {64,128,256}, p in {4,8,16}; wide banks M in {512,1024}, p in {4,8}): ragged
    frame counts, carried state; 20000 frames of 512 channels run several tiles per workgroup; branch lengths between the
    built sizes (p = 2, 6, 10, 12, 14) run on the next one with zero taps"""
    h = oracle.fir_design_kaiser(2 * M * m + 1, 0.5 / M, 60.0)
    x = oracle.gen_complex(SEED + 4, nfr * M)
    want = oracle.FirPfbCh(M, 2 * m, h).analyzer_execute(x)
    q = ya.FirPfbCh(M, 2 * m, h)
    k = (nfr * 2) // 3
    got = np.concatenate([q.analyzer_execute(x[: k * M]), q.analyzer_execute(x[k * M:])])
    assert rel_l2(got, want) <= 2e-6
    assert np.max(np.abs(got - want)) <= 2e-5 * float(np.max(np.abs(want)))


def test_firpfbch_kaiser_ctor_and_tone(ya, oracle):
    M, m = 64, 8
    q = ya.FirPfbCh.new_kaiser(M, m, 60.0)
    n = np.arange(M * 96)
    for k in (0, 1, 31, 32, 63):
        q.reset()
        y = q.analyzer_execute(np.exp(2j * np.pi * k / M * n).astype(np.complex64))[-1]
        p = 20 * np.log10(np.abs(y) / np.max(np.abs(y)) + 1e-12)
        assert np.argmax(p) == k
        assert np.all(np.delete(p, k) <= -55.0)


def test_firpfbch_equals_firdecim_per_channel(ya, oracle):
    """channel k of the analyzer == mix down by k/M, lowpass with the prototype, keep 1 of M
    sampled at n = f*M + M-1:  y_k[f] = sum_s h[fM+M-1-s] x[s] e^{-j 2 pi k s / M}."""
    M, m = 8, 3
    h = oracle.fir_design_kaiser(2 * M * m + 1, 0.5 / M, 60.0)[: 2 * M * m]
    nfr = 64
    x = oracle.gen_complex(SEED + 4, nfr * M)
    got = ya.FirPfbCh(M, 2 * m, h).analyzer_execute(x)
    n = np.arange(nfr * M)
    for k in range(M):
        mixed = x.astype(np.complex128) * np.exp(-2j * np.pi * k * n / M)
        full = np.convolve(mixed, h.astype(np.float64))[: nfr * M]
        want = full[M - 1:: M]                  # newest sample of frame f is f*M + M-1
        assert rel_l2(got[:, k], want) <= 1e-5, k


@pytest.mark.parametrize("M,m", [(4, 2), (8, 4), (64, 3), (256, 4), (6, 2), (10, 1)])
def test_firpfbch2_vs_oracle(ya, oracle, M, m):
    h = oracle.fir_design_kaiser(2 * M * m + 1, 1.0 / M, 60.0)
    h = (h * M / h.sum()).astype(np.float32)
    ns = 201
    x = oracle.gen_complex(SEED + 5, ns * (M // 2))
    want = oracle.FirPfbCh2(M, m, h).analyzer_execute(x)
    q = ya.FirPfbCh2(M, m, h)
    M2 = M // 2
    got = np.concatenate([q.analyzer_execute(x[: 33 * M2]), q.analyzer_execute(x[33 * M2: 34 * M2]),
                          q.analyzer_execute(x[34 * M2:])])           # odd split: parity carried
    assert rel_l2(got, want) <= 2e-6
    q.reset()
    assert rel_l2(q.analyzer_execute(x[: 9 * M2]), want[:9]) <= 2e-6


@pytest.mark.parametrize("M,m,ns", [(256, 4, 1000), (256, 2, 64), (64, 4, 4098), (128, 1, 333), (64, 2, 200), (64, 8, 700),
                                    (128, 8, 130), (512, 2, 333), (512, 1, 64), (512, 4, 90), (1024, 1, 200), (1024, 2, 77),
                                    (512, 2, 20000), (8, 2, 5000), (16, 1, 3001), (32, 4, 2000), (8, 4, 70000),
                                    (64, 3, 1000), (256, 5, 300), (128, 6, 500), (32, 7, 640), (256, 3, 2048),
                                    (512, 3, 400)])
def test_firpfbch2_column_kernel_long_runs(ya, oracle, M, m, ns):
    """the column-sliding kernel (M in {64,128,256}, 2m in {2,4,8,16}, even first step) incl. ragged tails"""
    h = oracle.fir_design_kaiser(2 * M * m + 1, 1.0 / M, 60.0)
    h = (h * M / h.sum()).astype(np.float32)
    M2 = M // 2
    x = oracle.gen_complex(SEED + 5, ns * M2)
    want = oracle.FirPfbCh2(M, m, h).analyzer_execute(x)
    q = ya.FirPfbCh2(M, m, h)
    k = ((ns * 2) // 3) & ~1                      # even split keeps the second call on the fast path
    got = np.concatenate([q.analyzer_execute(x[: k * M2]), q.analyzer_execute(x[k * M2:])])
    assert rel_l2(got, want) <= 2e-6
    # sharded form on the same kernel
    for R in ((2, 8) if ns <= 5000 else ()):
        dx = ya.DeviceArray.from_numpy(x)
        for r in (0, R - 1):
            qs = ya.FirPfbCh2(M, m, h)
            shard = ya.DeviceArray(ns * (M // R), np.complex64)
            qs.analyzer_execute_shard_dev(dx, ns, r, R, shard)
            ya.synchronize()
            assert rel_l2(shard.to_numpy().reshape(ns, M // R), want[:, r::R]) <= 3e-6, (R, r)


def test_firpfbch2_kaiser_tone_unit_gain(ya):
    M, m = 256, 4
    q = ya.FirPfbCh2.new_kaiser(M, m, 60.0)
    n = np.arange((M // 2) * 64)
    for k in (0, 3, 128, 255):
        q.reset()
        y = q.analyzer_execute(np.exp(2j * np.pi * k / M * n).astype(np.complex64))
        p = np.abs(y[-1])
        assert np.argmax(p) == k and abs(p[k] - 1.0) < 1e-3
        far = np.ones(M, bool)
        far[[(k - 1) % M, k, (k + 1) % M]] = False
        assert 20 * np.log10(np.max(p[far]) + 1e-12) <= -55.0
        assert abs(np.angle(y[-1][k] / y[-2][k])) < 1e-3        # channel-centre tone -> DC in both phases


@pytest.mark.parametrize("M,R", [(256, 8), (256, 2), (64, 4), (8, 8), (12, 3)])
def test_firpfbch2_sharded_equals_unsharded(ya, oracle, M, R):
    """config C5's partitioning: rank r computes sub-bands k = r + R*q; assembling the R shards
    (what the RCCL all-gather + assemble kernel produce) reproduces the full analyzer."""
    m = 4
    h = oracle.fir_design_kaiser(2 * M * m + 1, 1.0 / M, 60.0)
    h = (h * M / h.sum()).astype(np.float32)
    ns = 96
    x = oracle.gen_complex(SEED + 5, ns * (M // 2))
    full = ya.FirPfbCh2(M, m, h).analyzer_execute(x)
    dx = ya.DeviceArray.from_numpy(x)
    gathered = ya.DeviceArray(ns * M, np.complex64)          # [rank][step][M/R]
    Mr = M // R
    for r in range(R):
        q = ya.FirPfbCh2(M, m, h)
        shard = ya.DeviceArray(ns * Mr, np.complex64)
        q.analyzer_execute_shard_dev(dx, 40, r, R, shard)                       # two calls: state carried
        q.analyzer_execute_shard_dev(dx.ptr + 40 * (M // 2) * 8, ns - 40, r, R, shard.ptr + 40 * Mr * 8)
        ya.synchronize()
        s = shard.to_numpy().reshape(ns, Mr)
        assert rel_l2(s, full[:, r::R]) <= 2e-6, r
        ya.lib.yagi_hip_memcpy_h2d(gathered.ptr + r * ns * Mr * 8, s.ctypes.data, s.nbytes)
    out = ya.DeviceArray(ns * M, np.complex64)
    ya.FirPfbCh2.assemble_dev(gathered, ns, M, R, out)
    ya.synchronize()
    assert rel_l2(out.to_numpy().reshape(ns, M), full) <= 2e-6


def test_chan_config(ya):
    with pytest.raises(ya.ConfigError):
        ya.FirPfbCh.new_kaiser(0, 4, 60.0)
    with pytest.raises(ya.ConfigError):
        ya.FirPfbCh.new_kaiser(8, 0, 60.0)
    with pytest.raises(ya.ConfigError):
        ya.FirPfbCh2.new_kaiser(7, 4, 60.0)
    with pytest.raises(ya.ConfigError):
        ya.FirPfbCh2.new_kaiser(8, 0, 60.0)
    q = ya.FirPfbCh2.new_kaiser(8, 2, 60.0)
    with pytest.raises(ya.ConfigError):
        q.analyzer_execute(np.zeros(7, np.complex64))
    d = ya.DeviceArray(64, np.complex64)
    with pytest.raises(ya.ConfigError):
        q.analyzer_execute_shard_dev(d, 4, 0, 3, d)        # 8 channels do not shard 3 ways


def test_firpfbch2_sharded_driver_rccl_world1(ya, oracle):
    """yagi_amd.dist.firpfbch2_analyze_sharded end to end over RCCL (backend nccl) with the one GPU
    this box has: world_size 1 exercises shard kernel -> all_gather_into_tensor -> assemble kernel."""
    import os
    import socket
    import torch
    import torch.distributed as dist
    from yagi_amd import dist as yd
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        M, m, ns = 256, 4, 64
        q = ya.FirPfbCh2.new_kaiser(M, m, 60.0)
        x = oracle.gen_complex(SEED + 5, ns * (M // 2))
        xt = torch.from_numpy(x).cuda()
        y = yd.firpfbch2_analyze_sharded(q, xt, ns)
        torch.cuda.synchronize()
        want = ya.FirPfbCh2.new_kaiser(M, m, 60.0).analyzer_execute(x)
        assert rel_l2(y.cpu().numpy(), want) <= 1e-6
    finally:
        dist.destroy_process_group()


def test_firpfbch2_sharded_c_abi_rccl_chunked(ya, oracle):
    """the C-ABI communicator (yagi_hip_comm_*: RCCL bound by dlopen) and the chunked pipeline of
    analyzer_execute_sharded_dev: shard kernel on the object's stream, all-gather + assemble on the communicator's
    stream, chunk by chunk.  One GPU => a one-rank communicator with the pipeline forced (nchunks < 0): the R = 1
    sharded kernel, ncclAllGather and the assemble kernel all run; result == the plain analyzer, state carried over
    two calls, chunk counts that do not divide the block."""
    from yagi_amd.dist import Comm
    comm = Comm(Comm.unique_id(), 0, 1)
    assert (comm.rank, comm.nranks) == (0, 1)
    M, m = 256, 4
    ns1, ns2 = 3 * 4096 + 640, 2 * 4096
    x = oracle.gen_complex(SEED + 5, (ns1 + ns2) * (M // 2))
    want = ya.FirPfbCh2.new_kaiser(M, m, 60.0).analyzer_execute(x)
    dx = ya.DeviceArray.from_numpy(x)
    for nch in (-3, -1, -8):
        q = ya.FirPfbCh2.new_kaiser(M, m, 60.0)
        y = ya.DeviceArray((ns1 + ns2) * M, np.complex64)
        q.analyzer_execute_sharded_dev(dx, ns1, comm, y, nchunks=nch)
        q.analyzer_execute_sharded_dev(dx.ptr + ns1 * (M // 2) * 8, ns2, comm, y.ptr + ns1 * M * 8, nchunks=nch)
        ya.synchronize()
        assert rel_l2(y.to_numpy().reshape(-1, M), want) <= 2e-6, nch
    # nchunks >= 0 with one rank is the plain analyzer
    q = ya.FirPfbCh2.new_kaiser(M, m, 60.0)
    y = ya.DeviceArray((ns1 + ns2) * M, np.complex64)
    q.analyzer_execute_sharded_dev(dx, ns1 + ns2, comm, y)
    ya.synchronize()
    assert rel_l2(y.to_numpy().reshape(-1, M), want) <= 2e-6
    # the raw collective
    a = ya.DeviceArray.from_numpy(np.arange(1024, dtype=np.float32))
    b = ya.DeviceArray(1024, np.float32)
    comm.all_gather_dev(a, b, 4096)
    ya.synchronize()
    assert np.array_equal(b.to_numpy(), np.arange(1024, dtype=np.float32))
    comm.destroy()


def test_config_c4_firpfbch_64ch_full_size(ya, oracle):
    """BASELINE config C4 at full size: M 64, m 8 (p = 16), 2^26 complex samples generated on the device (runs of
    256 frames per column group: 16 half tiles each, the prefetch path).  Sampled frames -- first frames, every
    kind of run / tile boundary, the last frame -- against the oracle run on the p frames that reach them."""
    M, m = 64, 8
    p = 2 * m
    h = oracle.fir_design_kaiser(2 * M * m + 1, 0.5 / M, 60.0)
    n = 1 << 26
    nfr = n // M
    dx = ya.gen_complex_dev(SEED + 4, n)
    dy = ya.DeviceArray(n, np.complex64)
    q = ya.FirPfbCh(M, p, h)
    q.analyzer_execute_dev(dx, nfr, dy)
    ya.synchronize()
    for f in (0, 1, 7, 8, 15, 16, 255, 256, 257, 1023, 1024, 4 * 256 * 1000 + 9, nfr - 257, nfr - 1):
        lo = max(0, f - (p - 1))
        xs = dx.to_numpy((f + 1 - lo) * M, offset=lo * M)
        want = oracle.FirPfbCh(M, p, h).analyzer_execute(xs)[-1]
        got = dy.to_numpy(M, offset=f * M)
        assert rel_l2(got, want) <= 2e-6, f
    # second call continues the stream (history = last p-1 frames of the first block)
    dx2 = ya.gen_complex_dev(SEED + 4, 64 * M, first=n)
    dy2 = ya.DeviceArray(64 * M, np.complex64)
    q.analyzer_execute_dev(dx2, 64, dy2)
    ya.synchronize()
    xs = np.concatenate([dx.to_numpy((p - 1) * M, offset=n - (p - 1) * M), dx2.to_numpy(3 * M)])
    want = oracle.FirPfbCh(M, p, h).analyzer_execute(xs)[-3:]
    assert rel_l2(dy2.to_numpy(3 * M).reshape(3, M), want) <= 2e-6


def test_config_c5_firpfbch2_256ch_full_size(ya, oracle):
    """BASELINE config C5 (one GPU's part and the whole band): M 256, m 4, 2^26 complex samples = 524 288 steps
    (runs of 512 steps).  Sampled steps vs the oracle on the 2m + 1 steps that reach them; one sub-band shard of
    8 equals the matching channels."""
    M, m = 256, 4
    M2 = M // 2
    h = oracle.fir_design_kaiser(2 * M * m + 1, 1.0 / M, 60.0)
    h = (h * M / h.sum()).astype(np.float32)
    n = 1 << 26
    ns = n // M2
    dx = ya.gen_complex_dev(SEED + 5, n)
    dy = ya.DeviceArray(ns * M, np.complex64)
    q = ya.FirPfbCh2(M, m, h)
    q.analyzer_execute_dev(dx, ns, dy)
    ya.synchronize()
    reach = 4 * m + 2                                   # history is (2m-1) M + M/2 samples = 4m - 1 steps; even start
    for s in (0, 1, 2, 15, 16, 17, 511, 512, 513, 512 * 300 + 6, 512 * 300 + 7, ns - 513, ns - 2, ns - 1):
        lo = max(0, s - reach) & ~1                     # even first step keeps the oracle's step parity
        xs = dx.to_numpy((s + 1 - lo) * M2, offset=lo * M2)
        want = oracle.FirPfbCh2(M, m, h).analyzer_execute(xs)[-1]
        if lo > 0:                                      # zero history differs from the stream's only beyond reach
            got = dy.to_numpy(M, offset=s * M)
            assert rel_l2(got, want) <= 3e-6, s
        else:
            assert rel_l2(dy.to_numpy(M, offset=s * M), want) <= 3e-6, s
    R, r = 8, 3
    shard = ya.DeviceArray(ns * (M // R), np.complex64)
    qs = ya.FirPfbCh2(M, m, h)
    qs.analyzer_execute_shard_dev(dx, ns, r, R, shard)
    ya.synchronize()
    for s in (0, 513, ns - 1):
        full = dy.to_numpy(M, offset=s * M)
        assert rel_l2(shard.to_numpy(M // R, offset=s * (M // R)), full[r::R]) <= 3e-6, s


@pytest.mark.parametrize("M,m,nfr", [(4, 2, 50), (8, 4, 333), (64, 8, 200), (6, 3, 77), (10, 2, 100), (256, 4, 65), (1, 3, 20),
                                     (512, 2, 40), (48, 4, 90), (64, 8, 5000), (16, 2, 9001), (256, 2, 40000), (32, 4, 777),
                                     (64, 3, 1000), (256, 5, 300), (16, 1, 2000), (128, 7, 500)])
def test_firpfbch_synthesizer_vs_oracle(ya, oracle, M, m, nfr):
    """synthesizer (PARITY UNPINNED like the analyzer): frames of channel samples -> time samples, against
    the frame-by-frame restatement; state carried across calls; its state is independent of the analyzer's"""
    p = 2 * m
    h = oracle.fir_design_kaiser(2 * M * m + 1, 0.5 / M, 60.0)
    X = oracle.gen_complex(SEED + 6, nfr * M)
    want = oracle.FirPfbCh(M, p, h).synthesizer_execute(X)
    q = ya.FirPfbCh(M, p, h)
    k = nfr // 3
    q.analyzer_execute(X[: 2 * M])                       # must not disturb the synthesizer's window
    got = np.concatenate([q.synthesizer_execute(X[: k * M]), q.synthesizer_execute(X[k * M:(k + 1) * M]),
                          q.synthesizer_execute(X[(k + 1) * M:])])
    assert rel_l2(got, want) <= 3e-6
    q.reset()
    assert rel_l2(q.synthesizer_execute(X[: 3 * M]), want[: 3 * M]) <= 3e-6


def test_firpfbch_synthesizer_properties(ya, oracle):
    """an impulse in channel 0 plays out the prototype (the bank is an M-fold interpolator for channel 0); a constant in
    channel k alone gives a tone at k/M; linearity"""
    M, m = 16, 4
    p = 2 * m
    h = oracle.fir_design_kaiser(2 * M * m + 1, 0.5 / M, 60.0)
    q = ya.FirPfbCh(M, p, h)
    X = np.zeros((p + 2, M), np.complex64)
    X[0, 0] = 1.0
    y = q.synthesizer_execute(X)
    assert np.allclose(y[: M * p], h[: M * p], atol=1e-6) and np.allclose(y[M * p:], 0, atol=1e-7)
    for k in (1, 5, 15):
        q.reset()
        X = np.zeros((40, M), np.complex64)
        X[:, k] = 1.0
        y = q.synthesizer_execute(X)[M * p:]               # steady state
        n = np.arange(M * p, 40 * M)
        tone = np.exp(2j * np.pi * k * n / M)
        gain = np.vdot(tone, y) / len(y)
        assert abs(abs(gain) - abs(np.sum(h[: M * p]) / M)) <= 2e-3 * abs(np.sum(h)) / M + 1e-4
        assert np.linalg.norm(y - gain * tone) <= 0.05 * np.linalg.norm(y)
    rng = np.random.default_rng(3)
    A = (rng.standard_normal((30, M)) + 1j * rng.standard_normal((30, M))).astype(np.complex64)
    B = (rng.standard_normal((30, M)) + 1j * rng.standard_normal((30, M))).astype(np.complex64)
    qa, qb, qc = (ya.FirPfbCh(M, p, h) for _ in range(3))
    assert rel_l2(qc.synthesizer_execute(A + 2 * B), qa.synthesizer_execute(A) + 2 * qb.synthesizer_execute(B)) <= 1e-6


@pytest.mark.parametrize("M,p", [(16, 4), (64, 16), (256, 8), (512, 4), (12, 3)])
def test_firpfbch_analysis_synthesis_round_trip(ya, M, p):
    """size-independent property: with the block prototype h = [1]*M (+ zero taps so that the p-tap kernels run) the
    analyzer is a per-frame DFT and the synthesizer its inverse, so synthesizer(analyzer(x)) = M x over the whole
    stream -- 2^22 device-resident samples, every sample compared"""
    h = np.zeros(M * p, np.float32)
    h[:M] = 1.0
    n = (1 << 22) // M * M
    dx = ya.gen_complex_dev(SEED + 7, n)
    dX = ya.DeviceArray(n, np.complex64)
    dy = ya.DeviceArray(n, np.complex64)
    q = ya.FirPfbCh(M, p, h)
    q.analyzer_execute_dev(dx, n // M, dX)
    q.synthesizer_execute_dev(dX, n // M, dy)
    ya.synchronize()
    x, y = dx.to_numpy(), dy.to_numpy()
    assert rel_l2(y, M * x.astype(np.complex128)) <= 2e-6


@pytest.mark.parametrize("M,m,ns", [(4, 1, 40), (8, 2, 101), (16, 4, 64), (64, 3, 50), (6, 2, 33), (10, 1, 50), (256, 2, 21),
                                    (8, 2, 300), (16, 4, 257), (32, 2, 211), (64, 4, 230), (128, 2, 200), (256, 4, 215),
                                    (64, 3, 300), (256, 3, 215), (16, 1, 400), (128, 1, 257)])
def test_firpfbch2_synthesizer_vs_oracle(ya, oracle, M, m, ns):
    """firpfbch2 synthesizer (PARITY UNPINNED) against the step-by-step restatement; odd splits carry the
    step parity and the window across calls; independent of the analyzer's state.  Calls of >= 64 steps with M a power
    of two in 8 .. 256 and m in {2, 4} run the column-sliding kernel (the last six cases: first and third call), the
    rest the tiled one."""
    h = oracle.fir_design_kaiser(2 * M * m + 1, 0.5 / M, 60.0)
    h = (h * M / h.sum()).astype(np.float32)
    X = oracle.gen_complex(SEED + 10, ns * M)
    want = oracle.FirPfbCh2(M, m, h).synthesizer_execute(X)
    q = ya.FirPfbCh2(M, m, h)
    q.analyzer_execute(X[: 3 * (M // 2)])               # must not disturb the synthesizer
    k = ns // 3 | 1                                     # odd: the second call starts on an odd step
    got = np.concatenate([q.synthesizer_execute(X[: k * M]), q.synthesizer_execute(X[k * M:(k + 1) * M]),
                          q.synthesizer_execute(X[(k + 1) * M:])])
    assert got.shape == (ns * M // 2,)
    assert rel_l2(got, want) <= 3e-6
    q.reset()
    assert rel_l2(q.synthesizer_execute(X[: 5 * M]), want[: 5 * M // 2]) <= 3e-6


@pytest.mark.parametrize("M,m,tol_db", [(16, 4, -70.0), (64, 3, -55.0), (256, 4, -70.0), (12, 4, -70.0)])
def test_firpfbch2_perfect_reconstruction(ya, M, m, tol_db):
    """the property that pins both firpfbch2 conventions: analyzer (prototype cutoff 1/M) followed by the synthesizer
    (cutoff 0.5/M) reproduces the input delayed by 2 M m - M/2 + 1 samples with unit gain, over 2^20 device samples"""
    n = (1 << 20) // M * M
    ns = n // (M // 2)
    dx = ya.gen_complex_dev(SEED + 11, n)
    dX = ya.DeviceArray(ns * M, np.complex64)
    dy = ya.DeviceArray(n, np.complex64)
    ya.FirPfbCh2.new_kaiser(M, m, 80.0).analyzer_execute_dev(dx, ns, dX)
    ya.FirPfbCh2.new_kaiser_synthesizer(M, m, 80.0).synthesizer_execute_dev(dX, ns, dy)
    ya.synchronize()
    x, y = dx.to_numpy(), dy.to_numpy()
    d = 2 * M * m - M // 2 + 1
    err = np.linalg.norm(y[d:] - x[: n - d]) / np.linalg.norm(x[: n - d])
    assert 20 * np.log10(err) <= tol_db, 20 * np.log10(err)


def test_channelizer_randomised_shapes(ya, oracle):
    """seeded sweep over channel counts (powers of two on the column / wide kernels, anything else on the tiled ones),
    branch lengths and frame counts, analyzers and the firpfbch synthesizer against the restatements, with a cut"""
    rng = np.random.default_rng(99)
    Ms = [2, 4, 8, 16, 32, 64, 128, 256, 512, 1024, 6, 10, 12, 20, 48, 96, 100, 250]
    for case in range(36):
        M = int(rng.choice(Ms))
        m = int(rng.choice([1, 2, 3, 4, 8]))
        if M * m > 4096:
            m = 2
        nfr = int(rng.integers(64, 400))
        h = oracle.fir_design_kaiser(2 * M * m + 1, 0.5 / M, 60.0)
        x = oracle.gen_complex(1000 + case, nfr * M)
        k = int(rng.integers(1, nfr))
        q = ya.FirPfbCh(M, 2 * m, h)
        got = np.concatenate([q.analyzer_execute(x[: k * M]), q.analyzer_execute(x[k * M:])])
        assert rel_l2(got, oracle.FirPfbCh(M, 2 * m, h).analyzer_execute(x)) <= 3e-6, ("ch", M, m, nfr)
        if case % 3 == 0 and nfr * M <= 60000:
            got = np.concatenate([q.synthesizer_execute(x[: k * M]), q.synthesizer_execute(x[k * M:])])
            assert rel_l2(got, oracle.FirPfbCh(M, 2 * m, h).synthesizer_execute(x)) <= 3e-6, ("syn", M, m, nfr)
        if M % 2 == 0:
            h2 = oracle.fir_design_kaiser(2 * M * m + 1, 1.0 / M, 60.0)
            h2 = (h2 * M / h2.sum()).astype(np.float32)
            ns = 2 * (nfr // 2)
            x2 = x[: ns * (M // 2)]
            k2 = 2 * int(rng.integers(1, ns // 2))
            q2 = ya.FirPfbCh2(M, m, h2)
            got = np.concatenate([q2.analyzer_execute(x2[: k2 * (M // 2)]), q2.analyzer_execute(x2[k2 * (M // 2):])])
            assert rel_l2(got, oracle.FirPfbCh2(M, m, h2).analyzer_execute(x2)) <= 3e-6, ("ch2", M, m, ns)
