"""The headline composition through the C ABI: firfilt_crcf.execute_block -> consecutive 4096-sample
frames -> forward FFT, fused on the device (yagi_hip_firfft_crcf_*), vs the oracle."""
import numpy as np
import pytest

from gpu_util import SEED, fir_bound, rel_l2

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ya():
    import yagi_amd
    assert yagi_amd.device_count() > 0
    return yagi_amd


def spectra_truth(oracle, h, scale, x, nfft=4096):
    y = oracle.fir_block_f64("crcf", h, x, scale=scale)
    return np.array([np.fft.fft(y[f * nfft:(f + 1) * nfft]) for f in range(len(x) // nfft)])


@pytest.mark.parametrize("L", [1, 2, 4, 63, 130, 256, 257, 1000, 2049])
def test_stream_small_vs_oracle(ya, oracle, L):
    rng = np.random.default_rng(L)
    h = (rng.standard_normal(L) / np.sqrt(L)).astype(np.float32)
    x = oracle.gen_complex(SEED + 2, 12 * 4096)
    truth = spectra_truth(oracle, h, 0.4, x)
    # 1 sliding VALU FIR, 2 MFMA Toeplitz FIR, 3 fast convolution + FFT, 4 frequency-domain filter; 0 = auto
    for variant in ((1, 2, 3, 4, 0) if L <= 256 else (1, 3, 4, 0) if L <= 257 else (1, 3, 0) if L <= 1024 else (3, 0)):
        q = ya.FirFftStream(h)
        q.set_scale(0.4)
        q.set_variant(variant)
        got = np.concatenate([q.execute(x[: 5 * 4096]), q.execute(x[5 * 4096:])])     # state carried across calls
        for f in range(12):
            assert rel_l2(got[f], truth[f]) <= 1e-5, (variant, f)
    # same result as the unfused objects (FirFilter then Fft), to f32 rounding
    fir = ya.FirFilter("crcf", h)
    fir.set_scale(0.4)
    unfused = ya.Fft(4096, ya.Direction.Forward).run_batch(fir.execute_block(x))
    assert rel_l2(got, unfused) <= 2e-6
    # and the oracle's own f32 composition is within the same distance of the f64 truth
    want32 = oracle.stream_fir_fft(h, 0.4, x, 4096)
    assert rel_l2(got, truth) <= 3 * rel_l2(want32, truth) + 1e-7
    q.reset()
    assert rel_l2(q.execute(x[:4096])[0], truth[0]) <= 1e-5


def test_stream_config(ya):
    with pytest.raises(ya.ConfigError):
        ya.FirFftStream(np.zeros(0, np.float32))
    with pytest.raises(ya.ConfigError):
        ya.FirFftStream(np.ones(8, np.float32), nfft=0)
    q1 = ya.FirFftStream(np.ones(8, np.float32), nfft=1024)
    with pytest.raises(ya.ConfigError):
        q1.set_variant(4)                       # only nfft = 4096 has the fused kernels
    with pytest.raises(ya.ConfigError):
        ya.FirFftStream(np.ones(2050, np.float32))
    q = ya.FirFftStream(np.ones(8, np.float32))
    with pytest.raises(ya.ConfigError):
        q.execute(np.zeros(100, np.complex64))
    assert q.execute(np.zeros(0, np.complex64)).shape == (0, 4096)


def test_stream_integer_alignment_exact(ya):
    """frame/tap alignment: an impulse train through a rect filter gives exactly predictable spectra"""
    h = np.ones(256, np.float32)
    x = np.zeros(3 * 4096, np.complex64)
    x[4096 - 100] = 1.0          # its 256-sample response straddles the frame 0 / frame 1 boundary
    y = np.zeros(3 * 4096)
    y[4096 - 100: 4096 - 100 + 256] = 1.0
    for variant in (1, 2, 3, 4):
        q = ya.FirFftStream(h)
        q.set_variant(variant)
        got = q.execute(x)
        for f in range(3):
            truth = np.fft.fft(y[f * 4096:(f + 1) * 4096])
            assert np.max(np.abs(got[f] - truth)) <= 2e-3
        assert np.max(np.abs(got[2])) == 0.0


def test_headline_config_full_block(ya, oracle):
    """headline workload at bench size: kaiser(256,0.2,60), scale 0.4, one 2^24-sample block
    (4096 frames) of the C2 stream generated on the device; sampled frames vs f64 truth."""
    h = oracle.fir_design_kaiser(256, 0.2, 60.0)
    n = 1 << 24
    nframes = n // 4096
    dx = ya.gen_complex_dev(SEED + 2, n)
    dy = ya.DeviceArray(n, np.complex64)
    q = ya.FirFftStream(h)
    q.set_scale(0.4)
    q.execute_dev(dx, nframes, dy)
    ya.synchronize()
    # the time-domain forms (sliding, MFMA Toeplitz, fast convolution) give the same spectra as the default
    for variant in (1, 2, 3):
        q2 = ya.FirFftStream(h)
        q2.set_scale(0.4)
        q2.set_variant(variant)
        dy_alt = ya.DeviceArray(n, np.complex64)
        q2.execute_dev(dx, nframes, dy_alt)
        ya.synchronize()
        for f in (0, 5, 511, 512, 3000, 4095):
            assert rel_l2(dy_alt.to_numpy(4096, offset=f * 4096), dy.to_numpy(4096, offset=f * 4096)) <= 3e-6, (variant, f)
        # second call continues the stream (state + chunk offsets)
        q2.execute_dev(dx, 32, dy_alt)
        ya.synchronize()
        tail = dx.to_numpy(255, offset=n - 255)
        xs = np.concatenate([tail, dx.to_numpy(4096)])
        truth = np.fft.fft(oracle.fir_block_f64("crcf", h, xs, scale=0.4)[-4096:])
        assert rel_l2(dy_alt.to_numpy(4096), truth) <= 1e-5, variant
        dy_alt.free()
    for f in (0, 1, 2047, 4095):
        lo = max(0, f * 4096 - 255)
        xs = dx.to_numpy((f + 1) * 4096 - lo, offset=lo)
        y = oracle.fir_block_f64("crcf", h, xs, scale=0.4)[-4096:]
        truth = np.fft.fft(y)
        got = dy.to_numpy(4096, offset=f * 4096)
        assert rel_l2(got, truth) <= 1e-5, f
    # second block continues the stream: frame 0 of block 2 needs the last 255 samples of block 1
    dx2 = ya.gen_complex_dev(SEED + 2, 8 * 4096, first=n)
    dy2 = ya.DeviceArray(8 * 4096, np.complex64)
    q.execute_dev(dx2, 8, dy2)
    ya.synchronize()
    tail = dx.to_numpy(255, offset=n - 255)
    xs = np.concatenate([tail, dx2.to_numpy(4096)])
    truth = np.fft.fft(oracle.fir_block_f64("crcf", h, xs, scale=0.4)[-4096:])
    assert rel_l2(dy2.to_numpy(4096), truth) <= 1e-5


@pytest.mark.parametrize("nfft", [256, 1000, 2048, 8192])
def test_stream_other_frame_lengths(ya, oracle, nfft):
    """nfft != 4096: overlap-save FIR + batched transform, same stream semantics"""
    rng = np.random.default_rng(nfft)
    h = (rng.standard_normal(129) / 11).astype(np.float32)
    nframes = 9
    x = oracle.gen_complex(SEED + 2, nframes * nfft)
    truth = spectra_truth(oracle, h, 0.4, x, nfft)
    q = ya.FirFftStream(h, nfft=nfft)
    q.set_scale(0.4)
    got = np.concatenate([q.execute(x[: 4 * nfft]), q.execute(x[4 * nfft:])])
    for f in range(nframes):
        assert rel_l2(got[f], truth[f]) <= 1e-5, f


@pytest.mark.parametrize("delay", [0, 1, 100, 255])
def test_stream_pure_delay_whole_block(ya, delay):
    """size-independent property at bench size: with h = delta[k - delay] (256 taps) the filtered stream is the input
    delayed, so every spectrum must equal the transform of the delayed stream's frame -- all 4096 frames of a 2^24
    sample block compared, for the default (frequency-domain: this exercises the frame-boundary correction on every
    frame), overlap-save and direct-form variants, and across a second call (carried state)"""
    L, nfft, nframes, extra = 256, 4096, 4096, 8
    h = np.zeros(L, np.float32)
    h[delay] = 1.0
    n = nframes * nfft
    ntot = n + extra * nfft
    dx = ya.gen_complex_dev(SEED + 8, ntot)
    src = dx.to_numpy()
    # reference spectra: batched FFT of the delayed stream (zeros before the start)
    dd = ya.DeviceArray.from_numpy(np.concatenate([np.zeros(delay, np.complex64), src[: ntot - delay]]))
    dref = ya.DeviceArray(ntot, np.complex64)
    ya.Fft(nfft, ya.Direction.Forward).run_batch_dev(dd, dref, nframes + extra)
    ya.synchronize()
    ref = dref.to_numpy()
    dtail = ya.DeviceArray.from_numpy(src[n:])
    for variant in (0, 3, 2):
        q = ya.FirFftStream(h)
        q.set_variant(variant)
        dy = ya.DeviceArray(n, np.complex64)
        q.execute_dev(dx, nframes, dy)
        ya.synchronize()
        got = dy.to_numpy()
        assert rel_l2(got, ref[:n]) <= 2e-6, variant
        assert np.max(np.abs(got - ref[:n])) <= 2e-3, variant      # |X| ~ 64: a few ulp of the spectrum magnitude
        q.execute_dev(dtail, extra, dy)                            # second call continues the stream
        ya.synchronize()
        assert rel_l2(dy.to_numpy(extra * nfft), ref[n:]) <= 2e-6, variant
        dy.free()


def test_stream_pipelined_blocks_bit_identical(ya, oracle):
    """set_pipeline(1): consecutive execute_dev calls run on two streams of the object, each block reading its window
    from the previous call's input; after join() the spectra are bit for bit those of the unpipelined calls -- every block seam, ragged
    block lengths, a scale change, reset, and a host-pointer call in between"""
    h = oracle.fir_design_kaiser(256, 0.2, 60.0)
    sizes = [64, 1, 7, 64, 2, 128, 33, 64, 64, 64, 5, 64]
    nf = sum(sizes)
    dx = ya.gen_complex_dev(SEED + 2, nf * 4096)
    plain, piped = ya.FirFftStream(h), ya.FirFftStream(h)
    for q in (plain, piped):
        q.set_scale(0.4)
    piped.set_pipeline(True)
    dy0, dy1 = ya.DeviceArray(nf * 4096, np.complex64), ya.DeviceArray(nf * 4096, np.complex64)
    dy1.zero()
    ya.synchronize()
    for rep in range(3):
        for q, dy in ((plain, dy0), (piped, dy1)):
            f0 = 0
            for s in sizes:
                q.execute_dev(dx.ptr + 8 * 4096 * f0, s, dy.ptr + 8 * 4096 * f0)
                f0 += s
        piped.join()                                   # the null stream now waits for both lanes: no device sync needed
        a, b = dy0.to_numpy(), dy1.to_numpy()
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), rep
        if rep == 0:                                   # frames at a seam and inside a block against the f64 truth
            src = dx.to_numpy()
            for f in (0, 64, 65, 72, 500):
                lo = max(0, f * 4096 - 255)
                xs = src[lo:(f + 1) * 4096]
                if f == 0:
                    xs = np.concatenate([np.zeros(255, np.complex64), xs])
                truth = np.fft.fft(oracle.fir_block_f64("crcf", h, xs, scale=0.4)[-4096:])
                assert rel_l2(b[f * 4096:(f + 1) * 4096], truth) <= 1e-5, f
        if rep == 0:
            for q in (plain, piped):
                q.set_scale(0.25)                      # scaled FFT{h} is rebuilt on the object's stream: that call joins
        if rep == 1:
            for q in (plain, piped):
                q.reset()
    # host-pointer call after pipelined device calls continues the same stream
    x2 = oracle.gen_complex(SEED + 9, 3 * 4096)
    assert np.array_equal(plain.execute(x2).view(np.uint32), piped.execute(x2).view(np.uint32))
    piped.set_pipeline(False)
    plain.execute_dev(dx, 8, dy0)
    piped.execute_dev(dx, 8, dy1)
    ya.synchronize()
    assert np.array_equal(dy0.to_numpy(8 * 4096).view(np.uint32), dy1.to_numpy(8 * 4096).view(np.uint32))
