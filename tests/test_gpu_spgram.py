"""fft::Spgram through the C ABI vs the oracle restatement and the reference's own acceptance tests
(src/fft/spgram.rs:339-654: noise floor within +-0.5 dB over 2000*nfft samples, counters, config)."""
import numpy as np
import pytest

from gpu_util import SEED

pytestmark = pytest.mark.gpu
NOISE_FLOOR = -80.0


@pytest.fixture(scope="module")
def ya():
    import yagi_amd
    assert yagi_amd.device_count() > 0
    return yagi_amd


def noise(oracle, n, seed=SEED + 9):
    # Complex32::new(randnf(), randnf()) * nstd * sqrt(0.5)  (spgram.rs:352): re, im ~ N(0, nstd^2/2)
    return oracle.gen_complex(seed, n) * np.float32(10 ** (NOISE_FLOOR / 20))


CASES = [(440, 0, 0, "Unknown"), (1024, 0, 0, "Unknown"), (1200, 0, 0, "Unknown"),
         (400, 400, 100, "Hamming"), (512, 200, 120, "Hamming"), (640, 100, 10, "Hamming"),
         (960, 83, 17, "Hamming")]


@pytest.mark.parametrize("nfft,wlen,delay,wtype", CASES)
def test_spgramcf_noise(ya, oracle, nfft, wlen, delay, wtype):
    """testbench_spgramcf_noise (spgram.rs:339-367) at the reference's size: 2000*nfft samples"""
    n = 2000 * nfft
    q = ya.Spgram.default(nfft) if wlen == 0 else ya.Spgram(nfft, ya.WindowType[wtype], wlen, delay)
    x = noise(oracle, n)
    q.write(x[: n // 3])
    q.write(x[n // 3:])
    assert q.get_num_samples() == n and q.get_num_samples_total() == n
    d = q.get_delay()
    assert q.get_num_transforms() == n // d
    psd = q.get_psd()
    assert np.all(np.abs(psd - NOISE_FLOOR) <= 0.5), (psd.min(), psd.max())


@pytest.mark.parametrize("wtype", ["Hamming", "Hann", "BlackmanHarris", "BlackmanHarris7", "Kaiser", "FlatTop",
                                   "Triangular", "RcosTaper", "Kbd"])
def test_spgramcf_every_window_vs_oracle(ya, oracle, wtype):
    """all nine taper windows: device PSD == oracle restatement on the same samples (rel 2e-4)"""
    nfft, wlen, delay = 800, 400, 200
    wt = ya.WindowType[wtype]
    x = noise(oracle, 60 * nfft)
    q = ya.Spgram(nfft, wt, wlen, delay)
    q.write(x)
    ref = oracle.Spgram(nfft, int(wt), wlen, delay)
    ref.write(x)
    assert q.get_num_transforms() == ref.num_transforms
    a, b = q.get_psd_mag(), ref.get_psd_mag()
    assert np.max(np.abs(a - b) / b) <= 2e-4
    assert np.max(np.abs(q.get_psd() - ref.get_psd())) <= 2e-3
    assert q.get_wtype() == wt and q.get_window_len() == wlen and q.get_nfft() == nfft


def test_spgram_push_write_timer_and_modes(ya, oracle):
    """per-sample push == block write (timer across calls), clear/reset counters, alpha modes, f32 input"""
    nfft, wlen, delay = 64, 48, 7
    x = noise(oracle, 1000)
    ref = oracle.Spgram(nfft, 2, wlen, delay)
    q = ya.Spgram(nfft, ya.WindowType.Hann, wlen, delay)
    for lo, hi in [(0, 3), (3, 4), (4, 100), (100, 106), (106, 1000)]:
        ref.write(x[lo:hi])
        if hi - lo < 10:
            for v in x[lo:hi]:
                q.push(v)
        else:
            q.write(x[lo:hi])
        assert q.get_num_transforms() == ref.num_transforms, (lo, hi)
    assert np.max(np.abs(q.get_psd_mag() - ref.get_psd_mag()) / ref.get_psd_mag()) <= 2e-4
    # clear keeps the buffer and the totals (spgram.rs:135-147); reset zeroes everything
    q.clear()
    ref.clear()
    assert (q.get_num_samples(), q.get_num_samples_total(), q.get_num_transforms(), q.get_num_transforms_total()) == \
           (0, 1000, 0, ref.num_transforms_total)
    q.write(x[:delay])
    ref.write(x[:delay])
    assert np.max(np.abs(q.get_psd_mag() - ref.get_psd_mag()) / ref.get_psd_mag()) <= 2e-4
    q.reset()
    assert q.get_num_samples_total() == 0 and q.get_num_transforms_total() == 0
    # exponential averaging: the recurrence psd = gamma psd + alpha |X|^2; linear scale is 0 like the reference
    q.set_alpha(0.1)
    ref2 = oracle.Spgram(nfft, 2, wlen, delay)
    ref2.set_alpha(0.1)
    q.write(x)
    ref2.write(x)
    assert abs(q.get_alpha() - 0.1) < 1e-7
    assert np.all(q.get_psd_mag() == 0) and np.all(ref2.get_psd_mag() == 0)
    q.set_alpha(-1.0)                      # back to accumulate: exposes the recursively averaged psd
    ref2.set_alpha(-1.0)
    assert np.max(np.abs(q.get_psd_mag() - ref2.get_psd_mag()) / ref2.get_psd_mag()) <= 5e-4
    # real-valued samples (Spgram<f32>)
    xr = oracle.gen_real(5, 3000) * np.float32(1e-3)
    qf = ya.Spgram(128, ya.WindowType.Hamming, 100, 33, dtype=np.float32)
    rf = oracle.Spgram(128, 1, 100, 33, dtype=np.float32)
    qf.write(xr)
    rf.write(xr)
    assert np.max(np.abs(qf.get_psd_mag() - rf.get_psd_mag()) / rf.get_psd_mag()) <= 2e-4


def test_spgram_config_and_estimate(ya, oracle):
    """spgram.rs config errors + estimate_psd (:319-330)"""
    W = ya.WindowType
    for bad in [lambda: ya.Spgram(1, W.Hamming, 1, 1), lambda: ya.Spgram(64, W.Hamming, 65, 1),
                lambda: ya.Spgram(64, W.Hamming, 0, 1), lambda: ya.Spgram(64, W.Hamming, 32, 0),
                lambda: ya.Spgram(64, W.Kaiser, 33, 1), lambda: ya.Spgram(64, W.Unknown, 32, 1),
                lambda: ya.Spgram.default(1), lambda: ya.Spgram(1 << 21, W.Hamming, 64, 16)]:
        with pytest.raises(ya.ConfigError):
            bad()
    with pytest.raises(ya.ValueError_):                 # windows::kbd rejects odd lengths (windows.rs:161-163)
        ya.Spgram(64, W.Kbd, 33, 8)
    q = ya.Spgram.default(64)
    with pytest.raises(ya.ConfigError):
        q.set_alpha(1.5)
    with pytest.raises(ya.ConfigError):
        q.set_rate(0.0)
    q.set_rate(1e6)
    q.set_freq(2.4e9)
    assert (q.get_nfft(), q.get_window_len(), q.get_delay(), q.get_wtype()) == (64, 32, 16, W.Kaiser)
    x = noise(oracle, 64 * 40)
    got = ya.Spgram.estimate_psd(64, x)
    want = oracle.Spgram.estimate_psd(64, x)
    assert np.max(np.abs(got - want)) <= 2e-3
    short = x[:5]                                        # fewer samples than one delay: a forced step()
    assert np.max(np.abs(ya.Spgram.estimate_psd(64, short) - oracle.Spgram.estimate_psd(64, short))) <= 2e-3


def test_spgram_device_stream_4096(ya, oracle):
    """the stream -> 4096-point FFT consumer at scale: 2^24 device-resident samples, nfft 4096"""
    n = 1 << 24
    dx = ya.gen_complex_dev(SEED + 2, n)
    q = ya.Spgram(4096, ya.WindowType.Hann, 4096, 2048)
    q.write_dev(dx, n)
    assert q.get_num_transforms() == n // 2048
    psd = q.get_psd()                                   # unit-variance complex noise -> 0 dB flat
    assert np.all(np.abs(psd) <= 0.5)


@pytest.mark.parametrize("dtype", [np.complex64, np.float32])
@pytest.mark.parametrize("nfft", [256, 512, 1024, 2048, 4096])
@pytest.mark.parametrize("shape", [(1.0, 0.5, -1.0), (0.73, 0.19, -1.0), (1.0, 1.0, 0.2), (0.3, 0.025, 0.05)])
def test_spgram_fused_vs_oracle(ya, oracle, dtype, nfft, shape):
    """nfft in {256..4096} takes the fused taper -> FFT -> |X|^2 -> accumulate kernels: bin-level parity with the
    oracle's step-by-step restatement for both sample types, window shorter than nfft, odd delay, split writes
    (window carried across calls: the first frames of a call reach into it), and the recursive-average mode"""
    wlen = max(2, int(shape[0] * nfft))
    wlen += wlen % 2 == 1 and wlen < nfft            # even (Hann does not need it; keeps the shapes tidy)
    delay = max(1, int(shape[1] * nfft) | 1) if shape[1] < 0.5 else int(shape[1] * nfft)
    alpha = shape[2]
    n = 12 * 4096 + 321 if nfft < 4096 else 40 * 4096 + 321
    x = (oracle.gen_real(21, n) if dtype == np.float32 else noise(oracle, n)) * np.float32(3e-2)
    ref = oracle.Spgram(nfft, 2, wlen, delay, dtype=dtype)
    q = ya.Spgram(nfft, ya.WindowType.Hann, wlen, delay, dtype=dtype)
    if alpha >= 0:
        ref.set_alpha(alpha)
        q.set_alpha(alpha)
    for lo, hi in [(0, 5000), (5000, 5003), (5003, n // 2 + 7), (n // 2 + 7, n)]:
        ref.write(x[lo:hi])
        q.write(x[lo:hi])
        assert q.get_num_transforms() == ref.num_transforms
    if alpha >= 0:                         # linear scale is 0 while averaging (reference quirk): expose the estimate
        ref.set_alpha(-1.0)
        q.set_alpha(-1.0)
    a, b = q.get_psd_mag(), ref.get_psd_mag()
    assert np.max(np.abs(a - b) / b) <= 1e-3
    assert np.linalg.norm(a - b) / np.linalg.norm(b) <= 2e-5


@pytest.mark.parametrize("nfft,wlen,delay", [(16384, 16384, 8192), (10000, 7000, 3333)])
def test_spgram_large_nfft_vs_oracle(ya, oracle, nfft, wlen, delay):
    """transforms beyond one workgroup (four-step / Bluestein plans) through the unfused pipeline"""
    n = 12 * nfft + 77
    x = noise(oracle, n) * np.float32(1e-2)
    ref = oracle.Spgram(nfft, 2, wlen, delay)
    q = ya.Spgram(nfft, ya.WindowType.Hann, wlen, delay)
    ref.write(x)
    q.write(x)
    assert q.get_num_transforms() == ref.num_transforms
    a, b = q.get_psd_mag(), ref.get_psd_mag()
    assert np.linalg.norm(a - b) / np.linalg.norm(b) <= 5e-5
