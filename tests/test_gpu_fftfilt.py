"""FftFilt (overlap-add fast convolution) through the C ABI vs the oracle and the reference's 12 golden
sets -- mirrors src/filter/fftfilt.rs:150-330 (config, copy, *_data_h{4,7,13,23}x256; tol 1e-3)."""
import numpy as np
import pytest

from conftest import load_golden
from gpu_util import rand_samples, rand_taps, rel_l2

pytestmark = pytest.mark.gpu
KINDS = ["rrrf", "crcf", "cccf"]


@pytest.fixture(scope="module")
def ya():
    import yagi_amd
    assert yagi_amd.device_count() > 0
    return yagi_amd


def _nextpow2(x):            # math/mod.rs:80-92
    x -= 1
    n = 0
    while x > 0:
        x >>= 1
        n += 1
    return n


@pytest.mark.parametrize("kind", KINDS)
@pytest.mark.parametrize("case", ["h4x256", "h7x256", "h13x256", "h23x256"])
def test_fftfilt_golden(ya, oracle, kind, case):
    g = load_golden("fftfilt")
    h, x, y = (g[f"fftfilt_{kind}_data_{case}_{s}"] for s in "hxy")
    n = 1 << _nextpow2(len(h) - 1)
    q = ya.FftFilt(kind, h, n)
    per_call = np.concatenate([q.execute(x[i:i + n]) for i in range(0, len(x), n)])
    np.testing.assert_allclose(per_call[:len(y)], y, atol=1e-3, rtol=0)
    q.reset()
    batched = q.execute_blocks(x)                      # same samples as one device batch (FFT blocks fall differently)
    np.testing.assert_allclose(batched, per_call, atol=2e-6 * max(1.0, float(np.max(np.abs(per_call)))), rtol=0)
    ref = oracle.FftFilt(kind, h, n)
    want = np.concatenate([ref.execute(x[i:i + n]) for i in range(0, len(x), n)])
    np.testing.assert_allclose(per_call, want, atol=2e-6)


def test_fftfilt_config(ya):
    """fftfilt.rs:151-166"""
    h2 = np.arange(9, dtype=np.float32)
    with pytest.raises(ya.ConfigError):
        ya.FftFilt("rrrf", np.zeros(0, np.float32), 64)
    with pytest.raises(ya.ConfigError):
        ya.FftFilt("rrrf", h2, 7)
    with pytest.raises(ya.ConfigError):
        ya.FftFilt("rrrf", h2, 1 << 22)                # engine limit: 2n <= 2^22
    f = ya.FftFilt("rrrf", h2, 64)
    f.set_scale(3.0)
    assert abs(f.get_scale() - 3.0) < 1e-6
    assert f.get_length() == 9
    with pytest.raises(ya.ConfigError):                # fftfilt.rs:104-106
        f.execute(np.zeros(63, np.float32))


@pytest.mark.parametrize("kind", KINDS)
def test_fftfilt_copy(ya, kind):
    """fftfilt.rs:168-207: h_len 31, n 96 (FFT size 192 = 2^6*3), clone mid-stream, equal outputs"""
    rng = np.random.default_rng(5)
    h = rand_taps(rng, kind, 31)
    n = 96
    q0 = ya.FftFilt(kind, h, n)
    for _ in range(10):
        q0.execute(rand_samples(rng, kind, n))
    q1 = q0.clone()
    for _ in range(10):
        buf = rand_samples(rng, kind, n)
        assert np.array_equal(q0.execute(buf), q1.execute(buf))


@pytest.mark.parametrize("kind", KINDS)
@pytest.mark.parametrize("L,n", [(1, 1), (5, 4), (64, 64), (256, 2048), (257, 256), (1000, 4096), (100, 100), (2049, 2048),
                                 (2050, 2049), (3000, 4096), (2500, 5000),
                                 (9000, 16384)])      # > 2049 taps: the five-stage overlap-add path (2n > 8192: Bluestein / four-step)
def test_fftfilt_equals_direct_form(ya, oracle, kind, L, n):
    """fast convolution == firfilt (f64 truth) for ragged block counts and carried state"""
    rng = np.random.default_rng(L + n)
    h = rand_taps(rng, kind, L)
    nb = 7
    x = rand_samples(rng, kind, nb * n)
    q = ya.FftFilt(kind, h, n)
    q.set_scale(0.5)
    got = np.concatenate([q.execute_blocks(x[: 3 * n]), q.execute(x[3 * n: 4 * n]), q.execute_blocks(x[4 * n:])])
    truth = oracle.fir_block_f64(kind, h, x, scale=0.5)
    assert rel_l2(got, truth) <= 2e-6
    assert np.max(np.abs(got - truth)) <= 1e-5 * max(1.0, float(np.max(np.abs(truth))))


def test_fftfilt_large_batch_dev(ya, oracle):
    """device-pointer form at scale: 256-tap crcf, n = 2048 (FFT 4096), 2^22 samples"""
    h = oracle.fir_design_kaiser(256, 0.2, 60.0)
    n, nb = 2048, 2048
    dx = ya.gen_complex_dev(0x59414749 + 2, n * nb)
    dy = ya.DeviceArray(n * nb, np.complex64)
    q = ya.FftFilt("crcf", h, n)
    q.set_scale(0.4)
    q.execute_blocks_dev(dx, nb, dy)
    ya.synchronize()
    fir = ya.FirFilter("crcf", h)
    fir.set_scale(0.4)
    dz = ya.DeviceArray(n * nb, np.complex64)
    fir.execute_block_dev(dx, n * nb, dz)
    ya.synchronize()
    for off in (0, 12345, n * nb - 5000):
        a, b = dy.to_numpy(5000, offset=off), dz.to_numpy(5000, offset=off)
        assert rel_l2(a, b) <= 2e-6
