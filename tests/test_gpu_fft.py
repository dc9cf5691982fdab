"""HIP FFT through the C ABI vs the f64 DFT definition and the reference's 33 golden sizes
(src/fft/mod.rs:77-352; abs tol 2e-4 forward and on the inverse round trip)."""
import numpy as np
import pytest

from conftest import load_golden
from gpu_util import SEED, rel_l2

pytestmark = pytest.mark.gpu

FFT_SIZES = [2, 3, 4, 5, 6, 7, 8, 9, 10, 16, 17, 20, 21, 22, 24, 26, 30, 32, 35, 36, 43, 48, 63, 64,
             79, 92, 96, 120, 130, 157, 192, 317, 509]


@pytest.fixture(scope="module")
def ya():
    import yagi_amd
    assert yagi_amd.device_count() > 0
    return yagi_amd


@pytest.mark.parametrize("n", FFT_SIZES)
def test_fft_golden(ya, n):
    g = load_golden("fft")
    x, test = g[f"fft_test_x{n}"], g[f"fft_test_y{n}"]
    fwd = ya.Fft(n, ya.Direction.Forward)
    bwd = ya.Fft(n, ya.Direction.Backward)
    y = fwd.run(x)
    z = bwd.run(y) / np.float32(n)
    assert np.max(np.abs(y - test)) <= 2e-4
    assert np.max(np.abs(z - x)) <= 2e-4
    assert np.max(np.abs(ya.fft_run(x, ya.Direction.Forward) - test)) <= 2e-4


def test_fft_shift(ya):
    """fft/mod.rs:77-123 (+ odd n: last element stays, :51)"""
    f = ya.Fft(4, ya.Direction.Forward)
    for n, want in [(4, [2, 3, 0, 1]), (8, [4, 5, 6, 7, 0, 1, 2, 3]), (5, [2, 3, 0, 1, 4]), (1, [0])]:
        v = (np.arange(n) * (1 + 1j)).astype(np.complex64)
        assert np.array_equal(f.shift(v, n), (np.array(want) * (1 + 1j)).astype(np.complex64))


def test_fft_config(ya):
    with pytest.raises(ya.ConfigError):
        ya.Fft(0, ya.Direction.Forward)
    with pytest.raises(ya.ConfigError):
        ya.Fft(1 << 25, ya.Direction.Forward)          # documented size limit (2^24)
    f = ya.Fft(16, ya.Direction.Forward)
    with pytest.raises(ya.ConfigError):                # the reference panics (copy_from_slice)
        f.run(np.zeros(15, np.complex64))
    assert np.array_equal(ya.Fft(1, ya.Direction.Forward).run(np.complex64([3 - 2j])), np.complex64([3 - 2j]))


@pytest.mark.parametrize("n", [2, 16, 32, 64, 128, 256, 512, 1000, 1024, 2048, 3125, 4096, 8192, 4093,
                               1031, 2039, 2042, 2053, 4083, 4095 - 4])   # the last six: Bluestein in one kernel, m = 4096 / 8192
@pytest.mark.parametrize("direction", ["Forward", "Backward"])
def test_fft_vs_f64_definition(ya, oracle, n, direction):
    """parity unpinned by the reference above N = 509; pinned by the definition (f64 DFT)."""
    rng = np.random.default_rng(n)
    x = ((rng.standard_normal(3 * n) + 1j * rng.standard_normal(3 * n)) * np.sqrt(0.5)).astype(np.complex64)
    d = ya.Direction[direction]
    got = ya.Fft(n, d).run_batch(x)
    for b in range(3):
        truth = oracle.dft_f64(x[b * n:(b + 1) * n], backward=(d == ya.Direction.Backward))
        assert rel_l2(got[b], truth) <= 1e-5
        # reference abs tol 2e-4 at N<=509 scaled by the sqrt(N) growth of output magnitude
        assert np.max(np.abs(got[b] - truth)) <= 2e-4 * max(1.0, np.sqrt(n / 509))


def test_fft_4096_golden_fixture(ya, oracle):
    """committed fixture: 4 transforms of the C3 stream vs numpy f64 (tests/golden/fft4096.npz)"""
    g = load_golden("fft4096")
    x, Y = g["x"], g["y"]
    assert np.array_equal(x, oracle.gen_complex(SEED + 3, 4 * 4096))      # the generator is pinned too
    got = ya.Fft(4096, ya.Direction.Forward).run_batch(x)
    for b in range(4):
        assert rel_l2(got[b], Y[b]) <= 1e-5
        assert np.max(np.abs(got[b] - Y[b])) <= 5.7e-4
    back = ya.Fft(4096, ya.Direction.Backward).run_batch(got.reshape(-1)) / np.float32(4096)
    assert np.max(np.abs(back.reshape(-1) - x)) <= 2e-4


def test_fft_impulse_and_tone_are_exact_bins(ya):
    """indexing: impulse at n0 -> linear phase; tone at bin k -> energy in bin k only"""
    n = 4096
    f = ya.Fft(n, ya.Direction.Forward)
    x = np.zeros(n, np.complex64)
    x[0] = 1
    assert np.array_equal(f.run(x), np.ones(n, np.complex64))
    for k in (1, 17, 255, 256, 2049, 4095):
        t = np.exp(2j * np.pi * k * np.arange(n) / n).astype(np.complex64)
        y = f.run(t)
        assert np.argmax(np.abs(y)) == k and abs(y[k] - n) < 0.05
        y[k] = 0
        assert np.max(np.abs(y)) < 0.02


def test_config_c3_fft_4096_batch_65536(ya, oracle):
    """BASELINE config C3 at full size (2 GiB in, 2 GiB out) on device-generated input:
    sampled transforms vs the f64 definition + linearity (size-independent property)."""
    n, batch = 4096, 65536
    total = n * batch
    dx = ya.gen_complex_dev(SEED + 3, total)
    dy = ya.DeviceArray(total, np.complex64)
    plan = ya.Fft(n, ya.Direction.Forward)
    plan.run_batch_dev(dx, dy, batch)
    ya.synchronize()
    for b in (0, 1, 4095, 32768, 65535):
        xb = dx.to_numpy(n, offset=b * n)
        yb = dy.to_numpy(n, offset=b * n)
        truth = oracle.dft_f64(xb)
        assert rel_l2(yb, truth) <= 1e-5, b
        assert np.max(np.abs(yb - truth)) <= 5.7e-4, b
    # device generator == oracle generator up to libm ulps on the first transform
    assert np.max(np.abs(dx.to_numpy(n) - oracle.gen_complex(SEED + 3, n))) <= 1e-5
    # Parseval over a strided sample of transforms
    for b in (7, 12345, 54321):
        xb, yb = dx.to_numpy(n, offset=b * n), dy.to_numpy(n, offset=b * n)
        e_in = np.sum(np.abs(xb.astype(np.complex128)) ** 2)
        e_out = np.sum(np.abs(yb.astype(np.complex128)) ** 2) / n
        assert abs(e_in - e_out) <= 1e-5 * e_in
    dx.free()
    dy.free()


@pytest.mark.parametrize("n", [16, 128, 256, 512, 1024, 2048, 8192, 100, 1000, 509, 2039, 4093])
def test_fft_large_batch_on_device(ya, oracle, n):
    """many workgroups / several transforms per workgroup / a partial last workgroup (and, for 509, several
    passes through the Bluestein scratch; 2039 and 4093: the one-kernel Bluestein form, odd n = transforms that start
    on 8-byte boundaries only): 2^21+ points on device buffers, sampled transforms vs the f64 DFT"""
    batch = (1 << 21) // n + 3
    dx = ya.gen_complex_dev(SEED + 3, batch * n)
    dy = ya.DeviceArray(batch * n, np.complex64)
    plan = ya.Fft(n, ya.Direction.Forward)
    plan.run_batch_dev(dx, dy, batch)
    ya.synchronize()
    for b in (0, 1, 5, batch // 2, batch - 2, batch - 1):
        truth = oracle.dft_f64(dx.to_numpy(n, offset=b * n))
        assert rel_l2(dy.to_numpy(n, offset=b * n), truth) <= 1e-5, b


@pytest.mark.parametrize("n", [16384, 32768, 65536, 1 << 17, 1 << 20, 1 << 22, 1 << 24, 10000, 12289, 8193, 100003, 48000, 100000, 30030, 9973 * 4,
                               9216, 250000, 130321, 512 * 511, 1000000, 3000000])
@pytest.mark.parametrize("direction", ["Forward", "Backward"])
def test_fft_beyond_one_workgroup(ya, n, direction):
    """n > 8192: 2^14 and 2^15 by the two-launch column / row form over the LDS core, powers of two from 2^16 up as
    256 x n2 (256-point columns in registers, then the n2-point rows: two launches at 2^16, three above), smooth
    sizes with both factors up to 1024 by the two-launch mixed-radix form (fft_mixed_twopass_kernel; 10 000 = 100 x 100
    with a ragged last column tile, 130 321 = 361 x 361 with radix-19 direct sums, 10^6 = 1000 x 1000 with four columns
    per workgroup), larger smooth sizes by the four-step form (transposes around the one-kernel transforms: 3 x 10^6 =
    2000 x 1500), every other size by Bluestein over a power of two.
    Truth: numpy's f64 FFT of the same f32 samples."""
    rng = np.random.default_rng(n)
    batch = 2 if n < (1 << 22) else 1
    x = ((rng.standard_normal(batch * n) + 1j * rng.standard_normal(batch * n)) * np.sqrt(0.5)).astype(np.complex64)
    d = ya.Direction[direction]
    got = ya.Fft(n, d).run_batch(x)
    for b in range(batch):
        xb = x[b * n:(b + 1) * n].astype(np.complex128)
        truth = np.fft.fft(xb) if d == ya.Direction.Forward else np.fft.ifft(xb) * n
        assert rel_l2(got[b], truth) <= 1e-5, b
    if n in (16384, 65536, 10000, 48000):                     # in place on the device
        dx = ya.DeviceArray.from_numpy(x)
        ya.Fft(n, d).run_batch_dev(dx, dx, batch)
        ya.synchronize()
        assert np.array_equal(dx.to_numpy().reshape(batch, n), got.reshape(batch, n))


def test_fft_size_limits(ya):
    with pytest.raises(ya.ConfigError):
        ya.Fft((1 << 24) + 1, ya.Direction.Forward)         # not a power of two and 2n-1 > 2^24
    with pytest.raises(ya.ConfigError):
        ya.Fft(1 << 25, ya.Direction.Forward)


@pytest.mark.parametrize("n,total", [(4096, 1 << 26), (1024, 1 << 24), (8192, 1 << 24), (100, 100 * 50000), (65536, 1 << 24)])
def test_fft_round_trip_whole_buffer(ya, n, total):
    """size-independent property at scale: backward(forward(x)) = n x, every point of a large device buffer compared"""
    batch = total // n
    dx = ya.gen_complex_dev(SEED + 3, batch * n)
    dX = ya.DeviceArray(batch * n, np.complex64)
    dy = ya.DeviceArray(batch * n, np.complex64)
    ya.Fft(n, ya.Direction.Forward).run_batch_dev(dx, dX, batch)
    ya.Fft(n, ya.Direction.Backward).run_batch_dev(dX, dy, batch)
    ya.synchronize()
    x, y = dx.to_numpy(), dy.to_numpy()
    err = np.linalg.norm(y / np.float32(n) - x) / np.linalg.norm(x)
    assert err <= 2e-6, err
    # Parseval on the forward spectra: sum |X|^2 = n sum |x|^2
    X = dX.to_numpy()
    e_X = np.sum(np.abs(X).astype(np.float64) ** 2)
    e_x = np.sum(np.abs(x).astype(np.float64) ** 2)
    assert abs(e_X / (n * e_x) - 1.0) <= 1e-6


def test_fft_every_small_size_and_random_large_sizes(ya):
    """every n in 1..160 and 60 seeded sizes up to 40000 (register kernels, mixed radix, direct-sum primes, Bluestein,
    four-step -- whatever the plan picks), forward and backward, against numpy's f64 FFT"""
    rng = np.random.default_rng(7)
    sizes = list(range(1, 161)) + sorted(int(v) for v in rng.integers(161, 40001, 60))
    for n in sizes:
        x = ((rng.standard_normal(2 * n) + 1j * rng.standard_normal(2 * n)) * np.sqrt(0.5)).astype(np.complex64)
        for d, ref in ((ya.Direction.Forward, np.fft.fft), (ya.Direction.Backward, lambda v: np.fft.ifft(v) * len(v))):
            got = ya.Fft(n, d).run_batch(x)
            for b in range(2):
                truth = ref(x[b * n:(b + 1) * n].astype(np.complex128))
                assert rel_l2(got[b], truth) <= 1e-5, (n, d, b)
