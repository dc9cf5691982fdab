"""Pin the CPU oracle (oracle/yagi_oracle.c) against the reference's own golden vectors.

Each test mirrors one reference test (cited) with the reference's tolerance; in addition the
f32 restatement is compared with the f64 truth so the oracle hierarchy is consistent:
  reference golden  ~(ref tol)~  f32 restatement  ~(eps bound)~  f64 truth.
"""
import numpy as np
import pytest

from conftest import load_golden

KINDS = ["rrrf", "crcf", "cccf"]


# ---- firfilt: firfilt.rs:842-1039 (max_relative = 1e-3) ------------------------------------
@pytest.mark.parametrize("kind", KINDS)
@pytest.mark.parametrize("case", ["h4x8", "h7x16", "h13x32", "h23x64"])
def test_firfilt_golden(oracle, kind, case):
    g = load_golden("firfilt")
    h, x, y = (g[f"firfilt_{kind}_data_{case}_{s}"] for s in "hxy")
    q = oracle.FirFilter(kind, h)
    got = q.execute_block(x)
    np.testing.assert_allclose(got, y, rtol=1e-3, atol=1e-6)
    # per-sample path == block path, bit for bit (same code path in the reference)
    q2 = oracle.FirFilter(kind, h)
    per = np.array([q2.execute_one(v) for v in x], dtype=got.dtype)
    assert np.array_equal(per, got)
    truth = oracle.fir_block_f64(kind, h, x)
    assert np.max(np.abs(got - truth)) <= 4 * len(h) * np.finfo(np.float32).eps * np.max(np.abs(truth) + 1e-3)


# ---- firdecim: firdecim.rs:295-473 (epsilon = 1e-3) -----------------------------------------
@pytest.mark.parametrize("kind", KINDS)
@pytest.mark.parametrize("case,M", [("m2h4x20", 2), ("m3h7x30", 3), ("m4h13x40", 4), ("m5h23x50", 5)])
def test_firdecim_golden(oracle, kind, case, M):
    g = load_golden("firdecim")
    h, x, y = (g[f"firdecim_{kind}_data_{case}_{s}"] for s in "hxy")
    assert len(x) == M * len(y)
    q = oracle.FirDecimationFilter(kind, M, h)
    got = q.execute_block(x, len(y))
    np.testing.assert_allclose(got, y, atol=1e-3, rtol=0)
    truth = oracle.fir_block_f64(kind, h, x, M=M, n=len(y))
    np.testing.assert_allclose(got, truth, atol=1e-6)
    # firdecim_block (firdecim.rs:246-279): block == per-call, exact
    q2 = oracle.FirDecimationFilter(kind, M, h)
    per = np.array([q2.execute(x[i * M:(i + 1) * M]) for i in range(len(y))], dtype=got.dtype)
    assert np.array_equal(per, got)


# ---- firpfb: firpfb.rs:310-359 (1e-4) ------------------------------------------------------
def test_firpfb_impulse_response(oracle):
    g = load_golden("firpfb")
    h, noise, test = (g[f"firpfb_impulse_response__{s}"] for s in ("h", "noise", "test"))
    f = oracle.FirPfbFilter("rrrf", 4, h, 48)
    f.write(noise)
    for i, expected in enumerate(test):
        assert abs(f.execute(i) - expected) <= 1e-4
    with pytest.raises(ValueError):
        f.execute(4)          # firpfb.rs:278-280 index out of range -> Err(Config)


# ---- dotprod: dotprod/mod.rs:291-655 --------------------------------------------------------
def test_dotprod_basic(oracle):
    # mod.rs:291-303
    assert oracle.dotprod("rrrf", [1, 2, 3], [4, 5, 6]) == 32.0
    a = np.array([1 + 1j, 2 + 2j, 3 + 3j], np.complex64)
    b = np.array([4 - 4j, 5 - 5j, 6 - 6j], np.complex64)
    assert oracle.dotprod("ccc", a, b) == 64.0 + 0j
    # mod.rs:305-339 basic / uneven
    h = np.array([1, -1] * 8, np.float32)
    assert oracle.dotprod("rrrf", h, np.zeros(16)) == 0
    assert oracle.dotprod("rrrf", h, np.ones(16)) == 0
    assert oracle.dotprod("rrrf", h, np.arange(16) % 2) == -8
    assert oracle.dotprod("rrrf", h, 1 - np.arange(16) % 2) == 8
    assert oracle.dotprod("rrrf", h, h) == 16
    for n, want in [(1, 1), (2, 0), (3, 1), (11, 1), (13, 1), (15, 1)]:
        assert oracle.dotprod("rrrf", h[:n], np.ones(n)) == want


def test_dotprod_golden(oracle):
    g = load_golden("dotprod")

    def chk(got, want, tol):
        assert abs(got - want.item()) <= tol, (got, want)

    t = "test_dotprod_rrrf_rand01"
    chk(oracle.dotprod("rrrf", g[t + "__h"], g[t + "__x"]), g[t + "__test"], 1e-3)
    t = "test_dotprod_rrrf_rand02"
    chk(oracle.dotprod("rrrf", g[t + "__h"], g[t + "__x"]), g[t + "__test"], 1e-3)
    chk(oracle.dotprod("rrrf", g[t + "__h"][::-1], g[t + "__x"]), g[t + "__test_rev"], 1e-3)
    t = "test_dotprod_rrrf_struct_lengths"
    for n in (32, 33, 34, 35):
        chk(oracle.dotprod("rrrf", g[t + "__h"][:n], g[t + "__x"][:n]), g[f"{t}__len{n}"], 2e-6)
    # [f32].[Complex] tests (named crcf in liquid): mod.rs:455-524
    t = "test_dotprod_crcf_rand01"
    chk(oracle.dotprod("rcc", g[t + "__h"], g[t + "__x"]), g[t + "__test"], 1e-3)
    chk(oracle.dotprod("rcc", g[t + "__h"][::-1], g[t + "__x"]), g[t + "__test_rev"], 1e-3)
    t = "test_dotprod_crcf_rand02"
    chk(oracle.dotprod("rcc", g[t + "__h"], g[t + "__x"]), g[t + "__test"], 1e-3)
    t = "test_dotprod_cccf_rand16"
    chk(oracle.dotprod("ccc", g[t + "__h"], g[t + "__x"]), g[t + "__test"], 1e-3)
    chk(oracle.dotprod("ccc", g[t + "__h"][::-1], g[t + "__x"]), g[t + "__test_rev"], 1e-3)
    t = "test_dotprod_cccf_struct_lengths"
    for n in (32, 33, 34, 35):
        chk(oracle.dotprod("ccc", g[t + "__h"][:n], g[t + "__x"][:n]), g[f"{t}__v{n}"], 4e-6)


@pytest.mark.parametrize("kind", ["rrrf", "rcc", "crc", "ccc"])
def test_dotprod_struct_vs_ordinal(oracle, kind):
    """mod.rs:438-453,527-546,658-677: n = 1..512 random vs naive sum (seeded here)."""
    rng = np.random.default_rng(7)
    for n in range(1, 513, 7):
        a = rng.random(n).astype(np.float32) if kind in ("rrrf", "rcc") else (rng.random(n) + 1j * rng.random(n)).astype(np.complex64)
        b = rng.random(n).astype(np.float32) if kind in ("rrrf", "crc") else (rng.random(n) + 1j * rng.random(n)).astype(np.complex64)
        got = oracle.dotprod(kind, a, b)
        want = np.sum(a.astype(np.complex128) * b.astype(np.complex128))
        assert abs(got - want) <= 1e-4 * max(1.0, abs(want))


# ---- fft: fft/mod.rs:125-352 (abs 2e-4 forward and inverse round trip) ----------------------
FFT_SIZES = [2, 3, 4, 5, 6, 7, 8, 9, 10, 16, 17, 20, 21, 22, 24, 26, 30, 32, 35, 36, 43, 48, 63, 64,
             79, 92, 96, 120, 130, 157, 192, 317, 509]


@pytest.mark.parametrize("n", FFT_SIZES)
def test_fft_golden(oracle, n):
    g = load_golden("fft")
    x, y = g[f"fft_test_x{n}"], g[f"fft_test_y{n}"]
    Y = oracle.dft_f64(x)
    assert np.max(np.abs(Y - y)) <= 2e-4
    z = oracle.dft_f64(Y.astype(np.complex64), backward=True) / n
    assert np.max(np.abs(z - x)) <= 2e-4
    if n & (n - 1) == 0:       # the timed f32 radix-4 baseline agrees too
        y32 = oracle.FftPlanF32(n).run(x)
        assert np.max(np.abs(y32 - y)) <= 2e-4
        z32 = oracle.FftPlanF32(n, backward=True).run(y32) / n
        assert np.max(np.abs(z32 - x)) <= 2e-4


def test_fft_f32_4096_vs_definition(oracle):
    x = oracle.gen_complex(0x59414749 + 3, 4096)
    Y = oracle.dft_f64(x)
    y = oracle.FftPlanF32(4096).run(x)
    assert np.linalg.norm(y - Y) / np.linalg.norm(Y) <= 1e-6
    assert np.allclose(Y, np.fft.fft(x.astype(np.complex128)), atol=1e-9)


def test_fft_shift(oracle):
    # fft/mod.rs:77-123
    v = np.arange(4) * (1 + 1j)
    assert np.array_equal(oracle.fft_shift(v), np.array([2, 3, 0, 1]) * (1 + 1j))
    v = np.arange(8) * (1 + 1j)
    assert np.array_equal(oracle.fft_shift(v), np.array([4, 5, 6, 7, 0, 1, 2, 3]) * (1 + 1j))
    v = np.arange(5) * (1 + 1j)     # odd n: last element stays (mod.rs:51)
    assert np.array_equal(oracle.fft_shift(v), np.array([2, 3, 0, 1, 4]) * (1 + 1j))


# ---- Window: window.rs:111-187 (exact) ------------------------------------------------------
def test_windowf(oracle):
    v = [9, 8, 7, 6, 5, 4, 3, 2, 1, 0]
    w = oracle.Window(10)
    assert np.array_equal(w.read(), np.zeros(10))
    for _ in range(4):
        w.push(1.0)
    assert np.array_equal(w.read(), [0, 0, 0, 0, 0, 0, 1, 1, 1, 1])
    w.write(v[0:4])
    assert np.array_equal(w.read(), [0, 0, 1, 1, 1, 1, 9, 8, 7, 6])
    for _ in range(4):
        w.push(3.0)
    assert np.array_equal(w.read(), [1, 1, 9, 8, 7, 6, 3, 3, 3, 3])
    assert [w.index(i) for i in range(10)] == [1, 1, 9, 8, 7, 6, 3, 3, 3, 3]
    with pytest.raises(IndexError):
        w.index(999)
    for _ in range(4):
        w.push(5.0)
    assert np.array_equal(w.read(), [7, 6, 3, 3, 3, 3, 5, 5, 5, 5])
    w.resize(6)
    assert np.array_equal(w.read(), [3, 3, 5, 5, 5, 5])
    w.push(6.0)
    w.push(7.0)
    assert np.array_equal(w.read(), [5, 5, 5, 5, 6, 7])
    w.resize(10)
    assert np.array_equal(w.read(), [0, 0, 0, 0, 5, 5, 5, 5, 6, 7])
    w.reset()
    assert np.array_equal(w.read(), np.zeros(10))
    with pytest.raises(ValueError):
        oracle.Window(0)


def test_window_long_run_matches_tail(oracle):
    rng = np.random.default_rng(3)
    for n in (1, 2, 3, 7, 8, 9, 63, 64, 65):
        w = oracle.Window(n, np.complex64)
        xs = (rng.standard_normal(5 * n + 11) + 1j * rng.standard_normal(5 * n + 11)).astype(np.complex64)
        w.write(xs)
        assert np.array_equal(w.read(), xs[-n:])
        c = w.clone()
        c.push(1j)
        w.push(1j)
        assert np.array_equal(w.read(), c.read())


# ---- design mask: firfilt.rs:355-369 (new_kaiser(51, 0.2, 60, 0), scale 0.4) ----------------
def test_kaiser_design_mask(oracle):
    h = oracle.fir_design_kaiser(51, 0.2, 60.0, 0.0)
    f = np.linspace(-0.5, 0.5, 1201)
    H = np.array([np.sum(h * np.exp(-2j * np.pi * fk * np.arange(51))) for fk in f]) * 0.4
    dB = 20 * np.log10(np.abs(H) + 1e-300)
    assert np.all(np.abs(dB[np.abs(f) <= 0.15]) <= 0.1)
    assert np.all(dB[np.abs(f) >= 0.25] <= -60)
    # f64 cross-check of the f32 Bessel series
    from math import isclose
    assert isclose(float(np.sum(h)), 2.5, rel_tol=2e-3)
    with pytest.raises(ValueError):
        oracle.fir_design_kaiser(0, 0.2, 60.0)
    with pytest.raises(ValueError):
        oracle.fir_design_kaiser(10, 0.6, 60.0)
    with pytest.raises(ValueError):
        oracle.fir_design_kaiser(10, 0.2, 60.0, mu=0.7)


def test_generator_is_counter_based(oracle):
    a = oracle.gen_complex(42, 1000)
    b = oracle.gen_complex(42, 500, first=500)
    assert np.array_equal(a[500:], b)
    assert abs(np.var(a.real) - 0.5) < 0.06 and abs(np.var(a.imag) - 0.5) < 0.06
    r = oracle.gen_real(42, 4000)
    assert abs(np.var(r) - 1.0) < 0.08


# ---- fftfilt: filter/fftfilt.rs:150-330, data filter/test_data.rs (tol 1e-3) ------------------
def _nextpow2(x):            # math/mod.rs:80-92
    x -= 1
    n = 0
    while x > 0:
        x >>= 1
        n += 1
    return n


@pytest.mark.parametrize("kind", KINDS)
@pytest.mark.parametrize("case", ["h4x256", "h7x256", "h13x256", "h23x256"])
def test_fftfilt_golden(oracle, kind, case):
    g = load_golden("fftfilt")
    h, x, y = (g[f"fftfilt_{kind}_data_{case}_{s}"] for s in "hxy")
    n = 1 << _nextpow2(len(h) - 1)
    q = oracle.FftFilt(kind, h, n)
    got = np.concatenate([q.execute(x[i:i + n]) for i in range(0, len(x), n)])
    np.testing.assert_allclose(got[:len(y)], y, atol=1e-3, rtol=0)
    # fast convolution == direct form (the golden sets are firfilt outputs of the same h, x)
    np.testing.assert_allclose(got[:len(y)], oracle.fir_block_f64(kind, h, x)[:len(y)], atol=1e-5)
    with pytest.raises(ValueError):
        oracle.FftFilt(kind, h[:0], 64)
    with pytest.raises(ValueError):
        oracle.FftFilt(kind, np.arange(9), 7)


# ---- firinterp: firinterp.rs:277-387 (*_generic, tol 1e-6) -------------------------------------
@pytest.mark.parametrize("kind", ["rrrf", "crcf"])
def test_firinterp_generic(oracle, kind):
    g = load_golden("firinterp")
    h, x, test = (g[f"firinterp_{kind}_generic__{s}"] for s in ("h", "x", "test"))
    q = oracle.FirInterpolationFilter(kind, 4, h)
    y = np.concatenate([q.execute(v) for v in x])
    np.testing.assert_allclose(y, test, atol=1e-6, rtol=0)
    with pytest.raises(ValueError):
        oracle.FirInterpolationFilter(kind, 1, h)
    with pytest.raises(ValueError):
        oracle.FirInterpolationFilter(kind, 12, h)


# ---- spgram (oracle self-check against the reference's acceptance criterion, small sizes) -------
@pytest.mark.parametrize("nfft,wtype,wlen,delay", [(64, 5, 32, 16), (100, 1, 100, 25), (96, 9, 48, 24)])
def test_spgram_oracle_noise_floor(oracle, nfft, wtype, wlen, delay):
    """spgram.rs:339-367: AWGN at -80 dB -> every PSD bin within +-0.5 dB (here 400*nfft samples, +-1 dB)"""
    q = oracle.Spgram(nfft, wtype, wlen, delay)
    n = 400 * nfft
    x = oracle.gen_complex(77, n) * np.float32(10 ** (-80 / 20))
    q.write(x)
    assert q.num_samples == n and q.num_samples_total == n
    assert q.num_transforms == n // delay
    psd = q.get_psd()
    assert np.all(np.abs(psd + 80.0) <= 1.0)
    q.clear()
    assert q.num_samples == 0 and q.num_samples_total == n and q.num_transforms == 0


# ---- rresamp: the reference's own acceptance test for the schedule (rresamp.rs:198-238) ---------
@pytest.mark.parametrize("P", [1, 2, 3, 6, 8, 9])
def test_rresamp_oracle_partition(oracle, P, Q=5, m=15, n=20):
    """autotest_rresamp_crcf_part_P*_Q5 on the restatement: one 2n-block run == n blocks + a second resampler
    primed with the last m*Q inputs by write(); plus the P outputs / Q inputs bookkeeping of execute_primitive"""
    q0 = oracle.Rresamp.new_kaiser("crcf", P, Q, m, 0.5, 60.0)
    q1 = oracle.Rresamp.new_kaiser("crcf", P, Q, m, 0.5, 60.0)
    N = 2 * Q * n
    i = np.arange(N)
    ham = (0.53836 - 0.46164 * np.cos(2 * np.pi * i / (N - 1))).astype(np.float32)
    x = (ham * np.exp(2j * np.pi * 0.037 * i)).astype(np.complex64)
    y0 = q0.execute_block(x, 2 * n)
    assert y0.shape == (2 * P * n,)
    q0.reset()
    ya = q0.execute_block(x[:Q * n], n)
    for k in range(m):
        q1.write(x[Q * n - (m - k) * Q: Q * n - (m - k - 1) * Q])
    yb = q1.execute_block(x[Q * n:], n)
    np.testing.assert_allclose(np.concatenate([ya, yb]), y0, atol=1e-12, rtol=0)


def test_fft4096_fixture_is_reproducible():
    """tests/golden/fft4096.npz (the N = 4096 definition vectors: no reference vector exists above N = 509) is what
    tests/golden/make_golden.py::fft4096_definition_vectors regenerates from the repo alone"""
    import importlib.util
    from pathlib import Path
    path = Path(__file__).resolve().parent / "golden" / "make_golden.py"
    spec = importlib.util.spec_from_file_location("make_golden", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    assert mod.fft4096_definition_vectors(check_only=True)
