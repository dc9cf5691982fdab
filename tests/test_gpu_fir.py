"""HIP FIR objects through the C ABI vs the oracle: FirFilter / FirDecimationFilter / FirPfbFilter.
Mirrors the reference's tests (firfilt.rs:355-1039, firdecim.rs:216-473, firpfb.rs:310-396)."""
import numpy as np
import pytest

from conftest import load_golden
from gpu_util import SEED, fir_bound, int_samples, int_taps, rand_samples, rand_taps, rel_l2

pytestmark = pytest.mark.gpu
KINDS = ["rrrf", "crcf", "cccf"]


@pytest.fixture(scope="module")
def ya():
    import yagi_amd
    assert yagi_amd.device_count() > 0
    return yagi_amd


# ---------------------------------------------------------------------------------- firfilt
@pytest.mark.parametrize("kind", KINDS)
@pytest.mark.parametrize("case", ["h4x8", "h7x16", "h13x32", "h23x64"])
def test_firfilt_golden(ya, kind, case):
    """firfilt.rs:842-1039, max_relative 1e-3"""
    g = load_golden("firfilt")
    h, x, y = (g[f"firfilt_{kind}_data_{case}_{s}"] for s in "hxy")
    q = ya.FirFilter(kind, h)
    np.testing.assert_allclose(q.execute_block(x), y, rtol=1e-3, atol=1e-6)
    # per-sample path: push()+execute() == execute_one() == block
    q1, q2 = ya.FirFilter(kind, h), ya.FirFilter(kind, h)
    for xi, yi in zip(x, y):
        q1.push(xi)
        a, b = q1.execute(), q2.execute_one(xi)
        assert abs(a - yi) <= 1e-3 * abs(yi) + 1e-6 and abs(b - yi) <= 1e-3 * abs(yi) + 1e-6
    if kind == "crcf":
        for choice in (1, 2, 3):                    # general / sliding / MFMA Toeplitz kernels
            q = ya.FirFilter(kind, h)
            q.set_kernel(choice)
            np.testing.assert_allclose(q.execute_block(x), y, rtol=1e-3, atol=1e-6)


@pytest.mark.parametrize("kind", KINDS)
def test_firfilt_config_and_accessors(ya, kind):
    """firfilt.rs:435-457 + accessors :285-314"""
    T, Cdt = ya.KINDS[kind]
    with pytest.raises(ya.ConfigError):
        ya.FirFilter(kind, np.zeros(0, Cdt))
    with pytest.raises(ya.ConfigError):
        ya.FirFilter.new_rect(kind, 0)
    with pytest.raises(ya.ConfigError):
        ya.FirFilter.new_rect(kind, 1025)
    with pytest.raises(ya.ConfigError):
        ya.FirFilter.new_kaiser(kind, 0, 0.2, 60.0, 0.0)
    q = ya.FirFilter.new_rect(kind, 5)
    assert q.get_length() == 5 and np.all(q.get_coefficients() == 1)
    assert q.get_scale() == 1
    q.set_scale(3.0)
    assert q.get_scale() == 3.0
    with pytest.raises(ya.ConfigError):            # firfilt.rs:268-270
        q.execute_block(np.zeros(4, T), np.zeros(5, T))
    y = q.execute_block(np.ones(8, T))
    assert np.array_equal(y, 3.0 * np.minimum(np.arange(1, 9), 5).astype(T))
    assert q.execute_block(np.zeros(0, T)).size == 0          # empty block is a no-op


@pytest.mark.parametrize("kind", KINDS)
def test_firfilt_state_push_write_clone_reset(ya, oracle, kind):
    """firfilt_push_write (firfilt.rs:512-538), *_copy (:540-588), reset (:209-213)"""
    rng = np.random.default_rng(21)
    h = rand_taps(rng, kind, 21)
    x = rand_samples(rng, kind, 160)
    ref = oracle.FirFilter(kind, h)
    want = ref.execute_block(x)
    q = ya.FirFilter(kind, h)
    # mixed usage: block, then push/write + execute, then block again
    got = np.empty_like(x)
    got[:50] = q.execute_block(x[:50])
    for i in range(50, 60):
        q.push(x[i])
        got[i] = q.execute()
    q.write(x[60:70])                   # write() = push each, output only after the last
    got[69] = q.execute()
    got[60:69] = want[60:69]
    got[70:] = q.execute_block(x[70:])
    tol = fir_bound(kind, h, x)
    assert np.max(np.abs(got - oracle.fir_block_f64(kind, h, x))) <= tol
    # clone continues identically (bitwise: same kernels, same state)
    c = q.clone()
    extra = rand_samples(rng, kind, 40)
    assert np.array_equal(q.execute_block(extra), c.execute_block(extra))
    assert q.execute_one(1.0) == c.execute_one(1.0)
    # reset -> zero history again
    q.reset()
    np.testing.assert_allclose(q.execute_block(x[:30]), want[:30], atol=tol)
    # set_coefficients resets state and may change the length (firfilt.rs:193-206)
    h2 = rand_taps(rng, kind, 9)
    q.set_coefficients(h2)
    assert q.get_length() == 9
    np.testing.assert_allclose(q.execute_block(x[:30]), oracle.fir_block_f64(kind, h2, x[:30]), atol=fir_bound(kind, h2, x))


@pytest.mark.parametrize("kind", KINDS)
@pytest.mark.parametrize("L,n", [(1, 100), (2, 1), (31, 4097), (63, 10_000), (256, 20_000), (257, 5000), (1500, 9000)])
def test_firfilt_random_vs_f64(ya, oracle, kind, L, n):
    rng = np.random.default_rng(L * 7 + n)
    h, x = rand_taps(rng, kind, L), rand_samples(rng, kind, n)
    q = ya.FirFilter(kind, h)
    q.set_scale(0.4)
    # ragged blocks with carried state
    cuts = sorted(set([0, n // 3, n // 3 + 1, (2 * n) // 3, n]))
    got = np.concatenate([q.execute_block(x[a:b]) for a, b in zip(cuts[:-1], cuts[1:])])
    truth = oracle.fir_block_f64(kind, h, x, scale=0.4)
    assert np.max(np.abs(got - truth)) <= fir_bound(kind, h, x)
    assert rel_l2(got, truth) <= 2e-6


@pytest.mark.parametrize("kind", KINDS)
def test_firfilt_integer_inputs_bit_exact(ya, oracle, kind):
    """indexing (which sample meets which tap) is exact: integer data => bitwise equality"""
    rng = np.random.default_rng(99)
    for L, n in [(7, 300), (64, 5000), (256, 9000)]:
        h, x = int_taps(rng, kind, L), int_samples(rng, kind, n)
        want = oracle.FirFilter(kind, h).execute_block(x)
        assert np.array_equal(ya.FirFilter(kind, h).execute_block(x), want)
        if kind == "crcf":
            for choice in (1, 2, 3):
                q = ya.FirFilter(kind, h)
                q.set_kernel(choice)
                a = q.execute_block(x[: n // 2])
                b = q.execute_block(x[n // 2:])
                assert np.array_equal(np.concatenate([a, b]), want)


@pytest.mark.parametrize("kind", KINDS)
@pytest.mark.parametrize("L", [1, 7, 8, 9, 17, 63, 64, 300, 1025])
def test_firfilt_register_window_kernel(ya, oracle, kind, L):
    """Blocks of >= 512 samples take the register-window direct kernel (8 consecutive outputs per lane), shorter
    ones the interleaved-output kernel; both add the taps in the same order, so a stream cut either way must
    give the same bits, and both must meet the f64 truth."""
    rng = np.random.default_rng(7000 + L)
    n = 2 * 2048 + 2048 + 513 + 512 + 2049
    h, x = rand_taps(rng, kind, L), rand_samples(rng, kind, n)
    scale = (0.5 + 0.25j) if kind == "cccf" else -0.5
    q = ya.FirFilter(kind, h)
    q.set_scale(scale)
    cuts = np.cumsum([0, 2 * 2048, 2048, 513, 512, 2049])
    big = np.concatenate([q.execute_block(x[a:b]) for a, b in zip(cuts[:-1], cuts[1:])])
    q.reset()
    small = np.concatenate([q.execute_block(x[a:a + 500]) for a in range(0, n, 500)])
    assert np.array_equal(big, small)
    truth = oracle.fir_block_f64(kind, h, x, scale=scale)
    assert np.max(np.abs(big - truth)) <= fir_bound(kind, h, x) * abs(scale) * 1.5 + 1e-30
    assert rel_l2(big, truth) <= 2e-6
    # a NaN in the stream poisons outputs s0 .. s0+L-1 and nothing else (taps past L are skipped, not zeroed);
    # crcf's default is the sliding kernel, whose zero-padded taps widen the region -- kernel 1 is the exact one
    xn = x.copy()
    xn[5000] = np.nan
    q.reset()
    q.set_kernel(1)
    gn = q.execute_block(xn)
    bad = np.flatnonzero(np.isnan(gn.real) if np.iscomplexobj(gn) else np.isnan(gn))
    assert np.array_equal(bad, np.arange(5000, min(5000 + L, n)))


@pytest.mark.parametrize("L", [1, 3, 16, 63, 64, 65, 100, 128, 129, 200, 255, 256])
def test_firfilt_crcf_mfma_kernel_lengths(ya, oracle, L):
    """MFMA Toeplitz form (kernel 3) across the padded-length classes 64/128/256 and ragged blocks"""
    rng = np.random.default_rng(1000 + L)
    h, x = rand_taps(rng, "crcf", L), rand_samples(rng, "crcf", 3 * 4096 + 777)
    q = ya.FirFilter("crcf", h)
    q.set_kernel(3)
    q.set_scale(-1.5)
    got = np.concatenate([q.execute_block(x[:5000]), q.execute_block(x[5000:5001]), q.execute_block(x[5001:])])
    truth = oracle.fir_block_f64("crcf", h, x, scale=-1.5)
    assert np.max(np.abs(got - truth)) <= fir_bound("crcf", h, x) * 1.5
    assert rel_l2(got, truth) <= 2e-6


@pytest.mark.parametrize("kind", KINDS)
@pytest.mark.parametrize("L", [1, 2, 33, 256, 1000, 2049])
def test_firfilt_fast_convolution_kernel(ya, oracle, kind, L):
    """kernel 4: overlap-save fast convolution (FFT -> product -> IFFT in registers; rrrf: two blocks per
    transform; cccf: complex taps and scale).  Same tolerance as the direct kernels vs the f64 truth;
    integer exactness is NOT claimed for it."""
    rng = np.random.default_rng(4000 + L)
    h, x = rand_taps(rng, kind, L), rand_samples(rng, kind, 5 * 4096 + 123)
    scale = (0.75 - 0.5j) if kind == "cccf" else 0.75
    q = ya.FirFilter(kind, h)
    q.set_kernel(4)
    q.set_scale(scale)
    got = np.concatenate([q.execute_block(x[:7000]), q.execute_block(x[7000:7001]), q.execute_block(x[7001:])])
    truth = oracle.fir_block_f64(kind, h, x, scale=scale)
    assert np.max(np.abs(got - truth)) <= fir_bound(kind, h, x) * abs(scale) / 0.75
    assert rel_l2(got, truth) <= 2e-6
    q.set_coefficients(h[::-1].copy())            # new taps -> the cached FFT{h} must be rebuilt
    q.set_scale(scale)
    got = q.execute_block(x[:5000])
    assert rel_l2(got, oracle.fir_block_f64(kind, h[::-1], x[:5000], scale=scale)) <= 2e-6
    if L > 2049 - 1:
        with pytest.raises(ya.ConfigError):
            big = ya.FirFilter(kind, rand_taps(rng, kind, 2050))
            big.set_kernel(4)
            big.execute_block(x[:4096])


@pytest.mark.parametrize("kind", KINDS)
def test_firfilt_kernel_switch_keeps_state(ya, oracle, kind):
    """one stream, kernel switched between calls (overlap-save for the long block, direct form for the rest):
    state carried across the switch, odd number of convolution blocks (rrrf pairs them), output not a multiple
    of the block length; the auto choice is the direct form"""
    rng = np.random.default_rng(77)
    L, n = 129, 8 * 3968 + 1234
    h, x = rand_taps(rng, kind, L), rand_samples(rng, kind, n + 3000)
    truth = oracle.fir_block_f64(kind, h, x)
    q = ya.FirFilter(kind, h)
    q.set_kernel(4)
    a = q.execute_block(x[:n])
    q.set_kernel(0)
    got = np.concatenate([a, q.execute_block(x[n:n + 1000]), q.execute_block(x[n + 1000:])])
    assert np.max(np.abs(got - truth)) <= fir_bound(kind, h, x)
    assert rel_l2(got, truth) <= 2e-6
    d = ya.FirFilter(kind, h)
    direct = d.execute_block(x)                             # auto
    d1 = ya.FirFilter(kind, h)
    d1.set_kernel(1)
    if kind != "crcf":
        assert np.array_equal(direct, d1.execute_block(x))  # auto == general direct form
    assert rel_l2(got, direct) <= 2e-6
    assert not np.array_equal(got[:n], direct[:n])          # the long block really took the other kernel
    assert rel_l2(got[n:], direct[n:]) <= 1e-6
    with pytest.raises(ya.ConfigError):
        q.set_kernel(7)
    if kind != "crcf":
        with pytest.raises(ya.ConfigError):
            q.set_kernel(2)


def test_config_c1_firfilt_rrrf_63tap_1M(ya, oracle):
    """BASELINE config C1: kaiser(63, 0.2, 60), scale 0.4, 1 048 576 real samples"""
    h = oracle.fir_design_kaiser(63, 0.2, 60.0)
    x = oracle.gen_real(SEED + 1, 1 << 20)
    q = ya.FirFilter.new_kaiser("rrrf", 63, 0.2, 60.0, 0.0)
    np.testing.assert_allclose(q.get_coefficients(), h, rtol=2e-6, atol=1e-9)
    q.set_coefficients(h)
    q.set_scale(0.4)
    got = q.execute_block(x)
    ref = oracle.FirFilter("rrrf", h)
    ref.set_scale(0.4)
    want32 = ref.execute_block(x)
    truth = oracle.fir_block_f64("rrrf", h, x, scale=0.4)
    assert np.max(np.abs(got - truth)) <= fir_bound("rrrf", h, x)
    assert rel_l2(got, truth) <= 1e-6
    assert rel_l2(got, truth) <= rel_l2(want32, truth) * 1.5 + 1e-8       # no worse than the reference's own order


def test_config_c2_firfilt_crcf_256tap_stream(ya, oracle):
    """BASELINE config C2 (parity subset): kaiser(256,0.2,60), scale 0.4, first 2^20 samples in blocks"""
    h = oracle.fir_design_kaiser(256, 0.2, 60.0)
    n = 1 << 20
    x = oracle.gen_complex(SEED + 2, n)
    truth = oracle.fir_block_f64("crcf", h, x, scale=0.4)
    for choice in (0, 1, 2, 3, 4):               # 4 = fast convolution (overlap-save, tolerance only)
        q = ya.FirFilter("crcf", h)
        q.set_scale(0.4)
        q.set_kernel(choice)
        got = np.concatenate([q.execute_block(x[i:i + (1 << 18)]) for i in range(0, n, 1 << 18)])
        assert np.max(np.abs(got - truth)) <= fir_bound("crcf", h, x), choice
        assert rel_l2(got, truth) <= 1e-6, choice


# ---------------------------------------------------------------------------------- firdecim
@pytest.mark.parametrize("kind", KINDS)
@pytest.mark.parametrize("case,M", [("m2h4x20", 2), ("m3h7x30", 3), ("m4h13x40", 4), ("m5h23x50", 5)])
def test_firdecim_golden(ya, oracle, kind, case, M):
    """firdecim.rs:295-473, epsilon 1e-3; firdecim_block (:246-279): block == per call.  The per-call form runs on the
    host mirror with the reference's sequential sums (bitwise the oracle's), the block form on the device (fused
    multiply-adds: equal to f32 rounding)"""
    g = load_golden("firdecim")
    h, x, y = (g[f"firdecim_{kind}_data_{case}_{s}"] for s in "hxy")
    q = ya.FirDecimationFilter(kind, M, h)
    got = q.execute_block(x, len(y))
    np.testing.assert_allclose(got, y, atol=1e-3, rtol=0)
    q2, ref = ya.FirDecimationFilter(kind, M, h), oracle.FirDecimationFilter(kind, M, h)
    per = np.array([q2.execute(x[i * M:(i + 1) * M]) for i in range(len(y))]).astype(got.dtype)
    want = np.array([ref.execute(x[i * M:(i + 1) * M]) for i in range(len(y))]).astype(got.dtype)
    assert per.tobytes() == want.tobytes()
    np.testing.assert_allclose(per, got, atol=2e-6, rtol=1e-5)
    assert q.get_decim_rate() == M


@pytest.mark.parametrize("kind", KINDS)
def test_firdecim_config(ya, kind):
    """firdecim.rs:216-244"""
    T, Cdt = ya.KINDS[kind]
    with pytest.raises(ya.ConfigError):
        ya.FirDecimationFilter(kind, 0, np.ones(4, Cdt))
    with pytest.raises(ya.ConfigError):
        ya.FirDecimationFilter(kind, 2, np.zeros(0, Cdt))
    with pytest.raises(ya.ConfigError):
        ya.FirDecimationFilter.new_kaiser(kind, 1, 3, 60.0)
    with pytest.raises(ya.ConfigError):
        ya.FirDecimationFilter.new_kaiser(kind, 4, 0, 60.0)
    with pytest.raises(ya.ConfigError):
        ya.FirDecimationFilter.new_kaiser(kind, 4, 3, -1.0)
    q = ya.FirDecimationFilter.new_kaiser(kind, 4, 3, 60.0)
    with pytest.raises(ya.ConfigError):            # the reference panics on x.len() < M (firdecim.rs:182)
        q.execute(np.zeros(3, T))
    with pytest.raises(ya.ConfigError):
        q.execute_block(np.zeros(7, T), 2)


@pytest.mark.parametrize("kind", KINDS)
@pytest.mark.parametrize("M,L,n", [(2, 9, 1000), (3, 64, 5001), (5, 101, 3000), (8, 129, 4100), (64, 1025, 700), (300, 40, 50)])
def test_firdecim_random_vs_f64(ya, oracle, kind, M, L, n):
    rng = np.random.default_rng(M * 1000 + L)
    h, x = rand_taps(rng, kind, L), rand_samples(rng, kind, n * M)
    q = ya.FirDecimationFilter(kind, M, h)
    q.set_scale(0.5)
    k = n // 2
    got = np.concatenate([q.execute_block(x[:k * M], k), q.execute_block(x[k * M:], n - k)])
    truth = oracle.fir_block_f64(kind, h, x, M=M, n=n, scale=0.5)
    assert np.max(np.abs(got - truth)) <= fir_bound(kind, h, x)
    # clone + reset
    c = q.clone()
    e = rand_samples(rng, kind, 10 * M)
    assert np.array_equal(q.execute_block(e, 10), c.execute_block(e, 10))
    q.reset()
    np.testing.assert_allclose(q.execute_block(x[:20 * M], 20), truth[:20], atol=fir_bound(kind, h, x))


@pytest.mark.parametrize("kind", KINDS)
@pytest.mark.parametrize("M,L", [(2, 2), (2, 9), (2, 65), (3, 64), (3, 100), (4, 65), (4, 129), (5, 7), (7, 50), (8, 129),
                                 (8, 257), (8, 2049), (12, 200), (12, 400), (16, 33), (16, 513), (32, 513), (64, 1025)])
def test_firdecim_register_window_kernel(ya, oracle, kind, M, L):
    """Blocks of >= 512 outputs with >= 8 taps per decimation phase take the per-phase register-window kernel
    (8- or 4-sample window, workgroups of 256 / 128 / 64 lanes by phase length and what fits the LDS;
    YAGI_HIP_DECIM_WINDOW_MIN_STEPS=0 sends every shape here through it), the rest the general kernel; the
    stream is cut at ragged places, a NaN placed in
    the input must poison exactly the outputs whose window holds it, and integers stay exact."""
    rng = np.random.default_rng(9000 + 37 * M + L)
    cuts = np.cumsum([0, 4096, 2048, 513, 700, 512, 2049, 100])
    n = int(cuts[-1])
    h, x = rand_taps(rng, kind, L), rand_samples(rng, kind, n * M)
    q = ya.FirDecimationFilter(kind, M, h)
    q.set_scale(0.5)
    got = np.concatenate([q.execute_block(x[a * M:b * M], b - a) for a, b in zip(cuts[:-1], cuts[1:])])
    truth = oracle.fir_block_f64(kind, h, x, M=M, n=n, scale=0.5)
    assert np.max(np.abs(got - truth)) <= fir_bound(kind, h, x)
    assert rel_l2(got, truth) <= 2e-6
    # NaN containment: output o sees samples M*o-(L-1) .. M*o
    xn = x.copy()
    s0 = 3000 * M + 1
    xn[s0] = np.nan
    q.reset()
    gn = q.execute_block(xn, n)
    bad = np.flatnonzero(np.isnan(gn.real) if np.iscomplexobj(gn) else np.isnan(gn))
    o = np.arange(n)
    want = np.flatnonzero((M * o >= s0) & (M * o - (L - 1) <= s0))
    assert np.array_equal(bad, want)
    # integers: exact whatever the summation order
    T, Cdt = ya.KINDS[kind]
    hi = rng.integers(-4, 5, L).astype(Cdt)
    xi = rng.integers(-8, 9, 2048 * M).astype(T)
    qi = ya.FirDecimationFilter(kind, M, hi)
    assert np.array_equal(qi.execute_block(xi, 2048), oracle.fir_block_f64(kind, hi, xi, M=M, n=2048).astype(T))


@pytest.mark.parametrize("kind", KINDS)
def test_firdecim_integer_inputs_bit_exact(ya, oracle, kind):
    rng = np.random.default_rng(17)
    for M, L, n in [(2, 5, 100), (4, 33, 2000), (7, 50, 999)]:
        h, x = int_taps(rng, kind, L), int_samples(rng, kind, n * M)
        want = oracle.FirDecimationFilter(kind, M, h).execute_block(x, n)
        assert np.array_equal(ya.FirDecimationFilter(kind, M, h).execute_block(x, n), want)


# ---------------------------------------------------------------------------------- firpfb
def test_firpfb_impulse_response(ya):
    """firpfb.rs:310-359 (1e-4)"""
    g = load_golden("firpfb")
    h, noise, test = (g[f"firpfb_impulse_response__{s}"] for s in ("h", "noise", "test"))
    f = ya.FirPfbFilter("rrrf", 4, h, 48)
    for v in noise:
        f.push(v)
    for i, expected in enumerate(test):
        assert abs(f.execute(i) - expected) <= 1e-4
    with pytest.raises(ya.ConfigError):            # firpfb.rs:278-280
        f.execute(4)
    f2 = ya.FirPfbFilter("rrrf", 4, h, 48)
    f2.write(noise)
    assert all(f2.execute(i) == f.execute(i) for i in range(4))


def test_firpfb_config(ya):
    with pytest.raises(ya.ConfigError):
        ya.FirPfbFilter("crcf", 0, np.ones(8, np.float32))
    with pytest.raises(ya.ConfigError):
        ya.FirPfbFilter("crcf", 4, np.zeros(0, np.float32))
    with pytest.raises(ya.ConfigError):            # h_len / M == 0 -> Window::new(0) errs (window.rs:14)
        ya.FirPfbFilter("crcf", 16, np.ones(8, np.float32))
    with pytest.raises(ya.ConfigError):
        ya.FirPfbFilter.new_kaiser("crcf", 4, 0, 0.5, 60.0)
    with pytest.raises(ya.ConfigError):
        ya.FirPfbFilter.new_kaiser("crcf", 4, 3, 0.7, 60.0)


@pytest.mark.parametrize("kind", KINDS)
def test_firpfb_copy(ya, kind):
    """firpfb_crcf_copy (firpfb.rs:361-396): irregular sizes, clone mid-stream, equal outputs"""
    rng = np.random.default_rng(13)
    m, hh = 13, 7
    q0 = ya.FirPfbFilter.default(kind, m, hh)
    q0.write(rand_samples(rng, kind, 80))
    q1 = q0.clone()
    for _ in range(40):
        v = rand_samples(rng, kind, 1)[0]
        idx = int(rng.integers(0, m))
        q0.push(v)
        q1.push(v)
        assert q0.execute(idx) == q1.execute(idx)


@pytest.mark.parametrize("kind", KINDS)
@pytest.mark.parametrize("nf,hlen,n", [(4, 48, 300), (13, 13 * 14 + 1, 500), (64, 1025, 2000), (256, 2049, 700), (3, 10, 50)])
def test_firpfb_block_all_select_vs_oracle(ya, oracle, kind, nf, hlen, n):
    rng = np.random.default_rng(nf * 31 + hlen)
    h, x = rand_taps(rng, kind, hlen), rand_samples(rng, kind, n)
    hs = hlen // nf
    tol = 4 * hs * 1.2e-7 * float(np.sum(np.abs(h))) * float(np.max(np.abs(x))) + 1e-7
    # execute_block(i, x) for a few branches, state carried over two calls
    for i in sorted(set([0, 1, nf // 2, nf - 1])):
        ref = oracle.FirPfbFilter(kind, nf, h, hlen)
        q = ya.FirPfbFilter(kind, nf, h, hlen)
        got = np.concatenate([q.execute_block(i, x[: n // 2]), q.execute_block(i, x[n // 2:])])
        assert np.max(np.abs(got - ref.execute_block(i, x))) <= tol
    # all branches per pushed sample (interpolator form)
    q = ya.FirPfbFilter(kind, nf, h, hlen)
    q.set_scale(2.0)
    ya_all = np.concatenate([q.execute_all(x[:7]), q.execute_all(x[7:])])
    ref = oracle.FirPfbFilter(kind, nf, h, hlen)
    ref.set_scale(2.0)
    idx = rng.integers(0, nf, n).astype(np.uint32)
    want_sel = np.empty(n, x.dtype)
    for k in range(n):
        ref.push(x[k])
        if k % max(1, n // 16) == 0:
            want = np.array([ref.execute(i) for i in range(nf)])
            assert np.max(np.abs(ya_all[k] - want)) <= 2 * tol
        want_sel[k] = ref.execute(int(idx[k]))
    # branch index per sample
    q = ya.FirPfbFilter(kind, nf, h, hlen)
    q.set_scale(2.0)
    got_sel = np.concatenate([q.execute_select(idx[:11], x[:11]), q.execute_select(idx[11:], x[11:])])
    assert np.max(np.abs(got_sel - want_sel)) <= 2 * tol
    with pytest.raises(ya.ConfigError):
        q.execute_select(np.full(4, nf, np.uint32), x[:4])
    with pytest.raises(ya.ConfigError):
        q.execute_block(nf, x[:4])


@pytest.mark.parametrize("kind", KINDS)
def test_pure_delay_whole_buffer_exact(ya, kind):
    """size-independent property, bit-exact: h = delta[k - d] makes the direct-form kernels copy the input delayed by d
    (firfilt: every kernel choice; firdecim: every M-th sample), checked on every output of 2^22-sample device buffers;
    the overlap-save kernel reproduces it to f32 rounding"""
    L, d, n = 129, 77, 1 << 22
    h = np.zeros(L, np.complex64 if kind == "cccf" else np.float32)
    h[d] = 1.0
    dx = ya.gen_real_dev(SEED + 9, n) if kind == "rrrf" else ya.gen_complex_dev(SEED + 9, n)
    x = dx.to_numpy()
    delayed = np.concatenate([np.zeros(d, x.dtype), x[: n - d]])
    for choice in ((0, 1, 2, 3, 4) if kind == "crcf" else (0, 1, 4)):
        q = ya.FirFilter(kind, h)
        q.set_kernel(choice)
        dy = ya.DeviceArray(n, x.dtype)
        q.execute_block_dev(dx, n, dy)
        ya.synchronize()
        y = dy.to_numpy()
        if choice == 4:
            assert rel_l2(y, delayed) <= 1e-6
        else:
            assert np.array_equal(y, delayed), choice
        dy.free()
    for M in (2, 4, 5):
        q = ya.FirDecimationFilter(kind, M, h)
        dy = ya.DeviceArray(n // M, x.dtype)
        q.execute_block_dev(dx, n // M, dy)
        ya.synchronize()
        # firdecim.rs:179-205: the M samples of a frame are pushed, the output is taken after the FIRST one
        want = np.concatenate([np.zeros(d, x.dtype), x])[np.arange(n // M) * M]
        assert np.array_equal(dy.to_numpy(), want), M
        dy.free()


def test_firfilt_notch_and_dc_blocker(ya, oracle):
    """new_notch / new_dc_blocker (firfilt.rs:166-186, design/mod.rs:336-378): coefficients equal the restated design;
    the reference's acceptance tests: crcf / cccf spectral masks (firfilt.rs:406-433) and the cccf tone-rejection
    harness (:590-624, tolerance 1e-3) for its six notch frequencies; config errors (:444-446)"""
    for kind in ("crcf", "cccf", "rrrf"):
        q = ya.FirFilter.new_notch(kind, 20, 60.0, 0.125)
        np.testing.assert_allclose(q.get_coefficients(), oracle.notch_taps(kind, 20, 60.0, 0.125), rtol=2e-5, atol=2e-6)
        d = ya.FirFilter.new_dc_blocker(kind, 7, 40.0)
        np.testing.assert_allclose(d.get_coefficients(), oracle.fir_design_notch(7, 0.0, 40.0).astype(q.Cdt), rtol=2e-5, atol=2e-6)
        bads = [lambda: ya.FirFilter.new_dc_blocker(kind, 0, 0.0), lambda: ya.FirFilter.new_notch(kind, 0, 0.0, 0.0)]
        if kind != "cccf":       # complex taps mix the DC blocker by any f0 (firfilt.rs:34-43): no range check there
            bads.append(lambda: ya.FirFilter.new_notch(kind, 20, 60.0, 0.7))
        for bad in bads:
            with pytest.raises(ya.ConfigError):
                bad()
    # spectral masks on the frequency response of the taps (1200-point transform like validate_psd_firfilt)
    def response_db(h, nfft=1200):
        H = np.fft.fftshift(np.fft.fft(np.asarray(h, np.complex128), nfft))
        return np.arange(nfft) / nfft - 0.5, 20 * np.log10(np.abs(H) + 1e-300)
    f, p = response_db(ya.FirFilter.new_notch("crcf", 20, 60.0, 0.125).get_coefficients())
    for lo, hi in ((-0.5, -0.20), (-0.06, 0.06), (0.20, 0.5)):
        assert np.all(np.abs(p[(f >= lo) & (f <= hi)]) <= 0.1)
    for lo, hi in ((-0.126, -0.124), (0.124, 0.126)):
        assert np.all(p[(f >= lo) & (f <= hi)] <= -50.0)
    f, p = response_db(ya.FirFilter.new_notch("cccf", 20, 60.0, 0.125).get_coefficients())
    assert np.all(np.abs(p[(f <= 0.06)]) <= 0.1) and np.all(np.abs(p[f >= 0.20]) <= 0.1)
    assert np.all(p[(f >= 0.124) & (f <= 0.126)] <= -50.0)
    # tone at the notch frequency is removed (through the device filter)
    for f0 in (0.0, 0.1, 0.456, 0.5, -0.25, -0.389):
        m, n = 20, 600
        q = ya.FirFilter.new_notch("cccf", m, 60.0, f0)
        x = np.exp(2j * np.pi * np.float32(f0) * np.arange(n + 2 * m + 1)).astype(np.complex64)
        y = q.execute_block(x)[2 * m + 1:]
        assert abs(np.sqrt(np.mean(np.abs(x[2 * m + 1:]) ** 2)) - 1.0) <= 1e-3
        assert np.sqrt(np.mean(np.abs(y) ** 2)) <= 1e-3, f0


@pytest.mark.parametrize("kind", KINDS)
def test_freqresponse_and_groupdelay(ya, kind):
    """freqresponse (firfilt.rs:325-328, design/mod.rs:666-675), groupdelay (:339-342, design/mod.rs:687-704) and
    FirDecimationFilter::freqresp (firdecim.rs:164-168) against their definitions in f64"""
    rng = np.random.default_rng(12)
    h = rand_taps(rng, kind, 33)
    scale = (0.5 - 0.25j) if kind == "cccf" else 0.5
    q = ya.FirFilter(kind, h)
    q.set_scale(scale)
    d = ya.FirDecimationFilter(kind, 3, h)
    d.set_scale(scale)
    i = np.arange(33)
    for fc in (0.0, 0.1, -0.31, 0.5):
        want = np.sum(h.astype(np.complex128) * np.exp(-2j * np.pi * fc * i)) * scale
        assert abs(q.freqresponse(fc) - want) <= 1e-5 * (1 + abs(want))
        assert abs(d.freqresp(fc) - want) <= 1e-5 * (1 + abs(want))
        e = np.exp(2j * np.pi * fc * i)
        gd = (np.sum(h.real * e * i) / np.sum(h.real * e)).real
        assert abs(q.groupdelay(fc) - gd) <= 1e-3 * (1 + abs(gd))
    sym = ya.FirFilter.new_kaiser(kind, 41, 0.2, 60.0, 0.0)          # linear phase: delay (n-1)/2 everywhere in band
    assert abs(sym.groupdelay(0.05) - 20.0) <= 1e-3
    with pytest.raises(ya.ConfigError):
        q.groupdelay(0.6)


@pytest.mark.parametrize("kind", KINDS)
def test_randomised_shapes_vs_f64(ya, oracle, kind):
    """seeded sweep over filter length, block length, split points, scale and kernel choice (lengths 1..700, blocks
    1..9000): every kernel against the f64 truth, with the stream cut at random places so that state is carried"""
    rng = np.random.default_rng(2024)
    for case in range(40):
        L = int(rng.choice([1, 2, 3, 5, 16, 17, 31, 32, 33, 63, 64, 65, 127, 128, 129, 255, 256, 257, 300, 511, 700]))
        n = int(rng.integers(1, 9000))
        h, x = rand_taps(rng, kind, L), rand_samples(rng, kind, n)
        scale = (rng.standard_normal() + 1j * rng.standard_normal()) if kind == "cccf" else float(rng.standard_normal())
        truth = oracle.fir_block_f64(kind, h, x, scale=scale)
        choices = [0, 1, 4] + ([2, 3] if kind == "crcf" and L <= 256 else [])
        cuts = sorted(set(int(c) for c in rng.integers(0, n + 1, 3)) | {0, n})
        for choice in choices:
            q = ya.FirFilter(kind, h)
            q.set_scale(scale)
            q.set_kernel(choice)
            got = np.concatenate([q.execute_block(x[a:b]) for a, b in zip(cuts[:-1], cuts[1:]) if b > a])
            bound = fir_bound(kind, h, x) * max(1.0, abs(scale)) + 1e-30
            assert np.max(np.abs(got - truth)) <= bound, (case, L, n, choice)
            assert rel_l2(got, truth) <= 3e-6, (case, L, n, choice)
        M = int(rng.integers(2, 7))
        nd = n // M
        if nd:
            d = ya.FirDecimationFilter(kind, M, h)
            d.set_scale(scale)
            k = int(rng.integers(0, nd + 1))
            got = np.concatenate([d.execute_block(x[: k * M], k), d.execute_block(x[k * M: nd * M], nd - k)])
            assert rel_l2(got, oracle.fir_block_f64(kind, h, x[: nd * M], M=M, scale=scale)) <= 3e-6, (case, L, n, M)


@pytest.mark.parametrize("kind", KINDS)
def test_device_pointers_of_any_alignment(ya, oracle, kind):
    """execute_block_dev on input / output pointers that are NOT on 16-byte boundaries (offset by one element): the
    kernels' vector-load staging falls back to element loads; results must equal the aligned run bit for bit
    (filter, long-phase decimator, interpolator)."""
    T, Cdt = ya.KINDS[kind]
    rng = np.random.default_rng(77)
    n = 3 * 2048 + 777
    x = rand_samples(rng, kind, 4 * n + 8)
    dx = ya.DeviceArray.from_numpy(x)
    dy = ya.DeviceArray(4 * n + 8, T)
    isz = x.itemsize

    h = rand_taps(rng, kind, 63)
    ref = ya.FirFilter(kind, h).execute_block(x[1:1 + n])
    q = ya.FirFilter(kind, h)
    q.execute_block_dev(dx.ptr + isz, n, dy.ptr + isz)
    ya.synchronize()
    assert np.array_equal(dy.to_numpy(n, offset=1), ref)
    hd = rand_taps(rng, kind, 129)
    ref = ya.FirDecimationFilter(kind, 4, hd).execute_block(x[1:1 + 4 * n], n)
    d = ya.FirDecimationFilter(kind, 4, hd)
    d.execute_block_dev(dx.ptr + isz, n, dy.ptr + isz)
    ya.synchronize()
    assert np.array_equal(dy.to_numpy(n, offset=1), ref)


# ---------------------------------------------------------------------------------- per-sample calls on the host mirror
@pytest.mark.parametrize("kind", KINDS)
def test_per_sample_calls_interleaved_with_blocks(ya, oracle, kind):
    """push() / execute() / execute_one() are served from a host mirror of the window with the
    reference's own sequential sums (firfilt.rs:220-261), synchronised lazily with the device window when a block call
    follows (and back after one).  Integer data: every output of an arbitrary interleaving equals the oracle's bit for bit.
    Random data: the per-sample outputs are BITWISE the oracle's (same products, same order, same two-slice split of the
    VecDeque, whose ring position the block calls advance too); the block outputs within the f32 bound."""
    rng = np.random.default_rng(314)
    for L in (5, 64, 256):
        for integer in (True, False):
            h = int_taps(rng, kind, L) if integer else rand_taps(rng, kind, L)
            x = int_samples(rng, kind, 3000) if integer else rand_samples(rng, kind, 3000)
            ref, q = oracle.FirFilter(kind, h), ya.FirFilter(kind, h)
            if not integer:
                ref.set_scale(0.4 if kind != "cccf" else 0.4 - 0.25j)
                q.set_scale(0.4 if kind != "cccf" else 0.4 - 0.25j)
            want = ref.execute_block(x)
            got = np.empty_like(x)
            per_sample = np.zeros(x.size, bool)
            i = 0
            plan = [("one", 3), ("block", 700), ("push", 11), ("block", 1), ("one", L + 3), ("block", 513), ("write", 40),
                    ("one", 1), ("block", 900)]
            for what, n in plan:
                n = min(n, x.size - i)
                if what == "block":
                    got[i:i + n] = q.execute_block(x[i:i + n])
                elif what == "one":
                    for k in range(i, i + n):
                        got[k] = q.execute_one(x[k])
                    per_sample[i:i + n] = True
                elif what == "push":
                    for k in range(i, i + n):
                        q.push(x[k])
                        got[k] = q.execute()
                    per_sample[i:i + n] = True
                else:                                     # write(): pushes only; the output after the last one
                    q.write(x[i:i + n])
                    got[i:i + n - 1] = want[i:i + n - 1]
                    got[i + n - 1] = q.execute()
                    per_sample[i + n - 1] = True
                i += n
            rest = slice(i, x.size)
            got[rest] = q.execute_block(x[rest])
            if integer:
                assert np.array_equal(got, want), (kind, L)
            else:
                assert np.array_equal(got[per_sample].view(np.uint32), want[per_sample].view(np.uint32)), (kind, L)
                truth = oracle.fir_block_f64(kind, h, x, scale=ref_scale(kind))
                assert np.max(np.abs(got - truth)) <= fir_bound(kind, h, x)


def ref_scale(kind):
    return 0.4 if kind != "cccf" else 0.4 - 0.25j


@pytest.mark.parametrize("kind", KINDS)
def test_per_sample_firpfb_and_firdecim_on_the_host_mirror(ya, oracle, kind):
    """FirPfbFilter::push / execute(i) (firpfb.rs:255-286) and FirDecimationFilter::execute (firdecim.rs:179-191) per call:
    bitwise the oracle's sequential sums, interleaved with block calls that move the window on the device"""
    rng = np.random.default_rng(2718)
    nf, hs = 6, 9
    h = rand_taps(rng, kind, nf * hs + 2)                 # h_len % nf != 0: the floor of firpfb.rs:42
    x = rand_samples(rng, kind, 400)
    ref, q = oracle.FirPfbFilter(kind, nf, h), ya.FirPfbFilter(kind, nf, h)
    for k in range(60):
        ref.push(x[k]); q.push(x[k])
        a, b = ref.execute(k % nf), q.execute(k % nf)
        assert np.array([a]).astype(x.dtype).tobytes() == np.array([b]).astype(x.dtype).tobytes(), k
    blk_ref, blk = ref.execute_block(2, x[60:300]), q.execute_block(2, x[60:300])      # device block: f32 bound
    assert np.max(np.abs(blk - blk_ref)) <= 1e-5
    for k in range(300, 330):
        ref.push(x[k]); q.push(x[k])
        a, b = ref.execute(5), q.execute(5)
        assert np.array([a]).astype(x.dtype).tobytes() == np.array([b]).astype(x.dtype).tobytes(), k
    M, L = 4, 23
    hd = rand_taps(rng, kind, L)
    rd, qd = oracle.FirDecimationFilter(kind, M, hd), ya.FirDecimationFilter(kind, M, hd)
    for k in range(0, 80, M):
        a, b = rd.execute(x[k:k + M]), qd.execute(x[k:k + M])
        assert np.array([a]).astype(x.dtype).tobytes() == np.array([b]).astype(x.dtype).tobytes(), k
    yb_ref, yb = rd.execute_block(x[80:280], 50), qd.execute_block(x[80:280], 50)
    assert np.max(np.abs(yb - yb_ref)) <= 1e-5
    a, b = rd.execute(x[280:284]), qd.execute(x[280:284])
    assert np.array([a]).astype(x.dtype).tobytes() == np.array([b]).astype(x.dtype).tobytes()
    with pytest.raises(ya.ConfigError):
        qd.execute(x[:M - 1])                            # firdecim.rs:182 indexes x[i] for i < M


@pytest.mark.parametrize("kind", KINDS)
def test_firfilt_pipelined_blocks_bit_identical(ya, kind):
    """set_pipeline(1) on FirFilter: consecutive execute_block_dev calls run on two streams of the object, each reading its
    window from the tail of the previous call's input; after join() every output is bit for bit the unpipelined one --
    every kernel choice, ragged blocks (shorter than the filter: those calls join and run unpipelined), per-sample and
    host-pointer calls and a reset in between"""
    rng = np.random.default_rng(77)
    T = ya.KINDS[kind][0]
    for L, choices in ((31, (0, 1, 4)), (256, (0, 1, 2, 3, 4) if kind == "crcf" else (0, 1, 4))):
        h = rand_taps(rng, kind, L)
        sizes = [1 << 16, 5000, 17, 1 << 17, 300, 1 << 16, 1 << 21, 4096]
        x = rand_samples(rng, kind, sum(sizes))
        dx = ya.DeviceArray.from_numpy(x)
        for choice in choices:
            plain, piped = ya.FirFilter(kind, h), ya.FirFilter(kind, h)
            for q in (plain, piped):
                q.set_scale(0.4)
                q.set_kernel(choice)
            piped.set_pipeline(True)
            dy0, dy1 = ya.DeviceArray(x.size, T), ya.DeviceArray(x.size, T)
            isz = x.dtype.itemsize
            for rep in range(2):
                for q, dy in ((plain, dy0), (piped, dy1)):
                    o = 0
                    for n in sizes:
                        q.execute_block_dev(dx.ptr + isz * o, n, dy.ptr + isz * o)
                        o += n
                piped.join()
                a, b = dy0.to_numpy(), dy1.to_numpy()
                assert a.tobytes() == b.tobytes(), (kind, L, choice, rep)
                if rep == 0:
                    # per-sample call (host mirror: joins and fetches the window), host-pointer block, then reset
                    assert plain.execute_one(1.0) == piped.execute_one(1.0)
                    assert plain.execute_block(x[:1000]).tobytes() == piped.execute_block(x[:1000]).tobytes()
                    plain.reset(); piped.reset()
