"""Rresamp through the C ABI -- mirrors src/filter/resampler/rresamp.rs:185-385.

The reference holds no data vectors for the resampler: its tests are the partition test (:198-238, one block ==
two resamplers with the second primed by write()), and spectral masks driven by a QPSK symbol generator from the
framing/modem modules (out of scope here; the masks are checked with tones instead).  Sample-level parity is
against the oracle's restatement of execute_primitive on top of its golden-pinned FirPfbFilter."""
import numpy as np
import pytest

from gpu_util import int_samples, int_taps, rand_samples, rand_taps, rel_l2

pytestmark = pytest.mark.gpu
KINDS = ["rrrf", "crcf", "cccf"]


@pytest.fixture(scope="module")
def ya():
    import yagi_amd
    assert yagi_amd.device_count() > 0
    return yagi_amd


def test_config_and_getters(ya):
    for bad in [lambda: ya.Rresamp("crcf", 0, 5, 4, np.ones(64, np.float32)),
                lambda: ya.Rresamp("crcf", 3, 0, 4, np.ones(64, np.float32)),
                lambda: ya.Rresamp("crcf", 3, 5, 0, np.ones(64, np.float32)),
                lambda: ya.Rresamp("crcf", 3, 5, 4, np.ones(23, np.float32)),        # needs 2*3*4 taps
                lambda: ya.Rresamp.new_kaiser("crcf", 3, 5, 4, 0.6, 60.0)]:
        with pytest.raises(ya.ConfigError):
            bad()
    q = ya.Rresamp.new_kaiser("crcf", 6, 10, 15, -1.0, 60.0)       # gcd 2 -> 3/5, block_len 2 (:59-82)
    assert (q.get_interp(), q.get_decim(), q.get_delay(), q.get_block_len()) == (3, 5, 15, 2)
    assert (q.get_p(), q.get_q()) == (6, 10) and abs(q.get_rate() - 0.6) < 1e-7
    bw = np.float32(0.5) * np.float32(3) / np.float32(5)
    assert abs(q.get_scale() - 2 * bw * np.sqrt(np.float32(5) / np.float32(3))) < 1e-6
    d = ya.Rresamp.new_default("rrrf", 3, 2)                        # :99-104
    assert d.get_delay() == 12 and d.get_block_len() == 1 and abs(d.get_scale() - 2 * 0.5 * np.sqrt(2 / 3)) < 1e-6
    with pytest.raises(ya.RangeError):
        q.execute(np.zeros(9, np.complex64))                        # needs Q*block_len = 10 samples


@pytest.mark.parametrize("kind", KINDS)
@pytest.mark.parametrize("P,Q,m", [(1, 5, 3), (2, 5, 7), (3, 5, 15), (6, 5, 15), (8, 5, 4), (9, 5, 15), (7, 1, 2), (1, 1, 5),
                                   (160, 147, 6)])
def test_vs_oracle(ya, oracle, kind, P, Q, m):
    """execute / execute_block against the restated execute_primitive schedule; integer data bit-exact
    (branch order, branch <-> sample alignment), random data to f32 rounding"""
    rng = np.random.default_rng(100 * P + Q)
    nblk = 37 if P < 100 else 5
    h = int_taps(rng, kind, 2 * P * m + 3)                          # only the first 2*P*m are used (:40)
    x = int_samples(rng, kind, nblk * Q)
    ref = oracle.Rresamp(kind, P, Q, m, h)
    want = ref.execute_block(x, nblk)
    q = ya.Rresamp(kind, P, Q, m, h)
    k = nblk // 3
    got = np.concatenate([q.execute_block(x[:k * Q], k), q.execute(x[k * Q:(k + 1) * Q]),
                          q.execute_block(x[(k + 1) * Q:], nblk - k - 1)])      # state carried
    assert got.shape == (nblk * P,) and np.array_equal(got, want)
    h, x = rand_taps(rng, kind, 2 * P * m), rand_samples(rng, kind, nblk * Q)
    ref = oracle.Rresamp(kind, P, Q, m, h)
    scale = (0.3 + 0.2j) if kind == "cccf" else 0.7
    ref.set_scale(scale)
    q = ya.Rresamp(kind, P, Q, m, h)
    q.set_scale(scale)
    assert rel_l2(q.execute_block(x, nblk), ref.execute_block(x, nblk)) <= 1e-6
    q.reset()
    ref.reset()
    assert rel_l2(q.execute_block(x[:Q], 1), ref.execute_block(x[:Q], 1)) <= 1e-6


@pytest.mark.parametrize("kind", KINDS)
@pytest.mark.parametrize("P,Q,m", [(3, 5, 15), (5, 3, 6), (3, 2, 8), (2, 3, 4), (7, 4, 5), (1, 8, 3), (9, 1, 2), (16, 15, 4)])
def test_many_blocks_cut_both_ways(ya, oracle, kind, P, Q, m):
    """several LDS tiles per call with a ragged last tile: the same stream cut into long and short calls must
    give the same bits; integer data must match the restated schedule exactly"""
    rng = np.random.default_rng(31 * P + Q)
    nblk = 256 * 3 + 77
    h, x = rand_taps(rng, kind, 2 * P * m), rand_samples(rng, kind, nblk * Q)
    q = ya.Rresamp(kind, P, Q, m, h)
    q.set_scale(0.5)
    big = np.concatenate([q.execute_block(x[:300 * Q], 300), q.execute_block(x[300 * Q:], nblk - 300)])
    q.reset()
    small = np.concatenate([q.execute_block(x[b * Q:min(b + 100, nblk) * Q], min(100, nblk - b))
                            for b in range(0, nblk, 100)])
    assert np.array_equal(big, small)
    hi, xi = int_taps(rng, kind, 2 * P * m), int_samples(rng, kind, nblk * Q)
    ref = oracle.Rresamp(kind, P, Q, m, hi)
    qi = ya.Rresamp(kind, P, Q, m, hi)
    assert np.array_equal(qi.execute_block(xi, nblk), ref.execute_block(xi, nblk))


@pytest.mark.parametrize("P", [1, 2, 3, 6, 8, 9])
def test_partition(ya, P, Q=5, m=15, n=20):
    """rresamp.rs:198-238 (autotest_rresamp_crcf_part_P*_Q5): one 2n-block run == n blocks, then a second
    resampler primed with the last m*Q inputs by write() for the other n blocks"""
    q0 = ya.Rresamp.new_kaiser("crcf", P, Q, m, 0.5, 60.0)
    q1 = ya.Rresamp.new_kaiser("crcf", P, Q, m, 0.5, 60.0)
    N = 2 * Q * n
    i = np.arange(N)
    ham = (0.53836 - 0.46164 * np.cos(2 * np.pi * i / (N - 1))).astype(np.float32)
    x = (ham * np.exp(2j * np.pi * 0.037 * i)).astype(np.complex64)
    y0 = q0.execute_block(x, 2 * n)
    q0.reset()
    ya_ = q0.execute_block(x[:Q * n], n)
    for k in range(m):
        q1.write(x[Q * n - (m - k) * Q: Q * n - (m - k - 1) * Q])
    yb = q1.execute_block(x[Q * n:], n)
    np.testing.assert_allclose(np.concatenate([ya_, yb]), y0, atol=1e-12, rtol=0)


@pytest.mark.parametrize("P,Q", [(1, 5), (2, 5), (3, 5), (6, 5), (8, 5), (9, 5), (3, 2)])
def test_kaiser_response_and_block_len(ya, oracle, P, Q):
    """new_kaiser / new_default (rresamp.rs:240-290 masks, checked with tones: pass band 0 dB +-0.5 dB at the
    output rate, image/alias band below -60+0.5 dB) and block_len > 1 == the reduced-rate resampler"""
    q = ya.Rresamp.new_kaiser("crcf", P, Q, 15, -1.0, 60.0)          # the "baseline" objects of :294-315
    p, qq = q.get_interp(), q.get_decim()
    r = p / qq
    nblk = 4000
    n_in = nblk * qq
    t = np.arange(n_in)
    def gain(f_out):
        """tone at output-rate frequency f_out (|f| < 0.5): input frequency f_out * r"""
        q.reset()
        x = np.exp(2j * np.pi * f_out * r * t).astype(np.complex64)
        y = q.execute_block(x, nblk // q.get_block_len())[200 * p:]
        ref = np.exp(2j * np.pi * f_out * np.arange(200 * p, nblk * p))
        return 20 * np.log10(abs(np.vdot(ref, y)) / len(y) + 1e-30), 10 * np.log10(np.mean(np.abs(y) ** 2) + 1e-30)
    # new_kaiser's scale 2 bw sqrt(Q/P) (:78) keeps the power spectral DENSITY of a band-limited signal at 0 dB
    # (what the reference's masks measure); a tone's amplitude therefore changes by sqrt(Q/P)
    g0 = 10 * np.log10(qq / p)
    for f in (0.0, 0.03, -0.07):                       # inside 0.4 * bw of the reference's mask
        g, pw = gain(f)
        assert abs(g - g0) <= 0.5 and abs(pw - g0) <= 0.5, (f, g, pw, g0)
    if r < 1:                                          # decimating (bw = 0.5 P/Q): an input tone beyond the output Nyquist is rejected
        q.reset()
        x = np.exp(2j * np.pi * (0.5 * r + 0.6 * (0.5 - 0.5 * r)) * t).astype(np.complex64)
        y = q.execute_block(x, nblk // q.get_block_len())[200 * p:]
        assert 10 * np.log10(np.mean(np.abs(y) ** 2) + 1e-30) <= g0 - 60 + 0.5
    # gcd > 1: 2P/2Q reduces to P/Q with block_len 2 and gives the same samples
    q2 = ya.Rresamp.new_kaiser("crcf", 2 * P, 2 * Q, 15, -1.0, 60.0)
    assert q2.get_block_len() == 2 * q.get_block_len() and q2.get_interp() == p
    rng = np.random.default_rng(5)
    x = rand_samples(rng, "crcf", 64 * qq * q2.get_block_len())
    q.reset()
    assert np.array_equal(q2.execute_block(x, 64), q.execute_block(x, 64 * 2))
    ref = oracle.Rresamp.new_kaiser("crcf", 2 * P, 2 * Q, 15, -1.0, 60.0)
    q2.reset()
    assert rel_l2(q2.execute_block(x[: 8 * q2.get_q()], 8), ref.execute_block(x[: 8 * q2.get_q()], 8)) <= 1e-6


def test_long_block_device(ya, oracle):
    """2^22 inputs at 160/147 on device buffers; sampled blocks vs the f64 evaluation of the same schedule"""
    P, Q, m = 160, 147, 8
    q = ya.Rresamp.new_kaiser("crcf", P, Q, m, 0.45, 60.0)
    nblk = (1 << 22) // Q
    dx = ya.gen_complex_dev(11, nblk * Q)
    dy = ya.DeviceArray(nblk * P, np.complex64)
    q.execute_block_dev(dx, nblk, dy)
    ya.synchronize()
    hf = oracle.fir_design_kaiser(2 * P * m + 1, 0.45 / P, 60.0, 0.0).astype(np.float64)
    scale = float(q.get_scale())
    Ls = 2 * m
    for blk in (0, 1, 777, nblk - 1):
        lo = max(0, blk * Q - Ls)
        xs = np.concatenate([np.zeros(Ls - (blk * Q - lo), np.complex128), dx.to_numpy(blk * Q - lo + Q, offset=lo).astype(np.complex128)])
        want = np.empty(P, np.complex128)
        for n in range(P):
            i, br = divmod(n * Q, P)
            want[n] = scale * sum(hf[br + k * P] * xs[Ls + i - k] for k in range(Ls))
        got = dy.to_numpy(P, offset=blk * P)
        assert rel_l2(got, want) <= 2e-6, blk
