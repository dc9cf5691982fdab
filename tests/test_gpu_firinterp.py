"""FirInterpolationFilter through the C ABI -- mirrors src/filter/fir/firinterp.rs:262-500."""
import numpy as np
import pytest

from conftest import load_golden
from gpu_util import rand_samples, rand_taps

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ya():
    import yagi_amd
    assert yagi_amd.device_count() > 0
    return yagi_amd


def test_common_and_config(ya):
    assert ya.FirInterpolationFilter.new_kaiser("rrrf", 17, 4, 60.0).get_interp_rate() == 17   # :265-269
    q = ya.FirInterpolationFilter.new_kaiser("crcf", 7, 4, 60.0)                                 # :272-276
    assert q.get_interp_rate() == 7 and q.get_sub_len() == 8
    for bad in [lambda: ya.FirInterpolationFilter("rrrf", 1, np.ones(8, np.float32)),
                lambda: ya.FirInterpolationFilter("rrrf", 9, np.ones(8, np.float32)),
                lambda: ya.FirInterpolationFilter.new_kaiser("rrrf", 1, 4, 60.0),
                lambda: ya.FirInterpolationFilter.new_kaiser("rrrf", 4, 0, 60.0),
                lambda: ya.FirInterpolationFilter.new_kaiser("rrrf", 4, 4, -1.0),
                lambda: ya.FirInterpolationFilter.new_window("rrrf", 4, 0)]:
        with pytest.raises(ya.ConfigError):
            bad()
    lin = ya.FirInterpolationFilter.new_linear("rrrf", 4)          # :135-147
    np.testing.assert_allclose(lin.execute_block(np.float32([1, 1, 1])),
                               [0, .25, .5, .75, 1, 1, 1, 1, 1, 1, 1, 1], atol=1e-6)
    win = ya.FirInterpolationFilter.new_window("crcf", 3, 2)       # :159-174
    assert win.get_sub_len() == 4


@pytest.mark.parametrize("kind", ["rrrf", "crcf"])
def test_generic_golden(ya, kind):
    """firinterp.rs:277-387, tol 1e-6"""
    g = load_golden("firinterp")
    h, x, test = (g[f"firinterp_{kind}_generic__{s}"] for s in ("h", "x", "test"))
    q = ya.FirInterpolationFilter(kind, 4, h)
    y = np.concatenate([q.execute(v) for v in x])
    np.testing.assert_allclose(y, test, atol=1e-6, rtol=0)
    q.reset()
    np.testing.assert_allclose(q.execute_block(x), test, atol=1e-6, rtol=0)


def test_copy_and_flush(ya):
    """firinterp.rs:443-500"""
    rng = np.random.default_rng(2)
    q0 = ya.FirInterpolationFilter.new_kaiser("crcf", 3, 7, 60.0)
    q0.set_scale(0.12345)
    for _ in range(20):
        q0.execute(rand_samples(rng, "crcf", 1)[0])
    q1 = q0.clone()
    assert abs(q1.get_scale() - 0.12345) < 1e-7
    for _ in range(30):
        v = rand_samples(rng, "crcf", 1)[0]
        assert np.array_equal(q0.execute(v), q1.execute(v))
    buf = q0.execute(1 + 1j)
    assert np.sum(np.abs(buf) ** 2) > 0
    for _ in range(2 * 7):
        buf = q0.flush()
    assert np.sum(np.abs(buf) ** 2) == 0


@pytest.mark.parametrize("kind", ["rrrf", "crcf", "cccf"])
@pytest.mark.parametrize("interp,hl,n", [(2, 9, 300), (5, 53, 1000), (16, 16 * 12, 4096), (3, 3, 10)])
def test_block_vs_oracle(ya, oracle, kind, interp, hl, n):
    rng = np.random.default_rng(interp * 100 + hl)
    h, x = rand_taps(rng, kind, hl), rand_samples(rng, kind, n)
    q = ya.FirInterpolationFilter(kind, interp, h)
    q.set_scale(2.0)
    got = np.concatenate([q.execute_block(x[: n // 3]), q.execute_block(x[n // 3:])])
    ref = oracle.FirInterpolationFilter(kind, interp, h)
    ref.set_scale(2.0)
    m = min(n, 200)
    want = ref.execute_block(x[:m])
    tol = 8 * ((hl + interp - 1) // interp) * 1.2e-7 * float(np.sum(np.abs(h))) * float(np.max(np.abs(x))) + 1e-7
    assert np.max(np.abs(got[: m * interp] - want)) <= tol
    # == zero-stuffing + firfilt (f64 truth) over the whole block
    z = np.zeros(n * interp, x.dtype)
    z[::interp] = x
    truth = oracle.fir_block_f64(kind, h, z, scale=2.0)
    assert np.max(np.abs(got - truth)) <= tol
