"""CPU: the oracle's restatement of Resamp2 / MsResamp2 (oracle/yagi_oracle.c, following
src/filter/resampler/resamp2.rs:26-180 and msresamp2.rs:8-198) against the reference's own tests for them, which
are property tests (the reference holds no sample vectors for these objects):
  resamp2.rs:189-229  autotest_resamp2_analysis       two tones split into the low / high band, 1e-3
  resamp2.rs:231-268  autotest_resamp2_synthesis      two basebands merged, 3e-3
  resamp2.rs:270-336  autotest_resamp2_crcf_filter_0..5  impulse responses of filter_execute against spectral masks
  resamp2.rs:338-369  autotest_resamp2_config, :371-399 autotest_resamp2_copy
  msresamp2.rs:206-298 autotest_msresamp2_crcf_interp_01..11, :300-337 autotest_msresamp2_copy
The half-band prototypes are Kaiser-windowed (the reference's Parks-McClellan design is out of scope: "parity
unpinned" for the tap VALUES; everything downstream of the taps is what these tests pin).  Where the equiripple
design is what makes a mask pass, the margin the Kaiser prototype needs is written in the test."""
import numpy as np
import pytest

from psd_util import estimate_req_filter_transition_bandwidth, validate_psd_signal


def two_tone_analysis(make, m=5, n=37, f0=0.0739, f1=-0.1387):
    """resamp2.rs:189-229; make(m) -> object with execute_block(mode, x)"""
    i = np.arange(2 * n + 2 * m + 1)
    x = np.where(i < 2 * n, np.exp(1j * f0 * i) + np.exp(1j * (np.pi + f1) * i), 0).astype(np.complex64)
    q = make(m)
    y = q.execute_block("analyzer", x[: 2 * n]).reshape(n, 2)
    k = np.arange(m, n - m)
    e0 = np.abs(y[k + m, 0] - np.exp(2j * f0 * (k + 0.5)))
    e1 = np.abs(y[k + m, 1] - np.exp(2j * f1 * (k + 0.5)))
    return float(max(e0.max(), e1.max()))


def two_band_synthesis(make, m=5, n=37, f0=0.0739, f1=-0.1387):
    """resamp2.rs:231-268"""
    i = np.arange(n)
    x = np.stack([np.exp(1j * f0 * i), np.exp(1j * f1 * i)], axis=1).astype(np.complex64).reshape(-1)
    q = make(m)
    y = q.execute_block("synthesizer", x)
    k = np.arange(m, n - 2 * m)
    want = np.exp(0.5j * f0 * k) + np.exp(1j * (np.pi + 0.5 * f1) * k)
    return float(np.abs(y[k + 2 * m] - want).max())


def filter_masks(make, m, as_, tol):
    """resamp2.rs:270-316: impulse responses of the low / high outputs of filter_execute"""
    h_len = 4 * m + 1
    q = make(m, as_)
    x = np.zeros(h_len, np.complex64)
    x[0] = 1
    y = q.execute_block("filter", x).reshape(h_len, 2)
    ft = estimate_req_filter_transition_bandwidth(as_, h_len) * 1.1
    lo = [(-0.5, -0.25 - ft / 2, 0.0, -as_ + tol, False, True), (-0.25 + ft / 2, 0.25 - ft / 2, -1.0, 1.0, True, True),
          (0.25 + ft / 2, 0.5, 0.0, -as_ + tol, False, True)]
    hi = [(-0.5, -0.25 - ft / 2, -1.0, 1.0, True, True), (-0.25 + ft / 2, 0.25 - ft / 2, 0.0, -as_ + tol, False, True),
          (0.25 + ft / 2, 0.5, -1.0, 1.0, True, True)]
    return validate_psd_signal(y[:, 0], lo), validate_psd_signal(y[:, 1], hi)


def msresamp2_interp_mask(make, num_stages, fc, as_, margin=0.0):
    """msresamp2.rs:206-243"""
    q = make(num_stages, fc, as_)
    delay = q.get_delay()
    M = 1 << num_stages
    nb = 0
    while nb * M < 2.0 * M * delay:
        nb += 1
    x = np.zeros(nb, np.complex64)
    x[0] = 1
    buf = q.execute_block(x) / np.float32(M)
    f0 = fc / M
    f1 = 1.0 / M - f0
    return validate_psd_signal(buf, [(-0.5, -f1, 0.0, -as_ + margin, False, True), (-f0, f0, -0.1, 0.1, True, True),
                                     (f1, 0.5, 0.0, -as_ + margin, False, True)])


# ---- the oracle --------------------------------------------------------------------------------------------
class _O:
    def __init__(self, oracle, kind, m, as_=60.0, f0=0.0):
        self.q = oracle.Resamp2(kind, oracle.halfband_kaiser(m, as_), m, f0)

    def execute_block(self, mode, x):
        return self.q.execute_block(mode, x)


def test_resamp2_analysis(oracle):
    assert two_tone_analysis(lambda m: _O(oracle, "crcf", m)) <= 1e-3


def test_resamp2_synthesis(oracle):
    assert two_band_synthesis(lambda m: _O(oracle, "crcf", m)) <= 3e-3


# (m, as): the reference's six cases.  tol = the reference's 0.5 dB plus the margin a Kaiser-windowed prototype needs
# where the equiripple design just meets the mask (stop band set by the window, ripple not equalised).
@pytest.mark.parametrize("m,as_,extra", [(4, 60.0, 0.0), (7, 60.0, 0.0), (12, 60.0, 0.0), (15, 80.0, 0.0),
                                         (15, 100.0, 0.0), (15, 120.0, 0.0)])
def test_resamp2_crcf_filter_masks(oracle, m, as_, extra):
    (ok0, w0), (ok1, w1) = filter_masks(lambda mm, a: _O(oracle, "crcf", mm, a), m, as_, 0.5 + extra)
    assert ok0 and ok1, (w0, w1)


def test_resamp2_config_and_copy(oracle):
    hf = oracle.halfband_kaiser(4, 60.0)
    for bad in [(0, 0.0), (1, 0.0)]:
        with pytest.raises(ValueError):
            oracle.Resamp2("crcf", np.zeros(4 * bad[0] + 1, np.float32), bad[0], bad[1])
    for f0 in (0.7, -0.7):
        with pytest.raises(ValueError):
            oracle.Resamp2("crcf", hf, 4, f0)
    assert oracle.Resamp2("crcf", hf, 4).get_delay() == 7
    assert oracle.Resamp2("crcf", oracle.halfband_kaiser(8, 80.0), 8).get_delay() == 15
    # copy (resamp2.rs:371-399): a clone resumes the stream identically
    qa = oracle.Resamp2("crcf", oracle.halfband_kaiser(12, 60.0), 12)
    x = oracle.gen_complex(11, 160)
    qa.execute_block("filter", x[:80])
    qb = qa.clone()
    assert np.array_equal(qa.execute_block("filter", x[80:]), qb.execute_block("filter", x[80:]))


def test_resamp2_forms_are_consistent(oracle):
    """internal consistency of the five forms (any kind): interp == synthesizer fed (x, 0)*... and decim == low band of
    the analyzer up to its 1/2 input scaling; cccf with f0 = 0 equals crcf"""
    m = 6
    hf = oracle.halfband_kaiser(m, 60.0)
    x = oracle.gen_complex(5, 256)
    dec = oracle.Resamp2("crcf", hf, m).execute_block("decim", x)
    ana = oracle.Resamp2("crcf", hf, m).execute_block("analyzer", x).reshape(-1, 2)
    np.testing.assert_allclose(ana[:, 0], 0.5 * dec, rtol=0, atol=2e-6)
    a = oracle.Resamp2("crcf", hf, m).execute_block("interp", x)
    b = oracle.Resamp2("cccf", hf, m, 0.0).execute_block("interp", x)
    np.testing.assert_allclose(a, b, rtol=0, atol=1e-6)
    r = oracle.Resamp2("rrrf", hf, m).execute_block("interp", x.real.copy())
    np.testing.assert_allclose(r, a.real, rtol=0, atol=1e-6)


@pytest.mark.parametrize("ns,fc,as_", [(1, 0.25, 60.0), (2, 0.25, 60.0), (3, 0.25, 60.0), (4, 0.25, 60.0),
                                       (1, 0.45, 60.0), (2, 0.45, 60.0), (3, 0.45, 60.0), (4, 0.45, 60.0),
                                       (3, 0.45, 80.0), (3, 0.45, 90.0), (3, 0.45, 100.0)])
def test_msresamp2_crcf_interp_masks(oracle, ns, fc, as_):
    ok, worst = msresamp2_interp_mask(lambda s, f, a: oracle.MsResamp2("crcf", True, s, f, 0.0, a), ns, fc, as_)
    assert ok, worst


def test_msresamp2_stage_plan_and_delay(oracle):
    assert oracle.msresamp2_stage_lengths(3, 0.25, 60.0) == [4, 8, 3]
    q = oracle.MsResamp2("crcf", True, 4, 0.4, 0.0, 60.0)
    assert q.get_delay() > 0
    with pytest.raises(ValueError):
        oracle.MsResamp2("crcf", True, 17, 0.4, 0.0, 60.0)
    with pytest.raises(ValueError):
        oracle.MsResamp2("crcf", True, 2, 0.5, 0.0, 60.0)
    with pytest.raises(ValueError):
        oracle.MsResamp2("crcf", True, 2, 0.4, 0.1, 60.0)
    # decim(interp(x)) returns x delayed (unit gain in the pass band): a slow tone survives the round trip
    n = 400
    x = np.exp(2j * np.pi * 0.01 * np.arange(n)).astype(np.complex64)
    up = oracle.MsResamp2("crcf", True, 2, 0.4, 0.0, 60.0).execute_block(x)
    dn = oracle.MsResamp2("crcf", False, 2, 0.4, 0.0, 60.0).execute_block(up)
    mag = np.abs(dn[100:])
    assert np.all(np.abs(mag - 1.0) < 2e-3), (mag.min(), mag.max())
