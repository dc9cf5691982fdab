"""Resamp2 / MsResamp2 through the C ABI (HIP kernels) -- SURVEY.md section 8(f-4).

Mirrors src/filter/resampler/resamp2.rs:183-399 and msresamp2.rs:199-337: the reference's tests for these objects
are property tests (tone analysis / synthesis, spectral masks of impulse responses, config, copy) -- the same
harness as tests/test_oracle_resamp2.py, run here on the GPU objects -- plus sample-level parity with the oracle's
restatement on the same prototypes: bit-exact on integer-valued data (indexing: which sample meets which tap, the
delay branch, the toggle), rel. L2 <= 2e-6 on Gaussian data.  Prototype VALUES are Kaiser half-band (the reference's
Parks-McClellan design is out of scope): "parity unpinned" for the taps, pinned for everything after them."""
import numpy as np
import pytest

from gpu_util import SEED, int_samples, rand_samples, rel_l2
from test_oracle_resamp2 import filter_masks, msresamp2_interp_mask, two_band_synthesis, two_tone_analysis

pytestmark = pytest.mark.gpu
KINDS = ["rrrf", "crcf", "cccf"]
MODES = ["filter", "analyzer", "synthesizer", "decim", "interp"]


@pytest.fixture(scope="module")
def ya():
    import yagi_amd
    assert yagi_amd.device_count() > 0
    return yagi_amd


class _G:
    """GPU object with the oracle wrapper's execute_block(mode_name, x)"""
    def __init__(self, ya, kind, m, as_=60.0, f0=0.0, hf=None):
        self.ya = ya
        self.q = ya.Resamp2.new(kind, m, f0, as_) if hf is None else ya.Resamp2(kind, hf, m, f0)

    def execute_block(self, mode, x):
        return self.q.execute_block(MODES.index(mode), x)


def int_halfband(rng, m):
    """integer-valued 'prototype': exact arithmetic, so any indexing slip changes the output"""
    return rng.integers(-3, 4, 4 * m + 1).astype(np.float32)


@pytest.mark.parametrize("kind", KINDS)
@pytest.mark.parametrize("mode", MODES)
def test_forms_bit_exact_on_integers(ya, oracle, kind, mode):
    rng = np.random.default_rng(7)
    for m in (2, 5, 17):
        hf = int_halfband(rng, m)
        g = ya.Resamp2(kind, hf, m, 0.0)
        o = oracle.Resamp2(kind, hf, m, 0.0)
        g.set_scale(2.0)
        o.set_scale(2.0)
        # three calls: state carried; odd lengths for the sample-wise forms (the toggle flips between calls)
        sizes = (301, 1, 778) if mode in ("filter", "interp") else (300, 2, 778)
        for n in sizes:
            x = int_samples(rng, kind, n) * (2 if mode == "analyzer" else 1)      # analyzer halves its input
            got = g.execute_block(MODES.index(mode), x)
            want = o.execute_block(mode, x)
            assert got.shape == want.shape
            assert np.array_equal(got, want), (kind, mode, m, n)


@pytest.mark.parametrize("kind", KINDS)
def test_forms_vs_oracle_gaussian(ya, oracle, kind):
    rng = np.random.default_rng(8)
    for m, f0 in ((4, 0.0), (12, 0.11 if kind == "cccf" else 0.0), (30, 0.0)):
        hf = oracle.halfband_kaiser(m, 70.0)
        for mode in MODES:
            g = ya.Resamp2(kind, hf, m, f0)
            o = oracle.Resamp2(kind, hf, m, f0)
            x = rand_samples(rng, kind, 4096 + 2 * 333)
            got = np.concatenate([g.execute_block(MODES.index(mode), x[:666]), g.execute_block(MODES.index(mode), x[666:])])
            want = np.concatenate([o.execute_block(mode, x[:666]), o.execute_block(mode, x[666:])])
            assert rel_l2(got, want) <= 2e-6, (kind, mode, m)


def test_f0_modulation_rrrf_crcf(ya, oracle):
    """for_halfband (resamp2.rs:9-23): real coefficients take the cosine only"""
    m = 6
    hf = oracle.halfband_kaiser(m, 60.0)
    x = rand_samples(np.random.default_rng(3), "crcf", 512)
    for kind in ("crcf", "cccf"):
        got = ya.Resamp2(kind, hf, m, 0.2).execute_block(3, x)
        want = oracle.Resamp2(kind, hf, m, 0.2).execute_block("decim", x)
        assert rel_l2(got, want) <= 2e-6


def test_reference_tone_tests_on_gpu(ya):
    assert two_tone_analysis(lambda m: _G(ya, "crcf", m)) <= 1e-3               # resamp2.rs:189-229
    assert two_band_synthesis(lambda m: _G(ya, "crcf", m)) <= 3e-3              # :231-268


@pytest.mark.parametrize("m,as_", [(4, 60.0), (7, 60.0), (12, 60.0), (15, 80.0), (15, 100.0), (15, 120.0)])
def test_reference_filter_masks_on_gpu(ya, m, as_):                             # :270-336
    (ok0, w0), (ok1, w1) = filter_masks(lambda mm, a: _G(ya, "crcf", mm, a), m, as_, 0.5)
    assert ok0 and ok1, (w0, w1)


def test_config_copy_reset(ya, oracle):                                         # :338-399
    for bad in [lambda: ya.Resamp2.new("crcf", 0, 0.0, 60.0), lambda: ya.Resamp2.new("crcf", 1, 0.0, 60.0),
                lambda: ya.Resamp2.new("crcf", 2, 0.7, 60.0), lambda: ya.Resamp2.new("crcf", 2, -0.7, 60.0),
                lambda: ya.Resamp2.new("crcf", 2, 0.0, -1.0), lambda: ya.Resamp2("crcf", np.zeros(16, np.float32), 4)]:
        with pytest.raises(ya.ConfigError):
            bad()
    assert ya.Resamp2.new("crcf", 4, 0.0, 60.0).get_delay() == 7
    q = ya.Resamp2.new("crcf", 8, 0.0, 80.0)
    assert q.get_delay() == 15
    q.set_scale(7.22)
    assert q.get_scale() == np.float32(7.22)
    with pytest.raises(ya.ConfigError):
        q.execute_block(3, np.zeros(5, np.complex64))                           # decim consumes pairs
    with pytest.raises(ya.ConfigError):
        q.execute_block(9, np.zeros(4, np.complex64))
    # copy: a clone taken mid-stream (odd count: toggle set) continues identically; reset restarts
    qa = ya.Resamp2.new("crcf", 12, 0.0, 60.0)
    x = oracle.gen_complex(SEED + 9, 241)
    first = qa.execute_block(0, x[:81])
    qb = qa.clone()
    assert np.array_equal(qa.execute_block(0, x[81:]), qb.execute_block(0, x[81:]))
    qa.reset()
    assert np.array_equal(qa.execute_block(0, x[:81]), first)
    # the per-call forms
    qc, qd = ya.Resamp2.new("crcf", 5, 0.0, 60.0), ya.Resamp2.new("crcf", 5, 0.0, 60.0)
    blk = qd.execute_block(0, x[:6]).reshape(-1, 2)
    for i in range(6):
        y0, y1 = qc.filter_execute(x[i])
        assert (y0, y1) == (blk[i, 0], blk[i, 1])


def test_large_block_device_path(ya, oracle):
    """2^22 samples through the device entry point: interp then decim of the same half-band pair returns the input
    delayed by 2m - 1 + ... (unit pass-band gain), and the decimator's every output matches the oracle on a window"""
    m, n = 10, 1 << 22
    x = ya.gen_complex_dev(SEED + 9, n)
    up = ya.DeviceArray(2 * n, np.complex64)
    dn = ya.DeviceArray(n, np.complex64)
    qi, qd = ya.Resamp2.new("crcf", m, 0.0, 60.0), ya.Resamp2.new("crcf", m, 0.0, 60.0)
    qi.execute_block_dev(ya.Resamp2.INTERP, x, n, up)
    qd.execute_block_dev(ya.Resamp2.DECIM, up, 2 * n, dn)
    ya.synchronize()
    xs = x.to_numpy()
    hf = oracle.halfband_kaiser(m, 60.0)
    oi, od = oracle.Resamp2("crcf", hf, m), oracle.Resamp2("crcf", hf, m)
    k = 1 << 14
    want = od.execute_block("decim", oi.execute_block("interp", xs[:k]))
    assert rel_l2(dn.to_numpy(k), want) <= 2e-6
    # deep inside the block: recompute a window with the history it needs
    lo = n - 4096
    oi2, od2 = oracle.Resamp2("crcf", hf, m), oracle.Resamp2("crcf", hf, m)
    w = od2.execute_block("decim", oi2.execute_block("interp", xs[lo - 8 * m:]))
    assert rel_l2(dn.to_numpy(4096, lo), w[8 * m:]) <= 2e-6


@pytest.mark.parametrize("kind,m", [("crcf", 12), ("rrrf", 2), ("cccf", 7), ("crcf", 33), ("rrrf", 64), ("crcf", 65)])
def test_long_decimator_block_equals_short_blocks(ya, oracle, kind, m):
    """a decimator block of >= 2^19 samples runs as the one-stage case of the MsResamp2 chain kernels (its middle four
    outputs per lane through msresamp2_decim_fast_kernel; m > 64 stays on resamp2_kernel): the same input cut into
    blocks below that size goes through resamp2_kernel -- identical bits, windows carried across the calls either way"""
    hf = oracle.halfband_kaiser(m, 60.0)
    a, b = ya.Resamp2(kind, hf, m), ya.Resamp2(kind, hf, m)
    a.set_scale(0.37)
    b.set_scale(0.37)
    dt = np.float32 if kind == "rrrf" else np.complex64
    item = np.dtype(dt).itemsize
    n1, n2 = (1 << 21) + 2 * 12345, 2 * 777                     # input samples of the two calls
    x = ya.gen_complex_dev(SEED + 31, n1 + n2)                  # rrrf reads the first half of it as real samples
    got, want = ya.DeviceArray((n1 + n2) // 2, dt), ya.DeviceArray((n1 + n2) // 2, dt)
    a.execute_block_dev(ya.Resamp2.DECIM, x.ptr, n1, got.ptr)
    a.execute_block_dev(ya.Resamp2.DECIM, x.ptr + n1 * item, n2, got.ptr + (n1 // 2) * item)
    step, off = (1 << 18) + 2 * 101, 0
    while off < n1 + n2:
        cnt = min(step, n1 + n2 - off)
        b.execute_block_dev(ya.Resamp2.DECIM, x.ptr + off * item, cnt, want.ptr + (off // 2) * item)
        off += cnt
    ya.synchronize()
    assert np.array_equal(got.to_numpy(), want.to_numpy()), (kind, m)


# ---- MsResamp2 ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("ns,fc,as_", [(1, 0.25, 60.0), (2, 0.25, 60.0), (3, 0.25, 60.0), (4, 0.25, 60.0),
                                       (1, 0.45, 60.0), (2, 0.45, 60.0), (3, 0.45, 60.0), (4, 0.45, 60.0),
                                       (3, 0.45, 80.0), (3, 0.45, 90.0), (3, 0.45, 100.0)])
def test_msresamp2_reference_interp_masks_on_gpu(ya, ns, fc, as_):              # msresamp2.rs:206-298
    ok, worst = msresamp2_interp_mask(lambda s, f, a: ya.MsResamp2("crcf", ya.MsResamp2.INTERP, s, f, 0.0, a), ns, fc, as_)
    assert ok, worst


@pytest.mark.parametrize("kind", KINDS)
@pytest.mark.parametrize("interp", [True, False])
def test_msresamp2_vs_oracle(ya, oracle, kind, interp):
    rng = np.random.default_rng(12)
    for ns, fc in ((0, 0.3), (1, 0.25), (3, 0.45), (5, 0.4)):
        g = ya.MsResamp2(kind, ya.MsResamp2.INTERP if interp else ya.MsResamp2.DECIM, ns, fc, 0.0, 60.0)
        o = oracle.MsResamp2(kind, interp, ns, fc, 0.0, 60.0)
        assert g.get_stage_lengths() == o.m_stage and g.get_num_stages() == ns
        assert abs(g.get_delay() - o.get_delay()) < 1e-6 and g.get_rate() == (float(1 << ns) if interp else 1.0 / (1 << ns))
        rate = 1 << ns
        n1, n2 = 37, 200
        x = rand_samples(rng, kind, (n1 + n2) * (1 if interp else rate))
        cut = n1 * (1 if interp else rate)
        got = np.concatenate([g.execute_block(x[:cut]), g.execute_block(x[cut:])])          # state carried
        want = np.concatenate([o.execute_block(x[:cut]), o.execute_block(x[cut:])])
        assert got.shape == want.shape and rel_l2(got, want) <= 3e-6, (kind, interp, ns)


@pytest.mark.parametrize("kind,ms", [("crcf", [10, 5, 3]), ("rrrf", [6, 4]), ("cccf", [4, 3, 3, 2]), ("crcf", [7]),
                                     ("cccf", [2, 13, 6]), ("rrrf", [12, 2, 9]), ("crcf", [2, 2, 2]), ("rrrf", [30, 17])])
def test_msresamp2_fused_decimator_equals_the_chain_on_a_large_block(ya, oracle, kind, ms):
    """the decimator chain in one launch (<= 4 stages, LDS-resident intermediates, several tiles per workgroup from
    8192 tiles on; up to three stages: the middle of the block through msresamp2_decim_fast_kernel, its two ends through
    the general kernel) against the same stages run one after the other as Resamp2 objects: identical bits, over two
    calls (the windows of every stage written by the fused kernel's last workgroup carry into the second call)"""
    ns = len(ms)
    rate = 1 << ns
    hfs = [oracle.halfband_kaiser(m, 60.0) for m in ms]
    q = ya.MsResamp2.from_taps(kind, ya.MsResamp2.DECIM, ms, hfs)
    chain = [ya.Resamp2(kind, hfs[g], ms[g]) for g in range(ns)]
    chain[0].set_scale(1.0 / rate)                             # zeta multiplies the last stage's output (msresamp2.rs:197)
    cplx = kind != "rrrf"
    dt = np.complex64 if cplx else np.float32
    n1, n2 = (1 << 21) + 1000, 4099                            # outputs of the two calls: 8196 tiles, then a short block
    x = ya.gen_complex_dev(SEED + 21, (n1 + n2) * rate)          # rrrf reads the first half of it as real samples
    got = ya.DeviceArray(n1 + n2, dt)
    item = np.dtype(dt).itemsize
    q.execute_block_dev(x.ptr, n1, got.ptr)
    q.execute_block_dev(x.ptr + n1 * rate * item, n2, got.ptr + n1 * item)
    # the chain: stage g = S-1 first, each halves
    outs = []
    for off, cnt in ((0, n1), (n1, n2)):
        src, n = x.ptr + off * rate * item, cnt * rate
        for g in range(ns - 1, -1, -1):
            dst = ya.DeviceArray(n // 2, dt)
            chain[g].execute_block_dev(ya.Resamp2.DECIM, src, n, dst)
            keep = dst                                          # keep the buffer alive while the next stage reads it
            src, n = dst.ptr, n // 2
            outs.append(keep)
        ya.synchronize()
        want = keep.to_numpy()
        assert np.array_equal(got.to_numpy(cnt, off), want), (kind, off)


@pytest.mark.parametrize("interp", [False, True])
def test_msresamp2_fused_chain_randomised(ya, oracle, interp):
    """seeded sweep: 1-4 stages, semi-lengths 2..14, three calls of ragged lengths each (1 sample .. a few tiles); the
    one-launch decimator / interpolator chain equals the Resamp2 stages run one after the other bit for bit, state
    carried from call to call"""
    rng = np.random.default_rng(2024 + int(interp))
    for case in range(24):
        kind = ("rrrf", "crcf", "cccf")[case % 3]
        ns = int(rng.integers(1, 5))
        ms = [int(v) for v in rng.integers(2, 15, ns)]
        rate = 1 << ns
        hfs = [oracle.halfband_kaiser(m, 60.0) for m in ms]
        q = ya.MsResamp2.from_taps(kind, ya.MsResamp2.INTERP if interp else ya.MsResamp2.DECIM, ms, hfs)
        chain = [ya.Resamp2(kind, hfs[g], ms[g]) for g in range(ns)]
        if not interp:
            chain[0].set_scale(1.0 / rate)
        for n in (int(rng.integers(1, 40)), int(rng.integers(200, 1500)), int(rng.integers(1, 700))):
            x = rand_samples(rng, kind, n if interp else n * rate)
            got = q.execute_block(x)
            cur = x
            if interp:
                for g in range(ns):
                    cur = chain[g].execute_block(ya.Resamp2.INTERP, cur)
            else:
                for g in range(ns - 1, -1, -1):
                    cur = chain[g].execute_block(ya.Resamp2.DECIM, cur)
            assert np.array_equal(got, cur), (case, kind, ms, n)


def test_msresamp2_config_copy(ya, oracle):                                     # :38-48, :300-337
    for bad in [lambda: ya.MsResamp2("crcf", 1, 17, 0.4, 0.0, 60.0), lambda: ya.MsResamp2("crcf", 1, 2, 0.5, 0.0, 60.0),
                lambda: ya.MsResamp2("crcf", 1, 2, 0.0, 0.0, 60.0), lambda: ya.MsResamp2("crcf", 1, 2, 0.4, 0.1, 60.0)]:
        with pytest.raises(ya.ConfigError):
            bad()
    q0 = ya.MsResamp2("crcf", ya.MsResamp2.INTERP, 4, 0.4, 0.0, 60.0)
    x = oracle.gen_complex(SEED + 10, 70)
    q0.execute_block(x[:35])
    q1 = q0.clone()
    assert np.array_equal(q0.execute_block(x[35:]), q1.execute_block(x[35:]))
    q0.reset()
    fresh = ya.MsResamp2("crcf", ya.MsResamp2.INTERP, 4, 0.4, 0.0, 60.0)
    assert np.array_equal(q0.execute_block(x[:35]), fresh.execute_block(x[:35]))
    # externally designed stage prototypes (what a host with the reference's PM design would pass)
    ms = [4, 8]
    hfs = [oracle.halfband_kaiser(m, 65.0) for m in ms]
    a = ya.MsResamp2.from_taps("crcf", ya.MsResamp2.DECIM, ms, hfs)
    b = ya.MsResamp2("crcf", ya.MsResamp2.DECIM, 2, 0.25, 0.0, 60.0)
    assert b.get_stage_lengths() == ms
    xx = oracle.gen_complex(SEED + 11, 4 * 100)
    assert rel_l2(a.execute_block(xx), b.execute_block(xx)) <= 1e-6


def test_rresamp_clone(ya, oracle):
    """Rresamp is #[derive(Clone)] (rresamp.rs:8): a clone taken mid-stream continues identically"""
    q = ya.Rresamp.new_kaiser("crcf", 3, 5, 7, 0.4, 60.0)
    x = oracle.gen_complex(SEED + 12, 5 * 40)
    q.execute_block(x[: 5 * 15], 15)
    c = q.clone()
    assert np.array_equal(q.execute_block(x[5 * 15:], 25), c.execute_block(x[5 * 15:], 25))
    assert (c.get_interp(), c.get_decim(), c.get_delay()) == (3, 5, 7)


def test_output_array_checks(ya):
    """caller-supplied output arrays must be contiguous arrays of the object's type and size"""
    p = ya.Fft(16, ya.Direction.Forward)
    x = np.ones(16, np.complex64)
    with pytest.raises(ya.ConfigError):
        p.run(x, output=np.zeros(16, np.complex128))
    with pytest.raises(ya.ConfigError):
        p.run(x, output=np.zeros(32, np.complex64)[::2])
    out = np.zeros(16, np.complex64)
    assert p.run(x, output=out) is out and out[0] == 16


def test_device_entry_points_reject_overlapping_buffers(ya):
    """in-place / overlapping x and y must be refused (the kernels read halos and the carried window from x
    while other workgroups store y), with the object's state untouched"""
    n = 4096
    buf = ya.DeviceArray(3 * n, np.complex64)
    buf.zero()
    h = ya.fir_design_kaiser(31, 0.2, 60.0)
    q = ya.FirFilter("crcf", h)
    for xoff, yoff in ((0, 0), (0, n - 1), (n // 2, 0)):
        with pytest.raises(ya.ConfigError):
            q.execute_block_dev(buf.ptr + 8 * xoff, n, buf.ptr + 8 * yoff)
    q.execute_block_dev(buf.ptr, n, buf.ptr + 8 * n)                         # adjacent is fine
    d = ya.FirDecimationFilter("crcf", 4, h)
    with pytest.raises(ya.ConfigError):
        d.execute_block_dev(buf.ptr, n // 4, buf.ptr + 8 * (n - 1))         # output inside the 4n/4-sample input
    s = ya.FirFftStream(ya.fir_design_kaiser(256, 0.2, 60.0))
    with pytest.raises(ya.ConfigError):
        s.execute_dev(buf.ptr, 1, buf.ptr + 8 * 100)
    c = ya.FirPfbCh2.new_kaiser(8, 2, 60.0)
    with pytest.raises(ya.ConfigError):
        c.analyzer_execute_dev(buf.ptr, 64, buf.ptr + 8 * 10)
    r = ya.Resamp2.new("crcf", 4, 0.0, 60.0)
    with pytest.raises(ya.ConfigError):
        r.execute_block_dev(ya.Resamp2.INTERP, buf.ptr, n, buf.ptr + 8 * (n - 1))
    ya.synchronize()


@pytest.mark.parametrize("kind", ["rrrf", "crcf", "cccf"])
def test_resamp2_length_mismatch_is_a_config_error(ya, kind):
    """the resamp2 / msresamp2 block entry points carry both buffer lengths (the reference's slices do:
    msresamp2.rs:181 copy_from_slice panics on a mismatch); a wrong pair is YAGI_ERR_CONFIG and nothing is read or
    written, host and device forms alike, with the object's state untouched"""
    T = np.float32 if kind == "rrrf" else np.complex64
    rng = np.random.default_rng(5)
    x = (rng.standard_normal(64) + (0 if kind == "rrrf" else 1j * rng.standard_normal(64))).astype(T)
    md, md2 = (ya.MsResamp2(kind, ya.MsResamp2.DECIM, 2, 0.4, 0.0, 60.0) for _ in range(2))
    mi = ya.MsResamp2(kind, ya.MsResamp2.INTERP, 2, 0.4, 0.0, 60.0)
    with pytest.raises(ya.ConfigError):
        md.execute_block(x[:30], 8)                   # 8 outputs of a /4 decimator need 32 inputs
    with pytest.raises(ya.ConfigError):
        md.execute_block(x[:33], 8)
    with pytest.raises(ya.ConfigError):
        mi.execute_block(x[:7], 8)                    # 8 calls of an interpolator need 8 inputs
    assert np.array_equal(md.execute_block(x[:32], 8), md2.execute_block(x[:32], 8))    # state untouched by the refusals
    dx = ya.DeviceArray.from_numpy(x)
    dy = ya.DeviceArray(256, T)
    with pytest.raises(ya.ConfigError):
        md.execute_block_dev(dx, 8, dy, nx=31)
    with pytest.raises(ya.ConfigError):
        mi.execute_block_dev(dx, 8, dy, ny=31)
    r = ya.Resamp2.new(kind, 5, 0.0, 60.0)
    lib_fn = r._fn("execute_block")
    y = np.zeros(256, T)
    from yagi_amd import _ptr
    for mode, nx, ny in ((ya.Resamp2.FILTER, 8, 8), (ya.Resamp2.DECIM, 8, 8), (ya.Resamp2.INTERP, 8, 8),
                         (ya.Resamp2.ANALYZER, 8, 16), (ya.Resamp2.SYNTHESIZER, 8, 4), (ya.Resamp2.DECIM, 7, 3)):
        assert lib_fn(r._h, mode, _ptr(x), nx, _ptr(y), ny) == 2, (mode, nx, ny)      # YAGI_ERR_CONFIG
        with pytest.raises(ya.ConfigError):
            r.execute_block_dev(mode, dx, nx, dy, ny)
    assert not y.any()
