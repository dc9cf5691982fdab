"""HIP dotprod kernels (C ABI yagi_hip_dotprod_*) vs the oracle -- mirrors src/dotprod/mod.rs:291-677."""
import numpy as np
import pytest

from conftest import load_golden

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ya():
    import yagi_amd
    assert yagi_amd.device_count() > 0
    return yagi_amd


def test_basic_and_uneven(ya):
    assert ya.dotprod(np.float32([1, 2, 3]), np.float32([4, 5, 6])) == 32.0
    a = np.array([1 + 1j, 2 + 2j, 3 + 3j], np.complex64)
    b = np.array([4 - 4j, 5 - 5j, 6 - 6j], np.complex64)
    assert ya.dotprod(a, b) == 64.0 + 0j
    h = np.array([1, -1] * 8, np.float32)
    assert ya.dotprod(h, np.zeros(16, np.float32)) == 0
    assert ya.dotprod(h, np.ones(16, np.float32)) == 0
    assert ya.dotprod(h, (np.arange(16) % 2).astype(np.float32)) == -8
    assert ya.dotprod(h, (1 - np.arange(16) % 2).astype(np.float32)) == 8
    assert ya.dotprod(h, h) == 16
    for n, want in [(1, 1), (2, 0), (3, 1), (11, 1), (13, 1), (15, 1)]:
        assert ya.dotprod(h[:n], np.ones(n, np.float32)) == want
    assert ya.dotprod(np.zeros(0, np.float32), np.zeros(0, np.float32)) == 0   # empty input


def test_golden_vectors(ya):
    g = load_golden("dotprod")

    def chk(got, want, tol):
        assert abs(complex(got) - complex(want.item())) <= tol, (got, want)

    t = "test_dotprod_rrrf_rand01"
    chk(ya.dotprod(g[t + "__h"], g[t + "__x"]), g[t + "__test"], 1e-3)
    t = "test_dotprod_rrrf_rand02"
    chk(ya.dotprod(g[t + "__h"], g[t + "__x"]), g[t + "__test"], 1e-3)
    chk(ya.dotprod(g[t + "__h"][::-1], g[t + "__x"]), g[t + "__test_rev"], 1e-3)
    t = "test_dotprod_rrrf_struct_lengths"
    for n in (32, 33, 34, 35):
        chk(ya.dotprod(g[t + "__h"][:n], g[t + "__x"][:n]), g[f"{t}__len{n}"], 2e-6)
    t = "test_dotprod_crcf_rand01"
    chk(ya.dotprod(g[t + "__h"], g[t + "__x"]), g[t + "__test"], 1e-3)
    chk(ya.dotprod(g[t + "__h"][::-1], g[t + "__x"]), g[t + "__test_rev"], 1e-3)
    chk(ya.dotprod(g[t + "__x"], g[t + "__h"]), g[t + "__test"], 1e-3)     # [Complex].[f32] impl
    t = "test_dotprod_crcf_rand02"
    chk(ya.dotprod(g[t + "__h"], g[t + "__x"]), g[t + "__test"], 1e-3)
    t = "test_dotprod_cccf_rand16"
    chk(ya.dotprod(g[t + "__h"], g[t + "__x"]), g[t + "__test"], 1e-3)
    chk(ya.dotprod(g[t + "__h"][::-1], g[t + "__x"]), g[t + "__test_rev"], 1e-3)
    t = "test_dotprod_cccf_struct_lengths"
    for n in (32, 33, 34, 35):
        chk(ya.dotprod(g[t + "__h"][:n], g[t + "__x"][:n]), g[f"{t}__v{n}"], 8e-6)


@pytest.mark.parametrize("kind", ["rrrf", "rcc", "crc", "ccc"])
def test_struct_vs_ordinal(ya, oracle, kind):
    """n = 1..512 random vectors against the oracle's f64 sum (mod.rs:438-453 et al.)"""
    rng = np.random.default_rng(11)
    for n in list(range(1, 70)) + list(range(70, 513, 13)) + [512]:
        a = rng.random(n).astype(np.float32) if kind in ("rrrf", "rcc") else (rng.random(n) + 1j * rng.random(n)).astype(np.complex64)
        b = rng.random(n).astype(np.float32) if kind in ("rrrf", "crc") else (rng.random(n) + 1j * rng.random(n)).astype(np.complex64)
        got = ya.dotprod(a, b)
        want = oracle.dotprod_f64(kind, a, b)
        assert abs(complex(got) - complex(want)) <= 1e-4 * max(1.0, abs(want)), (n, got, want)
        seq = oracle.dotprod(kind, a, b)      # the reference's own f32 order is no closer to f64
        assert abs(complex(got) - complex(want)) <= abs(complex(seq) - complex(want)) + 8 * n * 1.2e-7


@pytest.mark.parametrize("n", [65_537, 1 << 20, 3_000_001])
def test_long_vectors_multi_workgroup(ya, oracle, n):
    rng = np.random.default_rng(n)
    a = (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(np.complex64)
    b = rng.standard_normal(n).astype(np.float32)
    got = ya.dotprod(a, b)
    want = oracle.dotprod_f64("crc", a, b)
    scale = np.sqrt(n)
    assert abs(complex(got) - complex(want)) <= 2e-5 * scale
    assert ya.dotprod(a, b) == got          # fixed-order combine: bitwise reproducible
    c = (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(np.complex64)
    got = ya.dotprod(a, c)
    want = oracle.dotprod_f64("ccc", a, c)
    assert abs(complex(got) - complex(want)) <= 4e-5 * scale


def test_integer_inputs_are_exact(ya):
    rng = np.random.default_rng(5)
    a = rng.integers(-8, 9, 4096).astype(np.float32)
    b = rng.integers(-8, 9, 4096).astype(np.float32)
    assert ya.dotprod(a, b) == np.float32(np.dot(a.astype(np.int64), b.astype(np.int64)))


@pytest.mark.parametrize("name,ta,tb", [("rrrf", np.float32, np.float32), ("crcf", np.complex64, np.float32),
                                        ("cccf", np.complex64, np.complex64)])
def test_device_pointers_of_any_alignment(ya, name, ta, tb):
    """*_dev entry points on device-resident operands: the 16-byte-load path (operands on 16-byte boundaries) and the
    element-wise path (operands offset by one element) must agree with each other to rounding and with the f64 sum;
    odd lengths leave a tail for the scalar loop."""
    import ctypes as C
    rng = np.random.default_rng(11)
    n = 3 * 256 * 64 + 1237

    def rnd(t, m):
        v = rng.standard_normal(m)
        return (v + 1j * rng.standard_normal(m)).astype(t) if t == np.complex64 else v.astype(t)

    a, b = rnd(ta, n + 1), rnd(tb, n + 1)
    da, db = ya.DeviceArray.from_numpy(a), ya.DeviceArray.from_numpy(b)
    ty = np.float32 if name == "rrrf" else np.complex64
    dy = ya.DeviceArray(2, ty)
    fn = getattr(ya.lib, f"yagi_hip_dotprod_{name}_dev")
    got = []
    for off in (0, 1):
        pa, pb = da.ptr + off * a.itemsize, db.ptr + off * b.itemsize
        assert fn(C.c_void_p(pa), C.c_void_p(pb), n, C.c_void_p(dy.ptr), None) == 0
        ya.synchronize()
        got.append(dy.to_numpy(1)[0])
        want = np.sum(a[off:off + n].astype(np.complex128) * b[off:off + n].astype(np.complex128))
        assert abs(complex(got[-1]) - want) <= 4e-5 * np.sqrt(n)
