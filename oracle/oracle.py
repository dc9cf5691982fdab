"""ctypes front-end of the CPU oracle (oracle/yagi_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
``cpu_baseline`` leg of bench.py -- never by yagi_amd/.  See the header of
yagi_oracle.c for what is pinned by the reference's golden vectors and what is
"parity unpinned".
"""
import ctypes as C
import subprocess
from pathlib import Path

import numpy as np

_DIR = Path(__file__).resolve().parent
_LIB_PATH = _DIR / "libyagi_oracle.so"


def build(force: bool = False) -> Path:
    src = _DIR / "yagi_oracle.c"
    if force or not _LIB_PATH.exists() or _LIB_PATH.stat().st_mtime < src.stat().st_mtime:
        subprocess.run(["make", "-C", str(_DIR), "-B", "libyagi_oracle.so"], check=True,
                       stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None


class cf32(C.Structure):
    _fields_ = [("re", C.c_float), ("im", C.c_float)]


class cf64(C.Structure):
    _fields_ = [("re", C.c_double), ("im", C.c_double)]


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(str(_LIB_PATH))
        vp, sz, fp = C.c_void_p, C.c_size_t, C.c_float
        L.yo_dotprod_rrrf.restype = C.c_float
        L.yo_dotprod_rrrf.argtypes = [vp, vp, sz]
        L.yo_dotprod_rrrf_f64.restype = C.c_double
        L.yo_dotprod_rrrf_f64.argtypes = [vp, vp, sz]
        for n in ("rcc", "crc", "ccc", "crc_f64", "ccc_f64"):
            f = getattr(L, f"yo_dotprod_{n}")
            f.restype = None
            f.argtypes = [vp, vp, sz, vp]
        for k, coeff in (("rrrf", fp), ("crcf", fp), ("cccf", cf32)):
            T = fp if k == "rrrf" else cf32
            g = lambda name: getattr(L, f"yo_{name}")
            g(f"firfilt_{k}_create").restype = vp
            g(f"firfilt_{k}_create").argtypes = [vp, sz]
            g(f"firfilt_{k}_clone").restype = vp
            g(f"firfilt_{k}_clone").argtypes = [vp]
            g(f"firfilt_{k}_destroy").argtypes = [vp]
            g(f"firfilt_{k}_reset").argtypes = [vp]
            g(f"firfilt_{k}_set_scale").argtypes = [vp, coeff]
            g(f"firfilt_{k}_push").argtypes = [vp, T]
            g(f"firfilt_{k}_execute").restype = T
            g(f"firfilt_{k}_execute").argtypes = [vp]
            g(f"firfilt_{k}_execute_block").restype = C.c_int
            g(f"firfilt_{k}_execute_block").argtypes = [vp, vp, sz, vp, sz]
            g(f"firdecim_{k}_create").restype = vp
            g(f"firdecim_{k}_create").argtypes = [sz, vp, sz]
            g(f"firdecim_{k}_destroy").argtypes = [vp]
            g(f"firdecim_{k}_reset").argtypes = [vp]
            g(f"firdecim_{k}_set_scale").argtypes = [vp, coeff]
            g(f"firdecim_{k}_execute").restype = T
            g(f"firdecim_{k}_execute").argtypes = [vp, vp]
            g(f"firdecim_{k}_execute_block").restype = None
            g(f"firdecim_{k}_execute_block").argtypes = [vp, vp, sz, vp]
            g(f"firpfb_{k}_create").restype = vp
            g(f"firpfb_{k}_create").argtypes = [sz, vp, sz]
            g(f"firpfb_{k}_destroy").argtypes = [vp]
            g(f"firpfb_{k}_reset").argtypes = [vp]
            g(f"firpfb_{k}_set_scale").argtypes = [vp, coeff]
            g(f"firpfb_{k}_push").argtypes = [vp, T]
            g(f"firpfb_{k}_execute").restype = C.c_int
            g(f"firpfb_{k}_execute").argtypes = [vp, sz, vp]
            g(f"firpfb_{k}_execute_block").restype = C.c_int
            g(f"firpfb_{k}_execute_block").argtypes = [vp, sz, vp, sz, vp]
        L.yo_fir_block_f64.restype = None
        L.yo_fir_block_f64.argtypes = [C.c_int, vp, sz, vp, vp, sz, sz, sz, vp]
        L.yo_dft_f64.restype = None
        L.yo_dft_f64.argtypes = [vp, sz, C.c_int, vp]
        L.yo_fft_shift.argtypes = [vp, sz]
        L.yo_fft_plan_create.restype = vp
        L.yo_fft_plan_create.argtypes = [sz, C.c_int]
        L.yo_fft_plan_destroy.argtypes = [vp]
        L.yo_fft_run_f32.restype = None
        L.yo_fft_run_f32.argtypes = [vp, vp, vp]
        L.yo_fir_design_kaiser.restype = C.c_int
        L.yo_fir_design_kaiser.argtypes = [sz, fp, fp, fp, vp]
        L.yo_fir_design_notch.restype = C.c_int
        L.yo_fir_design_notch.argtypes = [sz, fp, fp, vp]
        L.yo_besseli0f.restype = fp
        L.yo_besseli0f.argtypes = [fp]
        L.yo_gen_real.argtypes = [C.c_uint64, C.c_uint64, sz, vp]
        L.yo_gen_complex.argtypes = [C.c_uint64, C.c_uint64, sz, vp]
        L.yo_window_create.restype = vp
        L.yo_window_create.argtypes = [sz, sz]
        L.yo_window_destroy.argtypes = [vp]
        L.yo_window_reset.argtypes = [vp]
        L.yo_window_read.restype = vp
        L.yo_window_read.argtypes = [vp]
        L.yo_window_index.restype = C.c_int
        L.yo_window_index.argtypes = [vp, sz, vp]
        L.yo_window_push.argtypes = [vp, vp]
        L.yo_window_write.argtypes = [vp, vp, sz]
        L.yo_window_clone.restype = vp
        L.yo_window_clone.argtypes = [vp]
        L.yo_window_resize.restype = C.c_int
        L.yo_window_resize.argtypes = [vp, sz]
        L.yo_firpfbch_create.restype = vp
        L.yo_firpfbch_create.argtypes = [sz, sz, vp]
        L.yo_firpfbch_destroy.argtypes = [vp]
        L.yo_firpfbch_analyzer_execute.argtypes = [vp, vp, sz, vp]
        L.yo_firpfbch2_create.restype = vp
        L.yo_firpfbch2_create.argtypes = [sz, sz, vp]
        L.yo_firpfbch2_destroy.argtypes = [vp]
        L.yo_firpfbch2_analyzer_execute.argtypes = [vp, vp, sz, vp]
        L.yo_stream_fir_fft.argtypes = [vp, vp, vp, sz, vp, vp]
        L.yo_window_fn.restype = fp
        L.yo_window_fn.argtypes = [C.c_int, sz, sz, fp]
        L.yo_fftfilt_create.restype = vp
        L.yo_fftfilt_create.argtypes = [C.c_int, vp, sz, sz]
        L.yo_fftfilt_destroy.argtypes = [vp]
        L.yo_fftfilt_reset.argtypes = [vp]
        L.yo_fftfilt_set_scale.argtypes = [vp, fp, fp]
        L.yo_fftfilt_execute.argtypes = [vp, vp, vp]
        for k, (T, Cc, _) in (("rrrf", (fp, fp, 0)), ("crcf", (cf32, fp, 1)), ("cccf", (cf32, cf32, 2))):
            r = "yo_resamp2_" + k
            getattr(L, r + "_create").restype = vp
            getattr(L, r + "_create").argtypes = [vp, sz, fp]
            getattr(L, r + "_destroy").argtypes = [vp]
            getattr(L, r + "_clone").restype = vp
            getattr(L, r + "_clone").argtypes = [vp]
            getattr(L, r + "_reset").argtypes = [vp]
            getattr(L, r + "_set_scale").argtypes = [vp, Cc]
            getattr(L, r + "_execute_block").argtypes = [vp, C.c_int, vp, sz, vp]
            m_ = "yo_msresamp2_" + k
            getattr(L, m_ + "_create").restype = vp
            getattr(L, m_ + "_create").argtypes = [C.c_int, sz, vp, vp, Cc]
            getattr(L, m_ + "_destroy").argtypes = [vp]
            getattr(L, m_ + "_execute_block").argtypes = [vp, vp, sz, vp]
        L.yo_estimate_req_filter_len.restype = sz
        L.yo_estimate_req_filter_len.argtypes = [fp, fp]
        L.yo_msresamp2_stage_lengths.restype = C.c_int
        L.yo_msresamp2_stage_lengths.argtypes = [sz, fp, fp, vp]
        _lib = L
    return _lib


# ---------------------------------------------------------------------------------
# helpers
# ---------------------------------------------------------------------------------
KINDS = {"rrrf": (np.float32, np.float32, 0), "crcf": (np.complex64, np.float32, 1),
         "cccf": (np.complex64, np.complex64, 2)}


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _as(a, dt):
    return np.ascontiguousarray(np.asarray(a, dtype=dt))


def _scalar(kind, v, coeff=False):
    """numpy scalar -> ctypes by-value argument for T (or Coeff)."""
    dt = KINDS[kind][1 if coeff else 0]
    if dt == np.float32:
        return C.c_float(float(np.real(v)))
    v = complex(v)
    return cf32(v.real, v.imag)


def _from(kind, v):
    if isinstance(v, cf32):
        return np.complex64(complex(v.re, v.im))
    return np.float32(v)


def dotprod(kind, a, b):
    """kind in rrrf | rcc ([f32].[Complex]) | crc ([Complex].[f32]) | ccc; f32 sequential sum."""
    L = lib()
    if kind == "rrrf":
        a, b = _as(a, np.float32), _as(b, np.float32)
        return np.float32(L.yo_dotprod_rrrf(_p(a), _p(b), min(len(a), len(b))))
    dta = np.float32 if kind == "rcc" else np.complex64
    dtb = np.float32 if kind == "crc" else np.complex64
    a, b = _as(a, dta), _as(b, dtb)
    y = np.zeros(1, np.complex64)
    getattr(L, f"yo_dotprod_{kind}")(_p(a), _p(b), min(len(a), len(b)), _p(y))
    return y[0]


def dotprod_f64(kind, a, b):
    L = lib()
    if kind == "rrrf":
        a, b = _as(a, np.float32), _as(b, np.float32)
        return L.yo_dotprod_rrrf_f64(_p(a), _p(b), len(a))
    y = np.zeros(1, np.complex128)
    if kind == "crc":
        a, b = _as(a, np.complex64), _as(b, np.float32)
        L.yo_dotprod_crc_f64(_p(a), _p(b), len(a), _p(y))
    elif kind == "rcc":
        a, b = _as(a, np.float32), _as(b, np.complex64)
        L.yo_dotprod_crc_f64(_p(b), _p(a), len(a), _p(y))
    else:
        a, b = _as(a, np.complex64), _as(b, np.complex64)
        L.yo_dotprod_ccc_f64(_p(a), _p(b), len(a), _p(y))
    return y[0]


class Window:
    """Restatement of buffer::Window<T> (window.rs:4-92); T = f32 or Complex<f32>."""

    def __init__(self, n, dtype=np.float32, _h=None):
        self.dtype = np.dtype(dtype)
        self.L = lib()
        self.h = _h if _h is not None else self.L.yo_window_create(n, self.dtype.itemsize)
        if not self.h:
            raise ValueError("window size must be greater than zero")

    def __del__(self):
        if getattr(self, "h", None):
            self.L.yo_window_destroy(self.h)
            self.h = None

    def _len(self):
        # struct layout: v, esz, len, ...
        return C.cast(self.h, C.POINTER(C.c_size_t))[2]

    def read(self):
        n = self._len()
        ptr = self.L.yo_window_read(self.h)
        buf = (C.c_char * (n * self.dtype.itemsize)).from_address(ptr)
        return np.frombuffer(buf, dtype=self.dtype).copy()

    def push(self, v):
        a = np.asarray([v], dtype=self.dtype)
        self.L.yo_window_push(self.h, _p(a))

    def write(self, vals):
        a = _as(vals, self.dtype)
        self.L.yo_window_write(self.h, _p(a), len(a))

    def index(self, i):
        out = np.zeros(1, self.dtype)
        if self.L.yo_window_index(self.h, i, _p(out)):
            raise IndexError("index value out of range")
        return out[0]

    def reset(self):
        self.L.yo_window_reset(self.h)

    def resize(self, n):
        if self.L.yo_window_resize(self.h, n):
            raise ValueError("window size must be greater than zero")

    def clone(self):
        return Window(0, self.dtype, _h=self.L.yo_window_clone(self.h))


class _Obj:
    fam = None

    def __del__(self):
        if getattr(self, "h", None):
            getattr(self.L, f"yo_{self.fam}_{self.kind}_destroy")(self.h)
            self.h = None

    def _f(self, name):
        return getattr(self.L, f"yo_{self.fam}_{self.kind}_{name}")

    def reset(self):
        self._f("reset")(self.h)

    def set_scale(self, s):
        self._f("set_scale")(self.h, _scalar(self.kind, s, coeff=True))


class FirFilter(_Obj):
    """Restatement of FirFilter<T,Coeff> (firfilt.rs:10-15,63-79,209-278)."""
    fam = "firfilt"

    def __init__(self, kind, h, _h=None):
        self.kind, self.L = kind, lib()
        self.tdt, self.cdt, _ = KINDS[kind]
        if _h is not None:
            self.h = _h
            return
        h = _as(h, self.cdt)
        self.h = self._f("create")(_p(h), len(h))
        if not self.h:
            raise ValueError("filter length must be greater than zero")

    def clone(self):
        return FirFilter(self.kind, None, _h=self._f("clone")(self.h))

    def push(self, x):
        self._f("push")(self.h, _scalar(self.kind, x))

    def execute(self):
        return _from(self.kind, self._f("execute")(self.h))

    def execute_one(self, x):
        self.push(x)
        return self.execute()

    def execute_block(self, x):
        x = _as(x, self.tdt)
        y = np.empty_like(x)
        rc = self._f("execute_block")(self.h, _p(x), len(x), _p(y), len(y))
        assert rc == 0
        return y


class FirDecimationFilter(_Obj):
    """Restatement of FirDecimationFilter<T,Coeff> (firdecim.rs:38-57,179-205)."""
    fam = "firdecim"

    def __init__(self, kind, M, h):
        self.kind, self.L, self.M = kind, lib(), M
        self.tdt, self.cdt, _ = KINDS[kind]
        h = _as(h, self.cdt)
        self.h = self._f("create")(M, _p(h), len(h))
        if not self.h:
            raise ValueError("config")

    def execute(self, x):
        x = _as(x, self.tdt)
        assert len(x) >= self.M
        return _from(self.kind, self._f("execute")(self.h, _p(x)))

    def execute_block(self, x, n):
        x = _as(x, self.tdt)
        assert len(x) >= n * self.M
        y = np.empty(n, self.tdt)
        self._f("execute_block")(self.h, _p(x), n, _p(y))
        return y


class FirPfbFilter(_Obj):
    """Restatement of FirPfbFilter<T,Coeff> (firpfb.rs:34-65,255-301)."""
    fam = "firpfb"

    def __init__(self, kind, num_filters, h, h_len=None):
        self.kind, self.L = kind, lib()
        self.tdt, self.cdt, _ = KINDS[kind]
        h = _as(h, self.cdt)
        self.h = self._f("create")(num_filters, _p(h), len(h) if h_len is None else h_len)
        if not self.h:
            raise ValueError("config")

    def push(self, x):
        self._f("push")(self.h, _scalar(self.kind, x))

    def write(self, xs):
        for v in xs:
            self.push(v)

    def execute(self, i):
        y = np.zeros(1, self.tdt)
        if self._f("execute")(self.h, i, _p(y)):
            raise ValueError("filterbank index exceeds maximum")
        return y[0]

    def execute_block(self, i, x):
        x = _as(x, self.tdt)
        y = np.empty_like(x)
        if self._f("execute_block")(self.h, i, _p(x), len(x), _p(y)):
            raise ValueError("filterbank index exceeds maximum")
        return y


class FirInterpolationFilter:
    """Restatement of FirInterpolationFilter<T,Coeff> (firinterp.rs:36-60,177-253): zero-pad the taps
    to a multiple of interp, build a FirPfbFilter, execute = push then every branch in order."""

    def __init__(self, kind, interp, h, h_len=None):
        cdt = KINDS[kind][1]
        h = _as(h, cdt)
        h_len = len(h) if h_len is None else h_len
        if interp < 2 or h_len < interp:
            raise ValueError("config")
        sub = 0
        while interp * sub < h_len:
            sub += 1
        hp = np.zeros(interp * sub, cdt)
        hp[:h_len] = h[:h_len]
        self.kind, self.interp, self.h_sub_len = kind, interp, sub
        self.bank = FirPfbFilter(kind, interp, hp, len(hp))

    def set_scale(self, s):
        self.bank.set_scale(s)

    def reset(self):
        self.bank.reset()

    def execute(self, x):
        self.bank.push(x)
        return np.array([self.bank.execute(i) for i in range(self.interp)], dtype=self.bank.tdt)

    def execute_block(self, xs):
        return np.concatenate([self.execute(v) for v in xs])

    def flush(self):
        return self.execute(0.0)


class Rresamp:
    """Restatement of Rresamp<T,Coeff> (rresamp.rs:28-183): a FirPfbFilter of `interp` branches and 2m taps each;
    per primitive block push Q samples and, after each push, emit while index < P { branch index; index += Q },
    then index -= P (execute_primitive, :162-183).  execute_block advances by Q*block_len / P*block_len per
    execute (liquid-dsp's behaviour; the reference slices by Q / P there and panics for block_len > 1)."""

    def __init__(self, kind, interp, decim, m, h):
        if interp == 0 or decim == 0 or m == 0:
            raise ValueError("config")
        cdt = KINDS[kind][1]
        h = _as(h, cdt)
        self.kind, self.p, self.q, self.m, self.block_len = kind, interp, decim, m, 1
        self.pfb = FirPfbFilter(kind, interp, h, 2 * interp * m)

    @classmethod
    def new_kaiser(cls, kind, interp, decim, m, bw, as_):                    # :59-82
        g = int(np.gcd(interp, decim))
        interp, decim = interp // g, decim // g
        if bw < 0:
            bw = 0.5 if interp > decim else np.float32(0.5) * np.float32(interp) / np.float32(decim)
        elif bw > 0.5:
            raise ValueError("config")
        bw = np.float32(bw)
        hf = fir_design_kaiser(2 * interp * m + 1, float(bw / np.float32(interp)), as_, 0.0)
        q = cls(kind, interp, decim, m, hf.astype(KINDS[kind][1]))
        q.set_scale(np.float32(2.0) * bw * np.sqrt(np.float32(decim) / np.float32(interp)))
        q.block_len = g
        return q

    @classmethod
    def new_default(cls, kind, interp, decim):                               # :99-104
        return cls.new_kaiser(kind, interp, decim, 12, 0.5, 60.0)

    def set_scale(self, s):
        self.pfb.set_scale(s)

    def reset(self):
        self.pfb.reset()

    def write(self, buf):
        self.pfb.write(buf)

    def _primitive(self, x):
        y, index = [], 0
        for i in range(self.q):
            self.pfb.push(x[i])
            while index < self.p:
                y.append(self.pfb.execute(index))
                index += self.q
            index -= self.p
        assert index == 0 and len(y) == self.p
        return y

    def execute(self, x):
        y = []
        for i in range(self.block_len):
            y += self._primitive(x[i * self.q:(i + 1) * self.q])
        return np.array(y, dtype=self.pfb.tdt)

    def execute_block(self, x, n):
        step = self.q * self.block_len
        return np.concatenate([self.execute(x[i * step:(i + 1) * step]) for i in range(n)])


def fir_block_f64(kind, h, x, M=1, n=None, scale=1.0):
    """f64 truth: y[i] = scale * sum_k h[k] x[i*M-k], zero history."""
    tdt, cdt, code = KINDS[kind]
    h, x = _as(h, cdt), _as(x, tdt)
    if n is None:
        n = len(x) // M
    s = _as([scale], cdt)
    y = np.zeros(n, np.float64 if kind == "rrrf" else np.complex128)
    lib().yo_fir_block_f64(code, _p(h), len(h), _p(s), _p(x), len(x), M, n, _p(y))
    return y


def dft_f64(x, backward=False):
    x = _as(x, np.complex64)
    y = np.zeros(len(x), np.complex128)
    lib().yo_dft_f64(_p(x), len(x), 1 if backward else 0, _p(y))
    return y


def fft_shift(v):
    v = _as(v, np.complex64).copy()
    lib().yo_fft_shift(_p(v), len(v))
    return v


class FftPlanF32:
    """f32 radix-4 power-of-two FFT (timed CPU baseline; fft/mod.rs:39-48 semantics)."""

    def __init__(self, n, backward=False):
        self.L, self.n = lib(), n
        self.h = self.L.yo_fft_plan_create(n, 1 if backward else 0)
        if not self.h:
            raise ValueError("power of two only")

    def __del__(self):
        if getattr(self, "h", None):
            self.L.yo_fft_plan_destroy(self.h)
            self.h = None

    def run(self, x):
        x = _as(x, np.complex64)
        y = np.empty_like(x)
        for f in range(len(x) // self.n):
            self.L.yo_fft_run_f32(self.h, _p(x[f * self.n:]), _p(y[f * self.n:]))
        return y


def fir_design_kaiser(n, fc, as_, mu=0.0):
    h = np.zeros(max(n, 1), np.float32)
    if lib().yo_fir_design_kaiser(n, fc, as_, mu, _p(h)):
        raise ValueError("config")
    return h[:n]


def fir_design_notch(m, f0, as_):
    """design/mod.rs:336-378"""
    h = np.zeros(2 * max(m, 1) + 1, np.float32)
    if lib().yo_fir_design_notch(m, f0, as_, _p(h)):
        raise ValueError("config")
    return h[: 2 * m + 1]


def notch_taps(kind, m, as_, f0):
    """ComplexNotch (firfilt.rs:17-44): real taps = notch at +-f0; complex taps = DC blocker mixed to f0"""
    if kind != "cccf":
        return fir_design_notch(m, f0, as_)
    h = fir_design_notch(m, 0.0, as_)
    phi = (np.float32(2.0 * np.pi) * np.float32(f0) * (np.arange(2 * m + 1, dtype=np.float32) - np.float32(m))).astype(np.float32)
    return (h * (np.cos(phi) + 1j * np.sin(phi))).astype(np.complex64)


def gen_real(seed, n, first=0):
    x = np.empty(n, np.float32)
    lib().yo_gen_real(seed, first, n, _p(x))
    return x


def gen_complex(seed, n, first=0):
    x = np.empty(n, np.complex64)
    lib().yo_gen_complex(seed, first, n, _p(x))
    return x


class FirPfbCh:
    """firpfbch analyzer restatement (PARITY UNPINNED: absent from the reference)."""

    def __init__(self, M, p, h):
        self.L, self.M, self.p = lib(), M, p
        h = _as(h, np.float32)
        assert len(h) >= M * p
        self._taps = h.copy()
        self.h = self.L.yo_firpfbch_create(M, p, _p(h))

    def __del__(self):
        if getattr(self, "h", None):
            self.L.yo_firpfbch_destroy(self.h)
            self.h = None

    def analyzer_execute(self, x):
        x = _as(x, np.complex64)
        nf = len(x) // self.M
        y = np.empty(nf * self.M, np.complex64)
        self.L.yo_firpfbch_analyzer_execute(self.h, _p(x), nf, _p(y))
        return y.reshape(nf, self.M)

    def synthesizer_execute(self, X):
        """liquid-dsp firpfbch_crcf_synthesizer_execute, frame by frame (PARITY UNPINNED): inverse DFT of the M channel
        samples (unnormalised), branch i pushes v[i] into its window and y[i] = sum_n h[i + n M] * window_i[newest - n].
        Sequential f32 accumulation in tap order, like the dot products of the reference's FirPfbFilter."""
        M, p = self.M, self.p
        X = _as(X, np.complex64).reshape(-1, M)
        if not hasattr(self, "_syn_win"):
            self._syn_win = np.zeros((p, M), np.complex64)          # row 0 = newest inverse transform
            self._syn_h = np.asarray(self._taps, np.float32)[: M * p].reshape(p, M)   # [n][i] = h[i + n M]
        y = np.empty(X.shape, np.complex64)
        for f in range(X.shape[0]):
            v = (np.fft.ifft(X[f].astype(np.complex128)) * M).astype(np.complex64)
            self._syn_win = np.roll(self._syn_win, 1, axis=0)
            self._syn_win[0] = v
            acc = np.zeros(M, np.complex64)
            for n in range(p):
                acc = (acc + self._syn_h[n] * self._syn_win[n]).astype(np.complex64)
            y[f] = acc
        return y.reshape(-1)


class FirPfbCh2:
    """firpfbch2 analyzer restatement (PARITY UNPINNED: absent from the reference)."""

    def __init__(self, M, m, h):
        self.L, self.M, self.m = lib(), M, m
        h = _as(h, np.float32)
        assert len(h) >= 2 * M * m
        self._taps = h.copy()
        self.h = self.L.yo_firpfbch2_create(M, m, _p(h))
        if not self.h:
            raise ValueError("config")

    def __del__(self):
        if getattr(self, "h", None):
            self.L.yo_firpfbch2_destroy(self.h)
            self.h = None

    def analyzer_execute(self, x):
        x = _as(x, np.complex64)
        ns = len(x) // (self.M // 2)
        y = np.empty(ns * self.M, np.complex64)
        self.L.yo_firpfbch2_analyzer_execute(self.h, _p(x), ns, _p(y))
        return y.reshape(ns, self.M)

    def synthesizer_execute(self, X):
        """liquid-dsp firpfbch2_crcf_execute_synthesizer, step by step (PARITY UNPINNED): inverse DFT of the M channel
        samples (unnormalised, then / 2 so that analyzer -> synthesizer has unit gain), pushed into window set w1 on
        even steps and w0 on odd steps; output i < M/2 reads windows b = i (even) or i + M/2 (odd) of both sets, the
        freshly fed set going to sub-filter i, the other to sub-filter i + M/2.  Pinned by the reconstruction test."""
        M, m = self.M, self.m
        M2, p = M // 2, 2 * m
        X = _as(X, np.complex64).reshape(-1, M)
        if not hasattr(self, "_syn"):
            hs = np.asarray(self._taps, np.float32)[: M * p].reshape(p, M)          # hs[n][i] = h[i + n M]
            self._syn = dict(w0=np.zeros((M, p), np.complex64), w1=np.zeros((M, p), np.complex64), flag=0, hs=hs)
        st = self._syn
        out = np.empty((X.shape[0], M2), np.complex64)
        for s in range(X.shape[0]):
            v = (np.fft.ifft(X[s].astype(np.complex128)) * (M / 2.0)).astype(np.complex64)
            buf = st["w1"] if st["flag"] == 0 else st["w0"]
            buf[:, 1:] = buf[:, :-1].copy()
            buf[:, 0] = v
            for i in range(M2):
                b = i if st["flag"] == 0 else i + M2
                r0, r1 = st["w0"][b], st["w1"][b]
                p0, p1 = (r0, r1) if st["flag"] else (r1, r0)
                acc = np.complex64(0)
                for n in range(p):                       # f32, fresh-set tap then other-set tap: the device order
                    acc = np.complex64(acc + st["hs"][n, i] * p0[n])
                    acc = np.complex64(acc + st["hs"][n, i + M2] * p1[n])
                out[s, i] = acc
            st["flag"] = 1 - st["flag"]
        return out.reshape(-1)


class FftFilt:
    """Restatement of FftFilt<T,Coeff> (fftfilt.rs:46-142); the two FFTs are the f64 definition."""

    def __init__(self, kind, h, n):
        self.kind, self.L, self.n = kind, lib(), n
        self.tdt, self.cdt, code = KINDS[kind]
        h = _as(h, self.cdt)
        self.h = self.L.yo_fftfilt_create(code, _p(h), len(h), n)
        if not self.h:
            raise ValueError("config")

    def __del__(self):
        if getattr(self, "h", None):
            self.L.yo_fftfilt_destroy(self.h)
            self.h = None

    def reset(self):
        self.L.yo_fftfilt_reset(self.h)

    def set_scale(self, s):
        s = complex(s)
        self.L.yo_fftfilt_set_scale(self.h, s.real, s.imag)

    def execute(self, x):
        x = _as(x, self.tdt)
        assert len(x) == self.n
        y = np.empty_like(x)
        self.L.yo_fftfilt_execute(self.h, _p(x), _p(y))
        return y


def window_fn(wtype, i, wlen, arg=0.0):
    return np.float32(lib().yo_window_fn(int(wtype), i, wlen, arg))


class Spgram:
    """Restatement of fft::Spgram<T> (spgram.rs:49-330).  The FFT is numpy's f64 transform rounded to
    f32 (the reference's is rustfft f32); everything else follows the reference statement by statement,
    including the 0.0 linear scale when alpha != -1 (spgram.rs:295-299)."""
    PSD_MIN = np.float32(1e-12)

    def __init__(self, nfft, wtype, window_len, delay, dtype=np.complex64):
        if nfft < 2 or window_len > nfft or window_len == 0 or delay == 0:
            raise ValueError("config")
        if int(wtype) == 5 and window_len % 2:
            raise ValueError("config")
        if not 1 <= int(wtype) <= 9:
            raise ValueError("config")
        self.nfft, self.wtype, self.window_len, self.delay = nfft, int(wtype), window_len, delay
        self.dtype = np.dtype(dtype)
        self.buffer = Window(window_len, self.dtype)
        arg = {5: 10.0, 9: 3.0, 7: float(window_len), 8: float(window_len // 3)}.get(self.wtype, 0.0)
        w = np.array([window_fn(self.wtype, i, window_len, arg) for i in range(window_len)], np.float32)
        if np.any(np.isnan(w)):
            raise ValueError("value")
        g = np.float32(0)
        for v in w:
            g = np.float32(g + v * v)
        self.w = (np.float32(1.0) / np.sqrt(g)) * w
        self.set_alpha(-1.0)
        self.reset()

    @classmethod
    def default(cls, nfft, dtype=np.complex64):
        if nfft < 2:
            raise ValueError("config")
        return cls(nfft, 5, nfft // 2, nfft // 4, dtype)

    def clear(self):
        self.sample_timer = self.delay
        self.num_transforms = 0
        self.num_samples = 0
        self.psd = np.zeros(self.nfft, np.float32)

    def reset(self):
        self.clear()
        self.buffer.reset()
        self.num_samples_total = 0
        self.num_transforms_total = 0

    def set_alpha(self, a):
        if a != -1.0 and not 0.0 <= a <= 1.0:
            raise ValueError("config")
        self.accumulate = a == -1.0
        self.alpha = np.float32(1.0 if self.accumulate else a)
        self.gamma = np.float32(1.0 if self.accumulate else 1.0 - np.float32(a))

    def push(self, x):
        self.buffer.push(x)
        self.num_samples += 1
        self.num_samples_total += 1
        self.sample_timer -= 1
        if self.sample_timer == 0:
            self.sample_timer = self.delay
            self.step()

    def write(self, xs):
        for v in xs:
            self.push(v)

    def step(self):
        rc = self.buffer.read()
        t = np.zeros(self.nfft, np.complex64)
        t[: self.window_len] = (rc * self.w).astype(np.complex64)
        f = np.fft.fft(t.astype(np.complex128)).astype(np.complex64)
        mag = (f.real * f.real + f.imag * f.imag).astype(np.float32)
        self.psd = mag if self.num_transforms == 0 else (self.gamma * self.psd + self.alpha * mag).astype(np.float32)
        self.num_transforms += 1
        self.num_transforms_total += 1

    def get_psd_mag(self):
        scale = np.float32(1.0 / max(1, self.num_transforms)) if self.accumulate else np.float32(0.0)
        k = (np.arange(self.nfft) + self.nfft // 2) % self.nfft
        return (np.maximum(self.psd[k], self.PSD_MIN) * scale).astype(np.float32)

    def get_psd(self):
        with np.errstate(divide="ignore"):
            return (10.0 * np.log10(self.get_psd_mag())).astype(np.float32)

    @classmethod
    def estimate_psd(cls, nfft, x, dtype=np.complex64):
        q = cls.default(nfft, dtype)
        q.write(x)
        if q.num_transforms == 0:
            q.step()
        return q.get_psd()


def halfband_kaiser(m, as_):
    """a half-band prototype hf[4m+1] for Resamp2 (the reference designs its own with Parks-McClellan,
    fir_design_pm_halfband_stopband_attenuation -- design code, out of scope): Kaiser-windowed sinc at fc = 0.25.
    fir_design_kaiser returns sinc(2 fc t) w(t) (centre tap 1, DC gain 1 / (2 fc) = 2); a half-band prototype has
    centre tap 1/2 and unit DC gain, hence the factor 0.5 (exact in f32).  Only the odd-offset taps are ever used
    (resamp2.rs:66-70); the centre is implied by the delay branch (:113,140)."""
    return (np.float32(0.5) * fir_design_kaiser(4 * m + 1, 0.25, as_)).astype(np.float32)


class Resamp2:
    """resamp2.rs:26-174 over a given half-band prototype hf[4m+1]"""
    MODES = {"filter": 0, "analyzer": 1, "synthesizer": 2, "decim": 3, "interp": 4}

    def __init__(self, kind, hf, m, f0=0.0, _h=None):
        self.kind, self.m = kind, int(m)
        self.T = KINDS[kind][0]
        if _h is not None:
            self.h = _h
            return
        hf = _as(hf, np.float32)
        if len(hf) != 4 * m + 1:
            raise ValueError("config")
        self.h = getattr(lib(), f"yo_resamp2_{kind}_create")(_p(hf), m, f0)
        if not self.h:
            raise ValueError("config")

    def __del__(self):
        try:
            if getattr(self, "h", None):
                getattr(lib(), f"yo_resamp2_{self.kind}_destroy")(self.h)
        except Exception:              # interpreter shutting down
            pass

    def clone(self):
        return Resamp2(self.kind, None, self.m, _h=getattr(lib(), f"yo_resamp2_{self.kind}_clone")(self.h))

    def reset(self):
        getattr(lib(), f"yo_resamp2_{self.kind}_reset")(self.h)

    def set_scale(self, s):
        getattr(lib(), f"yo_resamp2_{self.kind}_set_scale")(self.h, _scalar(self.kind, s, coeff=True))

    def get_delay(self):
        return 2 * self.m - 1

    def execute_block(self, mode, x):
        """mode: filter (n -> 2n, (y0,y1) pairs), analyzer / synthesizer (n pairs -> n pairs), decim (2n -> n),
        interp (n -> 2n)"""
        x = _as(x, self.T)
        md = self.MODES[mode]
        n = len(x) if md in (0, 4) else len(x) // 2
        y = np.empty(n if md == 3 else 2 * n, self.T)
        getattr(lib(), f"yo_resamp2_{self.kind}_execute_block")(self.h, md, _p(x), n, _p(y))
        return y


def msresamp2_stage_lengths(num_stages, fc, as_):
    """msresamp2.rs:70-88: semi-length of every half-band stage"""
    m = np.zeros(max(num_stages, 1), np.uint64)
    if lib().yo_msresamp2_stage_lengths(num_stages, fc, as_, _p(m)):
        raise ValueError("config")
    return [int(v) for v in m[:num_stages]]


class MsResamp2:
    """msresamp2.rs:8-198 with Kaiser half-band stages (stage attenuation as_ + 5, :69)"""

    def __init__(self, kind, interp, num_stages, fc, f0, as_):
        if f0 != 0.0:
            raise ValueError("config")
        self.kind, self.interp, self.num_stages, self.rate = kind, bool(interp), num_stages, 1 << num_stages
        self.T = KINDS[kind][0]
        self.m_stage = msresamp2_stage_lengths(num_stages, fc, as_)
        hf = np.concatenate([halfband_kaiser(m, as_ + 5.0) for m in self.m_stage]) if num_stages else np.zeros(1, np.float32)
        ms = np.array(self.m_stage or [0], np.uint64)
        zeta = _scalar(kind, 1.0 / self.rate, coeff=True)
        self.h = getattr(lib(), f"yo_msresamp2_{kind}_create")(int(self.interp), num_stages, _p(ms), _p(_as(hf, np.float32)), zeta)
        if not self.h:
            raise ValueError("config")

    def __del__(self):
        try:
            if getattr(self, "h", None):
                getattr(lib(), f"yo_msresamp2_{self.kind}_destroy")(self.h)
        except Exception:
            pass

    def get_delay(self):                                          # :118-135
        d = 0.0
        if self.interp:
            for i in range(self.num_stages):
                d = d * 0.5 + self.m_stage[self.num_stages - i - 1]
        else:
            for i in range(self.num_stages):
                d = d * 2.0 + 2.0 * self.m_stage[i] - 1.0
        return d

    def execute_block(self, x):
        x = _as(x, self.T)
        n = len(x) if self.interp else len(x) // self.rate
        y = np.empty(n * self.rate if self.interp else n, self.T)
        getattr(lib(), f"yo_msresamp2_{self.kind}_execute_block")(self.h, _p(x), n, _p(y))
        return y


def stream_fir_fft(h, scale, x, nfft):
    """headline composition: firfilt_crcf.execute_block -> nfft frames -> forward f32 FFT."""
    L = lib()
    q = FirFilter("crcf", h)
    q.set_scale(scale)
    plan = FftPlanF32(nfft)
    x = _as(x, np.complex64)
    nf = len(x) // nfft
    scratch = np.empty(nfft, np.complex64)
    out = np.empty(nf * nfft, np.complex64)
    L.yo_stream_fir_fft(q.h, plan.h, _p(x), nf, _p(scratch), _p(out))
    return out.reshape(nf, nfft)
