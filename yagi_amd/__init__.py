"""yagi_amd -- Python host mirror of yagi's FIR/FFT object API over libyagi_hip.so (MI355X).

The classes keep the reference's names, argument meaning and error behaviour
(reference = EEGKit/yagi, paths relative to its root):

    DotProd / dotprod          src/dotprod/mod.rs:13-73
    FirFilter                  src/filter/fir/firfilt.rs:10-15,63-314
    FirDecimationFilter        src/filter/fir/firdecim.rs:12-17,38-205
    FirPfbFilter               src/filter/fir/firpfb.rs:10-15,34-301
    Fft, Direction, fft_run    src/fft/mod.rs:13-69
    Error variants             src/error.rs:4-14
    FirPfbCh / FirPfbCh2       (absent from the reference: src/multichannel/mod.rs is empty)
    FirFftStream               the headline composition FirFilter::execute_block -> Fft::run

Generic parameters <T, Coeff> are spelled with liquid-dsp's suffixes: "rrrf" = <f32,f32>,
"crcf" = <Complex32,f32>, "cccf" = <Complex32,Complex32>.  Every numeric result comes from a HIP
kernel through the C ABI (include/yagi_hip.h); importing this package fails if the library is
not built, and every call fails with DeviceError if no GPU is present.
"""
import ctypes as C
import enum

import numpy as np

from . import _capi
from ._capi import cf32, lib

__all__ = [
    "YagiError", "InternalError", "ConfigError", "ValueError_", "RangeError", "ModeError",
    "NoConvergenceError", "DeviceError", "Direction", "dotprod", "FirFilter", "FirDecimationFilter",
    "FirPfbFilter", "FirInterpolationFilter", "Rresamp", "FftFilt", "Fft", "fft_run", "Spgram", "WindowType", "FirFftStream", "FirPfbCh", "FirPfbCh2", "DeviceArray",
    "fir_design_kaiser", "device_count", "synchronize", "gen_complex_dev", "gen_real_dev",
]


# ---- error::Error (src/error.rs:7-14) ---------------------------------------------------------
class YagiError(Exception):
    pass


class InternalError(YagiError):
    pass


class ConfigError(YagiError):
    pass


class ValueError_(YagiError):
    pass


class RangeError(YagiError):
    pass


class ModeError(YagiError):
    pass


class NoConvergenceError(YagiError):
    pass


class DeviceError(YagiError):
    pass


_ERR = {1: InternalError, 2: ConfigError, 3: ValueError_, 4: RangeError, 5: ModeError,
        6: NoConvergenceError, 7: DeviceError}


def _check(rc):
    if rc != 0:
        raise _ERR.get(rc, YagiError)(lib.yagi_hip_last_error().decode())


class Direction(enum.Enum):
    """fft::Direction (src/fft/mod.rs:13-17)"""
    Forward = 0
    Backward = 1


KINDS = {"rrrf": (np.float32, np.float32), "crcf": (np.complex64, np.float32),
         "cccf": (np.complex64, np.complex64)}


def _arr(a, dt):
    return np.ascontiguousarray(np.asarray(a, dtype=dt))


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p) if isinstance(a, np.ndarray) else C.c_void_p(a)


def _out(y, n, dt):
    """a caller-supplied output array handed to the C ABI as a raw pointer: it must be exactly what the library
    will write -- dtype dt, C-contiguous, n elements -- or a fresh array when None"""
    if y is None:
        return np.empty(n, dt)
    if not (isinstance(y, np.ndarray) and y.dtype == np.dtype(dt) and y.flags.c_contiguous and y.size == n):
        raise ConfigError(f"output must be a C-contiguous {np.dtype(dt).name} array of {n} elements")
    return y


def _byval(v, ctype):
    if ctype is cf32:
        v = complex(v)
        return cf32(v.real, v.imag)
    return C.c_float(float(np.real(v)))


def _devptr(x):
    """device pointer from a DeviceArray, a torch tensor (data_ptr) or a raw int."""
    if isinstance(x, DeviceArray):
        return C.c_void_p(x.ptr)
    if hasattr(x, "data_ptr"):
        return C.c_void_p(x.data_ptr())
    return C.c_void_p(int(x))


# ---- device plumbing --------------------------------------------------------------------------
def device_count():
    n = C.c_int(0)
    rc = lib.yagi_hip_device_count(C.byref(n))
    return n.value if rc == 0 else 0


def synchronize():
    _check(lib.yagi_hip_device_synchronize())


class DeviceArray:
    """A typed HBM allocation owned through yagi_hip_malloc/free."""

    def __init__(self, n, dtype):
        self.dtype = np.dtype(dtype)
        self.n = int(n)
        p = C.c_void_p()
        _check(lib.yagi_hip_malloc(C.byref(p), self.n * self.dtype.itemsize))
        self.ptr = p.value

    @classmethod
    def from_numpy(cls, a):
        a = np.ascontiguousarray(a)
        d = cls(a.size, a.dtype)
        _check(lib.yagi_hip_memcpy_h2d(d.ptr, _ptr(a), a.nbytes))
        return d

    def to_numpy(self, n=None, offset=0):
        n = self.n - offset if n is None else n
        out = np.empty(n, self.dtype)
        _check(lib.yagi_hip_memcpy_d2h(_ptr(out), C.c_void_p(self.ptr + offset * self.dtype.itemsize), out.nbytes))
        return out

    def zero(self):
        _check(lib.yagi_hip_memset_dev(self.ptr, 0, self.n * self.dtype.itemsize))

    def free(self):
        if getattr(self, "ptr", None) and lib is not None:        # lib is None while the interpreter shuts down
            lib.yagi_hip_free(self.ptr)
            self.ptr = None

    def __del__(self):
        self.free()


def gen_complex_dev(seed, n, out=None, first=0, stream=None):
    out = out if out is not None else DeviceArray(n, np.complex64)
    _check(lib.yagi_hip_gen_complex_dev(seed, first, n, _devptr(out), stream))
    return out


def gen_real_dev(seed, n, out=None, first=0, stream=None):
    out = out if out is not None else DeviceArray(n, np.float32)
    _check(lib.yagi_hip_gen_real_dev(seed, first, n, _devptr(out), stream))
    return out


def fir_design_kaiser(n, fc, as_, mu=0.0):
    """filter::fir_design_kaiser (src/filter/fir/design/kaiser.rs:16-51)"""
    h = np.zeros(max(int(n), 1), np.float32)
    _check(lib.yagi_hip_fir_design_kaiser(n, fc, as_, mu, _ptr(h)))
    return h[:n]


# ---- dotprod (trait DotProd, src/dotprod/mod.rs:13-73) ----------------------------------------
def dotprod(a, b):
    """a.dotprod(b): the impl is picked from the element types like the Rust trait impls."""
    a, b = np.asarray(a), np.asarray(b)
    ca, cb = np.iscomplexobj(a), np.iscomplexobj(b)
    name = {(False, False): "rrrf", (False, True): "rccf", (True, False): "crcf", (True, True): "cccf"}[(ca, cb)]
    a = _arr(a, np.complex64 if ca else np.float32)
    b = _arr(b, np.complex64 if cb else np.float32)
    # zip() semantics: the shorter slice decides (slice impls, mod.rs:23-25)
    n = min(a.size, b.size)
    y = np.zeros(1, np.complex64 if (ca or cb) else np.float32)
    _check(getattr(lib, f"yagi_hip_dotprod_{name}")(_ptr(a), _ptr(b), n, _ptr(y)))
    return y[0]


# ---- handle base ------------------------------------------------------------------------------
class _Handle:
    _prefix = None

    def _fn(self, name):
        return getattr(lib, f"{self._prefix}{name}")

    def __del__(self):
        h = getattr(self, "_h", None)
        if h and lib is not None:                 # lib is None while the interpreter shuts down
            self._fn("destroy")(h)
            self._h = None

    def set_stream(self, stream):
        """stream: a hipStream_t as int (e.g. torch.cuda.current_stream().cuda_stream)"""
        _check(self._fn("set_stream")(self._h, stream))

    def reset(self):
        _check(self._fn("reset")(self._h))


class _FirBase(_Handle):
    def _init_kind(self, kind):
        if kind not in KINDS:
            raise ConfigError(f"unknown type combination {kind!r}")
        self.kind = kind
        self.T, self.Cdt = KINDS[kind]
        self._Tc, self._Cc = _capi.KIND_TYPES[kind]

    def set_scale(self, scale):
        _check(self._fn("set_scale")(self._h, _byval(scale, self._Cc)))

    def get_scale(self):
        s = np.zeros(1, self.Cdt)
        _check(self._fn("get_scale")(self._h, _ptr(s)))
        return s[0]

    def clone(self):
        new = object.__new__(type(self))
        new.__dict__.update({k: v for k, v in self.__dict__.items() if k != "_h"})
        h = C.c_void_p()
        _check(self._fn("clone")(self._h, C.byref(h)))
        new._h = h
        return new


class FirFilter(_FirBase):
    """FirFilter<T,Coeff> (src/filter/fir/firfilt.rs)."""

    def __init__(self, kind, h):
        self._init_kind(kind)
        self._prefix = f"yagi_hip_firfilt_{kind}_"
        h = _arr(h, self.Cdt)
        hd = C.c_void_p()
        _check(self._fn("create")(_ptr(h), h.size, C.byref(hd)))
        self._h = hd

    @classmethod
    def _from(cls, kind, creator, *args):
        self = object.__new__(cls)
        self._init_kind(kind)
        self._prefix = f"yagi_hip_firfilt_{kind}_"
        hd = C.c_void_p()
        _check(self._fn(creator)(*args, C.byref(hd)))
        self._h = hd
        return self

    @classmethod
    def new_kaiser(cls, kind, n, fc, as_, mu=0.0):          # firfilt.rs:93-97
        return cls._from(kind, "create_kaiser", n, fc, as_, mu)

    @classmethod
    def new_rect(cls, kind, n):                              # firfilt.rs:149-155
        return cls._from(kind, "create_rect", n)

    @classmethod
    def new_dc_blocker(cls, kind, m, as_):                   # firfilt.rs:166-170
        return cls._from(kind, "create_dc_blocker", m, as_)

    @classmethod
    def new_notch(cls, kind, m, as_, f0):                    # firfilt.rs:183-186
        return cls._from(kind, "create_notch", m, as_, f0)

    def set_coefficients(self, h):                           # :193-206
        h = _arr(h, self.Cdt)
        _check(self._fn("set_coefficients")(self._h, _ptr(h), h.size))

    def push(self, x):                                       # :220-223
        _check(self._fn("push")(self._h, _byval(x, self._Tc)))

    def write(self, x):                                      # :230-234
        x = _arr(x, self.T)
        _check(self._fn("write")(self._h, _ptr(x), x.size))

    def execute(self):                                       # :241-246
        y = np.zeros(1, self.T)
        _check(self._fn("execute")(self._h, _ptr(y)))
        return y[0]

    def execute_one(self, x):                                # :256-259
        y = np.zeros(1, self.T)
        _check(self._fn("execute_one")(self._h, _byval(x, self._Tc), _ptr(y)))
        return y[0]

    def execute_block(self, x, y=None):                      # :267-278
        x = _arr(x, self.T)
        if y is None:
            y = np.empty_like(x)
        elif y.dtype != self.T or not y.flags.c_contiguous:
            raise ConfigError("output must be a contiguous array of the sample type")
        _check(self._fn("execute_block")(self._h, _ptr(x), x.size, _ptr(y), y.size))
        return y

    def execute_block_dev(self, x_dev, n, y_dev):
        _check(self._fn("execute_block_dev")(self._h, _devptr(x_dev), n, _devptr(y_dev)))

    def set_pipeline(self, on=True):
        """pipelined block calls (include/yagi_hip.h): consecutive execute_block_dev calls overlap on two streams of the
        object; outputs (and the right to overwrite the inputs) are ordered on the object's stream after join()"""
        _check(self._fn("set_pipeline")(self._h, 1 if on else 0))

    def join(self):
        _check(self._fn("join")(self._h))

    def get_length(self):                                    # :301-303
        n = C.c_size_t()
        _check(self._fn("get_length")(self._h, C.byref(n)))
        return n.value

    def get_coefficients(self):                              # :310-312
        n = self.get_length()
        h = np.empty(n, self.Cdt)
        _check(self._fn("get_coefficients")(self._h, _ptr(h), n))
        return h

    def freqresponse(self, fc):                              # :325-328
        H = np.zeros(1, np.complex64)
        _check(self._fn("freqresponse")(self._h, fc, _ptr(H)))
        return H[0]

    def groupdelay(self, fc):                                # :339-342
        d = C.c_float()
        _check(self._fn("groupdelay")(self._h, fc, C.byref(d)))
        return d.value

    def set_kernel(self, choice):
        """0 auto, 1 general direct form, 4 overlap-save fast convolution; crcf also 2 register-sliding and
        3 MFMA Toeplitz direct forms (include/yagi_hip.h)."""
        _check(self._fn("set_kernel")(self._h, choice))


class FirDecimationFilter(_FirBase):
    """FirDecimationFilter<T,Coeff> (src/filter/fir/firdecim.rs)."""

    def __init__(self, kind, decimation_factor, h, h_len=None):
        self._init_kind(kind)
        self._prefix = f"yagi_hip_firdecim_{kind}_"
        h = _arr(h, self.Cdt)
        hd = C.c_void_p()
        _check(self._fn("create")(decimation_factor, _ptr(h), h.size if h_len is None else h_len, C.byref(hd)))
        self._h = hd

    @classmethod
    def new_kaiser(cls, kind, decimation_factor, m, as_):    # firdecim.rs:70-87
        self = object.__new__(cls)
        self._init_kind(kind)
        self._prefix = f"yagi_hip_firdecim_{kind}_"
        hd = C.c_void_p()
        _check(self._fn("create_kaiser")(decimation_factor, m, as_, C.byref(hd)))
        self._h = hd
        return self

    def get_decim_rate(self):                                # :130-132
        n = C.c_size_t()
        _check(self._fn("get_decim_rate")(self._h, C.byref(n)))
        return n.value

    def execute(self, x):                                    # :179-191
        x = _arr(x, self.T)
        y = np.zeros(1, self.T)
        _check(self._fn("execute")(self._h, _ptr(x), x.size, _ptr(y)))
        return y[0]

    def execute_block(self, x, n):                           # :200-205
        x = _arr(x, self.T)
        y = np.empty(n, self.T)
        _check(self._fn("execute_block")(self._h, _ptr(x), x.size, n, _ptr(y)))
        return y

    def execute_block_dev(self, x_dev, n, y_dev):
        _check(self._fn("execute_block_dev")(self._h, _devptr(x_dev), n, _devptr(y_dev)))

    def freqresp(self, fc):                                  # firdecim.rs:164-168
        H = np.zeros(1, np.complex64)
        _check(self._fn("freqresp")(self._h, fc, _ptr(H)))
        return H[0]


class FirPfbFilter(_FirBase):
    """FirPfbFilter<T,Coeff> (src/filter/fir/firpfb.rs)."""

    def __init__(self, kind, num_filters, h, h_len=None):
        self._init_kind(kind)
        self._prefix = f"yagi_hip_firpfb_{kind}_"
        h = _arr(h, self.Cdt)
        hd = C.c_void_p()
        _check(self._fn("create")(num_filters, _ptr(h), h.size if h_len is None else h_len, C.byref(hd)))
        self._h = hd
        self.num_filters = num_filters

    @classmethod
    def _from(cls, kind, num_filters, creator, *args):
        self = object.__new__(cls)
        self._init_kind(kind)
        self._prefix = f"yagi_hip_firpfb_{kind}_"
        hd = C.c_void_p()
        _check(self._fn(creator)(num_filters, *args, C.byref(hd)))
        self._h = hd
        self.num_filters = num_filters
        return self

    @classmethod
    def new_kaiser(cls, kind, num_filters, m, fc, as_):      # firpfb.rs:94-114
        return cls._from(kind, num_filters, "create_kaiser", m, fc, as_)

    @classmethod
    def default(cls, kind, num_filters, m):                  # firpfb.rs:79-81
        return cls._from(kind, num_filters, "create_default", m)

    def push(self, x):                                       # :255-257
        _check(self._fn("push")(self._h, _byval(x, self._Tc)))

    def write(self, x):                                      # :264-266
        x = _arr(x, self.T)
        _check(self._fn("write")(self._h, _ptr(x), x.size))

    def execute(self, i):                                    # :277-286
        y = np.zeros(1, self.T)
        _check(self._fn("execute")(self._h, i, _ptr(y)))
        return y[0]

    def execute_block(self, i, x):                           # :295-301
        x = _arr(x, self.T)
        y = np.empty_like(x)
        _check(self._fn("execute_block")(self._h, i, _ptr(x), x.size, _ptr(y), y.size))
        return y

    def execute_block_dev(self, i, x_dev, n, y_dev):
        _check(self._fn("execute_block_dev")(self._h, i, _devptr(x_dev), n, _devptr(y_dev)))

    def execute_all_dev(self, x_dev, n, y_dev):
        _check(self._fn("execute_all_dev")(self._h, _devptr(x_dev), n, _devptr(y_dev)))

    def execute_select_dev(self, idx_dev, x_dev, n, y_dev):
        _check(self._fn("execute_select_dev")(self._h, _devptr(idx_dev), _devptr(x_dev), n, _devptr(y_dev)))

    def execute_all(self, x):
        """host convenience over execute_all_dev: returns y[n, num_filters]"""
        x = _arr(x, self.T)
        dx = DeviceArray.from_numpy(x)
        dy = DeviceArray(x.size * self.num_filters, self.T)
        self.execute_all_dev(dx, x.size, dy)
        synchronize()
        return dy.to_numpy().reshape(x.size, self.num_filters)

    def execute_select(self, idx, x):
        x = _arr(x, self.T)
        idx = _arr(idx, np.uint32)
        if idx.size != x.size:
            raise ConfigError("index and sample blocks must have equal length")
        if idx.size and idx.max() >= self.num_filters:
            raise ConfigError(f"filterbank index ({idx.max()}) exceeds maximum ({self.num_filters})")
        dx, di = DeviceArray.from_numpy(x), DeviceArray.from_numpy(idx)
        dy = DeviceArray(x.size, self.T)
        self.execute_select_dev(di, dx, x.size, dy)
        synchronize()
        return dy.to_numpy()


class FirInterpolationFilter(_FirBase):
    """FirInterpolationFilter<T,Coeff> (src/filter/fir/firinterp.rs)."""

    def __init__(self, kind, interp, h, h_len=None):         # new(interp, h, h_len) :36-60
        self._init_kind(kind)
        self._prefix = f"yagi_hip_firinterp_{kind}_"
        h = _arr(h, self.Cdt)
        hd = C.c_void_p()
        _check(self._fn("create")(interp, _ptr(h), h.size if h_len is None else h_len, C.byref(hd)))
        self._h = hd

    @classmethod
    def _from(cls, kind, creator, *args):
        self = object.__new__(cls)
        self._init_kind(kind)
        self._prefix = f"yagi_hip_firinterp_{kind}_"
        hd = C.c_void_p()
        _check(self._fn(creator)(*args, C.byref(hd)))
        self._h = hd
        return self

    @classmethod
    def new_kaiser(cls, kind, interp, m, as_):               # :73-90
        return cls._from(kind, "create_kaiser", interp, m, as_)

    @classmethod
    def new_linear(cls, kind, interp):                       # :135-147
        return cls._from(kind, "create_linear", interp)

    @classmethod
    def new_window(cls, kind, interp, m):                    # :159-174
        return cls._from(kind, "create_window", interp, m)

    def get_interp_rate(self):                               # :186-188
        n = C.c_size_t()
        _check(self._fn("get_interp_rate")(self._h, C.byref(n)))
        return n.value

    def get_sub_len(self):                                   # :195-197
        n = C.c_size_t()
        _check(self._fn("get_sub_len")(self._h, C.byref(n)))
        return n.value

    def execute(self, x):                                    # :224-231 -> interp outputs
        y = np.empty(self.get_interp_rate(), self.T)
        _check(self._fn("execute")(self._h, _byval(x, self._Tc), _ptr(y), y.size))
        return y

    def execute_block(self, x):                              # :239-244 -> n*interp outputs
        x = _arr(x, self.T)
        y = np.empty(x.size * self.get_interp_rate(), self.T)
        _check(self._fn("execute_block")(self._h, _ptr(x), x.size, _ptr(y), y.size))
        return y

    def execute_block_dev(self, x_dev, n, y_dev):
        _check(self._fn("execute_block_dev")(self._h, _devptr(x_dev), n, _devptr(y_dev)))

    def flush(self):                                         # :251-253
        y = np.empty(self.get_interp_rate(), self.T)
        _check(self._fn("flush")(self._h, _ptr(y), y.size))
        return y


class Rresamp(_FirBase):
    """Rresamp<T,Coeff> (src/filter/resampler/rresamp.rs): rational-rate resampler, P outputs per Q inputs."""

    def __init__(self, kind, interp, decim, m, h):           # new(interp, decim, m, h) :28-57
        self._init_kind(kind)
        self._prefix = f"yagi_hip_rresamp_{kind}_"
        h = _arr(h, self.Cdt)
        hd = C.c_void_p()
        _check(self._fn("create")(interp, decim, m, _ptr(h), h.size, C.byref(hd)))
        self._h = hd

    @classmethod
    def _from(cls, kind, creator, *args):
        self = object.__new__(cls)
        self._init_kind(kind)
        self._prefix = f"yagi_hip_rresamp_{kind}_"
        hd = C.c_void_p()
        _check(self._fn(creator)(*args, C.byref(hd)))
        self._h = hd
        return self

    @classmethod
    def new_kaiser(cls, kind, interp, decim, m, bw, as_):    # :59-82
        return cls._from(kind, "create_kaiser", interp, decim, m, bw, as_)

    @classmethod
    def new_default(cls, kind, interp, decim):               # :99-104
        return cls._from(kind, "create_default", interp, decim)

    def _params(self):
        v = [C.c_size_t() for _ in range(4)]
        _check(self._fn("get_params")(self._h, *[C.byref(a) for a in v]))
        return tuple(a.value for a in v)                     # interp, decim, m, block_len

    def get_interp(self): return self._params()[0]           # :138-140
    def get_decim(self): return self._params()[1]            # :146-148
    def get_delay(self): return self._params()[2]            # :118-120
    def get_block_len(self): return self._params()[3]        # :122-124
    def get_rate(self):                                      # :126-128
        p, q, _, _ = self._params()
        return np.float32(p) / np.float32(q)
    def get_p(self): return self._params()[0] * self._params()[3]     # :130-132
    def get_q(self): return self._params()[1] * self._params()[3]     # :142-144

    def write(self, buf):                                    # :150-152
        buf = _arr(buf, self.T)
        _check(self._fn("write")(self._h, _ptr(buf), buf.size))

    def execute(self, x):                                    # :154-161  Q*block_len in -> P*block_len out
        x = _arr(x, self.T)
        y = np.empty(self.get_p(), self.T)
        _check(self._fn("execute")(self._h, _ptr(x), x.size, _ptr(y), y.size))
        return y

    def execute_block(self, x, n):                           # :163-170  n times execute
        x = _arr(x, self.T)
        y = np.empty(n * self.get_p(), self.T)
        _check(self._fn("execute_block")(self._h, _ptr(x), x.size, n, _ptr(y), y.size))
        return y

    def execute_block_dev(self, x_dev, n, y_dev):
        _check(self._fn("execute_block_dev")(self._h, _devptr(x_dev), n, _devptr(y_dev)))



class Resamp2(_FirBase):
    """Resamp2<T,Coeff> (src/filter/resampler/resamp2.rs): half-band filter / two-channel bank / x2 resampler.
    The reference designs its prototype with Parks-McClellan (design code, out of scope): `Resamp2(kind, hf, m, f0)`
    takes the designed prototype hf[4m+1]; `Resamp2.new(kind, m, f0, as_)` is the reference's constructor signature
    with a Kaiser-windowed half-band prototype."""
    FILTER, ANALYZER, SYNTHESIZER, DECIM, INTERP = range(5)

    def __init__(self, kind, hf, m, f0=0.0):                  # new() :44-88, from the designed prototype
        self._init_kind(kind)
        self._prefix = f"yagi_hip_resamp2_{kind}_"
        hf = _arr(hf, np.float32)
        if hf.size != 4 * m + 1:
            raise ConfigError("half-band prototype must hold 4*m+1 taps")
        hd = C.c_void_p()
        _check(self._fn("create")(_ptr(hf), m, f0, C.byref(hd)))
        self._h, self.m = hd, int(m)

    @classmethod
    def new(cls, kind, m, f0, as_):                           # Resamp2::new(m, f0, as_) :44
        """The reference's signature with a Kaiser-windowed half-band prototype.  NOT tap-compatible with the reference,
        whose prototype comes from its Parks-McClellan design code (resamp2.rs:58, out of scope): same structure, delay
        and stop-band specification, different tap values.  For the reference's taps pass them to the constructor."""
        self = object.__new__(cls)
        self._init_kind(kind)
        self._prefix = f"yagi_hip_resamp2_{kind}_"
        hd = C.c_void_p()
        _check(self._fn("create_kaiser")(m, f0, as_, C.byref(hd)))
        self._h, self.m = hd, int(m)
        return self

    def get_delay(self):                                      # :104-106
        d = C.c_size_t()
        _check(self._fn("get_delay")(self._h, C.byref(d)))
        return d.value

    @staticmethod
    def _out_count(mode, nx):
        return 2 * nx if mode in (0, 4) else nx // 2 if mode == 3 else nx

    def execute_block(self, mode, x):
        x = _arr(x, self.T)
        y = np.empty(self._out_count(mode, x.size), self.T)
        _check(self._fn("execute_block")(self._h, mode, _ptr(x), x.size, _ptr(y), y.size))
        return y

    def execute_block_dev(self, mode, x_dev, nx, y_dev, ny=None):
        """ny = samples the output buffer holds (default: what the form produces from nx)"""
        ny = self._out_count(mode, nx) if ny is None else ny
        _check(self._fn("execute_block_dev")(self._h, mode, _devptr(x_dev), nx, _devptr(y_dev), ny))

    # the reference's per-call forms
    def filter_execute(self, x):                              # :108-130 -> (y0, y1)
        y = self.execute_block(self.FILTER, np.array([x], self.T))
        return y[0], y[1]

    def analyzer_execute(self, x):                            # :132-143  x[2] -> y[2]
        return self.execute_block(self.ANALYZER, x)

    def synthesizer_execute(self, x):                         # :145-157  x[2] -> y[2]
        return self.execute_block(self.SYNTHESIZER, x)

    def decim_execute(self, x):                               # :159-169  x[2] -> y
        return self.execute_block(self.DECIM, x)[0]

    def interp_execute(self, x):                              # :171-180  x -> y[2]
        return self.execute_block(self.INTERP, np.array([x], self.T))


class MsResamp2(_FirBase):
    """MsResamp2<T,Coeff> (src/filter/resampler/msresamp2.rs): 2^num_stages interpolator / decimator, a chain of
    half-band stages (Kaiser prototypes; `from_taps` takes externally designed ones)."""
    DECIM, INTERP = 0, 1                                      # ResampType :27-31

    def __init__(self, kind, type_, num_stages, fc, f0, as_):  # new() :38-93
        self._init_kind(kind)
        self._prefix = f"yagi_hip_msresamp2_{kind}_"
        hd = C.c_void_p()
        _check(self._fn("create")(int(type_), num_stages, fc, f0, as_, C.byref(hd)))
        self._h = hd

    @classmethod
    def from_taps(cls, kind, type_, m_stage, hf_stages):
        self = object.__new__(cls)
        self._init_kind(kind)
        self._prefix = f"yagi_hip_msresamp2_{kind}_"
        ms = np.array(list(m_stage) or [0], np.uint64)
        hf = _arr(np.concatenate([np.asarray(h, np.float32) for h in hf_stages]) if len(m_stage) else np.zeros(1), np.float32)
        hd = C.c_void_p()
        _check(self._fn("create_taps")(int(type_), len(m_stage), _ptr(ms), _ptr(hf), C.byref(hd)))
        self._h = hd
        return self

    def _params(self):
        it, ns, d = C.c_int(), C.c_size_t(), C.c_float()
        ms = np.zeros(16, np.uint64)
        _check(self._fn("get_params")(self._h, C.byref(it), C.byref(ns), C.byref(d), _ptr(ms)))
        return it.value, ns.value, d.value, [int(v) for v in ms[: ns.value]]

    def get_type(self): return self._params()[0]              # :113-115
    def get_num_stages(self): return self._params()[1]        # :109-111
    def get_delay(self): return self._params()[2]             # :117-135
    def get_stage_lengths(self): return self._params()[3]
    def get_rate(self):                                       # :102-107
        it, ns, _, _ = self._params()
        return float(1 << ns) if it else 1.0 / (1 << ns)

    def set_scale(self, scale):
        raise ConfigError("MsResamp2 has no scale (msresamp2.rs)")

    get_scale = set_scale

    def execute_block(self, x, n=None):
        """n execute() calls (:137-152): interp: n inputs -> n*rate outputs; decim: n*rate inputs -> n outputs.
        x must hold exactly the n calls' input (the reference's copy_from_slice panics otherwise, :181)"""
        x = _arr(x, self.T)
        it, ns, _, _ = self._params()
        rate = 1 << ns
        n = (x.size if it else x.size // rate) if n is None else n
        y = np.empty(n * rate if it else n, self.T)
        _check(self._fn("execute_block")(self._h, _ptr(x), x.size, _ptr(y), y.size))
        return y

    def execute_block_dev(self, x_dev, n, y_dev, nx=None, ny=None):
        """n execute() calls on device buffers of nx / ny samples (defaults: exactly what n calls consume / produce)"""
        it, ns, _, _ = self._params()
        rate = 1 << ns
        nx = (n if it else n * rate) if nx is None else nx
        ny = (n * rate if it else n) if ny is None else ny
        _check(self._fn("execute_block_dev")(self._h, _devptr(x_dev), nx, _devptr(y_dev), ny))

    def execute(self, x):                                     # :137-152
        return self.execute_block(x, 1)


class FftFilt(_FirBase):
    """FftFilt<T,Coeff> (src/filter/fftfilt.rs): overlap-add fast convolution, block n, FFT size 2n."""

    def __init__(self, kind, h, n):                          # create(h, n) :46-84
        self._init_kind(kind)
        self._prefix = f"yagi_hip_fftfilt_{kind}_"
        h = _arr(h, self.Cdt)
        hd = C.c_void_p()
        _check(self._fn("create")(_ptr(h), h.size, n, C.byref(hd)))
        self._h = hd
        self.n = n

    create = classmethod(lambda cls, kind, h, n: cls(kind, h, n))

    def get_length(self):                                    # :140-142
        n = C.c_size_t()
        _check(self._fn("get_length")(self._h, C.byref(n)))
        return n.value

    def execute(self, x, y=None):                            # :103-138
        x = _arr(x, self.T)
        y = _out(y, self.n, self.T)
        _check(self._fn("execute")(self._h, _ptr(x), x.size, _ptr(y), y.size))
        return y

    def execute_blocks(self, x):
        """consecutive execute() calls over len(x)/n blocks as one device batch"""
        x = _arr(x, self.T)
        if x.size % self.n:
            raise ConfigError("input must hold a whole number of blocks")
        y = np.empty_like(x)
        _check(self._fn("execute_blocks")(self._h, _ptr(x), x.size // self.n, _ptr(y)))
        return y

    def execute_blocks_dev(self, x_dev, nblocks, y_dev):
        _check(self._fn("execute_blocks_dev")(self._h, _devptr(x_dev), nblocks, _devptr(y_dev)))


# ---- Fft (src/fft/mod.rs:33-69) ---------------------------------------------------------------
class Fft(_Handle):
    _prefix = "yagi_hip_fft_"

    def __init__(self, n, direction):
        hd = C.c_void_p()
        _check(lib.yagi_hip_fft_create(n, Direction(direction).value, C.byref(hd)))
        self._h = hd
        self.n = n
        self.direction = Direction(direction)

    def run(self, input, output=None):                       # fft/mod.rs:45-48
        x = _arr(input, np.complex64)
        y = _out(output, self.n, np.complex64)
        _check(lib.yagi_hip_fft_run(self._h, _ptr(x), x.size, _ptr(y), y.size))
        return y

    def run_batch_dev(self, in_dev, out_dev, batch, stream=None):
        _check(lib.yagi_hip_fft_run_batch_dev(self._h, _devptr(in_dev), _devptr(out_dev), batch, stream))

    def run_batch(self, x):
        """host convenience: x holds batch*n samples"""
        x = _arr(x, np.complex64)
        if x.size % self.n:
            raise ConfigError("batch input must hold a whole number of transforms")
        dx = DeviceArray.from_numpy(x)
        dy = DeviceArray(x.size, np.complex64)
        self.run_batch_dev(dx, dy, x.size // self.n)
        synchronize()
        return dy.to_numpy().reshape(-1, self.n)

    def shift(self, buf, n=None):                            # fft/mod.rs:50-57 (in place)
        if not (isinstance(buf, np.ndarray) and buf.dtype == np.complex64 and buf.flags.c_contiguous):
            raise ConfigError("shift works in place on a contiguous complex64 array")
        _check(lib.yagi_hip_fft_shift(_ptr(buf), buf.size if n is None else n))
        return buf

    def set_stream(self, stream):
        raise ModeError("Fft plans take the stream per call (run_batch_dev)")

    def reset(self):
        pass


def fft_run(input, direction):                               # fft/mod.rs:66-69
    x = _arr(input, np.complex64)
    y = np.empty_like(x)
    _check(lib.yagi_hip_fft_run_oneshot(_ptr(x), _ptr(y), x.size, Direction(direction).value))
    return y


class WindowType(enum.IntEnum):
    """math::WindowType (src/math/windows.rs:7-18)"""
    Unknown = 0
    Hamming = 1
    Hann = 2
    BlackmanHarris = 3
    BlackmanHarris7 = 4
    Kaiser = 5
    FlatTop = 6
    Triangular = 7
    RcosTaper = 8
    Kbd = 9


class Spgram(_Handle):
    """fft::Spgram<T> (src/fft/spgram.rs); T = Complex32 (dtype complex64) or f32 (float32)."""

    def __init__(self, nfft, wtype, window_len, delay, dtype=np.complex64):      # new() :49-125
        self._set_type(dtype)
        hd = C.c_void_p()
        _check(self._fn("create")(nfft, int(wtype), window_len, delay, C.byref(hd)))
        self._h = hd

    def _set_type(self, dtype):
        self.T = np.dtype(dtype).type
        if self.T not in (np.complex64, np.float32):
            raise ConfigError("Spgram sample type must be complex64 or float32")
        self._prefix = "yagi_hip_spgramcf_" if self.T is np.complex64 else "yagi_hip_spgramf_"
        self._Tc = cf32 if self.T is np.complex64 else C.c_float

    @classmethod
    def default(cls, nfft, dtype=np.complex64):                                  # :128-131
        self = object.__new__(cls)
        self._set_type(dtype)
        hd = C.c_void_p()
        _check(self._fn("create_default")(nfft, C.byref(hd)))
        self._h = hd
        return self

    def clear(self):
        _check(self._fn("clear")(self._h))

    def set_alpha(self, alpha):
        _check(self._fn("set_alpha")(self._h, alpha))

    def get_alpha(self):
        a = C.c_float()
        _check(self._fn("get_alpha")(self._h, C.byref(a)))
        return a.value

    def set_freq(self, f):
        _check(self._fn("set_freq")(self._h, f))

    def set_rate(self, r):
        _check(self._fn("set_rate")(self._h, r))

    def _params(self):
        a, b, c, d = C.c_size_t(), C.c_size_t(), C.c_size_t(), C.c_int()
        _check(self._fn("get_params")(self._h, C.byref(a), C.byref(b), C.byref(c), C.byref(d)))
        return a.value, b.value, c.value, WindowType(d.value)

    def get_nfft(self):
        return self._params()[0]

    def get_window_len(self):
        return self._params()[1]

    def get_delay(self):
        return self._params()[2]

    def get_wtype(self):
        return self._params()[3]

    def _counters(self):
        v = [C.c_uint64() for _ in range(4)]
        _check(self._fn("get_counters")(self._h, *[C.byref(t) for t in v]))
        return [t.value for t in v]

    def get_num_samples(self):
        return self._counters()[0]

    def get_num_samples_total(self):
        return self._counters()[1]

    def get_num_transforms(self):
        return self._counters()[2]

    def get_num_transforms_total(self):
        return self._counters()[3]

    def push(self, x):                                                           # :237-251
        _check(self._fn("push")(self._h, _byval(x, self._Tc)))

    def write(self, x):                                                          # :254-258
        x = _arr(x, self.T)
        _check(self._fn("write")(self._h, _ptr(x), x.size))

    def write_dev(self, x_dev, n):
        _check(self._fn("write_dev")(self._h, _devptr(x_dev), n))

    def get_psd_mag(self):                                                       # :292-304
        out = np.empty(self.get_nfft(), np.float32)
        _check(self._fn("get_psd_mag")(self._h, _ptr(out), out.size))
        return out

    def get_psd(self):                                                           # :308-316
        out = np.empty(self.get_nfft(), np.float32)
        with np.errstate(divide="ignore"):
            _check(self._fn("get_psd")(self._h, _ptr(out), out.size))
        return out

    @classmethod
    def estimate_psd(cls, nfft, x, dtype=np.complex64):                          # :319-330
        T = np.dtype(dtype).type
        x = _arr(x, T)
        out = np.empty(nfft, np.float32)
        fn = lib.yagi_hip_spgramcf_estimate_psd if T is np.complex64 else lib.yagi_hip_spgramf_estimate_psd
        _check(fn(nfft, _ptr(x), x.size, _ptr(out)))
        return out


# ---- headline stream ----------------------------------------------------------------------------
class FirFftStream(_Handle):
    """firfilt_crcf.execute_block -> consecutive nfft frames -> Fft::run(Forward), fused."""
    _prefix = "yagi_hip_firfft_crcf_"

    def __init__(self, h, nfft=4096):
        h = _arr(h, np.float32)
        hd = C.c_void_p()
        _check(lib.yagi_hip_firfft_crcf_create(_ptr(h), h.size, nfft, C.byref(hd)))
        self._h = hd
        self.nfft = nfft

    def set_scale(self, scale):
        _check(lib.yagi_hip_firfft_crcf_set_scale(self._h, scale))

    def set_variant(self, v):
        _check(lib.yagi_hip_firfft_crcf_set_variant(self._h, v))

    def execute(self, x):
        x = _arr(x, np.complex64)
        if x.size % self.nfft:
            raise ConfigError("input must hold a whole number of frames")
        nf = x.size // self.nfft
        y = np.empty(x.size, np.complex64)
        _check(lib.yagi_hip_firfft_crcf_execute(self._h, _ptr(x), nf, _ptr(y)))
        return y.reshape(nf, self.nfft)

    def execute_dev(self, x_dev, nframes, spectra_dev):
        _check(lib.yagi_hip_firfft_crcf_execute_dev(self._h, _devptr(x_dev), nframes, _devptr(spectra_dev)))

    def set_pipeline(self, on=True):
        """pipelined block calls: consecutive execute_dev calls overlap on two streams owned by the object;
        their outputs are ordered on the object's stream only after join() (include/yagi_hip.h)"""
        _check(lib.yagi_hip_firfft_crcf_set_pipeline(self._h, 1 if on else 0))

    def join(self):
        _check(lib.yagi_hip_firfft_crcf_join(self._h))


# ---- channelizers (absent from the reference; see include/yagi_hip.h) ---------------------------
class FirPfbCh(_Handle):
    _prefix = "yagi_hip_firpfbch_crcf_"

    def __init__(self, M, p, h):
        h = _arr(h, np.float32)
        if h.size < M * p:
            raise ConfigError("prototype shorter than M*p")
        hd = C.c_void_p()
        _check(lib.yagi_hip_firpfbch_crcf_create(M, p, _ptr(h), C.byref(hd)))
        self._h, self.M, self.p = hd, M, p

    @classmethod
    def new_kaiser(cls, M, m, as_):
        self = object.__new__(cls)
        hd = C.c_void_p()
        _check(lib.yagi_hip_firpfbch_crcf_create_kaiser(M, m, as_, C.byref(hd)))
        self._h, self.M, self.p = hd, M, 2 * m
        return self

    def analyzer_execute(self, x):
        x = _arr(x, np.complex64)
        if x.size % self.M:
            raise ConfigError("input must hold a whole number of M-sample frames")
        nf = x.size // self.M
        y = np.empty(x.size, np.complex64)
        _check(lib.yagi_hip_firpfbch_crcf_analyzer_execute(self._h, _ptr(x), nf, _ptr(y)))
        return y.reshape(nf, self.M)

    def analyzer_execute_dev(self, x_dev, nframes, y_dev):
        _check(lib.yagi_hip_firpfbch_crcf_analyzer_execute_dev(self._h, _devptr(x_dev), nframes, _devptr(y_dev)))

    def synthesizer_execute(self, X):
        """X: frames of M channel samples ([nframes, M] or flat) -> nframes*M output samples"""
        X = _arr(X, np.complex64)
        if X.size % self.M:
            raise ConfigError("input must hold a whole number of M-channel frames")
        nf = X.size // self.M
        y = np.empty(X.size, np.complex64)
        _check(lib.yagi_hip_firpfbch_crcf_synthesizer_execute(self._h, _ptr(X), nf, _ptr(y)))
        return y

    def synthesizer_execute_dev(self, x_dev, nframes, y_dev):
        _check(lib.yagi_hip_firpfbch_crcf_synthesizer_execute_dev(self._h, _devptr(x_dev), nframes, _devptr(y_dev)))


class FirPfbCh2(_Handle):
    _prefix = "yagi_hip_firpfbch2_crcf_"

    def __init__(self, M, m, h):
        h = _arr(h, np.float32)
        if h.size < 2 * M * m:
            raise ConfigError("prototype shorter than 2*M*m")
        hd = C.c_void_p()
        _check(lib.yagi_hip_firpfbch2_crcf_create(M, m, _ptr(h), C.byref(hd)))
        self._h, self.M, self.m = hd, M, m

    @classmethod
    def new_kaiser(cls, M, m, as_):
        self = object.__new__(cls)
        hd = C.c_void_p()
        _check(lib.yagi_hip_firpfbch2_crcf_create_kaiser(M, m, as_, C.byref(hd)))
        self._h, self.M, self.m = hd, M, m
        return self

    @classmethod
    def new_kaiser_synthesizer(cls, M, m, as_):
        """prototype of the matching synthesizer: kaiser(2Mm+1, 0.5/M, as_) scaled to sum M"""
        self = object.__new__(cls)
        hd = C.c_void_p()
        _check(lib.yagi_hip_firpfbch2_crcf_create_kaiser_synthesizer(M, m, as_, C.byref(hd)))
        self._h, self.M, self.m = hd, M, m
        return self

    def synthesizer_execute(self, X):
        """X: steps of M channel samples ([nsteps, M] or flat) -> nsteps*M/2 output samples"""
        X = _arr(X, np.complex64)
        if X.size % self.M:
            raise ConfigError("input must hold a whole number of M-channel steps")
        ns = X.size // self.M
        y = np.empty(ns * (self.M // 2), np.complex64)
        _check(lib.yagi_hip_firpfbch2_crcf_synthesizer_execute(self._h, _ptr(X), ns, _ptr(y)))
        return y

    def synthesizer_execute_dev(self, x_dev, nsteps, y_dev):
        _check(lib.yagi_hip_firpfbch2_crcf_synthesizer_execute_dev(self._h, _devptr(x_dev), nsteps, _devptr(y_dev)))

    def analyzer_execute(self, x):
        x = _arr(x, np.complex64)
        M2 = self.M // 2
        if x.size % M2:
            raise ConfigError("input must hold a whole number of M/2-sample steps")
        ns = x.size // M2
        y = np.empty(ns * self.M, np.complex64)
        _check(lib.yagi_hip_firpfbch2_crcf_analyzer_execute(self._h, _ptr(x), ns, _ptr(y)))
        return y.reshape(ns, self.M)

    def analyzer_execute_dev(self, x_dev, nsteps, y_dev):
        _check(lib.yagi_hip_firpfbch2_crcf_analyzer_execute_dev(self._h, _devptr(x_dev), nsteps, _devptr(y_dev)))

    def analyzer_execute_shard_dev(self, x_dev, nsteps, rank, nranks, yshard_dev):
        _check(lib.yagi_hip_firpfbch2_crcf_analyzer_execute_shard_dev(
            self._h, _devptr(x_dev), nsteps, rank, nranks, _devptr(yshard_dev)))

    def analyzer_execute_sharded_dev(self, x_dev, nsteps, comm, y_dev, nchunks=0):
        """sub-bands sharded over the ranks of `comm` (yagi_amd.dist.Comm): shard kernel -> RCCL all-gather ->
        assemble, chunked so the exchange overlaps the next chunk's kernel; y_dev = [nsteps][M] on every rank"""
        _check(lib.yagi_hip_firpfbch2_crcf_analyzer_execute_sharded_dev(
            self._h, _devptr(x_dev), nsteps, comm._h, nchunks, _devptr(y_dev)))

    @staticmethod
    def assemble_dev(gathered_dev, nsteps, M, nranks, y_dev, stream=None):
        _check(lib.yagi_hip_firpfbch2_crcf_assemble_dev(_devptr(gathered_dev), nsteps, M, nranks, _devptr(y_dev), stream))
