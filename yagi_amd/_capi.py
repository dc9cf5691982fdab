"""ctypes binding of libyagi_hip.so (include/yagi_hip.h).

Loading rules
  * The library is built in-tree (yagi_amd/libyagi_hip.so) by ``__graft_entry__.build()`` or
    ``make -C yagi_amd/csrc``.  If it is missing this module raises ImportError -- there is no
    CPU fallback anywhere in the package.
  * torch is imported first when available so the process holds ONE HIP runtime (torch's
    bundled libamdhip64 has the SONAME our DT_NEEDED asks for; see csrc/Makefile).
"""
import ctypes as C
import os
from pathlib import Path

_HERE = Path(__file__).resolve().parent
# YAGI_HIP_LIB lets the kernel-tuning scripts under tools/ load an alternative build of the SAME
# library (different compile flags); it is never a different back-end.
LIB_PATH = Path(os.environ.get("YAGI_HIP_LIB") or (_HERE / "libyagi_hip.so"))


class cf32(C.Structure):
    """num_complex::Complex<f32> / yagi_cf32"""
    _fields_ = [("re", C.c_float), ("im", C.c_float)]


def _load():
    if not LIB_PATH.exists():
        raise ImportError(
            f"{LIB_PATH} not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C yagi_amd/csrc`; yagi_amd has no CPU fallback")
    if os.environ.get("YAGI_NO_TORCH_PRELOAD", "0") != "1":
        try:
            import torch  # noqa: F401  (maps torch's HIP runtime before ours is resolved)
        except Exception:
            pass
    return C.CDLL(str(LIB_PATH))


lib = _load()

vp, sz, ci, f32, u64 = C.c_void_p, C.c_size_t, C.c_int, C.c_float, C.c_uint64
pvp = C.POINTER(C.c_void_p)


def _sig(name, *argtypes, restype=ci):
    fn = getattr(lib, name)
    fn.argtypes = list(argtypes)
    fn.restype = restype
    return fn


_sig("yagi_hip_last_error", restype=C.c_char_p)
_sig("yagi_hip_version", restype=C.c_char_p)
_sig("yagi_hip_device_count", C.POINTER(ci))
_sig("yagi_hip_set_device", ci)
_sig("yagi_hip_malloc", pvp, sz)
_sig("yagi_hip_free", vp)
_sig("yagi_hip_memcpy_h2d", vp, vp, sz)
_sig("yagi_hip_memcpy_d2h", vp, vp, sz)
_sig("yagi_hip_memset_dev", vp, ci, sz)
_sig("yagi_hip_device_synchronize")
_sig("yagi_hip_stream_synchronize", vp)
_sig("yagi_hip_gen_real_dev", u64, u64, sz, vp, vp)
_sig("yagi_hip_gen_complex_dev", u64, u64, sz, vp, vp)
_sig("yagi_hip_fir_design_kaiser", sz, f32, f32, f32, vp)

for _k in ("rrrf", "rccf", "crcf", "cccf"):
    _sig(f"yagi_hip_dotprod_{_k}", vp, vp, sz, vp)
    _sig(f"yagi_hip_dotprod_{_k}_dev", vp, vp, sz, vp, vp)

# (T, C) ctypes for by-value scalar arguments
KIND_TYPES = {"rrrf": (f32, f32), "crcf": (cf32, f32), "cccf": (cf32, cf32)}

for _k, (_T, _Cc) in KIND_TYPES.items():
    p = f"yagi_hip_firfilt_{_k}_"
    _sig(p + "create", vp, sz, pvp)
    _sig(p + "create_kaiser", sz, f32, f32, f32, pvp)
    _sig(p + "freqresponse", vp, f32, vp)
    _sig(p + "groupdelay", vp, f32, vp)
    _sig(p + "create_rect", sz, pvp)
    _sig(p + "create_dc_blocker", sz, f32, pvp)
    _sig(p + "create_notch", sz, f32, f32, pvp)
    _sig(p + "destroy", vp)
    _sig(p + "clone", vp, pvp)
    _sig(p + "set_stream", vp, vp)
    _sig(p + "set_coefficients", vp, vp, sz)
    _sig(p + "reset", vp)
    _sig(p + "push", vp, _T)
    _sig(p + "write", vp, vp, sz)
    _sig(p + "execute", vp, vp)
    _sig(p + "execute_one", vp, _T, vp)
    _sig(p + "execute_block", vp, vp, sz, vp, sz)
    _sig(p + "execute_block_dev", vp, vp, sz, vp)
    _sig(p + "set_pipeline", vp, ci)
    _sig(p + "join", vp)
    _sig(p + "set_scale", vp, _Cc)
    _sig(p + "get_scale", vp, vp)
    _sig(p + "get_length", vp, C.POINTER(sz))
    _sig(p + "get_coefficients", vp, vp, sz)
    p = f"yagi_hip_firdecim_{_k}_"
    _sig(p + "create", sz, vp, sz, pvp)
    _sig(p + "create_kaiser", sz, sz, f32, pvp)
    _sig(p + "destroy", vp)
    _sig(p + "clone", vp, pvp)
    _sig(p + "set_stream", vp, vp)
    _sig(p + "reset", vp)
    _sig(p + "get_decim_rate", vp, C.POINTER(sz))
    _sig(p + "set_scale", vp, _Cc)
    _sig(p + "get_scale", vp, vp)
    _sig(p + "execute", vp, vp, sz, vp)
    _sig(p + "execute_block", vp, vp, sz, sz, vp)
    _sig(p + "execute_block_dev", vp, vp, sz, vp)
    _sig(f"yagi_hip_firdecim_{_k}_freqresp", vp, f32, vp)
    p = f"yagi_hip_firpfb_{_k}_"
    _sig(p + "create", sz, vp, sz, pvp)
    _sig(p + "create_kaiser", sz, sz, f32, f32, pvp)
    _sig(p + "create_default", sz, sz, pvp)
    _sig(p + "destroy", vp)
    _sig(p + "clone", vp, pvp)
    _sig(p + "set_stream", vp, vp)
    _sig(p + "reset", vp)
    _sig(p + "set_scale", vp, _Cc)
    _sig(p + "get_scale", vp, vp)
    _sig(p + "push", vp, _T)
    _sig(p + "write", vp, vp, sz)
    _sig(p + "execute", vp, sz, vp)
    _sig(p + "execute_block", vp, sz, vp, sz, vp, sz)
    _sig(p + "execute_block_dev", vp, sz, vp, sz, vp)
    _sig(p + "execute_all_dev", vp, vp, sz, vp)
    _sig(p + "execute_select_dev", vp, vp, vp, sz, vp)

    p = f"yagi_hip_firinterp_{_k}_"
    _sig(p + "create", sz, vp, sz, pvp)
    _sig(p + "create_kaiser", sz, sz, f32, pvp)
    _sig(p + "create_linear", sz, pvp)
    _sig(p + "create_window", sz, sz, pvp)
    _sig(p + "destroy", vp)
    _sig(p + "clone", vp, pvp)
    _sig(p + "set_stream", vp, vp)
    _sig(p + "reset", vp)
    _sig(p + "get_interp_rate", vp, C.POINTER(sz))
    _sig(p + "get_sub_len", vp, C.POINTER(sz))
    _sig(p + "set_scale", vp, _Cc)
    _sig(p + "get_scale", vp, vp)
    _sig(p + "execute", vp, _T, vp, sz)
    _sig(p + "execute_block", vp, vp, sz, vp, sz)
    _sig(p + "execute_block_dev", vp, vp, sz, vp)
    _sig(p + "flush", vp, vp, sz)
    p = f"yagi_hip_rresamp_{_k}_"
    _sig(p + "create", sz, sz, sz, vp, sz, pvp)
    _sig(p + "create_kaiser", sz, sz, sz, f32, f32, pvp)
    _sig(p + "create_default", sz, sz, pvp)
    _sig(p + "destroy", vp)
    _sig(p + "set_stream", vp, vp)
    _sig(p + "reset", vp)
    _sig(p + "set_scale", vp, _Cc)
    _sig(p + "get_scale", vp, vp)
    _sig(p + "get_params", vp, C.POINTER(sz), C.POINTER(sz), C.POINTER(sz), C.POINTER(sz))
    _sig(p + "write", vp, vp, sz)
    _sig(p + "execute", vp, vp, sz, vp, sz)
    _sig(p + "execute_block", vp, vp, sz, sz, vp, sz)
    _sig(p + "execute_block_dev", vp, vp, sz, vp)
    p = f"yagi_hip_fftfilt_{_k}_"
    _sig(p + "create", vp, sz, sz, pvp)
    _sig(p + "destroy", vp)
    _sig(p + "clone", vp, pvp)
    _sig(p + "set_stream", vp, vp)
    _sig(p + "reset", vp)
    _sig(p + "set_scale", vp, _Cc)
    _sig(p + "get_scale", vp, vp)
    _sig(p + "get_length", vp, C.POINTER(sz))
    _sig(p + "execute", vp, vp, sz, vp, sz)
    _sig(p + "execute_blocks", vp, vp, sz, vp)
    _sig(p + "execute_blocks_dev", vp, vp, sz, vp)

for _k in ("rrrf", "crcf", "cccf"):
    _sig(f"yagi_hip_firfilt_{_k}_set_kernel", vp, ci)

for _k, _T in (("cf", cf32), ("f", f32)):
    p = f"yagi_hip_spgram{_k}_"
    _sig(p + "create", sz, ci, sz, sz, pvp)
    _sig(p + "create_default", sz, pvp)
    _sig(p + "destroy", vp)
    _sig(p + "set_stream", vp, vp)
    _sig(p + "clear", vp)
    _sig(p + "reset", vp)
    _sig(p + "set_alpha", vp, f32)
    _sig(p + "get_alpha", vp, C.POINTER(f32))
    _sig(p + "set_freq", vp, f32)
    _sig(p + "set_rate", vp, f32)
    _sig(p + "get_params", vp, C.POINTER(sz), C.POINTER(sz), C.POINTER(sz), C.POINTER(ci))
    _sig(p + "get_counters", vp, C.POINTER(u64), C.POINTER(u64), C.POINTER(u64), C.POINTER(u64))
    _sig(p + "push", vp, _T)
    _sig(p + "write", vp, vp, sz)
    _sig(p + "write_dev", vp, vp, sz)
    _sig(p + "get_psd_mag", vp, vp, sz)
    _sig(p + "get_psd", vp, vp, sz)
    _sig(p + "estimate_psd", sz, vp, sz, vp)

_sig("yagi_hip_fft_create", sz, ci, pvp)
_sig("yagi_hip_fft_destroy", vp)
_sig("yagi_hip_fft_clone", vp, pvp)
_sig("yagi_hip_fft_len", vp, C.POINTER(sz))
_sig("yagi_hip_fft_run", vp, vp, sz, vp, sz)
_sig("yagi_hip_fft_run_batch_dev", vp, vp, vp, sz, vp)
_sig("yagi_hip_fft_shift", vp, sz)
_sig("yagi_hip_fft_shift_dev", vp, sz, sz, vp)
_sig("yagi_hip_fft_run_oneshot", vp, vp, sz, ci)

_sig("yagi_hip_firfft_crcf_create", vp, sz, sz, pvp)
_sig("yagi_hip_firfft_crcf_destroy", vp)
_sig("yagi_hip_firfft_crcf_set_stream", vp, vp)
_sig("yagi_hip_firfft_crcf_set_scale", vp, f32)
_sig("yagi_hip_firfft_crcf_reset", vp)
_sig("yagi_hip_firfft_crcf_set_variant", vp, ci)
_sig("yagi_hip_firfft_crcf_execute", vp, vp, sz, vp)
_sig("yagi_hip_firfft_crcf_execute_dev", vp, vp, sz, vp)
_sig("yagi_hip_firfft_crcf_set_pipeline", vp, ci)
_sig("yagi_hip_firfft_crcf_join", vp)

_sig("yagi_hip_firpfbch_crcf_create", sz, sz, vp, pvp)
_sig("yagi_hip_firpfbch_crcf_create_kaiser", sz, sz, f32, pvp)
_sig("yagi_hip_firpfbch_crcf_destroy", vp)
_sig("yagi_hip_firpfbch_crcf_set_stream", vp, vp)
_sig("yagi_hip_firpfbch_crcf_reset", vp)
_sig("yagi_hip_firpfbch_crcf_analyzer_execute", vp, vp, sz, vp)
_sig("yagi_hip_firpfbch_crcf_analyzer_execute_dev", vp, vp, sz, vp)
_sig("yagi_hip_firpfbch_crcf_synthesizer_execute", vp, vp, sz, vp)
_sig("yagi_hip_firpfbch_crcf_synthesizer_execute_dev", vp, vp, sz, vp)

_sig("yagi_hip_firpfbch2_crcf_create", sz, sz, vp, pvp)
_sig("yagi_hip_firpfbch2_crcf_create_kaiser", sz, sz, f32, pvp)
_sig("yagi_hip_firpfbch2_crcf_create_kaiser_synthesizer", sz, sz, f32, pvp)
_sig("yagi_hip_firpfbch2_crcf_synthesizer_execute", vp, vp, sz, vp)
_sig("yagi_hip_firpfbch2_crcf_synthesizer_execute_dev", vp, vp, sz, vp)
_sig("yagi_hip_firpfbch2_crcf_destroy", vp)
_sig("yagi_hip_firpfbch2_crcf_set_stream", vp, vp)
_sig("yagi_hip_firpfbch2_crcf_reset", vp)
_sig("yagi_hip_firpfbch2_crcf_analyzer_execute", vp, vp, sz, vp)
_sig("yagi_hip_firpfbch2_crcf_analyzer_execute_dev", vp, vp, sz, vp)
_sig("yagi_hip_firpfbch2_crcf_analyzer_execute_shard_dev", vp, vp, sz, ci, ci, vp)
_sig("yagi_hip_firpfbch2_crcf_assemble_dev", vp, sz, sz, ci, vp, vp)

# ---- multi-GPU (RCCL through the C ABI) ---------------------------------------------------------
_sig("yagi_hip_comm_unique_id", vp)
_sig("yagi_hip_comm_create", vp, ci, ci, pvp)
_sig("yagi_hip_comm_destroy", vp)
_sig("yagi_hip_comm_rank", vp, C.POINTER(ci), C.POINTER(ci))
_sig("yagi_hip_comm_all_gather_dev", vp, vp, vp, sz, vp)
_sig("yagi_hip_firpfbch2_crcf_analyzer_execute_sharded_dev", vp, vp, sz, vp, ci, vp)

# ---- Resamp2 / MsResamp2 and Rresamp clone ---------------------------------
for _k, (_T, _Cc) in KIND_TYPES.items():
    _sig(f"yagi_hip_rresamp_{_k}_clone", vp, pvp)
    p = f"yagi_hip_resamp2_{_k}_"
    _sig(p + "create", vp, sz, f32, pvp)
    _sig(p + "create_kaiser", sz, f32, f32, pvp)
    _sig(p + "destroy", vp)
    _sig(p + "clone", vp, pvp)
    _sig(p + "reset", vp)
    _sig(p + "set_stream", vp, vp)
    _sig(p + "set_scale", vp, _Cc)
    _sig(p + "get_scale", vp, vp)
    _sig(p + "get_delay", vp, C.POINTER(sz))
    _sig(p + "execute_block", vp, ci, vp, sz, vp, sz)
    _sig(p + "execute_block_dev", vp, ci, vp, sz, vp, sz)
    p = f"yagi_hip_msresamp2_{_k}_"
    _sig(p + "create", ci, sz, f32, f32, f32, pvp)
    _sig(p + "create_taps", ci, sz, vp, vp, pvp)
    _sig(p + "destroy", vp)
    _sig(p + "clone", vp, pvp)
    _sig(p + "reset", vp)
    _sig(p + "set_stream", vp, vp)
    _sig(p + "get_params", vp, C.POINTER(ci), C.POINTER(sz), C.POINTER(f32), vp)
    _sig(p + "execute_block", vp, vp, sz, vp, sz)
    _sig(p + "execute_block_dev", vp, vp, sz, vp, sz)
