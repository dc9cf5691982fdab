"""helpers shared by the -m gpu parity tests (HIP path through the C ABI vs the CPU oracle)"""
import numpy as np

EPS = float(np.finfo(np.float32).eps)
SEED = 0x59414749          # seed + config index


def fir_bound(kind, h, x, M=1):
    """a-priori f32 error bound of a length-L inner product: 4 L eps sum|h| max|x|"""
    L = len(h)
    return 4.0 * L * EPS * float(np.sum(np.abs(h))) * float(np.max(np.abs(x)) + 1e-30)


def rel_l2(a, b):
    b = np.asarray(b)
    return float(np.linalg.norm(np.asarray(a, dtype=b.dtype) - b) / (np.linalg.norm(b) + 1e-300))


def rand_taps(rng, kind, L):
    if kind == "cccf":
        return (rng.standard_normal(L) + 1j * rng.standard_normal(L)).astype(np.complex64) / np.sqrt(L)
    return (rng.standard_normal(L) / np.sqrt(L)).astype(np.float32)


def rand_samples(rng, kind, n):
    if kind == "rrrf":
        return rng.standard_normal(n).astype(np.float32)
    return ((rng.standard_normal(n) + 1j * rng.standard_normal(n)) * np.sqrt(0.5)).astype(np.complex64)


def int_taps(rng, kind, L):
    """small integers: every product and partial sum is exact in f32 -> bit-exact indexing test"""
    if kind == "cccf":
        return (rng.integers(-3, 4, L) + 1j * rng.integers(-3, 4, L)).astype(np.complex64)
    return rng.integers(-3, 4, L).astype(np.float32)


def int_samples(rng, kind, n):
    if kind == "rrrf":
        return rng.integers(-8, 9, n).astype(np.float32)
    return (rng.integers(-8, 9, n) + 1j * rng.integers(-8, 9, n)).astype(np.complex64)
