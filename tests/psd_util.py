"""Restatement of the reference's spectral-mask test helper (src/utility/test_helpers.rs:9-75): validate_psd_signal
zero-pads the signal to nfft = 4 << nextpow2(max(len, 64)), takes 20 log10 |FFT| (fft-shifted) and checks every bin
of every region against its bounds.  Test infrastructure only."""
import numpy as np


def nextpow2(x):                                   # math/mod.rs:80-92
    x -= 1
    n = 0
    while x > 0:
        x >>= 1
        n += 1
    return n


def validate_psd_signal(buf, regions):
    """regions: (fmin, fmax, pmin, pmax, test_lo, test_hi); returns (ok, worst violation in dB)"""
    buf = np.asarray(buf, np.complex64)
    nfft = 4 << nextpow2(max(len(buf), 64))
    t = np.zeros(nfft, np.complex64)
    t[: len(buf)] = buf
    with np.errstate(divide="ignore"):
        psd = 20.0 * np.log10(np.abs(np.fft.fftshift(np.fft.fft(t.astype(np.complex128)))))
    f = np.arange(nfft, dtype=np.float32) / np.float32(nfft) - np.float32(0.5)
    worst = 0.0
    for fmin, fmax, pmin, pmax, lo, hi in regions:
        sel = (f >= fmin) & (f <= fmax)
        if lo and sel.any():
            worst = max(worst, float(np.max(pmin - psd[sel])))
        if hi and sel.any():
            worst = max(worst, float(np.max(psd[sel] - pmax)))
    return worst <= 0.0, worst


def estimate_req_filter_transition_bandwidth(as_, n):          # design/mod.rs:193-216 with Kaiser's formula :228-238
    df0, df1, df = np.float32(1e-3), np.float32(0.499), np.float32(0)
    for _ in range(20):
        df = np.float32(0.5) * (df1 + df0)
        n_hat = (np.float32(as_) - np.float32(7.95)) / (np.float32(14.26) * df)
        if n_hat < np.float32(n):
            df1 = df
        else:
            df0 = df
    return float(df)
