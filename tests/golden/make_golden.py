#!/usr/bin/env python3
"""Transcribe the reference's golden vectors (liquid-dsp autotest data) into fixtures.

Run ONCE in the build container (needs /root/reference, which never travels):

    python tests/golden/make_golden.py [/root/reference]

It reads only *numeric literals* out of the reference's Rust test-data tables and
inline test vectors and stores them as float32/complex64 arrays in ``.npz`` files next
to this script.  No reference source text is kept: a fixture is inputs + expected
outputs.  Sources (all relative to the reference root):

  src/filter/fir/firfilt_test_data.rs:3-969    FIRFILT_{RRRF,CRCF,CCCF}_DATA_H*_{H,X,Y}
  src/filter/fir/firdecim_test_data.rs:4-791   FIRDECIM_{RRRF,CRCF,CCCF}_DATA_M*_{H,X,Y}
  src/fft/test_data.rs:6-4690                  FFT_TEST_{X,Y}{2..509}
  src/filter/test_data.rs:3-6393               FFTFILT_{RRRF,CRCF,CCCF}_DATA_H*X256_{H,X,Y}
  src/dotprod/mod.rs:341-655                   inline h/x/test vectors of the rand/struct tests
  src/filter/fir/firpfb.rs:318-346             firpfb impulse-response vector
  src/filter/fir/firinterp.rs:277-387          firinterp_{rrrf,crcf}_generic h / x / test
"""
import re
import sys
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
NUM = r"[-+]?(?:\d+\.?\d*|\.\d+)(?:[eE][-+]?\d+)?"
CPLX = re.compile(r"(?:Complex32|Complex|Cf32)(?:::<f32>)?::new\(\s*(%s)\s*,\s*(%s)\s*\)" % (NUM, NUM))
REAL = re.compile(NUM)


def parse_body(body: str, is_complex: bool) -> np.ndarray:
    body = re.sub(r"//[^\n]*", "", body)
    if is_complex:
        vals = [complex(float(a), float(b)) for a, b in CPLX.findall(body)]
        return np.asarray(vals, dtype=np.complex64)
    vals = [float(v) for v in REAL.findall(body)]
    return np.asarray(vals, dtype=np.float32)


CONST_RE = re.compile(
    r"(?:pub\s+)?const\s+([A-Z0-9_]+)\s*:\s*\[\s*([A-Za-z0-9_:<>]+)\s*;\s*(\d+)\s*\]\s*=\s*\[(.*?)\];",
    re.S,
)


def const_tables(path: Path) -> dict:
    out = {}
    for name, ty, n, body in CONST_RE.findall(path.read_text()):
        arr = parse_body(body, "Complex" in ty)
        assert arr.size == int(n), (name, arr.size, n)
        out[name.lower()] = arr
    return out


LET_ARR_RE = re.compile(
    r"let\s+(\w+)\s*:\s*(?:Vec<\s*([A-Za-z0-9_:<>]+)\s*>|\[\s*([A-Za-z0-9_:<>]+)\s*;\s*\d+\s*\])\s*=\s*(?:vec!)?\[(.*?)\];",
    re.S,
)
LET_SCALAR_RE = re.compile(r"let\s+(\w+)\s*=\s*(%s)\s*;" % NUM)
LET_CSCALAR_RE = re.compile(r"let\s+(\w+)\s*=\s*Cf32::new\(\s*(%s)\s*,\s*(%s)\s*\)\s*;" % (NUM, NUM))
ASSERT_LEN_RE = re.compile(r"h\[\.\.(\d+)\]\.dotprod\(&x\[\.\.\d+\]\)\s*,\s*(%s)" % NUM)


def dotprod_vectors(path: Path) -> dict:
    """One entry per inline known-answer test: <test>__{h,x,test,test_rev,v32..}."""
    txt = path.read_text()
    out = {}
    for m in re.finditer(r"fn\s+(test_dotprod_\w+)\s*\(\)\s*\{", txt):
        name = m.group(1)
        # function body: up to the next "#[test]" or end of file
        nxt = txt.find("#[test]", m.end())
        body = txt[m.end(): nxt if nxt > 0 else len(txt)]
        if "thread_rng" in body or "vec![Complex::new(1.0" in body:
            continue  # unseeded random / trivial tests carry no vectors
        arrs = {}
        for var, ty1, ty2, abody in LET_ARR_RE.findall(body):
            ty = ty1 or ty2
            if "map(" in abody:
                continue
            arrs[var] = parse_body(abody, ("Cf32" in ty) or ("Complex" in ty))
        if "h" not in arrs or "x" not in arrs:
            continue
        out[f"{name}__h"] = arrs["h"]
        out[f"{name}__x"] = arrs["x"]
        for var, a, b in LET_CSCALAR_RE.findall(body):
            out[f"{name}__{var}"] = np.asarray([complex(float(a), float(b))], dtype=np.complex64)
        for var, v in LET_SCALAR_RE.findall(body):
            if var.startswith("test"):
                out[f"{name}__{var}"] = np.asarray([float(v)], dtype=np.float32)
        for n, v in ASSERT_LEN_RE.findall(body):
            out[f"{name}__len{n}"] = np.asarray([float(v)], dtype=np.float32)
    return out


def firpfb_vector(path: Path) -> dict:
    txt = path.read_text()
    i0 = txt.index("fn test_firpfb_impulse_response")
    body = txt[i0: txt.index("#[test]", i0)]
    out = {}
    for var, ty1, ty2, abody in LET_ARR_RE.findall(body):
        out[f"firpfb_impulse_response__{var}"] = parse_body(abody, False)
    assert set(k.split("__")[1] for k in out) == {"h", "noise", "test"}, out.keys()
    return out


def firinterp_vectors(path: Path) -> dict:
    """firinterp.rs:277-387: inline h / x / test of the two *_generic known-answer tests"""
    txt = path.read_text()
    out = {}
    for kind in ("rrrf", "crcf"):
        i0 = txt.index(f"fn test_firinterp_{kind}_generic")
        body = txt[i0: txt.index("#[test]", i0)]
        for var, ty1, ty2, abody in LET_ARR_RE.findall(body):
            if var in ("h", "x", "test"):
                ty = ty1 or ty2
                out[f"firinterp_{kind}_generic__{var}"] = parse_body(abody, "Complex" in ty)
        # the rrrf test spells x inline without a type annotation
        m = re.search(r"let\s+x\s*=\s*\[([^\]]*)\];", body)
        if m and f"firinterp_{kind}_generic__x" not in out:
            out[f"firinterp_{kind}_generic__x"] = parse_body(m.group(1), False)
        assert {f"firinterp_{kind}_generic__{v}" for v in ("h", "x", "test")} <= set(out), (kind, out.keys())
    return out


def main():
    ref = Path(sys.argv[1] if len(sys.argv) > 1 else "/root/reference")
    src = ref / "src"
    sets = {
        "firfilt": const_tables(src / "filter/fir/firfilt_test_data.rs"),
        "firdecim": const_tables(src / "filter/fir/firdecim_test_data.rs"),
        "fft": const_tables(src / "fft/test_data.rs"),
        "fftfilt": const_tables(src / "filter/test_data.rs"),
        "dotprod": dotprod_vectors(src / "dotprod/mod.rs"),
        "firpfb": firpfb_vector(src / "filter/fir/firpfb.rs"),
        "firinterp": firinterp_vectors(src / "filter/fir/firinterp.rs"),
    }
    for name, d in sets.items():
        assert d, name
        np.savez_compressed(HERE / f"{name}.npz", **d)
        print(f"{name}.npz: {len(d)} arrays, {sum(a.size for a in d.values())} values")


def fft4096_definition_vectors(check_only=False):
    """fft4096.npz is NOT transcribed from the reference (it holds no vector above N = 509, and its FFT arithmetic
    lives in the rustfft crate, not in the tree): 4 transforms of the C3 stream, input from the oracle's generator
    (seed 0x59414749 + 3), output by DEFINITION (unnormalised forward DFT) from numpy's f64 FFT.  Needs only the repo,
    so it also runs where /root/reference is absent; with check_only it verifies the committed file instead."""
    import sys
    sys.path.insert(0, str(HERE.parent.parent))
    from oracle import oracle
    x = oracle.gen_complex(0x59414749 + 3, 4 * 4096)
    y = np.array([np.fft.fft(x[b * 4096:(b + 1) * 4096].astype(np.complex128)) for b in range(4)])
    path = HERE / "fft4096.npz"
    if check_only:
        with np.load(path, allow_pickle=False) as z:
            return bool(np.array_equal(z["x"], x) and np.allclose(z["y"], y, rtol=0, atol=1e-9))
    np.savez_compressed(path, x=x, y=y)
    print(f"fft4096.npz: 4 transforms of 4096 points ({x.size} inputs)")
    return True


if __name__ == "__main__":
    import sys
    if "--fft4096" in sys.argv:
        fft4096_definition_vectors()
    else:
        main()
        fft4096_definition_vectors()
