"""ctypes binding of tools/jpegsynth.c: seeded synthetic baseline JPEGs for tests and bench."""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "libjpegsynth.so")


class _Params(C.Structure):
    _fields_ = [("width", C.c_int), ("height", C.c_int), ("ncomp", C.c_int),
                ("hs", C.c_int * 4), ("vs", C.c_int * 4),
                ("interleaved", C.c_int), ("restart_interval", C.c_int), ("quality", C.c_int),
                ("optimize", C.c_int), ("noise", C.c_int), ("fill_bytes", C.c_int), ("seed", C.c_uint64), ("qmax", C.c_int)]


def build(force=False):
    src = os.path.join(_HERE, "jpegsynth.c")
    if force or not os.path.exists(_LIB) or os.path.getmtime(_LIB) < os.path.getmtime(src):
        subprocess.check_call(["gcc", "-O2", "-shared", "-fPIC", "-o", _LIB, src, "-lm"])
    return _LIB


_lib = None


def _load():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_LIB)
        _lib.js_encode.restype = C.c_size_t
        _lib.js_encode.argtypes = [C.POINTER(_Params), C.c_void_p, C.c_size_t]
        _lib.js_encode_blocks_opt.restype = C.c_size_t
        _lib.js_encode_blocks_opt.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_size_t]
    return _lib


def encode_blocks(coef, blocks_x, q, restart_interval=0, optimize=False) -> bytes:
    """Grayscale JPEG whose data units are the given quantised coefficient blocks: `coef` int16 [n, 64] in
    natural order (n a multiple of blocks_x), `q` uint8 [64] natural order. Block b lands at block row
    b // blocks_x, column b % blocks_x. `optimize`: Huffman tables fitted to these blocks instead of Annex K's."""
    import numpy as np

    coef = np.ascontiguousarray(coef, dtype=np.int16).reshape(-1, 64)
    q = np.ascontiguousarray(q, dtype=np.uint8).reshape(64)
    n = len(coef)
    assert n % blocks_x == 0
    cap = 4096 + n * 64 * 4
    buf = C.create_string_buffer(cap)
    m = _load().js_encode_blocks_opt(coef.ctypes.data, blocks_x, n // blocks_x, q.ctypes.data, restart_interval, int(optimize), buf, cap)
    if m == 0:
        raise RuntimeError("jpegsynth: encode_blocks failed")
    return buf.raw[:m]


def encode(width, height, sampling=((2, 2), (1, 1), (1, 1)), interleaved=True, restart_interval=0,
           quality=75, optimize=False, noise=6, fill_bytes=0, seed=0, qmax=0) -> bytes:
    """sampling: one (h, v) pair per component (1..4 components)."""
    _load()
    p = _Params()
    p.width, p.height, p.ncomp = width, height, len(sampling)
    for c, (h, v) in enumerate(sampling):
        p.hs[c], p.vs[c] = h, v
    p.interleaved, p.restart_interval, p.quality = int(interleaved), restart_interval, quality
    p.optimize, p.noise, p.fill_bytes, p.seed, p.qmax = int(optimize), noise, fill_bytes, seed, qmax
    cap = 1024 + width * height * len(sampling) * 3
    buf = C.create_string_buffer(cap)
    n = _lib.js_encode(C.byref(p), buf, cap)
    if n == 0:
        raise RuntimeError("jpegsynth: encode failed (bad parameters or buffer too small)")
    return buf.raw[:n]


# The five BASELINE.json configurations, at full size and at a reduced size for oracle-speed tests.
def config(idx: int, seed: int = 0, small: bool = False) -> bytes:
    if idx in (1, 2, 3):  # 12 MP 4:2:0 interleaved, one restart interval per MCU row
        w, h = (4032, 3024) if not small else (496, 360)
        return encode(w, h, ((2, 2), (1, 1), (1, 1)), True, (w + 15) // 16, quality=88, noise=9, seed=seed)
    if idx == 4:  # 39 MP 4:4:4, three single-component scans, no restart markers
        w, h = (7216, 5408) if not small else (456, 344)
        return encode(w, h, ((1, 1), (1, 1), (1, 1)), False, 0, quality=80, noise=6, seed=seed)
    if idx == 5:  # 4 components (2x1,1x1,1x1,2x1), 4 DC + 4 AC tables, no restart markers
        w, h = (4032, 3024) if not small else (504, 376)
        return encode(w, h, ((2, 1), (1, 1), (1, 1), (2, 1)), True, 0, quality=85, optimize=True, noise=6, seed=seed)
    raise ValueError(idx)
