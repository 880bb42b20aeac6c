"""ctypes binding of the CPU oracle (oracle/jpeg_oracle.c).

TEST INFRASTRUCTURE: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "libjpegoracle.so")

QUIRK_SIGNED_Q = 1


class _JoImage(C.Structure):
    _fields_ = [
        ("width", C.c_int), ("height", C.c_int), ("ncomp", C.c_int),
        ("restart_interval", C.c_int), ("nscans", C.c_int),
        ("hs", C.c_int * 4), ("vs", C.c_int * 4), ("qidx", C.c_int * 4),
        ("plane_w", C.c_int * 4), ("plane_h", C.c_int * 4),
        ("blocks_w", C.c_int * 4), ("blocks_h", C.c_int * 4),
        ("coef", C.POINTER(C.c_int16) * 4),
        ("plane", C.POINTER(C.c_uint8) * 4),
        ("qtab", (C.c_uint16 * 64) * 4),
        ("stream_coef", C.POINTER(C.c_int16) * 4),
        ("stream_du", C.c_int * 4),
        ("scan_ncomp", C.c_int * 4),
        ("scan_du_per_mcu", C.c_int * 4),
        ("scan_comp", (C.c_int * 4) * 4),
    ]


class _JoScanLayout(C.Structure):
    _fields_ = [("num_subseq", C.c_int), ("num_segments", C.c_int), ("num_du", C.c_int),
                ("scan_begin", C.c_size_t), ("scan_end", C.c_size_t)]


def build(force=False):
    """Compile the oracle with gcc (building the checker is not using it)."""
    src = os.path.join(_HERE, "jpeg_oracle.c")
    if force or not os.path.exists(_LIB) or os.path.getmtime(_LIB) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _LIB


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_LIB)
        _lib.jo_decode.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(_JoImage), C.c_int]
        _lib.jo_free.argtypes = [C.POINTER(_JoImage)]
        _lib.jo_idct_block.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        _lib.jo_idct_vectors.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        _lib.jo_scan_info.argtypes = [C.c_char_p, C.c_size_t, C.c_int, C.c_int, C.POINTER(_JoScanLayout)]
        _lib.jo_scan_stages.argtypes = [C.c_char_p, C.c_size_t, C.c_int, C.c_int] + [C.c_void_p] * 12
    return _lib


class OracleError(Exception):
    def __init__(self, status):
        super().__init__("oracle status %d" % status)
        self.status = status


class Decoded:
    """Result of a sequential CPU decode: planes, per-component coefficient arrays, stream order."""


def decode(data: bytes, flags: int = 0) -> Decoded:
    img = _JoImage()
    rc = lib().jo_decode(data, len(data), C.byref(img), flags)
    if rc:
        raise OracleError(rc)
    try:
        out = Decoded()
        out.width, out.height, out.ncomp = img.width, img.height, img.ncomp
        out.restart_interval, out.nscans = img.restart_interval, img.nscans
        out.hs = list(img.hs)[: img.ncomp]
        out.vs = list(img.vs)[: img.ncomp]
        out.qidx = list(img.qidx)[: img.ncomp]
        out.qtab = np.ctypeslib.as_array(img.qtab).copy().reshape(4, 64)
        out.planes, out.coef = [], []
        for c in range(img.ncomp):
            w, h = img.plane_w[c], img.plane_h[c]
            out.planes.append(np.ctypeslib.as_array(img.plane[c], shape=(h, w)).copy())
            bw, bh = img.blocks_w[c], img.blocks_h[c]
            out.coef.append(np.ctypeslib.as_array(img.coef[c], shape=(bh, bw, 64)).copy())
        out.stream_coef = []
        out.scan_comp = []
        out.scan_du_per_mcu = []
        for s in range(img.nscans):
            out.stream_coef.append(np.ctypeslib.as_array(img.stream_coef[s], shape=(img.stream_du[s], 64)).copy())
            out.scan_comp.append(list(img.scan_comp[s])[: img.scan_ncomp[s]])
            out.scan_du_per_mcu.append(img.scan_du_per_mcu[s])
        return out
    finally:
        lib().jo_free(C.byref(img))


def idct_block(coef, q, flags: int = 0):
    coef = np.ascontiguousarray(coef, dtype=np.int16).reshape(64)
    q = np.ascontiguousarray(q, dtype=np.uint16).reshape(64)
    out = np.zeros(64, dtype=np.uint8)
    lib().jo_idct_block(coef.ctypes.data, q.ctypes.data, out.ctypes.data, flags)
    return out.reshape(8, 8)


def idct_vectors(vec):
    """idct_vector of the oracle on int32[n, 8]."""
    vec = np.ascontiguousarray(vec, dtype=np.int32).reshape(-1, 8)
    out = np.zeros_like(vec)
    lib().jo_idct_vectors(vec.ctypes.data, out.ctypes.data, len(vec))
    return out


def scan_info(data: bytes, scan_idx: int, subseq_bytes: int):
    lay = _JoScanLayout()
    rc = lib().jo_scan_info(data, len(data), scan_idx, subseq_bytes, C.byref(lay))
    if rc:
        raise OracleError(rc)
    return lay


class Stages:
    pass


def scan_stages(data: bytes, scan_idx: int, subseq_bytes: int) -> Stages:
    """Stage twins of one scan: destuffed bytes, segment table, sequential-decoder states at the
    subsequence boundaries, stream-order coefficients."""
    lay = scan_info(data, scan_idx, subseq_bytes)
    st = Stages()
    S, G = lay.num_subseq, lay.num_segments
    st.num_subseq, st.num_segments, st.num_du = S, G, lay.num_du
    st.scan_begin, st.scan_end = lay.scan_begin, lay.scan_end
    st.destuffed = np.zeros(S * subseq_bytes, dtype=np.uint8)
    st.seg_offset = np.zeros(G, dtype=np.int32)
    st.seg_count = np.zeros(G, dtype=np.int32)
    st.seg_index = np.zeros(S, dtype=np.int32)
    st.p = np.zeros(S, dtype=np.int32)
    st.n = np.zeros(S, dtype=np.int32)
    st.cz = np.zeros(S, dtype=np.int32)
    st.dc = [np.zeros(S, dtype=np.int32) for _ in range(4)]
    st.stream_coef = np.zeros((lay.num_du, 64), dtype=np.int16)
    ptr = lambda a: a.ctypes.data if a.size else None
    rc = lib().jo_scan_stages(
        data, len(data), scan_idx, subseq_bytes,
        ptr(st.destuffed), ptr(st.seg_offset), ptr(st.seg_count), ptr(st.seg_index),
        ptr(st.p), ptr(st.n), ptr(st.cz), ptr(st.dc[0]), ptr(st.dc[1]), ptr(st.dc[2]), ptr(st.dc[3]),
        ptr(st.stream_coef))
    if rc:
        raise OracleError(rc)
    return st


def planes_to_rgbi(planes, sub_x, sub_y, width, height):
    """CPU restatement of the reference's host helper conv_to_rgbi (util/util.h:62-104): nearest-neighbour
    chroma replication, r = y + 1.402 (cr - 128), g = y - .344136 (cb - 128) - .714136 (cr - 128),
    b = y + 1.772 (cb - 128) in float32, roundf (half away from zero), clamp to 0..255. `planes`: 1 or 3
    uint8 arrays as decoded; returns uint8 [height, width, 3]. The C expression may or may not be
    contracted into FMAs by the compiler, so a consumer compares within +-1."""
    import numpy as np

    n = len(planes)
    sxm, sym = max(sub_x[:n]), max(sub_y[:n])
    up = []
    for c in range(n):
        p = planes[c]
        ys = np.minimum(np.arange(height) * sub_y[c] // sym, p.shape[0] - 1)
        xs = np.minimum(np.arange(width) * sub_x[c] // sxm, p.shape[1] - 1)
        up.append(p[ys][:, xs].astype(np.float32))
    if n == 1:
        return np.repeat(up[0].astype(np.uint8)[:, :, None], 3, axis=2)
    y, cb, cr = up
    f = np.float32
    r = y + f(1.402) * (cr - f(128))
    g = y - f(.344136) * (cb - f(128)) - f(.714136) * (cr - f(128))
    b = y + f(1.772) * (cb - f(128))
    out = np.stack([r, g, b], axis=2)
    out = np.sign(out) * np.floor(np.abs(out) + f(0.5))  # roundf
    return np.clip(out, 0, 255).astype(np.uint8)
