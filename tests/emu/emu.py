"""Builds and binds tests/emu/emu_pipeline.cpp (host emulation of the device pipeline's logic)."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(_HERE))
_LIB = os.path.join(_HERE, "libjgemu.so")
_lib = None


def lib():
    global _lib
    if _lib is None:
        srcs = [os.path.join(_HERE, "emu_pipeline.cpp"), os.path.join(ROOT, "jpeggpu_amd", "csrc", "jg_reader.cpp")]
        deps = srcs + [os.path.join(ROOT, "jpeggpu_amd", "csrc", h) for h in ("jg_huff_core.h", "jg_defs.h", "jg_reader.hpp", "jg_bytes.h")]
        if not os.path.exists(_LIB) or any(os.path.getmtime(d) > os.path.getmtime(_LIB) for d in deps):
            subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-I" + os.path.join(ROOT, "include"),
                                   "-I" + os.path.join(ROOT, "jpeggpu_amd", "csrc")] + srcs + ["-o", _LIB])
        _lib = C.CDLL(_LIB)
        _lib.emu_decode_scan.argtypes = [C.c_char_p, C.c_size_t, C.c_int, C.c_int, C.c_int] + [C.c_void_p] * 10
    return _lib


class EmuScan:
    pass


def decode_scan(data: bytes, scan_idx: int, subseq_bytes: int, max_intra_iters: int = 256, multi_hypothesis: bool = False):
    """`multi_hypothesis`: the lone-decode path that starts the flows from the multi-hypothesis table (jg_defs.h), where
    the scan qualifies (several data units per MCU, restart segments); r.mh_known / r.mh_subseq then say how much of
    the table the chain of links supplied."""
    ns, nd, it = C.c_int(), C.c_int(), C.c_int()
    lib().emu_set_multi_hypothesis(int(multi_hypothesis))
    rc = lib().emu_decode_scan(data, len(data), subseq_bytes, max_intra_iters, scan_idx, C.byref(ns), C.byref(nd),
                               None, None, None, None, None, None, None, None)
    if rc:
        return rc, None
    S, D = ns.value, nd.value
    r = EmuScan()
    r.destuffed = np.zeros(S * subseq_bytes, np.uint8)
    r.seg_index = np.zeros(S, np.int32)
    r.p, r.n, r.cz = (np.zeros(S, np.int32) for _ in range(3))
    r.dc = np.zeros((4, S), np.int32)
    r.coef = np.zeros((D, 64), np.int16)
    ptr = lambda a: a.ctypes.data if a.size else None
    rc = lib().emu_decode_scan(data, len(data), subseq_bytes, max_intra_iters, scan_idx, None, None, ptr(r.destuffed),
                               ptr(r.seg_index), ptr(r.p), ptr(r.n), ptr(r.cz), ptr(r.dc), r.coef.ctypes.data,
                               C.byref(it))
    r.max_flow_iters = it.value
    r.mh_known = C.c_int.in_dll(lib(), "g_mh_known").value if multi_hypothesis else 0
    r.mh_subseq = C.c_int.in_dll(lib(), "g_mh_subseq").value if multi_hypothesis else 0
    return rc, r
