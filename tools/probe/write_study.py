#!/usr/bin/env python3
"""CPU study of the write pass's loop (host emulation, tests/emu, built from the product's decode_units): iterations a
WAVE of 64 consecutive subsequences stays in the loop (the largest of its lanes' counts: the loop is uniform) and the
symbols its lanes decode -- what a change of the loop (pair entries, slot periods) does to the number of iterations,
without a GPU.   python tools/probe/write_study.py [photo|cfg2] [extra g++ flags for the emulation, e.g. -DJG_WRITE_AC_BITS=11]"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
from tools import jpegsynth  # noqa: E402

what = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
flags = sys.argv[2:]
lib = "/tmp/libjgemu_study.so"
subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "jpeggpu_amd", "csrc")] + flags +
                      [os.path.join(ROOT, "tests", "emu", "emu_pipeline.cpp"), os.path.join(ROOT, "jpeggpu_amd", "csrc", "jg_reader.cpp"), "-o", lib])
L = C.CDLL(lib)
L.emu_decode_scan.argtypes = [C.c_char_p, C.c_size_t, C.c_int, C.c_int, C.c_int] + [C.c_void_p] * 10
L.emu_read_write_iters.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
data = open(os.path.join(ROOT, "tests", "golden", "IMG_6510.JPG"), "rb").read() if what == "photo" else jpegsynth.config(2, seed=0)
ns, nd, it = C.c_int(), C.c_int(), C.c_int()
assert L.emu_decode_scan(data, len(data), 256, 1, 0, C.byref(ns), C.byref(nd), None, None, None, None, None, None, None, None) == 0
coef = np.zeros((nd.value, 64), np.int16)
assert L.emu_decode_scan(data, len(data), 256, 1, 0, None, None, None, None, None, None, None, None, coef.ctypes.data, C.byref(it)) == 0
S = ns.value
iters, syms = np.zeros(S, np.int32), np.zeros(S, np.int32)
assert L.emu_read_write_iters(iters.ctypes.data, syms.ctypes.data, S) == S
# a sequence is 240 subsequences: waves of 64, 64, 64, 48 lanes
wave_iters = 0
for b in range(0, S, 240):
    seq = iters[b:b + 240]
    for w in range(0, len(seq), 64):
        wave_iters += int(seq[w:w + 64].max())
print("%s %s: %d subsequences, %d symbols, lane iterations mean %.1f | wave iterations %d (%.1f per wave), %.3f symbols per lane and wave iteration" % (
    what, " ".join(flags), S, syms.sum(), iters.mean(), wave_iters, wave_iters / ((S + 63) // 64), syms.sum() / (wave_iters * 64.0)))
