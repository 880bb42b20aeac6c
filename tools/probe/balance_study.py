#!/usr/bin/env python3
"""CPU study for length-balanced lane assignment in the write pass (host emulation, tests/emu: the product's decode_units,
one lane at a time): the iterations every lane needs, what a workgroup's four waves run today (the largest count of each
64 consecutive lanes) and what they would run with the sequence's subsequences ranked by a key and the w-th quartile
given to wave w -- by the exact count, and by keys the synchronisation passes could supply.
    python tools/probe/balance_study.py [photo|cfg2] [subseq_bytes]"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
from tools import jpegsynth  # noqa: E402

what = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
sb = int(sys.argv[2]) if len(sys.argv) > 2 else 256
lib = "/tmp/libjgemu_study.so"
subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "jpeggpu_amd", "csrc")] + sys.argv[3:] +
                      [os.path.join(ROOT, "tests", "emu", "emu_pipeline.cpp"), os.path.join(ROOT, "jpeggpu_amd", "csrc", "jg_reader.cpp"), "-o", lib])
L = C.CDLL(lib)
L.emu_decode_scan.argtypes = [C.c_char_p, C.c_size_t, C.c_int, C.c_int, C.c_int] + [C.c_void_p] * 10
L.emu_read_write_iters.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
data = open(os.path.join(ROOT, "tests", "golden", "IMG_6510.JPG"), "rb").read() if what == "photo" else jpegsynth.config(2, seed=0)
ns, nd, it = C.c_int(), C.c_int(), C.c_int()
assert L.emu_decode_scan(data, len(data), sb, 1, 0, C.byref(ns), C.byref(nd), None, None, None, None, None, None, None, None) == 0
S = ns.value
coef = np.zeros((nd.value, 64), np.int16)
n = np.zeros(S, np.int32)
p = np.zeros(S, np.int32)
assert L.emu_decode_scan(data, len(data), sb, 1, 0, None, None, None, None, p.ctypes.data, n.ctypes.data, None, None, coef.ctypes.data, C.byref(it)) == 0
iters, syms = np.zeros(S, np.int32), np.zeros(S, np.int32)
assert L.emu_read_write_iters(iters.ctypes.data, syms.ctypes.data, S) == S
SEQ = 255


def wave_sum(key):
    tot = 0
    for b in range(0, S, SEQ):
        seq = iters[b:b + SEQ]
        order = np.arange(len(seq)) if key is None else np.argsort(key[b:b + SEQ], kind="stable")
        s = seq[order]
        for w in range(0, len(s), 64):
            tot += int(s[w:w + 64].max())
    return tot


units = (n + 63) // 64
base = wave_sum(None)
print("%s, %d-byte subsequences: %d subsequences, %d symbols; lane iterations mean %.1f, p50 %d, p90 %d, p99 %d, max %d" % (
    what, sb, S, syms.sum(), iters.mean(), np.percentile(iters, 50), np.percentile(iters, 90), np.percentile(iters, 99), iters.max()))
print("  today: %d wave iterations, %.3f symbols per lane and wave iteration" % (base, syms.sum() / (base * 64.0)))
for name, key in (("exact iterations", iters), ("symbols", syms), ("data units (n / 64)", units), ("symbols + 1.5 units", syms + 1.5 * units),
                  ("max(symbols, 4 units)", np.maximum(syms, 4 * units))):
    t = wave_sum(key)
    print("  ranked by %-22s %d wave iterations (%.1f %%), %.3f symbols per lane and wave iteration; corr with iterations %.3f" % (
        name + ":", t, 100.0 * t / base, syms.sum() / (t * 64.0), np.corrcoef(key, iters)[0, 1]))


def wave_sum_groups(key, g):
    """Whole groups of g consecutive subsequences (a tile of the bitstream buffer holds 16 rows) ranked by the largest key
    among them, 64 / g groups to a wave."""
    tot = 0
    for b in range(0, S, SEQ):
        seq, k = iters[b:b + SEQ], key[b:b + SEQ]
        ng = (len(seq) + g - 1) // g
        gk = [k[i * g:(i + 1) * g].max() for i in range(ng)]
        order = np.argsort(gk, kind="stable")
        lanes = np.concatenate([np.arange(i * g, min((i + 1) * g, len(seq))) for i in order])
        s = seq[lanes]
        for w in range(0, len(s), 64):
            tot += int(s[w:w + 64].max())
    return tot


for g in (4, 8, 16, 32):
    for name, key in (("exact iterations", iters), ("data units", units)):
        t = wave_sum_groups(key, g)
        print("  groups of %2d ranked by %-18s %d wave iterations (%.1f %%)" % (g, name + ":", t, 100.0 * t / base))
