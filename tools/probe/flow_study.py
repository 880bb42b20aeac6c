#!/usr/bin/env python3
"""CPU study of the synchronisation flows (host emulation, tests/emu): what the tail pass finds when the sequence
kernel has run `iters` flow iterations -- how many entries of the state table are not yet the sequential decoder's, how
long the runs of such entries are (a run of L false entries is L lock-step trips of the tail pass, whatever else happens),
how many flows start.
    python tools/probe/flow_study.py [photo|cfg2] [subseq_bytes] [iters]"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
from tests.emu import emu  # noqa: E402
from tools import jpegsynth  # noqa: E402

what = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
sb = int(sys.argv[2]) if len(sys.argv) > 2 else 256
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 1
data = open(os.path.join(ROOT, "tests", "golden", "IMG_6510.JPG"), "rb").read() if what == "photo" else jpegsynth.config(2, seed=0)
rc, r = emu.decode_scan(data, 0, sb, iters)
assert rc == 0
S = len(r.p)
p, cz, pend = np.zeros(S, np.int32), np.zeros(S, np.int32), np.zeros(S, np.uint8)
emu.lib().emu_read_pre_tail.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
n = emu.lib().emu_read_pre_tail(p.ctypes.data, cz.ctypes.data, pend.ctypes.data, S)
assert n == S
false = (p != r.p) | (cz != r.cz)
seg = r.seg_index
runs, cur = [], 0
for i in range(S):
    if false[i] and (i == 0 or seg[i] == seg[i - 1] or cur == 0):
        cur += 1
    else:
        if cur:
            runs.append(cur)
        cur = 1 if false[i] else 0
if cur:
    runs.append(cur)
runs = np.array(runs if runs else [0])
print("%s, %d-byte subsequences, %d flow iteration(s) in the sequence kernel: %d subsequences" % (what, sb, iters, S))
print("  pending marks %d (%.1f %%), false entries %d (%.1f %%) of which only c differs: %d" % (
    pend.sum(), 100.0 * pend.sum() / S, false.sum(), 100.0 * false.sum() / S, ((p == r.p) & ((cz >> 8) == (r.cz >> 8)) & false).sum()))
print("  runs of false entries: %d, mean %.2f, p90 %d, max %d; histogram 1..8+: %s" % (
    len(runs), runs.mean(), np.percentile(runs, 90), runs.max(), [int((runs == k).sum()) for k in range(1, 8)] + [int((runs >= 8).sum())]))
