#!/usr/bin/env python3
"""Known-answer vectors of the reference's own Huffman symbol step, table builder and byte rule
-> tests/golden/huff_kats.npz.

TEST INFRASTRUCTURE, authoring container only. Inputs are generated here (tables read out of JPEGs made by
tools/jpegsynth, seeded windows); outputs come from oracle/_ref/libref_huff.so, i.e. from
/root/reference/src/decode_huffman.cu:148-286, reader.cpp:186-224 (+ reader.hpp:45-64) and decode_destuff.cu:37-44
compiled as they stand (oracle/ref_lift/build.sh). The fixture holds data only: inputs and the reference's outputs.

  tables: bits uint8[K,16], vals uint8[K,256], count int32[K], is_dc uint8[K], name str[K]
          ref_table uint8[K,896]: the reference's `huffman_table` struct as compute_huffman_table left it
  vectors: tbl int32[N] (table index), win uint32[N] (32 bits, MSB first), z int32[N] (0 for a DC table)
           -> length, symbol, run int32[N]: decode_next_symbol<true>
  byte rule: prev, byte uint8[65536] (every pair) -> is_data, written uint8[65536]
"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
LIB = os.path.join(ROOT, "oracle", "_ref", "libref_huff.so")
OUT = os.path.join(ROOT, "tests", "golden", "huff_kats.npz")


def dht_tables(jpeg: bytes):
    """(class, id, bits[16], vals) of every table defined in front of the first SOS."""
    out, i = [], 2
    while i + 4 <= len(jpeg):
        assert jpeg[i] == 0xFF
        m, n = jpeg[i + 1], int.from_bytes(jpeg[i + 2:i + 4], "big")
        if m == 0xDA:
            break
        if m == 0xC4:
            p, end = i + 4, i + 2 + n
            while p < end:
                tc, th = jpeg[p] >> 4, jpeg[p] & 15
                bits = list(jpeg[p + 1:p + 17])
                cnt = sum(bits)
                out.append((tc, th, bits, list(jpeg[p + 17:p + 17 + cnt])))
                p += 17 + cnt
        i += 2 + n
    return out


def codes_of(bits):
    code, out = 0, []
    for l in range(1, 17):
        for _ in range(bits[l - 1]):
            out.append((code, l))
            code += 1
        code <<= 1
    return out


def main():
    from tools import jpegsynth

    subprocess.check_call(["bash", os.path.join(HERE, "build.sh")])
    L = C.CDLL(LIB)
    L.ref_huff_build.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
    L.ref_huff_symbols.argtypes = [C.c_void_p] * 3 + [C.c_int] + [C.c_void_p] * 3
    L.ref_byte_rule.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    tb = L.ref_huff_table_bytes()
    assert tb == 896
    S420 = ((2, 2), (1, 1), (1, 1))
    sources = [
        ("annexk", jpegsynth.encode(64, 48, S420, seed=1)),                                  # the Annex K tables
        ("opt420", jpegsynth.encode(320, 240, S420, optimize=True, seed=22)),                # fitted, photo-like
        ("opt4c", jpegsynth.encode(264, 200, ((2, 1), (1, 1), (1, 1), (2, 1)), optimize=True, seed=20)),  # 4 + 4 fitted
        ("optq100", jpegsynth.encode(160, 128, S420, quality=100, noise=40, optimize=True, seed=23)),      # long codes
    ]
    rng = np.random.default_rng(20261005)
    names, bits_a, vals_a, count_a, isdc_a, ref_tab = [], [], [], [], [], []
    tbl, win, zz = [], [], []
    for src, jpeg in sources:
        for tc, th, bits, vals in dht_tables(jpeg):
            k = len(names)
            names.append("%s_%s%d" % (src, "dc" if tc == 0 else "ac", th))
            bits_a.append(bits)
            v = np.zeros(256, np.uint8)
            v[:len(vals)] = vals
            vals_a.append(v)
            count_a.append(len(vals))
            isdc_a.append(1 if tc == 0 else 0)
            w = []
            for code, l in codes_of(bits):  # every code of the table, three random tails each
                for _ in range(3):
                    w.append((code << (32 - l)) | int(rng.integers(0, 1 << (32 - l))))
            longest = max(l for _, l in codes_of(bits))
            w += [int(x) for x in rng.integers(0, 1 << 32, 2048, dtype=np.uint64)]        # anything
            w += [0xFFFFFFFF, 0xFFFF0000, 0xFFFE0000, 0xFFFFFFFE, 0x00000000, 0x80000000]  # prefixes no code has, corners
            w += [((0xFFFF << 16) | int(x)) for x in rng.integers(0, 1 << 16, 64)]        # all-ones 16-bit candidates
            w += [(((1 << longest) - 1) << (32 - longest)) | int(x) for x in rng.integers(0, 1 << (32 - longest), 64)]
            z = rng.integers(1, 64, len(w)) if tc == 1 else np.zeros(len(w), np.int64)
            tbl += [k] * len(w)
            win += w
            zz += [int(x) for x in z]
    K = len(names)
    bits_a = np.array(bits_a, np.uint8)
    vals_a = np.array(vals_a, np.uint8)
    tbl = np.array(tbl, np.int32)
    win = np.array(win, np.uint32)
    zz = np.array(zz, np.int32)
    length, symbol, run = (np.zeros(len(win), np.int32) for _ in range(3))
    for k in range(K):
        t = np.zeros(tb, np.uint8)
        L.ref_huff_build(bits_a[k].ctypes.data, vals_a[k].ctypes.data, int(count_a[k]), t.ctypes.data)
        ref_tab.append(t)
        m = np.nonzero(tbl == k)[0]
        a, b = int(m[0]), int(m[-1]) + 1
        assert (m == np.arange(a, b)).all()
        w, z = np.ascontiguousarray(win[a:b]), np.ascontiguousarray(zz[a:b])
        lo, so, ro = (np.zeros(b - a, np.int32) for _ in range(3))
        L.ref_huff_symbols(t.ctypes.data, w.ctypes.data, z.ctypes.data, b - a, lo.ctypes.data, so.ctypes.data, ro.ctypes.data)
        length[a:b], symbol[a:b], run[a:b] = lo, so, ro
    prev = np.repeat(np.arange(256, dtype=np.uint8), 256)
    byte = np.tile(np.arange(256, dtype=np.uint8), 256)
    is_data, written = np.zeros(65536, np.uint8), np.zeros(65536, np.uint8)
    L.ref_byte_rule(prev.ctypes.data, byte.ctypes.data, 65536, is_data.ctypes.data, written.ctypes.data)
    np.savez_compressed(OUT, names=np.array(names), bits=bits_a, vals=vals_a, count=np.array(count_a, np.int32),
                        is_dc=np.array(isdc_a, np.uint8), ref_table=np.array(ref_tab, np.uint8), tbl=tbl, win=win, z=zz,
                        length=length, symbol=symbol, run=run, prev=prev, byte=byte, is_data=is_data, written=written)
    print("%s: %d tables, %d windows (lengths %d..%d, %d of them 9..16-bit codes or longer prefixes), %d byte pairs, %d bytes" % (
        OUT, K, len(win), length.min(), length.max(), int((length - np.abs(0) >= 9).sum()), 65536, os.path.getsize(OUT)))


if __name__ == "__main__":
    main()
