"""Known-answer vectors of the reference's OWN Huffman symbol step, table builder and byte rule
(tests/golden/huff_kats.npz).

The expected outputs were produced by /root/reference/src/decode_huffman.cu:148-286 (u32_select_bits ...
decode_next_symbol), reader.cpp:186-224 + reader.hpp:45-64 (compute_huffman_table, struct huffman_table) and
decode_destuff.cu:37-44 (is_byte_data), compiled as they stand (oracle/ref_lift/build.sh: extracted by line range at
build time, g++ -D__device__= , no stand-in headers) and driven by oracle/ref_lift/make_huff_kats.py: 24 code tables
(Annex K and fitted ones from three encoders' settings), 54 315 windows -- every code of every table with random tails,
random windows, prefixes no code has, all-ones 16-bit candidates -- and all 65 536 (previous byte, byte) pairs. They pin
  * the oracle's restatement of the symbol step and of the byte rule, and
  * the PRODUCT's symbol step -- the parser's table builder and the look-up code of jg_huff_core.h that the kernels
    compile, built for the host (tests/emu): first-level table, second-level tables, the long-code path on its own, and the
    sync pack -- including what the reference does with bits no code matches (the 16-bit candidate always accepts, the
    huffval index wraps modulo 256: src/decode_huffman.cu:177-193), and the four-bytes-at-a-time byte rule of jg_bytes.h.
The GPU side of the same code is pinned through coefficients (tests/test_gpu_golden.py) and stage twins.
"""
import ctypes as C
import os

import numpy as np
import pytest

from tests.conftest import GOLDEN, ROOT


@pytest.fixture(scope="module")
def kats():
    z = dict(np.load(os.path.join(GOLDEN, "huff_kats.npz")))  # (a dict of arrays: an NpzFile reads a fresh copy at every access)
    z = {k: np.ascontiguousarray(v) for k, v in z.items()}
    assert len(z["names"]) == 24 and len(z["win"]) > 50000 and len(z["prev"]) == 65536
    assert z["length"].min() >= 1 and z["length"].max() <= 27
    return z


def _by_table(z):
    for k in range(len(z["names"])):
        m = np.nonzero(z["tbl"] == k)[0]
        a, b = int(m[0]), int(m[-1]) + 1
        yield k, str(z["names"][k]), slice(a, b)


def test_fixture_equals_reference_built_library(kats):
    """The pin itself (as tests/test_idct_kats.py): the library is rebuilt from the reference's sources where they are
    present, and every table, window and byte pair of the committed fixture goes through it again."""
    import subprocess

    lib = os.path.join(ROOT, "oracle", "_ref", "libref_huff.so")
    if os.path.isfile("/root/reference/src/decode_huffman.cu"):
        subprocess.check_call(["bash", os.path.join(ROOT, "oracle", "ref_lift", "build.sh")], stdout=subprocess.DEVNULL)
    if not os.path.exists(lib):
        pytest.skip("neither /root/reference nor a prebuilt oracle/_ref/libref_huff.so")
    L = C.CDLL(lib)
    L.ref_huff_build.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
    L.ref_huff_symbols.argtypes = [C.c_void_p] * 3 + [C.c_int] + [C.c_void_p] * 3
    L.ref_byte_rule.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    z = kats
    assert L.ref_huff_table_bytes() == z["ref_table"].shape[1] == 896
    for k, name, sl in _by_table(z):
        bits, vals = np.ascontiguousarray(z["bits"][k]), np.ascontiguousarray(z["vals"][k])
        t = np.zeros(896, np.uint8)
        L.ref_huff_build(bits.ctypes.data, vals.ctypes.data, int(z["count"][k]), t.ctypes.data)
        assert np.array_equal(t, z["ref_table"][k]), name
        w, zz = np.ascontiguousarray(z["win"][sl]), np.ascontiguousarray(z["z"][sl])
        lo, so, ro = (np.zeros(len(w), np.int32) for _ in range(3))
        L.ref_huff_symbols(t.ctypes.data, w.ctypes.data, zz.ctypes.data, len(w), lo.ctypes.data, so.ctypes.data, ro.ctypes.data)
        assert np.array_equal(lo, z["length"][sl]) and np.array_equal(so, z["symbol"][sl]) and np.array_equal(ro, z["run"][sl]), name
    d, wr = np.zeros(65536, np.uint8), np.zeros(65536, np.uint8)
    L.ref_byte_rule(z["prev"].ctypes.data, z["byte"].ctypes.data, 65536, d.ctypes.data, wr.ctypes.data)
    assert np.array_equal(d, z["is_data"]) and np.array_equal(wr, z["written"])


def test_reference_table_layout_is_what_the_survey_says(kats):
    """SURVEY.md Appendix A on `huffman_table` (896 bytes): entries[16]{int32 maxcode (-1: none), int32 valptr - mincode},
    lut[256]{u8 val, u8 nbits (0: not in the table)}, huffval[256] -- rebuilt here from the DHT payload with numpy and
    compared with the bytes compute_huffman_table left."""
    z = kats
    for k, name, _ in _by_table(z):
        bits, vals = z["bits"][k].astype(int), z["vals"][k]
        t = z["ref_table"][k]
        entries = t[:128].view(np.int32).reshape(16, 2)
        lut = t[128:640].reshape(256, 2)
        assert np.array_equal(t[640:], vals), name
        code, idx = 0, 0
        want_lut = np.zeros((256, 2), np.uint8)
        for l in range(1, 17):
            n = bits[l - 1]
            if n:
                assert entries[l - 1, 0] == code + n - 1 and entries[l - 1, 1] == idx - code, (name, l)
            else:
                assert entries[l - 1, 0] == -1, (name, l)
            for i in range(n):
                if l <= 8:
                    want_lut[(code + i) << (8 - l):(code + i + 1) << (8 - l)] = (vals[idx + i], l)
            code = (code + n) << 1
            idx += n
        assert np.array_equal(lut, want_lut), name


def test_oracle_symbol_step_equals_reference(kats):
    from oracle import oracle

    z = kats
    L = oracle.lib()
    L.jo_symbol_steps.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int] + [C.c_void_p] * 3
    for k, name, sl in _by_table(z):
        bits, vals = np.ascontiguousarray(z["bits"][k]), np.ascontiguousarray(z["vals"][k])
        w, zz = np.ascontiguousarray(z["win"][sl]), np.ascontiguousarray(z["z"][sl])
        lo, so, ro = (np.zeros(len(w), np.int32) for _ in range(3))
        rc = L.jo_symbol_steps(bits.ctypes.data, vals.ctypes.data, int(z["count"][k]), int(z["is_dc"][k]), w.ctypes.data, zz.ctypes.data,
                               len(w), lo.ctypes.data, so.ctypes.data, ro.ctypes.data)
        assert rc == 0
        assert np.array_equal(lo, z["length"][sl]), name
        assert np.array_equal(so, z["symbol"][sl]) and np.array_equal(ro, z["run"][sl]), name


def test_oracle_byte_rule_equals_reference(kats):
    from oracle import oracle

    z = kats
    L = oracle.lib()
    L.jo_byte_rule.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    d, wr = np.zeros(65536, np.uint8), np.zeros(65536, np.uint8)
    L.jo_byte_rule(z["prev"].ctypes.data, z["byte"].ctypes.data, 65536, d.ctypes.data, wr.ctypes.data)
    assert np.array_equal(d, z["is_data"]) and np.array_equal(wr, z["written"])


@pytest.mark.parametrize("path", [0, 1, 2])
def test_product_symbol_step_equals_reference(kats, path):
    """path 0: the write pass's tables (16-bit first level, second level, long codes); 1: the sync pack; 2: huff_long_code
    alone on the windows whose code is longer than 8 bits or matches nothing (the walk the reference makes for them,
    src/decode_huffman.cu:177-193) -- jg_huff_core.h claims the 16-bit candidate always accepts and the huffval index wraps."""
    from tests.emu import emu

    z = kats
    L = emu.lib()
    L.emu_symbol_steps.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int] + [C.c_void_p] * 3
    checked = 0
    for k, name, sl in _by_table(z):
        bits, vals = np.ascontiguousarray(z["bits"][k]), np.ascontiguousarray(z["vals"][k])
        w, zz = np.ascontiguousarray(z["win"][sl]), np.ascontiguousarray(z["z"][sl])
        want = [z["length"][sl], z["symbol"][sl], z["run"][sl]]
        if path == 2:  # the long-code path is only ever entered for codes of 9 bits and more (or no code at all)
            cat = np.abs(want[1]).astype(np.int64)
            ssss = np.where(cat > 0, np.floor(np.log2(np.maximum(cat, 1))).astype(int) + 1, 0)
            keep = want[0] - ssss >= 9
            w, zz = np.ascontiguousarray(w[keep]), np.ascontiguousarray(zz[keep])
            want = [a[keep] for a in want]
        lo, so, ro = (np.zeros(len(w), np.int32) for _ in range(3))
        L.emu_symbol_steps(bits.ctypes.data, vals.ctypes.data, int(z["count"][k]), int(z["is_dc"][k]), path, w.ctypes.data, zz.ctypes.data,
                           len(w), lo.ctypes.data, so.ctypes.data, ro.ctypes.data)
        assert np.array_equal(lo, want[0]), (name, "length")
        assert np.array_equal(so, want[1]) and np.array_equal(ro, want[2]), (name, "value / run")
        checked += len(w)
    assert checked > (3000 if path == 2 else 50000)


def test_product_byte_rule_equals_reference(kats):
    from tests.emu import emu

    z = kats
    L = emu.lib()
    L.emu_byte_rule.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    for across in (0, 1):
        d, wr = np.zeros(65536, np.uint8), np.zeros(65536, np.uint8)
        L.emu_byte_rule(z["prev"].ctypes.data, z["byte"].ctypes.data, 65536, across, d.ctypes.data, wr.ctypes.data)
        assert np.array_equal(d, z["is_data"]) and np.array_equal(wr, z["written"]), across
