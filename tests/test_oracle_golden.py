"""The CPU oracle against the committed golden vectors (oracle/pin/make_golden.py): this is what
pins the checker that every GPU parity test relies on. Runs without a GPU."""
import hashlib
import json
import os

import numpy as np
import pytest

from oracle import oracle
from tests.conftest import GOLDEN


@pytest.fixture(scope="module")
def vectors():
    z = np.load(os.path.join(GOLDEN, "pin_vectors.npz"))
    names = sorted({k.split("/")[0] for k in z.files})
    return z, names


def test_pin_vectors_present(vectors):
    z, names = vectors
    assert len(names) >= 14
    # inputs from three independent encoders
    assert any(n.startswith("pil_") for n in names) and any(n.startswith("cjpeg_") for n in names)
    assert any(n.startswith("syn_") for n in names)


def test_coefficients_equal_ijg(vectors):
    """Quantised coefficients == IJG libjpeg 9d jpeg_read_coefficients, exactly."""
    z, names = vectors
    for n in names:
        d = oracle.decode(z[n + "/jpeg"].tobytes())
        for c in range(d.ncomp):
            a = z["%s/coef%d" % (n, c)]
            assert np.array_equal(d.coef[c][:a.shape[0], :a.shape[1]], a), (n, c)


def test_planes_within_reference_accuracy_band(vectors):
    """The reference is not bit-exact with a standard decoder either: its README (README.md:76,81)
    reports MSE 0.15-0.23 per component against nvJPEG. Same band against IJG islow raw planes."""
    z, names = vectors
    for n in names:
        d = oracle.decode(z[n + "/jpeg"].tobytes())
        for c in range(d.ncomp):
            ref = z["%s/raw%d" % (n, c)].astype(int)
            diff = d.planes[c].astype(int) - ref
            assert diff.shape == ref.shape
            assert (diff ** 2).mean() <= 0.25 and abs(diff).max() <= 2, (n, c)


def test_reference_photo_pins(photo_bytes):
    """images/IMG_6510.JPG: counts published in the reference's README (README.md:37-38: 89 sequences
    of 256 subsequences), IJG coefficient hashes, and the oracle's own plane hashes."""
    pins = json.load(open(os.path.join(GOLDEN, "photo_pins.json")))
    assert hashlib.sha256(photo_bytes).hexdigest() == pins["sha256_file"]
    d = oracle.decode(photo_bytes)
    assert (d.width, d.height, d.ncomp, d.restart_interval) == (4032, 3024, 3, 252)
    for c, p in enumerate(pins["components"]):
        bh, bw = p["blocks"]
        assert hashlib.sha256(d.coef[c][:bh, :bw].tobytes()).hexdigest() == p["sha256_ijg_coefficients"]
        assert hashlib.sha256(d.planes[c].tobytes()).hexdigest() == p["sha256_oracle_plane"]
    li = oracle.scan_info(photo_bytes, 0, 128)
    assert (li.num_subseq, li.num_segments, li.num_du) == (22711, 189, 285768)
    assert (li.num_subseq + 255) // 256 == 89
    assert (li.scan_begin, li.scan_end) == (14049, 2921331)


def test_idct_known_answers():
    """Hand-checked vectors of the restated fixed-point IDCT (reference src/idct.cu:44-95,146-223)."""
    q1 = np.ones(64, np.uint8)
    z = np.zeros(64, np.int16)
    assert (oracle.idct_block(z, q1) == 128).all()
    dc = z.copy()
    dc[0] = 1024  # column pass: unfixh(1024 * 0x5a82) = 362; row pass: unfixh(362 * 0x5a82) = 128 -> 256 -> clamp
    assert (oracle.idct_block(dc, q1) == 255).all()
    dc[0] = -1024
    assert (oracle.idct_block(dc, q1) == 0).all()
    dc[0] = 8  # 8 * 0x5a82 = 185360 -> (185360 + 0x8000) >> 16 = 3; 3 * 0x5a82 + 0x8000 >> 16 = 1
    assert (oracle.idct_block(dc, q1) == 129).all()
    # dequantisation truncates to int16 before the transform (idct.cu:178-180)
    big = z.copy()
    big[0] = 1023
    q = q1.copy()
    q[0] = 127  # 1023 * 127 = 129921 -> int16 wrap = -1151
    wrapped = z.copy()
    wrapped[0] = np.int16(-1151)
    assert np.array_equal(oracle.idct_block(big, q), oracle.idct_block(wrapped, q1))
    # quantiser values >= 128: unsigned by default, the reference's signed read behind a switch (B-3)
    q[0] = 200
    one = z.copy()
    one[0] = 4
    a = oracle.idct_block(one, q)
    b = oracle.idct_block(one, q, oracle.QUIRK_SIGNED_Q)
    neg = z.copy()
    neg[0] = 4 * (200 - 256)
    assert np.array_equal(b, oracle.idct_block(neg, q1)) and not np.array_equal(a, b)


def test_negative_inputs():
    """Status codes of the reference's reader for broken / unsupported streams (src/reader.cpp)."""
    from tools import jpegsynth

    good = jpegsynth.encode(64, 48, seed=3)

    def status(data):
        try:
            oracle.decode(bytes(data))
            return 0
        except oracle.OracleError as e:
            return e.status

    assert status(good) == 0
    assert status(b"") == 2 and status(b"\xff\xd8") == 2
    assert status(good[:len(good) // 2]) == 2            # no EOI: invalid jpeg (reader.cpp:455-458)
    prog = bytearray(good)
    i = prog.find(b"\xff\xc0")
    prog[i + 1] = 0xC2
    assert status(prog) == 4                             # SOF2 progressive: not supported (:618-630)
    p12 = bytearray(good)
    p12[i + 4] = 12
    assert status(p12) == 4                              # 12-bit precision (:96-99)
    q16 = bytearray(good)
    j = q16.find(b"\xff\xdb")
    q16[j + 4] |= 0x10
    assert status(q16) in (2, 4)                         # 16-bit DQT: not supported (:517-520) or length error
    nodht = bytearray(good)
    k = nodht.find(b"\xff\xc4")
    nodht[k + 1] = 0xE5                                   # turn the first DHT into an APP5 segment
    assert status(nodht) == 2                            # undefined table reference (:377-385)
