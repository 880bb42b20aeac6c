"""Known-answer vectors of the reference's OWN IDCT (tests/golden/idct_kats.npz).

The expected outputs were produced by /root/reference/src/idct.cu:43-144 -- unfixh, unfixo, idct_vector,
idct_col, idct_row -- compiled as they stand (oracle/ref_lift/build.sh: extracted by line range at build
time, g++ -D__device__= , no stand-in headers) and driven by oracle/ref_lift/make_idct_kats.py. They pin
  * the oracle's restatement (CPU tests below), and
  * the HIP idct_kernel directly, without the oracle in between (GPU test below): the blocks are coded into
    grayscale JPEGs with the given quantisation table and decoded through the C ABI.
"""
import os

import numpy as np
import pytest

from tests.conftest import GOLDEN

BLOCKS_X = 32


@pytest.fixture(scope="module")
def kats():
    z = np.load(os.path.join(GOLDEN, "idct_kats.npz"))
    groups = sorted({k.split("/")[0] for k in z.files if "/" in k})
    assert len(groups) == 8 and len(z["vec_in"]) >= 4000
    return z, groups


def test_fixture_equals_reference_built_library(kats):
    """The pin itself: where the reference's sources are present (the authoring container) oracle/_ref/libref_idct.so
    is rebuilt from /root/reference/src/idct.cu:43-144 (oracle/ref_lift/build.sh) and every vector and block of the
    committed fixture is pushed through it again; where only the prebuilt library travelled (the GPU box) that one is
    used; with neither the test is skipped. A fixture that no longer matches the reference's code fails here."""
    import ctypes as C
    import subprocess

    from tests.conftest import ROOT

    lib = os.path.join(ROOT, "oracle", "_ref", "libref_idct.so")
    if os.path.isfile("/root/reference/src/idct.cu"):
        subprocess.check_call(["bash", os.path.join(ROOT, "oracle", "ref_lift", "build.sh")], stdout=subprocess.DEVNULL)
    if not os.path.exists(lib):
        pytest.skip("neither /root/reference nor a prebuilt oracle/_ref/libref_idct.so")
    L = C.CDLL(lib)
    L.ref_idct_vectors.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    L.ref_idct_blocks.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int]
    z, groups = kats
    vin = np.ascontiguousarray(z["vec_in"], np.int32)
    vout = np.zeros_like(vin)
    L.ref_idct_vectors(vin.ctypes.data, vout.ctypes.data, len(vin))
    assert np.array_equal(vout, z["vec_out"])
    for g in groups:
        q, coef = np.ascontiguousarray(z[g + "/q"], np.uint8), np.ascontiguousarray(z[g + "/coef"], np.int16)
        for signed, key in ((1, "/out"), (0, "/out_u")):
            out = np.zeros((len(coef), 64), np.uint8)
            L.ref_idct_blocks(coef.ctypes.data, q.ctypes.data, out.ctypes.data, len(coef), signed)
            assert np.array_equal(out, z[g + key]), (g, key)


def test_oracle_idct_vector_equals_reference(kats):
    from oracle import oracle

    z, _ = kats
    assert np.array_equal(oracle.idct_vectors(z["vec_in"]), z["vec_out"])


def test_oracle_idct_block_equals_reference(kats):
    """Column pass, row pass and every int16 store of the oracle against the reference's idct_col / idct_row;
    quantiser read as the reference does (int8, Appendix B-3) and as T.81 does (unsigned, the default)."""
    from oracle import oracle

    z, groups = kats
    for g in groups:
        q, coef = z[g + "/q"], z[g + "/coef"]
        for flags, key in ((oracle.QUIRK_SIGNED_Q, "/out"), (0, "/out_u")):
            got = np.stack([oracle.idct_block(c, q, flags).reshape(64) for c in coef])
            assert np.array_equal(got, z[g + key]), (g, key)
        if q.max() >= 128:
            assert not np.array_equal(z[g + "/out"], z[g + "/out_u"])


def test_oracle_decodes_kat_jpegs(kats):
    """The JPEGs the GPU test feeds to the HIP path carry exactly the KAT coefficients (oracle's Huffman
    decode == input blocks) and decode to the reference's pixels."""
    from oracle import oracle
    from tools import jpegsynth

    z, groups = kats
    for g in groups[:3]:
        q, coef = z[g + "/q"], z[g + "/coef"]
        d = oracle.decode(jpegsynth.encode_blocks(coef, BLOCKS_X, q))
        assert np.array_equal(d.coef[0].reshape(-1, 64), coef)
        want = z[g + "/out_u"].reshape(-1, BLOCKS_X, 8, 8).transpose(0, 2, 1, 3).reshape(d.planes[0].shape)
        assert np.array_equal(d.planes[0], want)


@pytest.mark.gpu
@pytest.mark.parametrize("restart_interval", [0, 7])
def test_hip_idct_equals_reference_kats(gpu_lib, kats, restart_interval):
    """idct_kernel (dequantisation, both passes, level shift, clamp) against the reference's outputs, no oracle
    involved. For q <= 127 the reference's literal output; above, the unsigned-quantiser variant of the same
    lifted code (documented deviation B-3)."""
    import jpeggpu_amd
    from tools import jpegsynth

    z, groups = kats
    bad = []
    for g in groups:
        q, coef = z[g + "/q"], z[g + "/coef"]
        data = jpegsynth.encode_blocks(coef, BLOCKS_X, q, restart_interval)
        planes, info = jpeggpu_amd.decode_to_planes(data)
        got = planes[0].cpu().numpy()
        key = "/out" if q.max() <= 127 else "/out_u"
        want = z[g + key].reshape(-1, BLOCKS_X, 8, 8).transpose(0, 2, 1, 3).reshape(got.shape)
        if not np.array_equal(got, want):
            bad.append((g, int((got != want).sum())))
    assert not bad, bad
