"""Host emulation of the device pipeline's logic (tests/emu) against the oracle's stage twins:
the subsequence-parallel algorithm, the host-built tables and work lists, without a GPU."""
import numpy as np
import pytest

from oracle import oracle
from tests import cases
from tests.emu import emu


@pytest.mark.parametrize("subseq_bytes,max_intra_iters", [(128, 256), (64, 256), (32, 256), (256, 1), (128, 3), (32, 1), (64, 2), (64, 255)])
def test_emulated_pipeline_equals_sequential_decode(subseq_bytes, max_intra_iters):
    """max_intra_iters < 256 cuts the lock-step loop of the sequence kernel short and hands the
    unfinished flows to the tail pass (1: only the first flow iteration runs in the sequence kernel)."""
    for name, data in cases.matrix().items():
        nscans = oracle.decode(data).nscans
        for s in range(nscans):
            rc, r = emu.decode_scan(data, s, subseq_bytes, max_intra_iters)
            assert rc == 0, name
            tw = oracle.scan_stages(data, s, subseq_bytes)
            assert np.array_equal(r.destuffed, tw.destuffed), (name, "destuffed bytes")
            assert np.array_equal(r.seg_index, tw.seg_index), (name, "segment index")
            ok = tw.p >= 0
            assert np.array_equal(r.p[ok], tw.p[ok]) and np.array_equal(r.n[ok], tw.n[ok]), (name, "p / n")
            assert np.array_equal(r.cz[ok], tw.cz[ok]), (name, "c, z")
            for k in range(4):
                assert np.array_equal(r.dc[k][ok].astype(np.int16), tw.dc[k][ok].astype(np.int16)), (name, "dc sums")
            assert np.array_equal(r.coef, tw.stream_coef), (name, "coefficients")


def test_multi_hypothesis_table_is_exact_and_shortens_the_flows():
    """The lone-decode path (jg_defs.h, multi-hypothesis speculation): candidates per data unit of the MCU, links,
    chain walk, and the reference's flows started from the chain's table. Whatever the table holds the flows must
    reach the sequential decoder's states; where restart segments let the chain run (it starts at a segment's first
    subsequence) it supplies the whole table and one flow pass verifies it."""
    m = cases.matrix()
    for name, data in m.items():
        for s in range(oracle.decode(data).nscans):
            for subseq_bytes in (64, 32):
                rc, r = emu.decode_scan(data, s, subseq_bytes, 256, multi_hypothesis=True)
                assert rc == 0, name
                tw = oracle.scan_stages(data, s, subseq_bytes)
                ok = tw.p >= 0
                assert np.array_equal(r.p[ok], tw.p[ok]) and np.array_equal(r.n[ok], tw.n[ok]) and np.array_equal(r.cz[ok], tw.cz[ok]), name
                assert np.array_equal(r.coef, tw.stream_coef), name
                if name in ("multi_seq_dri", "cfg2_small", "dri_row", "dri_7", "multi_seq_nodri", "cfg5_small") and subseq_bytes == 64:
                    _, plain = emu.decode_scan(data, s, subseq_bytes, 256)
                    assert r.mh_subseq == len(tw.p) and r.mh_known >= 0.95 * r.mh_subseq, (name, r.mh_known, r.mh_subseq)
                    assert r.max_flow_iters <= 2 < plain.max_flow_iters, (name, r.max_flow_iters, plain.max_flow_iters)
                if name in ("gray", "ni_444"):  # one data unit per MCU: plain speculation
                    assert r.mh_subseq == 0


def test_emulated_photo(photo_bytes):
    tw = oracle.scan_stages(photo_bytes, 0, 128)
    for cap in (256, 3):
        rc, r = emu.decode_scan(photo_bytes, 0, 128, cap)
        assert rc == 0 and np.array_equal(r.coef, tw.stream_coef)
        assert r.max_flow_iters >= 1


@pytest.mark.parametrize("simd", ["auto", "avx2", "sse2", "scalar"])
def test_marker_walk_at_every_alignment(simd, monkeypatch):
    """The host walk scans 64-byte blocks on the destuff-window grid (jg_reader.cpp, scan_window):
    shift the entropy-coded bytes through every position of a block (a COM segment of growing length
    in front of them) and across window edges, for streams rich in stuffed FF bytes, restart markers
    and fill bytes; destuffed bytes, segment index and coefficients must not change."""
    if simd != "auto":
        monkeypatch.setenv("JPEGGPU_HOST_SIMD", simd)
    m = cases.matrix()
    for name in ("multi_seq_dri", "dri_fill", "q100_noisy", "dri_1"):
        data = m[name]
        tw = oracle.scan_stages(data, 0, 64)
        assert data[:2] == b"\xff\xd8"
        for shift in list(range(0, 66)) + [4096 - 20 + k for k in range(0, 40, 3)]:
            com = b"\xff\xfe" + (2 + shift).to_bytes(2, "big") + bytes(shift)
            shifted = data[:2] + com + data[2:]
            rc, r = emu.decode_scan(shifted, 0, 64, 256)
            assert rc == 0, (name, shift)
            assert np.array_equal(r.destuffed, tw.destuffed), (name, shift, "destuffed bytes")
            assert np.array_equal(r.seg_index, tw.seg_index), (name, shift, "segment index")
            assert np.array_equal(r.coef, tw.stream_coef), (name, shift, "coefficients")
