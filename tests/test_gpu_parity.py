"""GPU parity: the HIP path, called through the C ABI, against the CPU oracle. Bit-exact."""
import os

import numpy as np
import pytest

from tests import cases, gpu_util

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_cuda():
    import torch

    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    return torch


@pytest.fixture(scope="module")
def inputs():
    return cases.matrix()


def _decode_gpu(data, subseq_bytes=None):
    import jpeggpu_amd

    planes, info = jpeggpu_amd.decode_to_planes(data, subseq_bytes=subseq_bytes)
    return [p.cpu().numpy() for p in planes], info


@pytest.mark.parametrize("subseq_bytes", [256, 128, 64, 32])
def test_matrix_planes_bit_exact(gpu_lib, torch_cuda, inputs, subseq_bytes):
    from oracle import oracle

    bad = []
    for name, data in inputs.items():
        ref = oracle.decode(data)
        got, info = _decode_gpu(data, subseq_bytes)
        assert info.num_components == ref.ncomp
        for c in range(ref.ncomp):
            assert got[c].shape == ref.planes[c].shape, name
            if not np.array_equal(got[c], ref.planes[c]):
                bad.append((name, c, int((got[c] != ref.planes[c]).sum())))
    assert not bad, bad


_tmp_view = gpu_util.tmp_view


@pytest.mark.parametrize("name", ["dri_row", "multi_seq_nodri", "ni_420_dri", "cfg5_small", "dri_fill"])
@pytest.mark.parametrize("subseq_bytes", [256, 128, 32])
@pytest.mark.parametrize("device_scan", [False, True])
def test_stage_parity(gpu_lib, torch_cuda, inputs, name, subseq_bytes, device_scan):
    """Every intermediate buffer against its CPU twin: destuffed bytes, subsequence->segment map,
    synchronised states, stream-order coefficients -- with the segment table and the destuff work list from the
    host walk, and with the ones the device-side front end builds (jpeggpu_ext_set_device_scan)."""
    import jpeggpu_amd
    from oracle import oracle

    torch = torch_cuda
    data = inputs[name]
    planes, info, tmp, base, lay = jpeggpu_amd.decode_to_planes(data, subseq_bytes=subseq_bytes, return_tmp=True,
                                                                device_scan=device_scan)
    for s in range(lay.num_scans):
        sl = lay.scans[s]
        tw = oracle.scan_stages(data, s, subseq_bytes)
        S, G = sl.num_subsequences, sl.num_segments
        if sl.device_scan:  # the layout holds capacities; the device reports what it found
            words = _tmp_view(torch, tmp, base, sl.off_device_status, 5, torch.int32)
            assert words[0] == 0 and S >= words[1]
            S, G = int(words[1]), int(words[2])
            segs = _tmp_view(torch, tmp, base, sl.off_segments, 2 * G, torch.int32).reshape(G, 2)
            assert np.array_equal(segs[:, 0], tw.seg_offset) and np.array_equal(segs[:, 1], tw.seg_count), "segment table"
        else:
            assert not device_scan or lay.num_scans > 1
        assert S == tw.num_subseq and G == tw.num_segments and sl.num_data_units == tw.num_du
        # the device keeps the destuffed bytes in tiles of 32 subsequences, word-major, every 32-bit word most
        # significant byte first; a subsequence's row holds, around its own W words, the last word of the
        # previous subsequence (slot 0) and the first two of the next one (slots W + 1, W + 2) (jg_defs.h)
        W = subseq_bytes // 4
        R = 16 if W >= 64 else 32  # rows per tile (jg_defs.h)
        tiles = (S + R - 1) // R
        tiled = _tmp_view(torch, tmp, base, sl.off_destuffed, tiles * R * (subseq_bytes + 12), torch.uint8)
        rows = tiled.reshape(tiles, W + 3, R, 4)[..., ::-1].transpose(0, 2, 1, 3).reshape(tiles * R, W + 3, 4)[:S]
        dst = rows[:, 1:W + 1].reshape(-1)
        if S > 1:
            assert np.array_equal(rows[1:, 0], rows[:-1, W]), "slot 0 mirrors the previous row's last word"
            assert np.array_equal(rows[:-1, W + 1:W + 3], rows[1:, 1:3]), "slots W+1, W+2 mirror the next row's first words"
        assert np.array_equal(dst, tw.destuffed), "destuffed bytes"
        seg = _tmp_view(torch, tmp, base, sl.off_segment_index, S, torch.int32)
        assert np.array_equal(seg, tw.seg_index), "segment index"
        ok = tw.p >= 0
        for nm, off, ref in (("p", sl.off_state_p, tw.p), ("n", sl.off_state_n, tw.n), ("cz", sl.off_state_cz, tw.cz)):
            got = _tmp_view(torch, tmp, base, off, S, torch.int32)
            assert np.array_equal(got[ok], ref[ok]), "state " + nm
        # DC-difference sums are kept modulo 2^16, two scan components per 32-bit word
        d01 = _tmp_view(torch, tmp, base, sl.off_state_dc01, S, torch.int32).view(np.uint32)
        d23 = _tmp_view(torch, tmp, base, sl.off_state_dc23, S, torch.int32).view(np.uint32)
        halves = [d01 & 0xFFFF, d01 >> 16, d23 & 0xFFFF, d23 >> 16]
        for k in range(sl.num_components):
            assert np.array_equal(halves[k][ok], tw.dc[k][ok].astype(np.uint32) & 0xFFFF), "state dc%d" % k
        # the write pass emits a symbol stream + a table {first entry, count} per data unit: rebuild the
        # dense stream-order coefficients from it
        coef = gpu_util.stream_coefficients(torch, tmp, base, sl, S)
        assert np.array_equal(coef, tw.stream_coef), "coefficients"


def test_reference_photo_full_size(gpu_lib, torch_cuda, photo_bytes):
    """BASELINE config 1/2 input: the reference's own 12 MP photo, bit-exact vs the oracle."""
    from oracle import oracle

    import jpeggpu_amd

    ref = oracle.decode(photo_bytes)
    got, info = _decode_gpu(photo_bytes)
    assert list(info.sizes_x)[:3] == [4032, 2016, 2016] and list(info.sizes_y)[:3] == [3024, 1512, 1512]
    for c in range(3):
        assert np.array_equal(got[c], ref.planes[c]), "component %d" % c
    # that decode ran the multi-hypothesis speculation (six candidates per subsequence); with it switched off, and at
    # the other subsequence sizes, the planes are the same
    _, _, _tmp, _base, lay = jpeggpu_amd.decode_to_planes(photo_bytes, return_tmp=True)
    assert lay.scans[0].hypotheses == 6 and lay.subsequence_bytes == 64
    for sb in (32, 128):
        planes, _, _tmp, _base, lay = jpeggpu_amd.decode_to_planes(photo_bytes, subseq_bytes=sb, return_tmp=True)
        assert lay.scans[0].hypotheses == 6
        for c in range(3):
            assert np.array_equal(planes[c].cpu().numpy(), ref.planes[c]), "multi-hypothesis, %d bytes, component %d" % (sb, c)
    os.environ["JPEGGPU_MULTI_HYPOTHESIS"] = "0"
    try:
        planes, _, _tmp, _base, lay = jpeggpu_amd.decode_to_planes(photo_bytes, return_tmp=True)
    finally:
        del os.environ["JPEGGPU_MULTI_HYPOTHESIS"]
    assert lay.scans[0].hypotheses == 0
    for c in range(3):
        assert np.array_equal(planes[c].cpu().numpy(), ref.planes[c]), "plain speculation, component %d" % c
    # the same with the marker scan on the device. The 1.17 MB of padding behind EOI is not transferred (host search
    # for the last EOI from the back, reference src/decoder.cpp:175-180 copies the whole file) ...
    planes, _, _tmp, _base, lay = jpeggpu_amd.decode_to_planes(photo_bytes, device_scan=True, return_tmp=True)
    assert lay.scans[0].device_scan and lay.transferred_bytes < 2_910_000
    for c in range(3):
        assert np.array_equal(planes[c].cpu().numpy(), ref.planes[c]), "device scan, component %d" % c
    # ... and a trailer that holds markers of its own (an appended second JPEG) must not confuse the search for the
    # end of the scan
    tail = photo_bytes[:20000] + b"\xff\xd9"
    planes, _ = jpeggpu_amd.decode_to_planes(photo_bytes[:2921333] + tail, device_scan=True)
    for c in range(3):
        assert np.array_equal(planes[c].cpu().numpy(), ref.planes[c]), "device scan with trailer, component %d" % c
