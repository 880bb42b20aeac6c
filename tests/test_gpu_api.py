"""GPU-side API contract, the additive upsample kernel, and the full-size BASELINE configurations."""
import os
import threading

import numpy as np
import pytest

from tests import cases

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_cuda(gpu_lib):
    import torch

    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    return torch


def _alloc(torch, dec, info, pitch_extra=0):
    n = dec.get_buffer_size()
    tmp = torch.empty(n + 256, dtype=torch.uint8, device="cuda:0")
    base = (tmp.data_ptr() + 255) // 256 * 256
    planes = [torch.full((info.sizes_y[c], info.sizes_x[c] + pitch_extra), 0xAB, dtype=torch.uint8, device="cuda:0")
              for c in range(info.num_components)]
    return n, tmp, base, planes


def test_pitch_and_argument_checks(torch_cuda):
    import jpeggpu_amd
    from jpeggpu_amd import Status
    from oracle import oracle

    torch = torch_cuda
    data = cases.matrix()["odd_partial_mcu"]
    ref = oracle.decode(data)
    dec = jpeggpu_amd.Decoder()
    info = dec.parse_header(data)
    n, tmp, base, planes = _alloc(torch, dec, info, pitch_extra=13)  # unaligned pitch: byte-store path
    dec.transfer(base, n, 0)
    dec.decode([p.data_ptr() for p in planes], [p.stride(0) for p in planes], base, n, 0)
    torch.cuda.synchronize()
    for c in range(ref.ncomp):
        got = planes[c].cpu().numpy()
        assert np.array_equal(got[:, :info.sizes_x[c]], ref.planes[c])
        assert (got[:, info.sizes_x[c]:] == 0xAB).all(), "wrote outside the visible plane"
    # pitch smaller than the width, null plane, unaligned / too small tmp (reference decoder.cpp:345-348)
    ptrs, pitches = [p.data_ptr() for p in planes], [p.stride(0) for p in planes]
    for bad_ptrs, bad_pitches, bad_base, bad_n, want in (
            (ptrs, [info.sizes_x[0] - 1] + pitches[1:], base, n, Status.INVALID_ARGUMENT),
            ([0] + ptrs[1:], pitches, base, n, Status.INVALID_ARGUMENT),
            (ptrs, pitches, base + 8, n, Status.INVALID_ARGUMENT),
            (ptrs, pitches, base, n - 256, Status.INTERNAL_ERROR)):
        with pytest.raises(jpeggpu_amd.JpegGpuError) as ei:
            dec.decode(bad_ptrs, bad_pitches, bad_base, bad_n, 0)
        assert ei.value.status == want
    with pytest.raises(jpeggpu_amd.JpegGpuError) as ei:
        dec.transfer(base, n - 256, 0)
    assert ei.value.status == Status.INTERNAL_ERROR
    dec.cleanup()


def test_decoder_reuse_across_geometries_and_streams(torch_cuda):
    import jpeggpu_amd
    from oracle import oracle

    torch = torch_cuda
    m = cases.matrix()
    dec = jpeggpu_amd.Decoder()
    stream = torch.cuda.Stream()
    for name in ("ss_2x2", "four_comp_opt", "gray", "ni_420_dri", "ss_2x2"):
        ref = oracle.decode(m[name])
        info = dec.parse_header(m[name])
        n, tmp, base, planes = _alloc(torch, dec, info)
        dec.transfer(base, n, stream.cuda_stream)
        dec.decode([p.data_ptr() for p in planes], [p.stride(0) for p in planes], base, n, stream.cuda_stream)
        stream.synchronize()
        for c in range(ref.ncomp):
            assert np.array_equal(planes[c].cpu().numpy(), ref.planes[c]), name
    dec.cleanup()


def test_decode_from_bytearray(torch_cuda):
    """A bytearray input is read in place at transfer time (a temporary copy used to dangle)."""
    import gc

    import jpeggpu_amd
    from oracle import oracle

    src = cases.matrix()["dri_7"]
    data = bytearray(src)
    ref = oracle.decode(src)
    for _ in range(3):
        gc.collect()
        _ = [bytes(len(src)) for _ in range(8)]  # churn the allocator where a freed copy would have lived
        planes, _info = jpeggpu_amd.decode_to_planes(data)
        for c in range(ref.ncomp):
            assert np.array_equal(planes[c].cpu().numpy(), ref.planes[c])


def test_two_decoders_on_two_host_threads(torch_cuda):
    import jpeggpu_amd
    from oracle import oracle

    torch = torch_cuda
    m = cases.matrix()
    names = ["multi_seq_dri", "multi_seq_nodri"]
    errors = []

    def work(name):
        try:
            ref = oracle.decode(m[name])
            st = torch.cuda.Stream()
            with torch.cuda.stream(st):
                for _ in range(5):
                    planes, _ = jpeggpu_amd.decode_to_planes(m[name])
                    for c in range(ref.ncomp):
                        if not np.array_equal(planes[c].cpu().numpy(), ref.planes[c]):
                            errors.append((name, c))
        except Exception as e:  # pragma: no cover
            errors.append((name, repr(e)))

    ts = [threading.Thread(target=work, args=(n,)) for n in names]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert not errors, errors


def test_upsample_planes(torch_cuda):
    """Additive chroma-replication kernel vs the integer part of the reference's host helper
    (util/util.h:62-91): dst[y][x] = src[y * sy / sy_max][x * sx / sx_max]."""
    import ctypes as C

    import jpeggpu_amd
    from jpeggpu_amd.api import Img, lib

    torch = torch_cuda
    for name in ("ss_2x2", "ss_4x1", "ss_1x2", "odd_partial_mcu"):
        data = cases.matrix()[name]
        planes, info = jpeggpu_amd.decode_to_planes(data)
        W, H = info.sizes_x[0] * max(info.subsampling.x[:3]) // info.subsampling.x[0], 0
        sxm, sym = max(info.subsampling.x[:3]), max(info.subsampling.y[:3])
        W, H = info.sizes_x[[i for i in range(3) if info.subsampling.x[i] == sxm][0]], \
            info.sizes_y[[i for i in range(3) if info.subsampling.y[i] == sym][0]]
        src, dst = Img(), Img()
        outs = []
        for c in range(3):
            src.image[c], src.pitch[c] = planes[c].data_ptr(), planes[c].stride(0)
            o = torch.zeros((H, W + 3), dtype=torch.uint8, device="cuda:0")
            outs.append(o)
            dst.image[c], dst.pitch[c] = o.data_ptr(), o.stride(0)
        rc = lib().jpeggpu_ext_upsample_planes(C.byref(info), C.byref(src), C.byref(dst), W, H, None)
        assert rc == 0
        torch.cuda.synchronize()
        for c in range(3):
            p = planes[c].cpu().numpy()
            ys = np.minimum(np.arange(H) * info.subsampling.y[c] // sym, p.shape[0] - 1)
            xs = np.minimum(np.arange(W) * info.subsampling.x[c] // sxm, p.shape[1] - 1)
            assert np.array_equal(outs[c].cpu().numpy()[:, :W], p[ys][:, xs]), (name, c)


def test_planes_to_rgbi(torch_cuda):
    """Additive colour kernel vs the oracle's restatement of the reference's host helper conv_to_rgbi
    (util/util.h:62-104). Float arithmetic: the C expression may be contracted into FMAs, so the tolerance
    is 1 LSB, and almost every sample must be exact."""
    import ctypes as C

    import jpeggpu_amd
    from jpeggpu_amd.api import Img, lib
    from oracle import oracle

    torch = torch_cuda
    for name in ("ss_2x2", "ss_1x1", "ss_2x1", "odd_partial_mcu", "odd_17x9", "gray", "q100_noisy"):
        data = cases.matrix()[name]
        planes, info = jpeggpu_amd.decode_to_planes(data)
        n = info.num_components
        sxm, sym = max(info.subsampling.x[:n]), max(info.subsampling.y[:n])
        W = info.sizes_x[[i for i in range(n) if info.subsampling.x[i] == sxm][0]]
        H = info.sizes_y[[i for i in range(n) if info.subsampling.y[i] == sym][0]]
        src = Img()
        for c in range(n):
            src.image[c], src.pitch[c] = planes[c].data_ptr(), planes[c].stride(0)
        pitch = 3 * W + 5  # unaligned rows on purpose
        out = torch.full((H, pitch), 0x5A, dtype=torch.uint8, device="cuda:0")
        rc = lib().jpeggpu_ext_planes_to_rgbi(C.byref(info), C.byref(src), out.data_ptr(), pitch, W, H, None)
        assert rc == 0, name
        torch.cuda.synchronize()
        got = out.cpu().numpy()
        assert (got[:, 3 * W:] == 0x5A).all(), (name, "wrote past the row")
        got = got[:, :3 * W].reshape(H, W, 3).astype(np.int32)
        ref = oracle.planes_to_rgbi([p.cpu().numpy() for p in planes], list(info.subsampling.x), list(info.subsampling.y), W, H)
        diff = np.abs(got - ref.astype(np.int32))
        assert diff.max() <= 1, (name, int(diff.max()))
        assert (diff == 0).mean() > 0.999, (name, float((diff == 0).mean()))
    # 4 components: not supported, as the reference's helper
    data = cases.matrix()["four_comp_444"]
    planes, info = jpeggpu_amd.decode_to_planes(data)
    src = Img()
    for c in range(4):
        src.image[c], src.pitch[c] = planes[c].data_ptr(), planes[c].stride(0)
    out = torch.zeros((8, 64), dtype=torch.uint8, device="cuda:0")
    assert lib().jpeggpu_ext_planes_to_rgbi(C.byref(info), C.byref(src), out.data_ptr(), 64, 8, 8, None) == int(jpeggpu_amd.Status.NOT_SUPPORTED)


def test_batch_decode_equals_single(torch_cuda):
    """jpeggpu_ext_decode_batch (one launch per stage, grid.y = scan) against the oracle, for a batch of
    images of different geometry, scan count and table sets, decoded twice through the same handle."""
    import jpeggpu_amd
    from oracle import oracle

    torch = torch_cuda
    m = cases.matrix()
    names = ["multi_seq_dri", "ni_420_dri", "four_comp_opt", "gray", "multi_seq_nodri", "odd_1x1px", "cfg4_small", "dri_1"]
    names = names * 3 + list(m.keys())[:12]  # > 32 scans: enough for four concurrent parts (set_overlap)
    keep, entries, refs = [], [], []
    total_scans = 0
    sizes = set()
    for k, name in enumerate(names):
        # the subsequence size is the library's per-image choice (lone or batched call type) or the caller's: a batch
        # takes any mix of sizes, one group of launches per size
        dec = jpeggpu_amd.Decoder(128 if k % 5 == 0 else None)
        dec.set_batched(k % 2 == 0)
        dec.set_device_scan(k % 3 == 1)  # a batch may mix host-walked images and images the device scans for markers
        info = dec.parse_header(m[name])
        n, tmp, base, planes = _alloc(torch, dec, info)
        dec.transfer(base, n, 0)
        total_scans += dec.layout().num_scans
        sizes.add(dec.layout().subsequence_bytes)
        keep.append((dec, tmp, planes))
        entries.append((dec, [p.data_ptr() for p in planes], [p.stride(0) for p in planes], base, n))
        refs.append(oracle.decode(m[name]))
    assert len(sizes) >= 3, sizes
    batch = jpeggpu_amd.Batch(total_scans)
    scratch = torch.empty(batch.scratch_size, dtype=torch.uint8, device="cuda:0")
    batch.set_items(entries)
    # cap of the sequence kernel's lock-step loop (>= 1); 255 / 256: one below and at the lane count -- a caller's cap always
    # leaves marks for the tail kernel (ScanParams::tail_marks), however high it is (the kernel used to infer
    # "no marks" from the cap)
    for rep, iters in enumerate([3, 1, 2, 256, 255]):
        batch.set_sync_iterations(iters)
        batch.set_overlap(1 + rep % 4)  # 1..4 concurrent parts on internal streams
        for _, _, planes in keep:
            for p in planes:
                p.fill_(0xCD)
        batch.decode(scratch.data_ptr(), 0)
        torch.cuda.synchronize()
        for name, ref, (dec, _, planes), ent in zip(names, refs, keep, entries):
            assert dec.device_status(ent[3], 0) == jpeggpu_amd.Status.SUCCESS
            for c in range(ref.ncomp):
                assert np.array_equal(planes[c].cpu().numpy(), ref.planes[c]), (name, c, rep)
    # too small a scratch / handle is rejected
    small = jpeggpu_amd.Batch(1)
    small.set_items(entries)
    with pytest.raises(jpeggpu_amd.JpegGpuError):
        small.decode(scratch.data_ptr(), 0)
    small.destroy()
    batch.destroy()
    for dec, _, _ in keep:
        dec.cleanup()


@pytest.mark.parametrize("images", [1, 2, 4, 8, 16])
def test_small_batches_with_the_plans_the_library_picks(torch_cuda, images, monkeypatch):
    """jpeggpu_ext_set_batch_hint: calls of 1 / 2 / 4 / 8 / 16 images, every decoder told the call's size, decoded with
    whatever the library then picks -- one image: the lone decode's plan (multi-hypothesis tables) and its kernels; a
    handful: shorter subsequences and the sequence kernel that keeps every flow in its workgroup; and, forced here for
    the same items, the full batch's one flow iteration + tail kernel -- planes against the oracle each time. The layout
    says what the last call used (a decoder set up for one call type and passed to the other)."""
    import jpeggpu_amd
    from oracle import oracle
    from tools import jpegsynth

    torch = torch_cuda
    m = cases.matrix()
    # two mid-sized 4:2:0 files with restart rows (several sequences, flows that cross them) among small ones of every kind
    mid = [jpegsynth.encode(1616, 1208, cases.S420, True, 101, quality=88, noise=9, seed=70 + k) for k in range(2)]
    pool = mid + [m[k] for k in ("multi_seq_dri", "ni_420_dri", "four_comp_opt", "gray", "multi_seq_nodri", "cfg4_small", "dri_1", "ss_4x1")]
    datas = [pool[i % len(pool)] for i in range(images)]
    refs = [oracle.decode(d) for d in datas]
    # "marks": the full batch's path -- since round 5 with the tail kernel's parts and the write pass's sequences in ONE launch
    # (huff_tail_write: ready queues, tickets); "marks_two_launches": the same with the two kernels one after the other
    for mode in ("auto", "marks", "marks_two_launches"):
        if mode != "auto":
            monkeypatch.setenv("JPEGGPU_EXP_KEEP_FLOWS_BELOW", "0")  # read at jpeggpu_ext_batch_create
        keep, entries = [], []
        for i, d in enumerate(datas):
            dec = jpeggpu_amd.Decoder()
            if i % 3 != 2:
                dec.set_batch_hint(images)  # (every third decoder keeps the lone plan: any decoder may go into any call)
            if images == 1 and mode != "auto":
                dec.set_device_scan(2)  # the checked mode blocks jpeggpu_decoder_decode only: an item of a batch is never waited for
            info = dec.parse_header(d)
            n, tmp, base, planes = _alloc(torch, dec, info)
            dec.transfer(base, n, 0)
            keep.append((dec, tmp, planes))
            entries.append((dec, [p.data_ptr() for p in planes], [p.stride(0) for p in planes], base, n))
        lay0 = keep[0][0].layout()
        if images == 1:
            assert lay0.subsequence_bytes in (32, 64) and lay0.scans[0].hypotheses == 6, (lay0.subsequence_bytes, lay0.scans[0].hypotheses)
        else:
            assert lay0.subsequence_bytes == 128 and lay0.scans[0].hypotheses == 0, (lay0.subsequence_bytes, lay0.scans[0].hypotheses)
            assert lay0.subsequences_per_sequence == 255  # before any call: what a full batch would use
        batch = jpeggpu_amd.Batch(sum(k[0].layout().num_scans for k in keep))
        scratch = torch.empty(batch.scratch_size, dtype=torch.uint8, device="cuda:0")
        batch.set_items(entries)
        if mode == "marks_two_launches":
            batch.set_fused_tail(False)
        for rep in range(2):
            for _, _, planes in keep:
                for p in planes:
                    p.fill_(0xCD)
            batch.decode(scratch.data_ptr(), 0)
            torch.cuda.synchronize()
            for i, (ref, (dec, _, planes)) in enumerate(zip(refs, keep)):
                assert dec.device_status(entries[i][3], 0) == jpeggpu_amd.Status.SUCCESS
                for c in range(ref.ncomp):
                    assert np.array_equal(planes[c].cpu().numpy(), ref.planes[c]), (images, mode, i, c, rep)
        # what the call used: a small call keeps its flows in the sequence kernel (240 + 16 lanes), a full one does not
        used = [k[0].layout().subsequences_per_sequence for k in keep]
        assert used == [240 if mode == "auto" or images == 1 else 255] * images, (mode, used)
        assert jpeggpu_amd.fused_tail_timeouts() == 0
        if images >= 2:
            # ... and the drop-in call on a decoder that was set up for batches reports its own geometry afterwards
            dec, _, planes = keep[0]
            ent = entries[0]
            dec.decode(ent[1], ent[2], ent[3], ent[4], 0)
            torch.cuda.synchronize()
            assert dec.layout().subsequences_per_sequence == 240
            for c in range(refs[0].ncomp):
                assert np.array_equal(planes[c].cpu().numpy(), refs[0].planes[c])
        batch.destroy()
        for dec, _, _ in keep:
            dec.cleanup()


def test_corrupt_entropy_data_is_memory_safe(torch_cuda, monkeypatch):
    """Random damage inside the entropy-coded segment (no new markers): the planes are garbage by
    definition, but every decode must complete, stay inside its buffers, and leave the decoder and the
    device usable -- the guards that matter for serving untrusted files (quota, region capacity,
    table clamps, bounded flows). Canary bytes around tmp and the planes must survive. Every damaged file also goes
    through the batched call, with the kernels of a small call and of one that fills the chip (round 5: the fused
    launch, whose writers wait for parts -- which must finish whatever the bits say)."""
    import jpeggpu_amd
    from oracle import oracle

    torch = torch_cuda
    rng = np.random.default_rng(1234)
    m = cases.matrix()
    for name in ("multi_seq_dri", "multi_seq_nodri", "four_comp_opt", "ni_420_dri"):
        good = m[name]
        ref = oracle.decode(good)
        lo, hi = oracle.scan_info(good, 0, 128).scan_begin, oracle.scan_info(good, 0, 128).scan_end
        for trial in range(6):
            bad = bytearray(good)
            for pos in rng.integers(lo + 4, hi - 4, size=int(rng.integers(1, 40))):
                v = int(rng.integers(0, 255))          # never 0xFF: the marker structure stays intact
                if bad[pos] != 0xFF and bad[pos - 1] != 0xFF:
                    bad[pos] = v
            dec = jpeggpu_amd.Decoder(int(rng.choice([32, 64, 128])))
            dec.set_device_scan(bool(trial & 1))  # half of the trials find the markers on the device
            try:
                info = dec.parse_header(bytes(bad))
            except jpeggpu_amd.JpegGpuError:
                dec.cleanup()
                continue
            n = dec.get_buffer_size()
            guard = 4096
            tmp = torch.full((n + 256 + 2 * guard,), 0x5A, dtype=torch.uint8, device="cuda:0")
            base = (tmp.data_ptr() + guard + 255) // 256 * 256
            planes = []
            for c in range(info.num_components):
                buf = torch.full((info.sizes_y[c] + 2, info.sizes_x[c]), 0x5A, dtype=torch.uint8, device="cuda:0")
                planes.append(buf)
            dec.transfer(base, n, 0)
            dec.decode([p[1:-1].data_ptr() for p in planes], [p.stride(0) for p in planes], base, n, 0)
            torch.cuda.synchronize()
            off = base - tmp.data_ptr()
            assert (tmp[:off] == 0x5A).all() and (tmp[off + n:] == 0x5A).all(), (name, trial, "tmp overrun")
            for p in planes:
                assert (p[0] == 0x5A).all() and (p[-1] == 0x5A).all(), (name, trial, "plane overrun")
            dec.cleanup()
            # the same bytes as an item of a batched call
            for below in ("220000", "0"):
                monkeypatch.setenv("JPEGGPU_EXP_KEEP_FLOWS_BELOW", below)  # read at jpeggpu_ext_batch_create
                dec = jpeggpu_amd.Decoder(int(rng.choice([32, 64, 128, 256])))
                dec.set_batch_hint(64)
                dec.set_device_scan(bool(trial & 2))
                info = dec.parse_header(bytes(bad))
                n = dec.get_buffer_size()
                tmp = torch.full((n + 256 + 2 * guard,), 0x5A, dtype=torch.uint8, device="cuda:0")
                base = (tmp.data_ptr() + guard + 255) // 256 * 256
                planes = [torch.full((info.sizes_y[c] + 2, info.sizes_x[c]), 0x5A, dtype=torch.uint8, device="cuda:0") for c in range(info.num_components)]
                dec.transfer(base, n, 0)
                batch = jpeggpu_amd.Batch(dec.layout().num_scans)
                scratch = torch.empty(batch.scratch_size, dtype=torch.uint8, device="cuda:0")
                batch.set_items([(dec, [p[1:-1].data_ptr() for p in planes], [p.stride(0) for p in planes], base, n)])
                batch.decode(scratch.data_ptr(), 0)
                torch.cuda.synchronize()
                off = base - tmp.data_ptr()
                assert (tmp[:off] == 0x5A).all() and (tmp[off + n:] == 0x5A).all(), (name, trial, below, "tmp overrun (batch)")
                for p in planes:
                    assert (p[0] == 0x5A).all() and (p[-1] == 0x5A).all(), (name, trial, below, "plane overrun (batch)")
                batch.destroy()
                dec.cleanup()
        assert jpeggpu_amd.fused_tail_timeouts() == 0
        # the device still decodes correctly afterwards
        planes, _ = jpeggpu_amd.decode_to_planes(good)
        for c in range(ref.ncomp):
            assert np.array_equal(planes[c].cpu().numpy(), ref.planes[c]), name


def test_corrupt_headers_are_memory_safe(torch_cuda):
    """Random damage in the header segments (tables, frame and scan headers, restart interval): whatever the
    parser still accepts must decode inside its buffers. Absurd geometries are skipped (allocation size)."""
    import jpeggpu_amd

    torch = torch_cuda
    rng = np.random.default_rng(4321)
    m = cases.matrix()
    accepted = 0
    for name in ("ss_2x2", "dri_7", "four_comp_opt", "ni_420_dri", "opt_tables_420", "q16_tables"):
        good = m[name]
        sos = good.index(b"\xff\xda")
        for trial in range(40):
            bad = bytearray(good)
            for pos in rng.integers(2, sos + 12, size=int(rng.integers(1, 4))):
                bad[pos] = int(rng.integers(0, 256)) if rng.random() < 0.5 else bad[pos] ^ (1 << int(rng.integers(8)))
            dec = jpeggpu_amd.Decoder(int(rng.choice([32, 64, 128, 256])))
            dec.set_device_scan(bool(trial & 1))
            try:
                info = dec.parse_header(bytes(bad))
                n = dec.get_buffer_size()
            except jpeggpu_amd.JpegGpuError:
                dec.cleanup()
                continue
            px = sum(info.sizes_x[c] * info.sizes_y[c] for c in range(info.num_components))
            if px > 48 << 20 or n > 1 << 30:
                dec.cleanup()
                continue
            accepted += 1
            guard = 4096
            tmp = torch.full((n + 256 + 2 * guard,), 0x5A, dtype=torch.uint8, device="cuda:0")
            base = (tmp.data_ptr() + guard + 255) // 256 * 256
            planes = [torch.full((info.sizes_y[c] + 2, info.sizes_x[c]), 0x5A, dtype=torch.uint8, device="cuda:0")
                      for c in range(info.num_components)]
            dec.transfer(base, n, 0)
            dec.decode([p[1:-1].data_ptr() for p in planes], [p.stride(0) for p in planes], base, n, 0)
            torch.cuda.synchronize()
            off = base - tmp.data_ptr()
            assert (tmp[:off] == 0x5A).all() and (tmp[off + n:] == 0x5A).all(), (name, trial, "tmp overrun")
            for p in planes:
                assert (p[0] == 0x5A).all() and (p[-1] == 0x5A).all(), (name, trial, "plane overrun")
            dec.cleanup()
    assert accepted > 20  # the test must actually exercise the device
    planes, _ = jpeggpu_amd.decode_to_planes(m["ss_2x2"])
    from oracle import oracle

    ref = oracle.decode(m["ss_2x2"])
    for c in range(ref.ncomp):
        assert np.array_equal(planes[c].cpu().numpy(), ref.planes[c])


def _decode_device_scan(torch, jpeggpu_amd, data, subseq_bytes):
    dec = jpeggpu_amd.Decoder(subseq_bytes)
    dec.set_device_scan(True)
    info = dec.parse_header(data)
    lay = dec.layout()
    n = dec.get_buffer_size()
    tmp = torch.full((n + 256 + 8192,), 0x5A, dtype=torch.uint8, device="cuda:0")
    base = (tmp.data_ptr() + 4096 + 255) // 256 * 256
    planes = [torch.full((info.sizes_y[c] + 2, info.sizes_x[c]), 0x5A, dtype=torch.uint8, device="cuda:0")
              for c in range(info.num_components)]
    dec.transfer(base, n, 0)
    dec.decode([p[1:-1].data_ptr() for p in planes], [p.stride(0) for p in planes], base, n, 0)
    status = dec.device_status(base, 0)
    torch.cuda.synchronize()
    off = base - tmp.data_ptr()
    assert (tmp[:off] == 0x5A).all() and (tmp[off + n:] == 0x5A).all(), "tmp overrun"
    for p in planes:
        assert (p[0] == 0x5A).all() and (p[-1] == 0x5A).all(), "plane overrun"
    words = None
    last = lay.scans[lay.num_scans - 1]  # the scan the device walks, if any
    if last.device_scan:
        w = tmp[off + last.off_device_status: off + last.off_device_status + 32].cpu().numpy().view(np.uint32)
        words = [int(x) for x in w[:5]]
    dec.cleanup()
    return status, [p[1:-1].cpu().numpy() for p in planes], lay, words


def test_device_side_marker_scan(torch_cuda):
    """jpeggpu_ext_set_device_scan: the device finds the restart markers and builds segment table,
    destuff work list and tail parts; planes bit-exact vs the oracle, counts equal to the host walk's; streams the
    host walk would have refused at parse time report their status from the device and leave the planes alone."""
    import jpeggpu_amd
    from jpeggpu_amd import Status
    from oracle import oracle

    torch = torch_cuda
    m = cases.matrix()
    took_device_path = multi_scan = 0
    for name, data in m.items():
        ref = oracle.decode(data)
        for sb in (128, 32):
            status, planes, lay, words = _decode_device_scan(torch, jpeggpu_amd, data, sb)
            k = lay.num_scans - 1
            # the last scan is walked on the device -- of several only if it is the bulk of the bytes --, the scans in front
            # of it on the host
            assert lay.num_scans == ref.nscans and not any(lay.scans[i].device_scan for i in range(k)), name
            assert lay.scans[k].device_scan == (0 if name.startswith(("ni_4", "cfg4")) else 1), name
            multi_scan += k > 0 and lay.scans[k].device_scan
            if not lay.scans[k].device_scan:
                assert status == Status.SUCCESS
            else:
                host = jpeggpu_amd.Decoder(sb)
                host.parse_header(data)
                hl = host.layout().scans[k]
                host.cleanup()
                assert status == Status.SUCCESS, (name, sb, status)
                took_device_path += 1
                assert words[0] == 0 and words[1] == hl.num_subsequences and words[2] == hl.num_segments, (name, sb, words)
                assert words[3] == hl.num_chunks, (name, sb, words, hl.num_chunks)
            if status == Status.SUCCESS:
                for c in range(ref.ncomp):
                    assert np.array_equal(planes[c], ref.planes[c]), (name, sb, c)
    assert took_device_path > 40 and multi_scan == 4  # every single-scan case, however dense its restart markers; two files whose last scan is the bulk

    # what the host walk refuses at parse time comes back from the device; the planes are not written
    good = m["multi_seq_dri"]
    cut = good[: len(good) * 2 // 3]                       # no terminating marker
    status, planes, _, _ = _decode_device_scan(torch, jpeggpu_amd, cut, 128)
    assert status == Status.INVALID_JPEG and all((p == 0x5A).all() for p in planes)  # the host walk's code for it
    sos = good.index(b"\xff\xda")
    dri = good.index(b"\xff\xdd")
    assert dri < sos
    wrong = bytearray(good)
    wrong[dri + 5] ^= 0x20                                 # a restart interval (64 -> 96 MCUs) the markers do not follow
    status, planes, _, _ = _decode_device_scan(torch, jpeggpu_amd, bytes(wrong), 128)
    assert status == Status.INVALID_JPEG and all((p == 0x5A).all() for p in planes)


def test_device_scan_agrees_with_host_walk_on_refusals(torch_cuda, monkeypatch):
    """What the host walk refuses at parse time gets the same status from the device: an FF FF 00 inside
    the scan (neither fill byte + marker nor stuffing), a scan no marker ends. And the checked mode that
    JPEGGPU_DEVICE_SCAN=1 / 2 / checked selects for callers of the drop-in API alone: decode itself returns that status
    (=async keeps decode asynchronous: success at enqueue time whatever the stream holds)."""
    import jpeggpu_amd
    from jpeggpu_amd import JpegGpuError, Status

    torch = torch_cuda
    good = cases.matrix()["dri_fill"]  # restart markers with FF fill bytes in front: legal
    rst = good.index(b"\xff\xff\xff\xd0")
    ffzero = good[:rst] + b"\xff\xff\x00" + good[rst:]          # fill byte followed by a stuffed FF
    cut = good[: len(good) * 2 // 3]
    behind_eoi = good + b"\xff\xff\x00" + bytes(100)             # the same sequence behind the image: not the scan's business
    empty_seg = cases.empty_segment_case()                       # two restart markers back to back, segment COUNT as the geometry wants
    three = cases.big_last_scan_case(restart_interval=5, seed=42) # two scans, the last one the device's: refused like a lone one
    three_cut = three[: len(three) - 40]                          # ... its terminating marker is gone
    three_bad = three[:-2] + b"\xff\xff\x00" + three[-2:]         # ... FF FF 00 in front of it
    for name, data, want in (("good", good, Status.SUCCESS), ("ffzero", ffzero, Status.INVALID_JPEG),
                             ("cut", cut, Status.INVALID_JPEG), ("behind_eoi", behind_eoi, Status.SUCCESS),
                             ("empty_segment", empty_seg, Status.INVALID_JPEG), ("two_scans", three, Status.SUCCESS),
                             ("two_scans_cut", three_cut, Status.INVALID_JPEG), ("two_scans_ffff00", three_bad, Status.INVALID_JPEG)):
        host = jpeggpu_amd.Decoder()
        try:
            host.parse_header(data)
            host_status = Status.SUCCESS
        except JpegGpuError as e:
            host_status = e.status
        host.cleanup()
        assert host_status == want, (name, host_status)
        status, planes, lay, _ = _decode_device_scan(torch, jpeggpu_amd, data, 128)
        assert lay.scans[lay.num_scans - 1].device_scan and status == want, (name, status)
        # nothing of a refused scan is written (the host-walked scans in front of it, if any, are decoded as usual)
        assert want == Status.SUCCESS or all((p == 0x5A).all() for p in planes[lay.num_scans - 1:])
        # checked mode through the environment variable, drop-in calls only
        for env, expect in (("2", want), ("1", want), ("checked", want), ("async", Status.SUCCESS)):
            monkeypatch.setenv("JPEGGPU_DEVICE_SCAN", env)
            dec = jpeggpu_amd.Decoder()
            monkeypatch.delenv("JPEGGPU_DEVICE_SCAN")
            info = dec.parse_header(data)
            assert dec.layout().scans[dec.layout().num_scans - 1].device_scan
            n, tmp, base, pl = _alloc(torch, dec, info)
            dec.transfer(base, n, 0)
            try:
                dec.decode([p.data_ptr() for p in pl], [p.stride(0) for p in pl], base, n, 0)
                got = Status.SUCCESS
            except JpegGpuError as e:
                got = e.status
            assert got == expect, (name, env, got)
            torch.cuda.synchronize()
            dec.cleanup()


@pytest.mark.parametrize("full_batch_kernels", [False, True])
def test_batch_with_bad_device_scanned_items(torch_cuda, monkeypatch, full_batch_kernels):
    """A batch in which some device-scanned items turn out to be unusable on the device (scan without terminating
    marker, restart markers that do not follow DRI): those report their status and leave their planes alone, the
    other items of the same launches decode bit-exact. Also through the kernels of a call that fills the chip (round 5:
    huff_tail_write hands out its roles by the launch's extents; the refused jobs hold no part and no sequence)."""
    if full_batch_kernels:
        monkeypatch.setenv("JPEGGPU_EXP_KEEP_FLOWS_BELOW", "0")  # read at jpeggpu_ext_batch_create
    import jpeggpu_amd
    from jpeggpu_amd import Status
    from oracle import oracle

    torch = torch_cuda
    m = cases.matrix()
    good = m["multi_seq_dri"]
    cut = good[: len(good) * 2 // 3]
    wrong = bytearray(good)
    wrong[good.index(b"\xff\xdd") + 5] ^= 0x20
    items = [("good_host", good, False, Status.SUCCESS), ("cut", cut, True, Status.INVALID_JPEG),
             ("good_dev", good, True, Status.SUCCESS), ("wrong_dri", bytes(wrong), True, Status.INVALID_JPEG),
             ("other_dev", m["dri_row"], True, Status.SUCCESS), ("gray_host", m["gray"], False, Status.SUCCESS)] * 3
    keep, entries = [], []
    for name, data, dev, _ in items:
        dec = jpeggpu_amd.Decoder()
        dec.set_device_scan(dev)
        info = dec.parse_header(data)
        n, tmp, base, planes = _alloc(torch, dec, info)
        dec.transfer(base, n, 0)
        keep.append((dec, tmp, planes))
        entries.append((dec, [p.data_ptr() for p in planes], [p.stride(0) for p in planes], base, n))
    batch = jpeggpu_amd.Batch(len(items))
    scratch = torch.empty(batch.scratch_size, dtype=torch.uint8, device="cuda:0")
    batch.set_items(entries)
    for overlap in (1, 2):
        batch.set_overlap(overlap)
        for _, _, planes in keep:
            for p in planes:
                p.fill_(0xAB)
        batch.decode(scratch.data_ptr(), 0)
        torch.cuda.synchronize()
        for (name, data, dev, want), (dec, _, planes), ent in zip(items, keep, entries):
            assert dec.device_status(ent[3], 0) == want, name
            if want == Status.SUCCESS:
                ref = oracle.decode(data)
                for c in range(ref.ncomp):
                    assert np.array_equal(planes[c].cpu().numpy(), ref.planes[c]), (name, c, overlap)
            else:
                assert all((p == 0xAB).all() for p in planes), name
    assert jpeggpu_amd.fused_tail_timeouts() == 0
    batch.destroy()
    for dec, _, _ in keep:
        dec.cleanup()


def test_c_caller_decodes_the_reference_photo(torch_cuda, tmp_path):
    """examples/decode_file.c (plain C, HIP runtime, no Python in the process) on the reference's photo: the
    planes it writes carry the committed hashes of the oracle's planes (tests/golden/photo_pins.json)."""
    import hashlib
    import json
    import os
    import subprocess

    from tests.test_host_api import _build_c_example

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "decode_file")
    _build_c_example(exe)
    prefix = str(tmp_path / "photo")
    out = subprocess.run([exe, os.path.join(root, "tests", "golden", "IMG_6510.JPG"), prefix, "--rgb"],
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
    assert out.returncode == 0, out.stderr.decode()
    assert b"3 components, 4032x3024" in out.stdout
    pins = json.load(open(os.path.join(root, "tests", "golden", "photo_pins.json")))
    for c, (w, h) in enumerate(((4032, 3024), (2016, 1512), (2016, 1512))):
        raw = open("%s_%d.pgm" % (prefix, c), "rb").read()
        header = b"P5\n%d %d\n255\n" % (w, h)
        assert raw.startswith(header) and len(raw) == len(header) + w * h
        assert hashlib.sha256(raw[len(header):]).hexdigest() == pins["components"][c]["sha256_oracle_plane"], c
    assert os.path.getsize(prefix + ".ppm") == len(b"P6\n4032 3024\n255\n") + 4032 * 3024 * 3


def test_random_soak_short(torch_cuda, monkeypatch):
    """A few seconds of tools/soak_gpu.py: random geometry / sampling / restart interval / quality / tables,
    all subsequence sizes, random sync iterations and overlap parts, batch and drop-in calls, bit-exact."""
    import sys

    from tools import soak_gpu

    monkeypatch.setattr(sys, "argv", ["soak_gpu.py", "8", "20261004"])
    soak_gpu.main()


def test_extreme_geometry(torch_cuda):
    """The largest dimensions a JPEG frame header can carry, in both orientations, with partial MCUs at the
    far edge, restart intervals that do not divide the MCU count, and a 1-pixel-wide column: planes
    bit-exact vs the oracle."""
    import jpeggpu_amd
    from oracle import oracle
    from tools import jpegsynth

    S420, S444 = ((2, 2), (1, 1), (1, 1)), ((1, 1),) * 3
    for w, h, ss, kw in ((65535, 17, S420, dict(restart_interval=1000)), (17, 65535, S420, dict(restart_interval=7)),
                         (1, 4099, S444, {}), (40000, 9, ((4, 1), (1, 1), (1, 1)), dict(optimize=True))):
        data = jpegsynth.encode(w, h, ss, seed=w + h, noise=10, **kw)
        ref = oracle.decode(data)
        for device_scan in (False, True):
            planes, info = jpeggpu_amd.decode_to_planes(data, device_scan=device_scan)
            assert (info.sizes_x[0], info.sizes_y[0]) == (w, h)
            for c in range(ref.ncomp):
                assert np.array_equal(planes[c].cpu().numpy(), ref.planes[c]), (w, h, c, device_scan)


def test_dense_restart_markers_full_size(torch_cuda):
    """A restart marker behind every MCU of a 12 MP image (47 628 segments of ~36 bytes) and behind every second
    MCU of a 48 MP one (93 750 segments): host walk and device-side marker scan, bit-exact vs the oracle."""
    import jpeggpu_amd
    from oracle import oracle
    from tools import jpegsynth

    S420 = ((2, 2), (1, 1), (1, 1))
    for w, h, dri in ((4032, 3024, 1), (8000, 6000, 2)):
        data = jpegsynth.encode(w, h, S420, restart_interval=dri, quality=80, noise=6, seed=w + dri)
        ref = oracle.decode(data)
        for device_scan in (False, True):
            planes, _ = jpeggpu_amd.decode_to_planes(data, device_scan=device_scan)
            for c in range(3):
                assert np.array_equal(planes[c].cpu().numpy(), ref.planes[c]), (w, h, dri, c, device_scan)


@pytest.mark.parametrize("cfg", [2, 4, 5])
def test_baseline_configs_full_size(torch_cuda, cfg):
    """BASELINE.json configs 2, 4 (39 MP, three non-interleaved scans) and 5 (4 components, 4+4
    tables, no restart markers) at full size, bit-exact against the oracle."""
    import jpeggpu_amd
    from oracle import oracle
    from tools import jpegsynth

    data = jpegsynth.config(cfg, seed=5)
    ref = oracle.decode(data)
    for device_scan in (False, True):  # (cfg 4's last scan is a fifth of its bytes: the host walks all three either way)
        planes, info = jpeggpu_amd.decode_to_planes(data, device_scan=device_scan)
        assert info.num_components == ref.ncomp
        for c in range(ref.ncomp):
            assert np.array_equal(planes[c].cpu().numpy(), ref.planes[c]), (cfg, c, device_scan)


def test_self_test_passes_and_its_constants_are_the_oracles(torch_cuda):
    """jpeggpu_ext_self_test: the built-in image decodes to the stored plane hashes on this system; and those constants
    (jpeggpu_amd/csrc/jg_selftest_data.h, data made once) are what the oracle says about that image: the array is read
    back out of the header and decoded by the oracle here."""
    import re

    import jpeggpu_amd
    from oracle import oracle
    from tests.conftest import ROOT

    jpeggpu_amd.self_test()
    text = open(os.path.join(ROOT, "jpeggpu_amd", "csrc", "jg_selftest_data.h")).read()
    body = text[text.index("kSelfTestJpeg[] = {"):]
    data = bytes(int(x, 16) for x in re.findall(r"0x([0-9a-f]{2})\b", body[:body.index("};")]))
    want = [int(x, 16) for x in re.findall(r"0x([0-9a-f]{16})ull", text)]
    ref = oracle.decode(data)
    assert len(data) == 5325 and ref.ncomp == 3 and len(want) == 3

    def fnv(b):
        h = 0xcbf29ce484222325
        for x in b:
            h = ((h ^ x) * 0x100000001b3) & 0xFFFFFFFFFFFFFFFF
        return h

    assert [fnv(p.tobytes()) for p in ref.planes] == want
    planes, _ = jpeggpu_amd.decode_to_planes(data)
    for c in range(3):
        assert np.array_equal(planes[c].cpu().numpy(), ref.planes[c])


def test_block_wise_multi_hypothesis_walk_without_restart_markers(torch_cuda, monkeypatch):
    """Scans without restart markers -- most JPEGs in the wild -- are ONE segment: the multi-hypothesis chain is walked
    block by block (huff_mh_block_maps / _chain / huff_mh_resolve per block, jg_defs.h). A 12 MP 4:2:0 file without DRI
    (~45 blocks of 1024 subsequences at 64 bytes), BASELINE configs[4] (4 components, 8 tables) and a restart interval
    long enough for segments of more than 1024 subsequences: the layout says the blocks are used, planes bit-exact vs the
    oracle at three subsequence sizes, and the same with the speculation switched off."""
    import jpeggpu_amd
    from oracle import oracle
    from tools import jpegsynth

    S420 = ((2, 2), (1, 1), (1, 1))
    inputs = {"12mp_nodri": jpegsynth.encode(4032, 3024, S420, True, 0, quality=88, noise=9, seed=77),
              "411_nodri": jpegsynth.encode(2560, 1920, ((4, 1), (1, 1), (1, 1)), True, 0, quality=90, noise=10, seed=79),
              "long_segments": jpegsynth.encode(2048, 1536, S420, True, 128 * 24, quality=92, noise=12, seed=78)}
    for name, data in inputs.items():
        ref = oracle.decode(data)
        for sb in (0, 32, 128):
            planes, _, _tmp, _base, lay = jpeggpu_amd.decode_to_planes(data, subseq_bytes=sb, return_tmp=True)
            sc = lay.scans[0]
            assert sc.hypotheses == sc.data_units_per_mcu == 6, (name, sb)
            assert sc.hypothesis_blocks >= (sc.num_subsequences + 1023) // 1024 > 1, (name, sb, sc.hypothesis_blocks)
            for c in range(ref.ncomp):
                assert np.array_equal(planes[c].cpu().numpy(), ref.planes[c]), (name, sb, c)
        if name.endswith("nodri"):
            # the same with the device-side marker scan (round 5): the scan is one segment whose length only the device
            # knows -- it builds the block list itself (jg_front.hip, front_plan), the host sizes it from the header's bound
            for sb in (0, 32):
                planes, _, _tmp, _base, lay = jpeggpu_amd.decode_to_planes(data, subseq_bytes=sb, return_tmp=True, device_scan=True)
                sc = lay.scans[0]
                assert sc.device_scan == 1 and sc.hypotheses == 6, (name, sb, sc.hypotheses)
                assert sc.hypothesis_blocks >= (sc.num_subsequences + 1023) // 1024 > 1, (name, sb, sc.hypothesis_blocks)
                for c in range(ref.ncomp):
                    assert np.array_equal(planes[c].cpu().numpy(), ref.planes[c]), (name, sb, c, "device scan")
    # BASELINE configs[4] -- runs of two data units with the same tables, no restart markers -- synchronises faster without
    # (measured, jg_decoder.cpp make_plan): the library does not apply the speculation there
    planes, _, _tmp, _base, lay = jpeggpu_amd.decode_to_planes(jpegsynth.config(5, small=True), return_tmp=True)
    assert lay.scans[0].hypotheses == 6 and lay.scans[0].hypothesis_blocks == 0  # (a short scan: one segment, walked whole)
    dec = jpeggpu_amd.Decoder()
    dec.parse_header(jpegsynth.config(5, seed=3))
    assert dec.layout().scans[0].hypotheses == 0
    dec.cleanup()
    monkeypatch.setenv("JPEGGPU_MULTI_HYPOTHESIS", "0")
    data = inputs["12mp_nodri"]
    ref = oracle.decode(data)
    planes, _, _tmp, _base, lay = jpeggpu_amd.decode_to_planes(data, return_tmp=True)
    assert lay.scans[0].hypotheses == 0
    for c in range(3):
        assert np.array_equal(planes[c].cpu().numpy(), ref.planes[c]), c


def test_config3_batch_of_64_twelve_megapixel_images(torch_cuda):
    """BASELINE.json configs[2] on one GPU: 64 x 12 MP 4:2:0 (8 distinct seeds) through jpeggpu_ext_decode_batch in one
    call, plane hashes against the oracle. (Across GPUs the same batch is sharded by image: bench.py's `gather`.)"""
    import hashlib

    import jpeggpu_amd
    from oracle import oracle
    from tools import jpegsynth

    torch = torch_cuda
    datas = [jpegsynth.config(2, seed=100 + s) for s in range(8)]
    want = []
    for d in datas:
        ref = oracle.decode(d)
        want.append([hashlib.sha256(p.tobytes()).hexdigest() for p in ref.planes])
    keep, entries = [], []
    for i in range(64):
        dec = jpeggpu_amd.Decoder()
        dec.set_batched(True)  # the library picks the subsequence size: 256 bytes for these
        if i % 2:
            dec.set_device_scan(True)
        info = dec.parse_header(datas[i % 8])
        n, tmp, base, planes = _alloc(torch, dec, info)
        dec.transfer(base, n, 0)
        keep.append((dec, tmp, planes))
        entries.append((dec, [p.data_ptr() for p in planes], [p.stride(0) for p in planes], base, n))
    batch = jpeggpu_amd.Batch(64)
    scratch = torch.empty(batch.scratch_size, dtype=torch.uint8, device="cuda:0")
    batch.set_items(entries)
    batch.set_overlap(4)
    batch.decode(scratch.data_ptr(), 0)
    torch.cuda.synchronize()
    bad = []
    for i, (dec, _tmp, planes) in enumerate(keep):
        got = [hashlib.sha256(p.cpu().numpy().tobytes()).hexdigest() for p in planes]
        if got != want[i % 8]:
            bad.append(i)
        dec.cleanup()
    batch.destroy()
    assert not bad, bad


def test_bench_multi_rank_control_flow_over_gloo(torch_cuda):
    """bench.py's own N > 1 path -- sharding by rank, max-over-ranks timing, oracle verification reduced over the
    ranks, the configs[2] leg with its gather -- as two ranks on this one GPU (JPEGGPU_BENCH_BACKEND=gloo: the
    collectives run on CPU tensors; the driver's multi-GPU runs use RCCL)."""
    import json
    import os
    import socket
    import subprocess
    import sys

    from tests.conftest import ROOT

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    # started exactly as the driver starts N = 1 -- plain `python bench.py --gpus 2 ...`, no WORLD_SIZE in the environment:
    # bench.py launches its own ranks as a child process (torch.distributed.run) and exits with its code
    env = dict(os.environ, JPEGGPU_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_PORT=str(port))
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--batch", "8", "--rounds", "1", "--unique", "2", "--latency-iters", "0", "--other-configs", "0", "--no-cpu", "--e2e-rounds", "0",
           "--roofline-launches", "2", "--gather-rounds", "2", "--segment-shard-rounds", "2"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 2 and d["verified"] is True and d["verified_images"] == 2
    assert d["config"]["images_per_step"] == 16 and d["value"] > 0
    g = d["gather"]
    assert g["images_per_round"] == 64 or g["images_per_round"] == 16  # 64 // 2 per rank, capped by the batch
    assert g["gathered_buffers_match_senders"] is True and g["value"] > 0
    # the leg is timed three ways -- decode alone, gather alone, and the pipeline in which the gather of round k runs
    # beside the decode of round k + 1 (two plane buffers) --, all present and consistent
    assert g["decode_ms"] > 0 and g["gather_ms"] > 0 and g["overlapped_ms"] > 0
    assert g["ms_per_round"] == g["overlapped_ms"] and abs(g["value"] - g["images_per_round"] / (g["overlapped_ms"] * 1e-3)) < 1e-6 * g["value"]
    assert d["roofline"]["frac"] > 0
    sh = d["segment_shard"]  # one 39 MP image over the two ranks by restart segments
    assert sh["assembled_equals_whole_decode"] is True and len(sh["rows_of_plane_0_per_rank"]) == 2 and sh["value"] > 0


def test_scan_larger_than_16_mib(torch_cuda):
    """one scan of 19.4 MiB in ONE restart segment -- 158 632 subsequences of 128
    bytes, more than the 131 072 one block of the reference's inter-sequence kernel covers (its supersequences are
    not synchronised with each other, src/decode_huffman.cu:548-558). Planes bit-exact vs the oracle, host walk and
    device scan."""
    import jpeggpu_amd
    from oracle import oracle
    from tools import jpegsynth

    data = jpegsynth.encode(7216, 5408, ((1, 1), (1, 1), (1, 1)), True, 0, quality=93, noise=10, seed=3)
    lay = oracle.scan_info(data, 0, 128)
    assert lay.scan_end - lay.scan_begin > (16 << 20) and lay.num_segments == 1 and lay.num_subseq > 131072
    ref = oracle.decode(data)
    for device_scan in (False, True):
        planes, _info = jpeggpu_amd.decode_to_planes(data, subseq_bytes=128, device_scan=device_scan)
        for c in range(3):
            assert np.array_equal(planes[c].cpu().numpy(), ref.planes[c]), (device_scan, c)
        del planes


def test_segment_shard_bands_make_the_image(torch_cuda):
    """jpeggpu_ext_set_segment_shard: `world` decoders each take a share of the restart segments and write only their
    band of every plane; together the bands are the oracle's image, and nothing outside a band is touched."""
    import jpeggpu_amd
    from jpeggpu_amd import JpegGpuError, Status
    from oracle import oracle
    from tools import jpegsynth

    torch = torch_cuda
    m = cases.matrix()
    inputs = {"dri_row": m["dri_row"], "cfg2_small": m["cfg2_small"], "gray_rows": jpegsynth.encode(200, 152, ((1, 1),), restart_interval=50, seed=77),
              "two_rows": jpegsynth.encode(333, 251, cases.S420, restart_interval=42, seed=78)}
    for name, data in inputs.items():
        ref = oracle.decode(data)
        for world in (2, 3, 7, 50) if name == "dri_row" else (2, 3, 7):  # 50: more decoders than segments, some get nothing
            planes = [torch.full(p.shape, 0xAB, dtype=torch.uint8, device="cuda:0") for p in ref.planes]
            for rank in range(world):
                dec = jpeggpu_amd.Decoder(32 if rank % 2 else 64)
                dec.set_segment_shard(rank, world)
                info = dec.parse_header(data)
                n = dec.get_buffer_size()
                tmp = torch.empty(n + 256, dtype=torch.uint8, device="cuda:0")
                base = (tmp.data_ptr() + 255) // 256 * 256
                before = [p.clone() for p in planes]
                dec.transfer(base, n, 0)
                dec.decode([p.data_ptr() for p in planes], [p.stride(0) for p in planes], base, n, 0)
                torch.cuda.synchronize()
                for c in range(info.num_components):
                    a, cnt = dec.shard_rows(c)
                    assert torch.equal(planes[c][:a], before[c][:a]) and torch.equal(planes[c][a + cnt:], before[c][a + cnt:]), (name, world, rank, c)
                    assert np.array_equal(planes[c][a:a + cnt].cpu().numpy(), ref.planes[c][a:a + cnt]), (name, world, rank, c)
                dec.cleanup()
            for c in range(ref.ncomp):
                assert np.array_equal(planes[c].cpu().numpy(), ref.planes[c]), (name, world, c)
    # a decoder that was sharded and is switched back (world = 1) decodes whole images again, device scan included;
    # the layout says which mode and which walk an image got
    data = m["dri_row"]
    ref = oracle.decode(data)
    dec = jpeggpu_amd.Decoder(64)
    dec.set_device_scan(True)
    for rank, world in ((1, 2), (0, 1)):
        dec.set_segment_shard(rank, world)
        info = dec.parse_header(data)
        lay = dec.layout()
        assert (lay.shard_rank, lay.shard_world) == (rank, world) and bool(lay.scans[0].device_scan) == (world == 1)
        n, tmp, base, planes = _alloc(torch, dec, info)
        for p in planes:
            p.fill_(0xAB)
        dec.transfer(base, n, 0)
        dec.decode([p.data_ptr() for p in planes], [p.stride(0) for p in planes], base, n, 0)
        torch.cuda.synchronize()
        whole = all(np.array_equal(planes[c][:ref.planes[c].shape[0], :ref.planes[c].shape[1]].cpu().numpy(), ref.planes[c]) for c in range(ref.ncomp))
        assert whole == (world == 1), (rank, world)
    dec.cleanup()
    # what cannot be cut into bands says so at parse time
    for name in ("dri_7", "multi_seq_nodri", "ni_420_dri"):
        dec = jpeggpu_amd.Decoder()
        dec.set_segment_shard(0, 2)
        with pytest.raises(JpegGpuError) as e:
            dec.parse_header(m[name])
        assert e.value.status == Status.NOT_SUPPORTED
        dec.cleanup()
