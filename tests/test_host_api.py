"""Host side of the C ABI without a GPU: the library loads, exports every symbol the headers declare,
parses like the oracle, sizes its buffers, and fails loudly where a device is needed."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import jpeggpu_amd
from jpeggpu_amd import Status
from jpeggpu_amd import build as jbuild
from oracle import oracle
from tests import cases
from tests.conftest import ROOT


@pytest.fixture(scope="module")
def L():
    jbuild.build()
    return jpeggpu_amd.lib()


def test_exports_every_declared_symbol(L):
    declared = set()
    for h in ("jpeggpu.h", "jpeggpu_ext.h"):
        text = open(os.path.join(ROOT, "include", "jpeggpu", h)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        declared |= set(re.findall(r"\b(jpeggpu_[a-z_0-9]+|is_css_444)\s*\(", text))
    assert {"jpeggpu_decoder_startup", "jpeggpu_decoder_parse_header", "jpeggpu_decoder_get_buffer_size",
            "jpeggpu_decoder_transfer", "jpeggpu_decoder_decode", "jpeggpu_decoder_cleanup",
            "jpeggpu_set_logging", "jpeggpu_get_status_string", "is_css_444"} <= declared
    for name in sorted(declared):
        assert hasattr(L, name), "library does not export " + name


def test_status_strings(L):
    # same strings as the reference (src/jpeggpu.cpp:41-60)
    want = ["success", "invalid argument", "invalid jpeg", "internal jpeggpu error", "jpeg is not supported",
            "out of host memory", "incomplete bitstream"]
    assert [jpeggpu_amd.status_string(i) for i in range(7)] == want
    assert jpeggpu_amd.status_string(99) == "unknown status"


def test_null_arguments(L):
    dec = C.c_void_p()
    assert L.jpeggpu_decoder_startup(None) == Status.INVALID_ARGUMENT
    assert L.jpeggpu_decoder_startup(C.byref(dec)) == Status.SUCCESS
    info = jpeggpu_amd.ImgInfo()
    n = C.c_size_t()
    assert L.jpeggpu_set_logging(None, 1) == Status.INVALID_ARGUMENT
    assert L.jpeggpu_decoder_parse_header(None, C.byref(info), b"x", 1) == Status.INVALID_ARGUMENT
    assert L.jpeggpu_decoder_parse_header(dec, None, b"x", 1) == Status.INVALID_ARGUMENT
    assert L.jpeggpu_decoder_parse_header(dec, C.byref(info), None, 0) == Status.INVALID_ARGUMENT
    assert L.jpeggpu_decoder_get_buffer_size(dec, None) == Status.INVALID_ARGUMENT
    assert L.jpeggpu_decoder_get_buffer_size(None, C.byref(n)) == Status.INVALID_ARGUMENT
    assert L.jpeggpu_decoder_get_buffer_size(dec, C.byref(n)) == Status.INVALID_ARGUMENT  # nothing parsed yet
    assert L.jpeggpu_decoder_transfer(None, None, 0, None) == Status.INVALID_ARGUMENT
    assert L.jpeggpu_decoder_decode(None, None, None, 0, None) == Status.INVALID_ARGUMENT
    assert L.jpeggpu_decoder_decode(dec, None, None, 0, None) == Status.INVALID_ARGUMENT
    assert L.jpeggpu_decoder_cleanup(None) == Status.INVALID_ARGUMENT
    assert L.jpeggpu_decoder_cleanup(dec) == Status.SUCCESS


def test_is_css_444(L):
    s = jpeggpu_amd.api.Subsampling()
    for c in range(3):
        s.x[c] = s.y[c] = 1
    assert L.is_css_444(s, 3) == 1 and L.is_css_444(s, 1) == 1
    s.x[0] = 2
    assert L.is_css_444(s, 3) == 0
    assert L.is_css_444(s, 0) == 0 and L.is_css_444(s, 5) == 0


def test_parse_matches_oracle_geometry(L):
    for name, data in cases.matrix().items():
        ref = oracle.decode(data)
        dec = jpeggpu_amd.Decoder()
        info = dec.parse_header(data)
        assert info.num_components == ref.ncomp, name
        for c in range(ref.ncomp):
            assert (info.sizes_y[c], info.sizes_x[c]) == ref.planes[c].shape, name
            assert (info.subsampling.x[c], info.subsampling.y[c]) == (ref.hs[c], ref.vs[c]), name
        for c in range(ref.ncomp, 4):
            assert info.sizes_x[c] == info.sizes_y[c] == 0
        dec.cleanup()


@pytest.mark.parametrize("subseq_bytes", [128, 64, 32])
def test_layout_matches_oracle_segment_walk(L, photo_bytes, subseq_bytes):
    """The host walk of the entropy-coded bytes (reference src/reader.cpp:447-489) against the oracle's."""
    inputs = dict(cases.matrix())
    inputs["photo"] = photo_bytes
    for name, data in inputs.items():
        dec = jpeggpu_amd.Decoder(subseq_bytes)
        dec.parse_header(data)
        lay = dec.layout()
        assert lay.subsequence_bytes == subseq_bytes
        for s in range(lay.num_scans):
            li = oracle.scan_info(data, s, subseq_bytes)
            sl = lay.scans[s]
            assert (sl.num_subsequences, sl.num_segments, sl.num_data_units) == (li.num_subseq, li.num_segments, li.num_du), name
            per = lay.subsequences_per_sequence
            assert sl.num_sequences == (li.num_subseq + per - 1) // per
        assert dec.get_buffer_size() % 256 == 0
        dec.cleanup()


def test_photo_buffer_is_smaller_than_the_references(L, photo_bytes):
    dec = jpeggpu_amd.Decoder(128)
    dec.parse_header(photo_bytes)
    lay = dec.layout()
    assert lay.scans[0].num_subsequences == 22711 and (22711 + 255) // 256 == 89  # README.md:37-38
    assert lay.scans[0].num_sequences == -(-22711 // lay.subsequences_per_sequence)
    assert lay.transferred_bytes < 2_907_282 + 64           # scan bytes only, not the 4 MB file (B-7)
    assert dec.get_buffer_size() < 80 * 2 ** 20             # the reference needs ~116 MB
    dec.cleanup()


def test_negative_inputs_same_status_as_oracle(L):
    from tools import jpegsynth

    good = bytearray(jpegsynth.encode(64, 48, seed=3))

    def both(data):
        dec = jpeggpu_amd.Decoder()
        try:
            dec.parse_header(bytes(data))
            got = 0
        except jpeggpu_amd.JpegGpuError as e:
            got = int(e.status)
        finally:
            dec.cleanup()
        try:
            oracle.decode(bytes(data))
            want = 0
        except oracle.OracleError as e:
            want = e.status
        return got, want

    variants = {"good": good, "empty": b"\xff", "soi_only": b"\xff\xd8", "truncated": good[:len(good) // 2]}
    i = good.find(b"\xff\xc0")
    v = bytearray(good); v[i + 1] = 0xC2; variants["progressive"] = v
    v = bytearray(good); v[i + 4] = 12; variants["12bit"] = v
    v = bytearray(good); v[i + 9] = 5; variants["too_many_components"] = v
    v = bytearray(good); v[i + 11] = 0x51; variants["bad_sampling"] = v
    j = good.find(b"\xff\xdb")
    v = bytearray(good); v[j + 4] |= 0x10; variants["dqt16"] = v
    k = good.find(b"\xff\xc4")
    v = bytearray(good); v[k + 1] = 0xE5; variants["missing_dht"] = v
    m = good.find(b"\xff\xda")
    v = bytearray(good); v[m + 6] = 0x33; variants["undefined_table_id"] = v
    v = bytearray(good); v[m + 5] = 9; variants["unknown_component"] = v
    from tests import cases

    variants["empty_restart_segment"] = cases.empty_segment_case()
    for name, data in variants.items():
        got, want = both(data)
        assert got == want, (name, got, want)
    assert both(variants["progressive"])[0] == Status.NOT_SUPPORTED
    assert both(variants["truncated"])[0] == Status.INVALID_JPEG
    assert both(variants["empty_restart_segment"])[0] == Status.INVALID_JPEG


def test_bytearray_input_is_borrowed_not_copied(L):
    """The decoder keeps the address it was given until transfer(): for a bytearray that must be the bytearray's
    own storage, kept alive by the Decoder object (a temporary bytes copy would dangle)."""
    import gc

    from jpeggpu_amd import api

    data = bytearray(cases.matrix()["ss_2x2"])
    ptr, n, keep = api._host_buffer(data)
    assert n == len(data) and ptr == np.frombuffer(data, np.uint8).ctypes.data
    dec = jpeggpu_amd.Decoder()
    dec.parse_header(data)
    assert dec._keep[0] is data
    gc.collect()
    lay = dec.layout()
    assert lay.num_scans == 1
    bufs = [bytearray(cases.matrix()["gray"]), cases.matrix()["ss_1x1"]]
    decs = [jpeggpu_amd.Decoder(), jpeggpu_amd.Decoder()]
    jpeggpu_amd.parse_headers(decs, bufs, num_threads=2)
    assert decs[0]._keep[0] is bufs[0] and decs[1]._keep is bufs[1]
    for d in decs + [dec]:
        d.cleanup()


def test_decoder_reuse_and_subsequence_knob(L):
    m = cases.matrix()
    dec = jpeggpu_amd.Decoder()
    sizes = []
    for name in ("ss_2x2", "gray", "four_comp_opt", "ss_2x2"):
        dec.parse_header(m[name])
        sizes.append(dec.get_buffer_size())
    assert sizes[0] == sizes[3]
    with pytest.raises(jpeggpu_amd.JpegGpuError):
        dec.set_subsequence_bytes(48)
    dec.set_subsequence_bytes(32)
    with pytest.raises(jpeggpu_amd.JpegGpuError):
        dec.get_buffer_size()  # the knob invalidates the parsed image
    dec.cleanup()


def test_subsequence_size_is_chosen_per_image(L, photo_bytes, monkeypatch):
    """SURVEY.md 8f-4, the reference's own TODO (src/decoder_defs.hpp:28-34): the size follows the call type, the
    scan's size and its restart density; an explicit size (call or environment) wins; 0 gives the choice back."""
    from tools import jpegsynth

    m = cases.matrix()
    got = {}
    for batched in (False, True):
        dec = jpeggpu_amd.Decoder()
        dec.set_batched(batched)
        for name, data in (("photo", photo_bytes), ("dri_1", m["dri_1"]), ("dri_7", m["dri_7"]), ("small", m["ss_2x2"]),
                           ("multi_seq_nodri", m["multi_seq_nodri"]), ("cfg4_small", m["cfg4_small"])):
            dec.parse_header(data)
            lay = dec.layout()
            got[(name, batched)] = lay.subsequence_bytes
            for s in range(lay.num_scans):  # the tables are those of the chosen size
                assert lay.scans[s].num_subsequences == oracle.scan_info(data, s, lay.subsequence_bytes).num_subseq
        dec.cleanup()
    assert got[("photo", False)] == 64 and got[("photo", True)] == 256        # 12 MP, segments of ~15 KB
    assert got[("multi_seq_nodri", False)] == 64 and got[("multi_seq_nodri", True)] in (128, 256)
    assert got[("dri_1", True)] == 32 and got[("dri_1", False)] == 32       # a restart marker behind every MCU: ~60-byte segments
    assert got[("dri_7", True)] <= 64
    assert got[("small", True)] == 64                                          # a few kilobytes: not even one sequence
    dec = jpeggpu_amd.Decoder(128)
    dec.set_batched(True)
    dec.parse_header(photo_bytes)
    assert dec.layout().subsequence_bytes == 128
    dec.set_subsequence_bytes(0)
    dec.parse_header(photo_bytes)
    assert dec.layout().subsequence_bytes == 256
    dec.cleanup()
    monkeypatch.setenv("JPEGGPU_SUBSEQ_BYTES", "32")
    dec = jpeggpu_amd.Decoder()
    dec.parse_header(photo_bytes)
    assert dec.layout().subsequence_bytes == 32
    dec.cleanup()


def test_multi_hypothesis_applies_to_lone_decodes_of_interleaved_restart_scans(L, photo_bytes, monkeypatch):
    """jpeggpu_ext_scan_layout.hypotheses (jg_defs.h, multi-hypothesis speculation): one candidate per data unit of the
    MCU for an image decoded on its own (device-scanned or not; a scan without restart markers, or with segments of more
    than 1024 subsequences, walks its chain block-wise: hypothesis_blocks -- for a device-scanned scan without restart
    markers from a list the device builds, sized from the header's bound); not for batches, single-unit MCUs, or when
    the environment switches it off."""
    m = cases.matrix()

    def hyp(data, batched=False, device_scan=False):
        dec = jpeggpu_amd.Decoder()
        dec.set_batched(batched)
        dec.set_device_scan(device_scan)
        dec.parse_header(data)
        lay = dec.layout()
        out = [lay.scans[s].hypotheses for s in range(lay.num_scans)]
        blocks[:] = [lay.scans[s].hypothesis_blocks for s in range(lay.num_scans)]
        subseq[:] = [lay.scans[s].num_subsequences for s in range(lay.num_scans)]
        dec.cleanup()
        return out

    blocks, subseq = [], []

    assert hyp(photo_bytes) == [6] and hyp(m["dri_row"]) == [6] and hyp(m["cfg2_small"]) == [6]
    assert hyp(photo_bytes, batched=True) == [0] and hyp(photo_bytes, device_scan=True) == [6]
    assert blocks == [0]
    assert hyp(m["gray"]) == [0]
    assert hyp(m["cfg5_small"]) == [6] and blocks == [0]       # no restart markers, one short segment: walked whole
    assert hyp(m["multi_seq_nodri"]) == [6] and subseq[0] > 2048 and blocks == [(subseq[0] + 1023) // 1024]  # one long segment
    # the device finds the one segment's length: the block list is its work (round 5), its capacity comes from the bound
    assert hyp(m["multi_seq_nodri"], device_scan=True) == [6] and blocks == [(subseq[0] + 1023) // 1024] and blocks[0] > 1
    assert hyp(m["ni_420_dri"]) == [0, 0, 0]          # one data unit per MCU in every scan
    assert hyp(m["dri_7"]) == [6]
    monkeypatch.setenv("JPEGGPU_MULTI_HYPOTHESIS", "0")
    assert hyp(photo_bytes) == [0]


def test_device_scan_knob_and_environment(L, monkeypatch):
    """jpeggpu_ext_set_device_scan / JPEGGPU_DEVICE_SCAN: the LAST scan of a file (the only one of most) is parsed up to
    its scan header only (capacities instead of counts, a bigger buffer) -- the scans in front of it keep the host walk,
    which has to find the next scan header behind them anyway, and the last of several is the device's only if it holds
    at least as many bytes as they do; the environment variable switches it on for decoders created afterwards."""
    m = cases.matrix()

    def layouts(dec):
        out = []
        for name in ("multi_seq_dri", "ni_420_dri", "ni_big_last_dri"):
            dec.parse_header(m[name])
            lay = dec.layout()
            out.append((lay.scans[0], dec.get_buffer_size(), [lay.scans[k].device_scan for k in range(lay.num_scans)]))
        return out

    plain = jpeggpu_amd.Decoder()
    a = layouts(plain)
    assert not a[0][0].device_scan and not a[1][0].device_scan
    plain.set_device_scan(True)
    b = layouts(plain)
    # (the buffer sizes are not comparable: the host-walked lone decode also carries the multi-hypothesis tables)
    assert b[0][0].device_scan and b[0][0].num_subsequences >= a[0][0].num_subsequences and b[0][1] != a[0][1]
    assert a[1][2] == [0, 0, 0] and b[1][2] == [0, 0, 0] and b[1][1] == a[1][1]  # three scans, the last one a sixth of the bytes: host walk
    assert a[2][2] == [0, 0] and b[2][2] == [0, 1] and b[2][1] != a[2][1]        # two scans, the last one the bulk: on the device
    plain.cleanup()
    monkeypatch.setenv("JPEGGPU_DEVICE_SCAN", "1")
    env = jpeggpu_amd.Decoder()
    c = layouts(env)
    assert c[0][0].device_scan and c[0][1] == b[0][1] and c[1][2] == [0, 0, 0] and c[2][2] == [0, 1]
    env.cleanup()


def test_parallel_parse_equals_serial(L):
    """jpeggpu_ext_parse_headers on a thread pool: same geometry, sizes and layouts as one-by-one parsing,
    per-item statuses for bad inputs, duplicate decoders rejected."""
    import ctypes as C

    import jpeggpu_amd
    from jpeggpu_amd.api import ParseItem

    m = cases.matrix()
    names = list(m.keys())
    bufs = [m[k] for k in names] * 2
    serial = []
    for b in bufs:
        d = jpeggpu_amd.Decoder()
        info = d.parse_header(b)
        serial.append((info.num_components, list(info.sizes_x), list(info.sizes_y), d.get_buffer_size(),
                       d.layout().scans[0].num_subsequences, d.layout().scans[0].num_segments))
        d.cleanup()
    for threads in (1, 3, 8):
        decs = [jpeggpu_amd.Decoder() for _ in bufs]
        infos = jpeggpu_amd.parse_headers(decs, bufs, num_threads=threads)
        got = [(i.num_components, list(i.sizes_x), list(i.sizes_y), d.get_buffer_size(),
                d.layout().scans[0].num_subsequences, d.layout().scans[0].num_segments) for i, d in zip(infos, decs)]
        assert got == serial, threads
        for d in decs:
            d.cleanup()
    # one truncated file among good ones: its own status, the others still parsed
    decs = [jpeggpu_amd.Decoder() for _ in range(3)]
    datas = [m["ss_2x2"], m["ss_2x2"][:200], m["gray"]]
    items = (ParseItem * 3)()
    infos = [jpeggpu_amd.ImgInfo() for _ in range(3)]
    for i in range(3):
        items[i].decoder, items[i].img_info = decs[i]._h.value, C.pointer(infos[i])
        items[i].data, items[i].size = C.cast(C.c_char_p(datas[i]), C.c_void_p).value, len(datas[i])
    st = (C.c_int * 3)()
    rc = L.jpeggpu_ext_parse_headers(items, 3, 2, st)
    assert list(st) == [0, rc, 0] and rc != 0
    assert infos[0].num_components == 3 and infos[2].num_components == 1
    items[2].decoder = decs[0]._h.value  # the same decoder twice
    assert L.jpeggpu_ext_parse_headers(items, 3, 2, st) == int(jpeggpu_amd.Status.INVALID_ARGUMENT)
    for d in decs:
        d.cleanup()


def _build_c_example(out):
    import subprocess

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib_dir = os.path.join(root, "jpeggpu_amd", "lib")
    subprocess.check_call(["gcc", "-std=c11", "-Wall", "-Wextra", "-Werror", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include",
                           "-I" + os.path.join(root, "include"), os.path.join(root, "examples", "decode_file.c"),
                           "-L" + lib_dir, "-ljpeggpu", "-L/opt/rocm/lib", "-lamdhip64",
                           "-Wl,-rpath," + lib_dir, "-Wl,-rpath,/opt/rocm/lib", "-o", out])


def test_c_caller_compiles_and_links(L, tmp_path):
    """The public headers are C (not only C++ / ctypes): examples/decode_file.c, the reference example tool's
    call sequence, builds as C11 with -Wall -Wextra -Werror against the library."""
    _build_c_example(str(tmp_path / "decode_file"))


def test_fails_loudly_without_a_device(L):
    """No CPU fallback: without a HIP device transfer/decode return an error, never fake planes."""
    import torch

    if torch.cuda.is_available():
        pytest.skip("a device is present")
    dec = jpeggpu_amd.Decoder()
    dec.parse_header(cases.matrix()["ss_2x2"])
    n = dec.get_buffer_size()
    buf = np.zeros(n + 256, np.uint8)
    base = (buf.ctypes.data + 255) // 256 * 256
    with pytest.raises(jpeggpu_amd.JpegGpuError) as ei:
        dec.transfer(base, n, 0)
    assert ei.value.status == Status.INTERNAL_ERROR
    dec.cleanup()


def test_build_checks_the_counted_wait_refill_of_the_write_pass(L):
    """RowWindow::top() loads the next bitstream word from inline assembly and waits with
    `s_waitcnt vmcnt(1)`; the compiler must never touch that register between the assembly blocks. jpeggpu_amd/build.py
    checks the generated gfx950 code at every build (no scratch, no spills, the register named only inside the blocks) and
    falls back to -DJG_SAFE_REFILL otherwise. The check passes on the tree as it is, and it does find a copy of the
    register planted into the loop, a spill count and a missing block."""
    import re

    from jpeggpu_amd import build as jbuild

    text = jbuild.device_assembly()
    assert jbuild.check_refill_text(text) == []
    # plant `v_mov_b32 v255, <nxt>` behind the first refill block of the first huff_write
    start = next(i for i, ln in enumerate(text) if re.match(r"^_ZN2jg\S*huff_write\S*:", ln))
    k = next(i for i in range(start, len(text)) if "global_load_dword" in text[i] and ";;#ASMSTART" in "".join(text[i - 12:i]))
    reg = re.match(r"\s*global_load_dword (v\d+),", text[k]).group(1)
    end = next(i for i in range(k, len(text)) if ";;#ASMEND" in text[i])
    doctored = text[:end + 1] + ["\tv_mov_b32_e32 v255, %s" % reg] + text[end + 1:]
    bad = jbuild.check_refill_text(doctored)
    assert len(bad) == 1 and "touched outside the assembly blocks" in bad[0] and reg in bad[0]
    spilled = [ln.replace(".vgpr_spill_count: 0", ".vgpr_spill_count: 3") if ".vgpr_spill_count" in ln else ln for ln in text]
    assert any("vgpr_spill_count: 3" in p for p in jbuild.check_refill_text(spilled))
    assert jbuild.check_refill_text([ln for ln in text if "global_load_dword" not in ln or "s[" not in ln]) != []
