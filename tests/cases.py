"""Shared matrix of synthetic inputs (reference coverage: test/test.sh:31-43 sampling variants and
grayscale; plus what it never tests)."""
from tools import jpegsynth

S420 = ((2, 2), (1, 1), (1, 1))
S444 = ((1, 1), (1, 1), (1, 1))


def dense_escape_case(groups=96, seed=5):
    """The most entries per bit a code table allows (jg_defs.h, sym_region_entries): a grayscale file whose fitted
    tables give the DC symbol and the AC symbol (0, 1) one-bit codes, in groups of four data units of 63 coefficients
    +-1 (64 entries in 127 bits each) followed by one unit of 63 coefficients of magnitude >= 512 (127 entries: every
    one takes an escape). A subsequence that commits four dense units and the DC symbol of the escaped one emits
    4.03 B + 127 entries."""
    import numpy as np

    rng = np.random.default_rng(seed)
    n = 5 * groups
    coef = np.zeros((n, 64), np.int16)
    sign = lambda shape: rng.integers(0, 2, shape).astype(np.int16) * 2 - 1
    coef[:, 1:] = sign((n, 63))
    coef[4::5, 1:] = sign((groups, 63)) * rng.integers(512, 1024, (groups, 63)).astype(np.int16)
    return jpegsynth.encode_blocks(coef, 24, np.ones(64, np.uint8), optimize=True)


def empty_segment_case():
    """A 4:2:0 file with restart markers in which one restart marker was taken out and put back right behind the next
    one: the number of segments still matches the geometry, but one segment holds no byte at all (the chain
    walk of huff_mh_resolve had no first subsequence to start from). Both walks refuse it."""
    good = jpegsynth.encode(240, 168, S420, restart_interval=7, seed=9)
    sos = good.index(b"\xff\xda")
    marks = [i for i in range(sos, len(good) - 1) if good[i] == 0xFF and 0xD0 <= good[i + 1] <= 0xD7]
    assert len(marks) >= 4
    a, b = marks[1], marks[2]
    # segment 1 and 2 become one (their marker is gone), an empty one follows the next marker
    return good[:a] + good[a + 2:b + 2] + good[a:a + 2] + good[b + 2:]


def big_last_scan_case(restart_interval=0, seed=41):
    """Two non-interleaved scans of which the LAST one holds most of the bytes (component 1 has four times the samples of
    component 0): the shape of file whose last scan the device-side marker scan takes over (jpeggpu_ext_set_device_scan)."""
    return jpegsynth.encode(232, 176, ((1, 1), (2, 2)), interleaved=False, restart_interval=restart_interval, quality=90, noise=12, seed=seed)


def matrix():
    """name -> bytes. Small enough for the oracle to finish in well under a second each."""
    e = jpegsynth.encode
    cases = {
        # the reference's own variant matrix (test/test.sh): 1x1, 2x1, 2x2, 1x2, 4x1, grayscale
        "ss_1x1": e(200, 152, S444, seed=1),
        "ss_2x1": e(200, 152, ((2, 1), (1, 1), (1, 1)), seed=2),
        "ss_2x2": e(200, 152, S420, seed=3),
        "ss_1x2": e(200, 152, ((1, 2), (1, 1), (1, 1)), seed=4),
        "ss_4x1": e(200, 152, ((4, 1), (1, 1), (1, 1)), seed=5),
        "gray": e(200, 152, ((1, 1),), seed=6),
        "gray_hdr_2x2": e(120, 88, ((2, 2),), seed=7),  # single component: factors are ignored
        # restart intervals
        "dri_1": e(160, 96, S420, restart_interval=1, seed=8),
        "dri_7": e(240, 168, S420, restart_interval=7, seed=9),
        "dri_row": e(496, 360, S420, restart_interval=31, seed=10),
        "dri_nondiv": e(248, 200, S420, restart_interval=11, seed=11),
        "dri_fill": e(248, 200, S420, restart_interval=5, fill_bytes=2, seed=12),
        # odd sizes
        "odd_1x1px": e(1, 1, S420, seed=13),
        "odd_17x9": e(17, 9, S420, seed=14),
        "odd_partial_mcu": e(333, 251, S420, seed=15),
        # non-interleaved
        "ni_444": e(232, 176, S444, interleaved=False, seed=16),
        "ni_420": e(232, 176, S420, interleaved=False, seed=17),
        "ni_420_dri": e(233, 171, S420, interleaved=False, restart_interval=9, seed=18),
        "ni_big_last": big_last_scan_case(),
        "ni_big_last_dri": big_last_scan_case(restart_interval=5, seed=42),
        # components and tables
        "two_comp": e(160, 120, ((2, 1), (1, 1)), seed=19),
        "four_comp_opt": e(264, 200, ((2, 1), (1, 1), (1, 1), (2, 1)), optimize=True, seed=20),
        "four_comp_444": e(136, 104, ((1, 1),) * 4, seed=21),
        "opt_tables_420": e(320, 240, S420, optimize=True, seed=22),
        # entropy extremes
        "q100_noisy": e(160, 128, S420, quality=100, noise=40, seed=23),   # long codes, long blocks
        "q5_flat": e(640, 480, S420, quality=5, noise=0, seed=24),         # almost all EOB
        "dense_escapes": dense_escape_case(),  # worst-case entries per subsequence of the write pass
        "q16_tables": e(328, 248, S420, restart_interval=7, quality=3, noise=30, seed=31, qmax=65535),  # 16-bit DQT (Pq = 1)
        # more than one sequence without restart markers (inter-sequence flows), ~150 KB of scan
        "multi_seq_nodri": e(1024, 768, S420, quality=92, noise=12, seed=25),
        "multi_seq_dri": e(1024, 768, S420, quality=92, noise=12, restart_interval=64, seed=26),
        # BASELINE.json configs at reduced size
        "cfg2_small": jpegsynth.config(2, small=True),
        "cfg4_small": jpegsynth.config(4, small=True),
        "cfg5_small": jpegsynth.config(5, small=True),
    }
    return cases
