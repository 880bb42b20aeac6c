"""Mutation fuzzing of the host parser / table builder / marker walk and of the shared symbol loop under
AddressSanitizer + UBSan (CPU build only: GPU sanitizers are not available on the pool). The product's
parser and jg_huff_core.h are compiled as they are; the device pipeline's logic comes from tests/emu."""
import os
import subprocess
import sys
import tempfile

import pytest

from tests import cases

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.timeout(600)
def test_mutated_streams_under_asan_ubsan():
    with tempfile.TemporaryDirectory() as d:
        exe = os.path.join(d, "fuzz_main")
        srcs = [os.path.join(ROOT, "tests", "emu", f) for f in ("fuzz_main.cpp", "emu_pipeline.cpp")]
        srcs.append(os.path.join(ROOT, "jpeggpu_amd", "csrc", "jg_reader.cpp"))
        subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-fwrapv", "-fsanitize=address,undefined",
                               "-fno-sanitize-recover=all", "-I" + os.path.join(ROOT, "include"),
                               "-I" + os.path.join(ROOT, "jpeggpu_amd", "csrc")] + srcs + ["-o", exe])
        m = cases.matrix()
        files = []
        for name in ("ss_2x2", "dri_1", "dri_fill", "ni_420_dri", "four_comp_opt", "gray", "odd_17x9", "q16_tables",
                     "multi_seq_dri"):
            p = os.path.join(d, name + ".jpg")
            with open(p, "wb") as f:
                f.write(m[name])
            files.append(p)
        env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=1", UBSAN_OPTIONS="print_stacktrace=1")
        r = subprocess.run([exe, "1500", "20261004"] + files, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                           timeout=540)
        sys.stdout.write(r.stdout.decode())
        assert r.returncode == 0, r.stderr.decode()[-4000:]
        assert b"decoded" in r.stdout
