#!/usr/bin/env python3
"""Pins the CPU oracle against an independent standard decoder and writes tests/golden/.

Runs ONLY in the authoring container (needs IJG libjpeg 9d headers/library from /opt/conda and, for a
few inputs, Pillow / cjpeg as independent ENCODERS). The GPU box never runs this; it reads the
committed fixtures. The reference itself cannot produce vectors here (it needs nvcc + CUDA, see
DESIGN.md), and its own tests hold no golden outputs (test/test.cpp:299-314 only prints an MSE), so
the pins are:
  * quantised coefficients  == IJG jpeg_read_coefficients (exact, any correct decoder agrees),
  * planes within the accuracy band the reference's README states against a standard decoder
    (README.md:76,81) -> recorded IJG islow raw planes, checked as MSE <= 0.25 and max |diff| <= 2,
  * the reference photo's subsequence / sequence counts printed in README.md:37-38.
Outputs:
  tests/golden/pin_vectors.npz   per case: jpeg bytes, IJG coefficients, IJG raw planes
  tests/golden/photo_pins.json   sha256 of IJG coefficients and of the oracle's planes, MSE vs IJG,
                                 segment / subsequence counts for the reference's photo
"""
import hashlib
import io
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle  # noqa: E402
from tools import jpegsynth  # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden")
DUMP = os.path.join(tempfile.gettempdir(), "ijg_dump")


def build_dumper():
    src = os.path.join(ROOT, "oracle", "pin", "ijg_dump.c")
    subprocess.check_call(["gcc", "-O2", "-o", DUMP, src, "-I/opt/conda/include", "-L/opt/conda/lib", "-ljpeg",
                           "-Wl,-rpath,/opt/conda/lib"])


def ijg(data, mode):
    with tempfile.TemporaryDirectory() as d:
        jp, out = os.path.join(d, "a.jpg"), os.path.join(d, "a.bin")
        open(jp, "wb").write(data)
        subprocess.check_call([DUMP, mode, jp, out])
        b = open(out, "rb").read()
    o, res = 4, []
    nc = int(np.frombuffer(b, np.int32, 1, 0)[0])
    for _ in range(nc):
        w, h = (int(v) for v in np.frombuffer(b, np.int32, 2, o))
        o += 8
        if mode == "coef":
            res.append(np.frombuffer(b, np.int16, w * h * 64, o).reshape(h, w, 64).copy())
            o += w * h * 128
        else:
            res.append(np.frombuffer(b, np.uint8, w * h, o).reshape(h, w).copy())
            o += w * h
    return res


def pil_cases():
    """Inputs from independent encoders (Pillow's libjpeg-turbo, IJG cjpeg) so the pins do not rest on
    this repository's own synthetic encoder alone."""
    from PIL import Image

    rng = np.random.default_rng(7)
    y, x = np.mgrid[0:120, 0:168]
    base = (np.stack([128 + 90 * np.sin(x / 17.0) * np.cos(y / 11.0), 128 + 60 * np.cos(x / 9.0),
                      128 + 70 * np.sin((x + y) / 13.0)], -1) + rng.normal(0, 9, (120, 168, 3))).clip(0, 255)
    img = Image.fromarray(base.astype(np.uint8), "RGB")
    out = {}
    for name, kw in {
        "pil_420": dict(subsampling=2, quality=85),
        "pil_422": dict(subsampling=1, quality=90),
        "pil_444": dict(subsampling=0, quality=75),
        "pil_420_opt": dict(subsampling=2, quality=92, optimize=True),
        "pil_420_rst": dict(subsampling=2, quality=80, restart_marker_blocks=3),
        "pil_gray": dict(quality=80),
    }.items():
        buf = io.BytesIO()
        (img.convert("L") if name == "pil_gray" else img).save(buf, "JPEG", **kw)
        out[name] = buf.getvalue()
    cmyk = Image.fromarray(np.concatenate([base, base[:, :, :1]], -1).astype(np.uint8), "CMYK")
    buf = io.BytesIO()
    cmyk.save(buf, "JPEG", quality=85)
    out["pil_cmyk"] = buf.getvalue()
    # IJG cjpeg: non-interleaved scans via a scan script, restart in MCU rows
    with tempfile.TemporaryDirectory() as d:
        ppm = os.path.join(d, "a.ppm")
        img.save(ppm)
        scans = os.path.join(d, "s.txt")
        open(scans, "w").write("0;\n1;\n2;\n")
        for name, args in {"cjpeg_ni_420": ["-sample", "2x2", "-scans", scans],
                           "cjpeg_rst_rows": ["-sample", "2x1", "-restart", "1"]}.items():
            o = os.path.join(d, name + ".jpg")
            subprocess.check_call(["/opt/conda/bin/cjpeg", "-quality", "88", "-baseline"] + args + ["-outfile", o, ppm])
            out[name] = open(o, "rb").read()
    return out


def main():
    build_dumper()
    os.makedirs(GOLDEN, exist_ok=True)
    e = jpegsynth.encode
    S420, S444 = ((2, 2), (1, 1), (1, 1)), ((1, 1),) * 3
    cases = {
        "syn_420": e(200, 152, S420, seed=101),
        "syn_444_odd": e(57, 43, S444, seed=102),
        "syn_420_dri": e(248, 200, S420, restart_interval=11, seed=103),
        "syn_ni_420_dri": e(233, 171, S420, interleaved=False, restart_interval=9, seed=104),
        "syn_4comp_opt": e(264, 200, ((2, 1), (1, 1), (1, 1), (2, 1)), optimize=True, seed=105),
        "syn_4x1": e(200, 152, ((4, 1), (1, 1), (1, 1)), seed=106),
        "syn_q100": e(160, 128, S420, quality=100, noise=40, seed=107),
        "syn_q16_tables": e(200, 152, S420, restart_interval=5, quality=3, noise=30, seed=108, qmax=65535),  # Pq = 1
    }
    cases.update(pil_cases())
    store, report = {}, {}
    for name, data in sorted(cases.items()):
        coef, raw = ijg(data, "coef"), ijg(data, "raw")
        d = oracle.decode(data)
        store[name + "/jpeg"] = np.frombuffer(data, np.uint8)
        mse = []
        for c in range(d.ncomp):
            a = coef[c]
            assert np.array_equal(d.coef[c][:a.shape[0], :a.shape[1]], a), (name, c, "coefficients differ from IJG")
            store["%s/coef%d" % (name, c)] = a
            store["%s/raw%d" % (name, c)] = raw[c]
            diff = d.planes[c].astype(int) - raw[c].astype(int)
            assert (diff ** 2).mean() <= 0.25 and abs(diff).max() <= 2, (name, c, (diff ** 2).mean(), abs(diff).max())
            mse.append(float((diff ** 2).mean()))
        report[name] = {"bytes": len(data), "ncomp": d.ncomp, "nscans": d.nscans, "mse_vs_ijg": mse}
    np.savez_compressed(os.path.join(GOLDEN, "pin_vectors.npz"), **store)

    photo = open(os.path.join(GOLDEN, "IMG_6510.JPG"), "rb").read()
    coef, raw = ijg(photo, "coef"), ijg(photo, "raw")
    d = oracle.decode(photo)
    pins = {"sha256_file": hashlib.sha256(photo).hexdigest(), "components": []}
    for c in range(3):
        a = coef[c]
        assert np.array_equal(d.coef[c][:a.shape[0], :a.shape[1]], a)
        diff = d.planes[c].astype(int) - raw[c].astype(int)
        pins["components"].append({
            "blocks": [int(a.shape[0]), int(a.shape[1])],
            "sha256_ijg_coefficients": hashlib.sha256(a.tobytes()).hexdigest(),
            "sha256_oracle_plane": hashlib.sha256(d.planes[c].tobytes()).hexdigest(),
            "mse_vs_ijg_islow": float((diff ** 2).mean()), "max_abs_diff_vs_ijg_islow": int(abs(diff).max())})
    for sb in (128, 64, 32):
        li = oracle.scan_info(photo, 0, sb)
        pins["subsequences_%d" % sb] = li.num_subseq
    pins["segments"] = oracle.scan_info(photo, 0, 128).num_segments
    pins["data_units"] = oracle.scan_info(photo, 0, 128).num_du
    pins["readme_sequences_of_256x128B"] = (pins["subsequences_128"] + 255) // 256  # README.md:37 says 89
    json.dump(pins, open(os.path.join(GOLDEN, "photo_pins.json"), "w"), indent=1)
    json.dump(report, open(os.path.join(GOLDEN, "pin_report.json"), "w"), indent=1)
    print(json.dumps(report, indent=1))
    print(json.dumps(pins, indent=1))


if __name__ == "__main__":
    main()
