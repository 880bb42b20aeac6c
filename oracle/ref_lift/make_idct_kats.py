#!/usr/bin/env python3
"""Known-answer vectors of the reference's own IDCT -> tests/golden/idct_kats.npz.

TEST INFRASTRUCTURE, authoring container only. Inputs are generated here (seeded numpy); outputs come from
oracle/_ref/libref_idct.so, i.e. from /root/reference/src/idct.cu:43-144 compiled as it stands
(oracle/ref_lift/build.sh). The fixture holds data only: inputs and the reference's outputs.

  vec_in  int32[N,8]  -> vec_out int32[N,8]      idct_vector (idct.cu:50-95)
  per group g (one quantisation table each, so that a group is one JPEG for the GPU test):
    q[g]      uint8[64]          natural order
    coef[g]   int16[B,64]        natural order; DC in [-1024, 1023], AC in [-1023, 1023] (baseline-codable)
    out[g]    uint8[B,64]        ref_idct_block with the reference's literal int8 quantiser read
    out_u[g]  uint8[B,64]        the same with the quantiser read as unsigned (differs only where q >= 128)
"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
LIB = os.path.join(ROOT, "oracle", "_ref", "libref_idct.so")
OUT = os.path.join(ROOT, "tests", "golden", "idct_kats.npz")

ZIGZAG = [0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21,
          28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61,
          54, 47, 55, 62, 63]


def blocks(rng, n, kind):
    """n coefficient blocks (natural order) of one flavour."""
    c = np.zeros((n, 64), np.int16)
    if kind == "sparse":  # photo-like: a handful of low-frequency coefficients
        for b in range(n):
            k = int(rng.integers(1, 12))
            pos = rng.choice(20, size=k, replace=False)
            c[b, [ZIGZAG[p] for p in pos]] = np.clip(np.rint(rng.laplace(0, 30, k)), -1023, 1023)
    elif kind == "dense":
        c[:] = np.clip(np.rint(rng.laplace(0, 12, (n, 64))), -1023, 1023)
    elif kind == "extreme":  # saturating: every coefficient at the ends of its range
        c[:] = rng.choice(np.array([-1023, 1023, -1024 + 1, 1023, 0], np.int16), (n, 64))
    elif kind == "dc_only":
        c[:, 0] = rng.integers(-1024, 1024, n)
    elif kind == "one_hot":  # a single coefficient per block, every position, both signs
        for b in range(n):
            c[b, b % 64] = (1 if (b // 64) % 2 == 0 else -1) * int(rng.integers(1, 1024))
    elif kind == "uniform":
        c[:] = rng.integers(-1023, 1024, (n, 64))
    c[:, 0] = np.clip(c[:, 0], -1024, 1023)
    return c


def main():
    subprocess.check_call(["bash", os.path.join(HERE, "build.sh")])
    L = C.CDLL(LIB)
    L.ref_idct_vectors.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    L.ref_idct_blocks.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int]
    rng = np.random.default_rng(20261004)

    # 8-vectors: what a column pass sees (dequantised int16) and what a row pass sees, plus edge patterns
    v = [rng.integers(-32768, 32768, (2048, 8)), rng.integers(-2048, 2048, (1024, 8)),
         np.rint(rng.laplace(0, 40, (1024, 8))).astype(np.int64)]
    edge = np.array([-32768, -32767, -1, 0, 1, 32767])
    v.append(edge[rng.integers(0, len(edge), (512, 8))])
    v.append(np.eye(8, dtype=np.int64)[np.arange(64) % 8] * np.repeat(np.array([1, -1, 1024, -1024, 32767, -32768, 181, 8]), 8)[:, None])
    vec_in = np.ascontiguousarray(np.concatenate(v).astype(np.int32))
    vec_out = np.zeros_like(vec_in)
    L.ref_idct_vectors(vec_in.ctypes.data, vec_out.ctypes.data, len(vec_in))

    groups = {}
    qs = {
        "q_ones": np.ones(64, np.uint8),
        "q_photo": np.clip(np.arange(64).reshape(8, 8).T // 5 + 2, 1, 13).astype(np.uint8).reshape(64),
        "q_annexk": np.array([16, 11, 10, 16, 24, 40, 51, 61, 12, 12, 14, 19, 26, 58, 60, 55, 14, 13, 16, 24, 40, 57, 69, 56,
                              14, 17, 22, 29, 51, 87, 80, 62, 18, 22, 37, 56, 68, 109, 103, 77, 24, 35, 55, 64, 81, 104, 113,
                              92, 49, 64, 78, 87, 103, 121, 120, 101, 72, 92, 95, 98, 112, 100, 103, 99], np.uint8),
        "q_127": np.full(64, 127, np.uint8),                      # coef * q wraps int16 (idct.cu:180)
        "q_rand_le127": rng.integers(1, 128, 64).astype(np.uint8),
        "q_rand_ge128": rng.integers(128, 256, 64).astype(np.uint8),  # Appendix B-3: int8 read goes negative
        "q_mixed": rng.integers(1, 256, 64).astype(np.uint8),
        "q_255": np.full(64, 255, np.uint8),
    }
    kinds = ["sparse", "dense", "extreme", "dc_only", "one_hot", "uniform"]
    out = {"vec_in": vec_in, "vec_out": vec_out}
    for name, q in qs.items():
        c = np.ascontiguousarray(np.concatenate([blocks(rng, 128, k) for k in kinds]))
        o = np.zeros((len(c), 64), np.uint8)
        ou = np.zeros((len(c), 64), np.uint8)
        L.ref_idct_blocks(c.ctypes.data, q.ctypes.data, o.ctypes.data, len(c), 1)
        L.ref_idct_blocks(c.ctypes.data, q.ctypes.data, ou.ctypes.data, len(c), 0)
        if q.max() <= 127:
            assert np.array_equal(o, ou)
        out[name + "/q"], out[name + "/coef"], out[name + "/out"], out[name + "/out_u"] = q, c, o, ou
    np.savez_compressed(OUT, **out)
    print(OUT, os.path.getsize(OUT), "bytes;", len(vec_in), "vectors,", sum(len(out[k]) for k in out if k.endswith("/coef")), "blocks")


if __name__ == "__main__":
    sys.exit(main())
