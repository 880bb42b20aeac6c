"""Build driver: compiles the C-ABI shared library for gfx950 with hipcc, in-tree."""
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(_HERE)
CSRC = os.path.join(_HERE, "csrc")
LIB_DIR = os.path.join(_HERE, "lib")
LIB_PATH = os.path.join(LIB_DIR, "libjpeggpu.so")
SOURCES = ["jg_kernels.hip", "jg_front.hip", "jg_decoder.cpp", "jg_reader.cpp"]


def _stale() -> bool:
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    # every source and header under csrc/ (a header missing from a hand-kept list once left a stale library)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".h", ".hpp", ".hip", ".cpp"))]
    deps += [os.path.join(ROOT, "include", "jpeggpu", h) for h in ("jpeggpu.h", "jpeggpu_ext.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False, out: str = None, extra_flags=()) -> str:
    """hipcc --offload-arch=gfx950 -> jpeggpu_amd/lib/libjpeggpu.so (cross-compiles without a GPU).
    `out` / `extra_flags`: experimental builds for in-run A/B comparisons (tools/probe/ab.sh, JPEGGPU_LIB)."""
    if out is not None:
        return _compile(out, verbose, extra_flags)
    if not force and not _stale():
        return LIB_PATH
    return _compile(LIB_PATH, verbose, extra_flags)


def _compile(lib_path, verbose, extra_flags):
    os.makedirs(LIB_DIR, exist_ok=True)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    # -fwrapv: the reference's GPU integer arithmetic wraps; signed overflow must not be undefined (same flag as the oracle)
    cmd = [hipcc, "-O3", "-std=c++17", "-fwrapv", "--offload-arch=gfx950", "-fPIC", "-shared",
           "-I" + os.path.join(ROOT, "include"), "-I" + CSRC]
    cmd += list(extra_flags)
    cmd += [os.path.join(CSRC, s) for s in SOURCES]
    cmd += ["-o", lib_path]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return lib_path


if __name__ == "__main__":
    import sys

    # python jpeggpu_amd/build.py [out.so [extra hipcc flags...]]
    if len(sys.argv) > 1:
        print(build(out=os.path.abspath(sys.argv[1]), extra_flags=sys.argv[2:], verbose=True))
    else:
        print(build(force=True, verbose=True))
