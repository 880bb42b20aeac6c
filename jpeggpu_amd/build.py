"""Build driver: compiles the C-ABI shared library for gfx950 with hipcc, in-tree."""
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(_HERE)
CSRC = os.path.join(_HERE, "csrc")
LIB_DIR = os.path.join(_HERE, "lib")
LIB_PATH = os.path.join(LIB_DIR, "libjpeggpu.so")
SOURCES = ["jg_kernels.hip", "jg_front.hip", "jg_decoder.cpp", "jg_reader.cpp"]


def _mode_path(lib_path):
    return lib_path + ".mode"


def _wanted_mode():
    """What the environment asks of a build: the refill's wait (JPEGGPU_SAFE_REFILL) and whether the check of the generated code
    runs (JPEGGPU_SKIP_BUILD_CHECK). Recorded beside the library so that changing either rebuilds it."""
    return "safe_refill=%s skip_check=%s" % (os.environ.get("JPEGGPU_SAFE_REFILL") == "1", os.environ.get("JPEGGPU_SKIP_BUILD_CHECK") == "1")


def _stale() -> bool:
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    # every source and header under csrc/ (a header missing from a hand-kept list once left a stale library)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".h", ".hpp", ".hip", ".cpp"))]
    deps += [os.path.join(ROOT, "include", "jpeggpu", h) for h in ("jpeggpu.h", "jpeggpu_ext.h")]
    if any(os.path.getmtime(d) > t for d in deps):
        return True
    try:  # built under other settings of the environment: a fast-refill library must not outlive a request for the safe one
        with open(_mode_path(LIB_PATH)) as f:
            return f.read().split("|")[0].strip() != _wanted_mode()
    except OSError:
        return os.environ.get("JPEGGPU_SAFE_REFILL") == "1" or os.environ.get("JPEGGPU_SKIP_BUILD_CHECK") == "1"


def build(force: bool = False, verbose: bool = False, out: str = None, extra_flags=()) -> str:
    """hipcc --offload-arch=gfx950 -> jpeggpu_amd/lib/libjpeggpu.so (cross-compiles without a GPU).
    `out` / `extra_flags`: experimental builds for in-run A/B comparisons (tools/probe/ab.sh, JPEGGPU_LIB)."""
    if out is not None:
        return _compile(out, verbose, extra_flags)
    if not force and not _stale():
        return LIB_PATH
    return _compile(LIB_PATH, verbose, extra_flags)


def _base_cmd(extra_flags):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    # -fwrapv: the reference's GPU integer arithmetic wraps; signed overflow must not be undefined (same flag as the oracle)
    return [hipcc, "-O3", "-std=c++17", "-fwrapv", "--offload-arch=gfx950", "-I" + os.path.join(ROOT, "include"), "-I" + CSRC] + list(extra_flags)


def check_refill(extra_flags=(), verbose=False):
    """The write pass refills its bit window from inline assembly and waits for all but the most recent load
    (jg_kernels.hip, RowWindow: `s_waitcnt vmcnt(1)`). That is only right if the compiler never touches the register the
    loads target while one is in flight -- which it does not know. So the generated code is checked, for every
    instantiation of huff_write: no scratch memory, no spilled registers, and between the loop's first refill block and
    the `s_waitcnt vmcnt(0)` of RowWindow::done() no instruction outside the hand-written assembly blocks names that
    register. Returns a list of problems (empty: fine)."""
    return check_refill_text(device_assembly(extra_flags, verbose))


def device_assembly(extra_flags=(), verbose=False):
    """The gfx950 assembly of jg_kernels.hip, as lines."""
    import tempfile

    with tempfile.TemporaryDirectory() as tmp:
        asm = os.path.join(tmp, "jg_kernels.s")
        cmd = _base_cmd(extra_flags) + ["--offload-device-only", "-S", os.path.join(CSRC, "jg_kernels.hip"), "-o", asm]
        if verbose:
            print(" ".join(cmd))
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
        if r.returncode != 0:  # the compiler's own words, not a bare CalledProcessError
            import sys

            sys.stderr.write(r.stderr)
            raise subprocess.CalledProcessError(r.returncode, cmd, r.stdout, r.stderr)
        return open(asm).read().split("\n")


def check_refill_text(text):
    """check_refill on assembly text (a list of lines); tests feed it doctored code."""
    import re

    problems, seen = [], 0
    for k, ln in enumerate(text):  # the code object's metadata: spill counts per kernel
        if ".name:" in ln and "huff_write" in ln:
            for x in text[k:k + 14]:
                if ("spill_count" in x or ".private_segment_fixed_size" in x) and not x.strip().endswith(" 0"):
                    problems.append("%s: %s" % (ln.split(".name:")[1].strip(), x.strip()))
    i = 0
    while i < len(text):
        m = re.match(r"^(_ZN2jg\S*huff_write\S*):", text[i])
        if not m:
            i += 1
            continue
        name, body, j = m.group(1), [], i + 1
        while j < len(text) and not text[j].startswith(".Lfunc_end"):
            body.append(text[j])
            j += 1
        meta = []
        while j < len(text) and not re.match(r"^_ZN2jg\S*:", text[j]) and len(meta) < 200:
            meta.append(text[j])
            j += 1
        i = j
        seen += 1
        for key in (".amdhsa_private_segment_fixed_size", "; ScratchSize:"):
            for ln in meta:
                if key in ln and not re.search(r"[ :]0\s*$", ln.strip()):
                    problems.append("%s: %s" % (name, ln.strip()))
        # the hand-written blocks and the register their load targets
        blocks, inside, start = [], False, 0
        for k, ln in enumerate(body):
            if ";;#ASMSTART" in ln:
                inside, start = True, k
            elif ";;#ASMEND" in ln and inside:
                inside = False
                blocks.append((start, k))
        refills = [(a, b) for a, b in blocks if any("global_load_dword" in x for x in body[a:b]) and any("s_waitcnt vmcnt(" in x for x in body[a:b])]
        dones = [(a, b) for a, b in blocks if any(x.strip() == "s_waitcnt vmcnt(0)" for x in body[a:b]) and not any("global_load" in x for x in body[a:b])]
        if not refills or not dones:
            problems.append("%s: refill / done() assembly blocks not found" % name)
            continue
        regs = set()
        for a, b in refills:
            for x in body[a:b]:
                mm = re.match(r"\s*global_load_dword (v\d+),", x)
                if mm:
                    regs.add(mm.group(1))
        if len(regs) != 1:
            problems.append("%s: the refill blocks load into %s" % (name, sorted(regs)))
            continue
        reg = int(regs.pop()[1:])
        lo, hi = refills[0][0], max(b for _, b in dones)
        in_asm = [False] * len(body)
        for a, b in blocks:
            for k in range(a, b + 1):
                in_asm[k] = True
        for k in range(lo, hi + 1):
            ln = body[k].split(";")[0]
            if in_asm[k] or not ln.startswith("\t") or ln.strip().startswith("."):
                continue
            hit = re.search(r"\bv%d\b" % reg, ln) is not None
            for mm in re.finditer(r"v\[(\d+):(\d+)\]", ln):
                hit |= int(mm.group(1)) <= reg <= int(mm.group(2))
            if hit:
                problems.append("%s: v%d is touched outside the assembly blocks: %s" % (name, reg, ln.strip()))
    if seen == 0:
        problems.append("no huff_write kernel found in the generated code")
    return problems


def _compile(lib_path, verbose, extra_flags):
    os.makedirs(LIB_DIR, exist_ok=True)
    extra_flags = list(extra_flags)
    if os.environ.get("JPEGGPU_SAFE_REFILL") == "1" and "-DJG_SAFE_REFILL" not in extra_flags:
        extra_flags.append("-DJG_SAFE_REFILL")  # the selectable safe build: the refill waits for every load (vmcnt(0))
    if "-DJG_SAFE_REFILL" not in extra_flags and os.environ.get("JPEGGPU_SKIP_BUILD_CHECK") != "1":
        problems = check_refill(extra_flags, verbose)
        if problems:
            import sys

            sys.stderr.write("jpeggpu build: the counted-wait refill of the write pass does not pass its check with this compiler:\n  "
                             + "\n  ".join(problems[:8]) + "\n  -> building with -DJG_SAFE_REFILL (s_waitcnt vmcnt(0): a few per cent slower, always right)\n")
            extra_flags.append("-DJG_SAFE_REFILL")
    cmd = _base_cmd(extra_flags) + ["-fPIC", "-shared"]
    cmd += [os.path.join(CSRC, s) for s in SOURCES]
    cmd += ["-o", lib_path]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    with open(_mode_path(lib_path), "w") as f:  # what the library was built as: `_stale` compares it with the environment
        f.write("%s | refill: %s\n" % (_wanted_mode(), "vmcnt(0), safe" if "-DJG_SAFE_REFILL" in extra_flags else "vmcnt(1), checked"))
    return lib_path


if __name__ == "__main__":
    import sys

    # python jpeggpu_amd/build.py [out.so [extra hipcc flags...]]
    if len(sys.argv) > 1:
        print(build(out=os.path.abspath(sys.argv[1]), extra_flags=sys.argv[2:], verbose=True))
    else:
        print(build(force=True, verbose=True))
