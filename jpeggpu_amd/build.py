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
    instantiation of huff_write and huff_tail_write: no scratch memory, no spilled registers, and between the loop's first refill block and
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
        if ".name:" in ln and ("huff_write" in ln or "huff_tail_write" in ln):
            for x in text[k:k + 14]:
                if ("spill_count" in x or ".private_segment_fixed_size" in x) and not x.strip().endswith(" 0"):
                    problems.append("%s: %s" % (ln.split(".name:")[1].strip(), x.strip()))
    i = 0
    while i < len(text):
        m = re.match(r"^(_ZN2jg\S*huff_(?:tail_)?write\S*):", text[i])
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
        # Where may a load into that register be in flight? On the paths from a refill block to the next refill or to done():
        # the body as a graph of straight-line pieces (cut at labels, behind branches and around the assembly blocks), the
        # pieces reachable FROM a refill block that also REACH a refill or a done() block. (By position in the text -- from
        # the first refill to the last done() -- the check took in code the compiler had laid out in between without it
        # lying on any such path: the other role of huff_tail_write.)
        cuts = {0, len(body)}
        for a, b in blocks:
            cuts.add(a)
            cuts.add(b + 1)
        label_at = {}
        for k, ln in enumerate(body):
            mm = re.match(r"^(\.LBB\S+):", ln)
            if mm:
                cuts.add(k)
                label_at[mm.group(1)] = k
            op = ln.split(";")[0].split()
            if op and ln.startswith("\t") and (op[0].startswith("s_cbranch") or op[0] in ("s_branch", "s_endpgm", "s_setpc_b64")):
                cuts.add(k + 1)
        starts = sorted(c for c in cuts if c < len(body))
        node_of = {}
        for n, st in enumerate(starts):
            for k in range(st, starts[n + 1] if n + 1 < len(starts) else len(body)):
                node_of[k] = n
        succ = [set() for _ in starts]
        for n, st in enumerate(starts):
            en = starts[n + 1] if n + 1 < len(starts) else len(body)
            last = None
            for k in range(st, en):
                op = body[k].split(";")[0].split()
                if op and body[k].startswith("\t") and not op[0].startswith("."):
                    last = op
            falls = True
            if last and (last[0].startswith("s_cbranch") or last[0] == "s_branch"):
                tgt = last[1].rstrip(",") if len(last) > 1 else ""
                if tgt in label_at:
                    succ[n].add(node_of[label_at[tgt]])
                else:
                    problems.append("%s: branch to an unknown label: %s" % (name, " ".join(last)))
                falls = last[0] != "s_branch"
            elif last and last[0] in ("s_endpgm", "s_setpc_b64"):
                falls = last[0] != "s_endpgm"
                if last[0] == "s_setpc_b64":
                    problems.append("%s: an indirect branch inside the kernel" % name)
            if falls and n + 1 < len(starts):
                succ[n].add(n + 1)
        pred = [set() for _ in starts]
        for n, ss in enumerate(succ):
            for m in ss:
                pred[m].add(n)

        def closure(seeds, edges, stop=()):
            seen_n, todo = set(), list(seeds)
            while todo:
                x = todo.pop()
                for y in edges[x]:
                    if y not in seen_n:
                        seen_n.add(y)
                        if y not in stop:
                            todo.append(y)
            return seen_n

        refill_nodes = {node_of[a] for a, _ in refills}
        done_nodes = {node_of[a] for a, _ in dones}
        # (forwards no further than a done() block: behind it nothing is in flight -- the structured exit of one role of
        # huff_tail_write leads, on paper, into the other's code)
        zone = closure(refill_nodes, succ, done_nodes) & (closure(refill_nodes | done_nodes, pred) | refill_nodes | done_nodes)
        in_asm = [False] * len(body)
        for a, b in blocks:
            for k in range(a, b + 1):
                in_asm[k] = True
        for k in range(len(body)):
            if node_of[k] not in zone:
                continue
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
