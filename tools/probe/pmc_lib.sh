#!/bin/bash
# rocprofv3 counter passes over the batched bench for a given build of the library (JPEGGPU_LIB):
#   bash tools/probe/pmc_lib.sh <tag> <lib name under jpeggpu_amd/lib without .so> "COUNTERS A" "COUNTERS B" ...
set -e
tag=$1; lib=$2; shift; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
export JPEGGPU_LIB=$root/jpeggpu_amd/lib/$lib.so
cd /tmp && export TMPDIR=/tmp
i=0
for group in "$@"; do
    out=$root/gpurun_out/${tag}_$i
    rm -rf "$out"
    timeout -k 10 150 rocprofv3 --kernel-trace --pmc $group -d "$out" -o bench --output-format csv -- \
        python3 "$root/bench.py" --steps 1 --warmup 1 --batch 64 --rounds 1 --streams 1 --unique 2 --no-cpu --no-verify --latency-iters 0 --other-configs 0 --e2e-rounds 0 --roofline-launches 3 \
        > "$out.json" 2> "$out.err"
    python3 "$root/tools/pmc_summary.py" "$out" | tee "$out.txt"
    i=$((i + 1))
done
