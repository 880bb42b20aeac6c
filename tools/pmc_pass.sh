#!/bin/bash
# Runs rocprofv3 counter passes over the batched bench (one pass per counter group; --pmc only with
# --kernel-trace, as the pool requires) and prints per-kernel averages. Usage on the GPU box:
#   bash tools/pmc_pass.sh <tag> "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_INSTS_VALU ..."
set -e
tag=$1; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
i=0
for group in "$@"; do
    out=$root/gpurun_out/${tag}_$i
    rm -rf "$out"
    timeout -k 10 400 rocprofv3 --kernel-trace --pmc $group -d "$out" -o bench --output-format csv -- \
        python3 "$root/bench.py" --steps 1 --warmup 1 --batch 64 --rounds 1 --streams 1 --unique 2 --no-cpu --no-verify --latency-iters 0 --other-configs 0 --e2e-rounds 0 --roofline-launches 3 \
        > "$out.json" 2> "$out.err"
    python3 "$root/tools/pmc_summary.py" "$out" | tee "$out.txt"
    i=$((i + 1))
done
