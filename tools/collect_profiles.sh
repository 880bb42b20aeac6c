#!/bin/bash
# Collects the profiles of one round on the GPU box (run through gpurun from the repository root):
#   bash tools/collect_profiles.sh r02
# 1. the bench line of the default command                              -> gpurun_out/<round>_bench_default.json
# 2. rocprofv3 --kernel-trace --stats of the default command             -> gpurun_out/<round>_stats_default/
# 3. the same of a run whose batched launches are all SERIALIZED (one stream, 64 images per launch: the shape
#    of bench.py's roofline leg)                                         -> gpurun_out/<round>_stats_serialized/
# 4. counter passes over the serialized run, one --pmc group per pass     -> gpurun_out/<round>_pmc_<i>/
# 5. the FETCH_SIZE / WRITE_SIZE calibration of tools/probe/fetch_probe.hip  -> gpurun_out/fetch_calibration.json
# tools/summarize_profiles.py then writes the summaries into profiles/ (run it here, commit the result).
set -e
rnd=$1
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
ser="--steps 1 --warmup 1 --batch 64 --rounds 2 --streams 1 --unique 4 --no-cpu --no-verify --latency-iters 0 --other-configs 0 --e2e-rounds 0 --curve-iters 0 --shard-iters 0 --roofline-launches 8"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 python3 "$root/bench.py" > "$out/${rnd}_bench_default.json" 2> "$out/${rnd}_bench_default.err"
echo "bench done"
rm -rf "$out/${rnd}_stats_default" "$out/${rnd}_stats_serialized"
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d "$out/${rnd}_stats_default" -o bench --output-format csv -- \
    python3 "$root/bench.py" --steps 5 --warmup 2 --no-cpu --e2e-rounds 0 --other-configs 0 --curve-iters 0 --shard-iters 0 > "$out/${rnd}_stats_default.json" 2> "$out/${rnd}_stats_default.err"
echo "stats default done"
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d "$out/${rnd}_stats_serialized" -o bench --output-format csv -- \
    python3 "$root/bench.py" $ser > "$out/${rnd}_stats_serialized.json" 2> "$out/${rnd}_stats_serialized.err"
echo "stats serialized done"
i=0
for group in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD"; do
    d="$out/${rnd}_pmc_$i"
    rm -rf "$d"
    timeout -k 10 200 rocprofv3 --kernel-trace --pmc $group -d "$d" -o bench --output-format csv -- \
        python3 "$root/bench.py" $ser > "$d.json" 2> "$d.err"
    python3 "$root/tools/pmc_summary.py" "$d" | tee "$d.txt"
    i=$((i + 1))
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 "$root/tools/probe/fetch_probe.hip" -o "$out/fetch_probe"
rm -rf "$out/fetch_probe_rd" "$out/fetch_probe_wr"
timeout -k 10 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d "$out/fetch_probe_rd" -o p --output-format csv -- "$out/fetch_probe" > "$out/fetch_probe_rd.log" 2>&1
timeout -k 10 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d "$out/fetch_probe_wr" -o p --output-format csv -- "$out/fetch_probe" > "$out/fetch_probe_wr.log" 2>&1
python3 "$root/tools/probe/fetch_probe_summary.py" "$out/fetch_probe_rd" "$out/fetch_probe_wr" "$out/fetch_calibration.json"
