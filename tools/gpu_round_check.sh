#!/bin/bash
# One gpurun call that checks a tree: the GPU tests, then (optional, by argument) calibration, bench and sweeps.
#   gpurun --timeout 1200 -- 'bash tools/gpu_round_check.sh tests calib bench sweep'
set -o pipefail
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
mkdir -p "$out"
for what in "$@"; do
case $what in
tests)
    timeout -k 10 900 python -m pytest tests -m gpu -x -q > "$out/pytest_gpu.log" 2>&1 || { tail -30 "$out/pytest_gpu.log"; exit 1; }
    tail -3 "$out/pytest_gpu.log" ;;
calib)
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 "$root/tools/probe/fetch_probe.hip" -o "$out/fetch_probe" || exit 1
    (cd /tmp && export TMPDIR=/tmp && rm -rf "$out/fetch_probe_rd" "$out/fetch_probe_wr" &&
     timeout -k 10 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d "$out/fetch_probe_rd" -o p --output-format csv -- "$out/fetch_probe" > "$out/fetch_probe_rd.log" 2>&1 &&
     timeout -k 10 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d "$out/fetch_probe_wr" -o p --output-format csv -- "$out/fetch_probe" > "$out/fetch_probe_wr.log" 2>&1) || { tail -5 "$out"/fetch_probe_*.log; exit 1; }
    python "$root/tools/probe/fetch_probe_summary.py" "$out/fetch_probe_rd" "$out/fetch_probe_wr" "$out/fetch_calibration.json" ;;
bench)
    timeout -k 10 500 python "$root/bench.py" > "$out/bench_default.json" 2> "$out/bench_default.err" || { tail -20 "$out/bench_default.err"; exit 1; }
    python "$root/tools/probe/show.py" default "$out/bench_default.json" || true ;;
sweep)
    (cd "$root" && timeout -k 10 300 python tools/probe/latency_by_size.py > "$out/latency_by_size.log" 2>&1) || { tail -5 "$out/latency_by_size.log"; exit 1; }
    cat "$out/latency_by_size.log" ;;
esac
done
