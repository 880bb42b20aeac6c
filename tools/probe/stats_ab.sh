#!/bin/bash
# Per-kernel average durations AT THROUGHPUT (4 streams, kernels of different groups overlap) for several builds of
# the library on one box: rocprofv3 --kernel-trace --stats of the same short bench run each.
#   gpurun -- 'bash tools/probe/stats_ab.sh exp_base libjpeggpu'
root=$PWD
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
for lib in "$@"; do
    d=$root/gpurun_out/stats_$lib
    rm -rf "$d"
    JPEGGPU_LIB=$root/jpeggpu_amd/lib/$lib.so timeout -k 10 200 rocprofv3 --kernel-trace --stats -d "$d" -o s --output-format csv -- \
        python3 "$root/bench.py" --steps 4 --warmup 1 --no-cpu --e2e-rounds 0 --latency-iters 0 --unique 4 --no-verify --other-configs 0 --roofline-launches 0 > "$d.json" 2> "$d.err" || { tail -5 "$d.err"; exit 1; }
    echo "== $lib: $(python3 -c "import json,sys; print(round(json.loads(open('$d.json').read().strip().splitlines()[-1])['value']))") img/s"
    python3 - "$d" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    n = r["Name"].split("(")[0].replace("void jg::(anonymous namespace)::", "")[:60]
    print("   %-60s calls %6s avg %9.1f us total %8.1f ms" % (n, r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
done
