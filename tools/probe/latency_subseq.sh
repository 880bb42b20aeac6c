# p50 of the reference's protocol for each subsequence size the library supports (environment override)
for sb in 32 64 128 256; do
JPEGGPU_SUBSEQ_BYTES=$sb timeout -k 10 200 python bench.py --steps 1 --warmup 1 --batch 8 --rounds 1 --unique 1 --no-cpu --e2e-rounds 0 --latency-iters 100 --other-configs 0 --no-verify --roofline-launches 0 --subseq-bytes 128 $BENCH_ARGS > gpurun_out/b_x.log 2>&1 && python - <<PY
import json
d=json.loads(open('gpurun_out/b_x.log').read().strip().splitlines()[-1])
for k in ('latency_ms','latency_ms_device_scan'):
    l=d[k]; print($sb, k, 'sb', l['subsequence_bytes'], 'p50 %.3f parse %.3f' % (l['p50'], l['p50_host_parse']), {a:round(b) for a,b in l['stage_us_device'].items()})
PY
done
