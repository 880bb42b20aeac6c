run() { timeout -k 10 150 python bench.py --steps 6 --warmup 2 --no-cpu --e2e-rounds 0 --latency-iters 2 "$@" > gpurun_out/b_x.log 2>&1 && python -c "
import json,sys; d=json.loads(open('gpurun_out/b_x.log').read().strip().splitlines()[-1]); print('$*', round(d['value']))"; }
run
run --sync-iters 2
run --streams 3
run --streams 6 --batch 96
run --streams 8 --batch 128
run --streams 4 --batch 128
run --subseq-bytes 256
run
