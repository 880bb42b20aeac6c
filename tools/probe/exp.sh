for lib in "$@"; do
JPEGGPU_LIB=$PWD/jpeggpu_amd/lib/$lib.so timeout -k 10 150 python bench.py --steps 4 --warmup 2 --batch 32 --streams 1 --no-cpu --e2e-rounds 0 --latency-iters 2 > gpurun_out/b_x.log 2>&1 && python -c "
import json,sys; d=json.loads(open('gpurun_out/b_x.log').read().strip().splitlines()[-1]); print('$lib 1 stream x32:', round(d['value']), {k:round(v) for k,v in d['stage_us_under_load'].items()})"
done
