# In-run A/B of two builds of the library on ONE box (box-to-box variation of the same build is +-4 %):
#   git worktree add /tmp/wt HEAD && (cd /tmp/wt && python jpeggpu_amd/build.py) && cp /tmp/wt/jpeggpu_amd/lib/libjpeggpu.so jpeggpu_amd/lib/exp_base.so
#   gpurun -- 'bash tools/probe/ab.sh'      # exp_base = the committed tree, libjpeggpu = the working tree
# bench.py picks the library from JPEGGPU_LIB. Remove jpeggpu_amd/lib/exp_*.so afterwards.
for rep in 1 2; do for lib in exp_base libjpeggpu; do
JPEGGPU_LIB=$PWD/jpeggpu_amd/lib/$lib.so timeout -k 10 150 python bench.py --steps 6 --warmup 2 --no-cpu --e2e-rounds 0 > gpurun_out/b_x.log 2>&1 && python -c "
import json,sys; d=json.loads(open('gpurun_out/b_x.log').read().strip().splitlines()[-1]); print('$lib', round(d['value']), round(d['latency_ms']['p50'],3), {k:round(v) for k,v in d['stage_us_solo'].items()})"
JPEGGPU_LIB=$PWD/jpeggpu_amd/lib/$lib.so timeout -k 10 150 python bench.py --steps 4 --warmup 2 --batch 32 --streams 1 --no-cpu --e2e-rounds 0 --latency-iters 2 > gpurun_out/b_x.log 2>&1 && python -c "
import json,sys; d=json.loads(open('gpurun_out/b_x.log').read().strip().splitlines()[-1]); print('   1 stream x32:', round(d['value']), {k:round(v) for k,v in d['stage_us_under_load'].items()})"
done; done
