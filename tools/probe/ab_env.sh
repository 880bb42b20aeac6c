# In-run A/B of ONE build under two settings of an environment variable (the library reads its JPEGGPU_EXP_* switches when a
# batch is created):   gpurun -- 'bash tools/probe/ab_env.sh JPEGGPU_EXP_FUSE_TAIL_WRITE 0 1'
var=$1; shift
for rep in 1 2; do for val in "$@"; do
env $var=$val timeout -k 10 200 python bench.py --steps 4 --warmup 1 --no-cpu --e2e-rounds 0 --latency-iters 30 --unique 4 --no-verify --other-configs 3 --photo-steps 2 --curve-iters 0 --shard-iters 0 $BENCH_ARGS > gpurun_out/b_x.log 2>&1 && python tools/probe/show.py "$var=$val" gpurun_out/b_x.log
done; done
