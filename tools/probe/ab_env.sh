#!/bin/bash
# In-run comparison of settings an environment variable selects, on one box, against the library of HEAD (exp_base.so):
#   gpurun -- 'bash tools/probe/ab_env.sh JPEGGPU_EXP_TAIL_PART "2048 1024 512 256"'
mkdir -p gpurun_out
common="--steps 4 --warmup 1 --no-cpu --e2e-rounds 0 --latency-iters 30 --unique 4 --other-configs 3 --photo-steps 2 --curve-iters 0 --shard-iters 0 $BENCH_ARGS"
var=$1
for rep in 1 2; do
if [ -f jpeggpu_amd/lib/exp_base.so ]; then
JPEGGPU_LIB=$PWD/jpeggpu_amd/lib/exp_base.so timeout -k 10 200 python bench.py $common > gpurun_out/b_base.log 2>&1 && python tools/probe/show.py base gpurun_out/b_base.log
fi
for v in $2; do
env $var=$v timeout -k 10 200 python bench.py $common > gpurun_out/b_$v.log 2>&1 && python tools/probe/show.py $var=$v gpurun_out/b_$v.log
done; done
