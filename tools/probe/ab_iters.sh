#!/bin/bash
# In-run A/B on one box: the library of HEAD (exp_base.so, tools/probe/ab.sh says how it is built) against the working
# tree's at several caps of the sequence kernel's flow iterations; verified against the oracle.
#   gpurun -- 'bash tools/probe/ab_iters.sh "1 2 3 8"'
mkdir -p gpurun_out
common="--steps 4 --warmup 1 --no-cpu --e2e-rounds 0 --latency-iters 30 --unique 4 --other-configs 3 --photo-steps 2 --curve-iters 0 --shard-iters 0"
for rep in 1 2; do
JPEGGPU_LIB=$PWD/jpeggpu_amd/lib/exp_base.so timeout -k 10 200 python bench.py $common > gpurun_out/b_base.log 2>&1 && python tools/probe/show.py base gpurun_out/b_base.log
for it in $1; do
timeout -k 10 200 python bench.py $common --sync-iters $it > gpurun_out/b_it$it.log 2>&1 && python tools/probe/show.py iters$it gpurun_out/b_it$it.log
done; done
