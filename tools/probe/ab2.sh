# In-run A/B over builds AND one environment switch: bash tools/probe/ab2.sh VAR "lib:val lib:val ..."
var=$1; shift
for rep in 1 2; do for lv in $@; do lib=${lv%%:*}; val=${lv##*:}
env JPEGGPU_LIB=$PWD/jpeggpu_amd/lib/$lib.so $var=$val timeout -k 10 200 python bench.py --steps 4 --warmup 1 --no-cpu --e2e-rounds 0 --latency-iters 30 --unique 4 --no-verify --other-configs 3 --photo-steps 2 --curve-iters 0 --shard-iters 0 $BENCH_ARGS > gpurun_out/b_x.log 2>&1 && python tools/probe/show.py "$lib,$var=$val" gpurun_out/b_x.log
done; done
