mkdir -p gpurun_out
C="--no-cpu --e2e-rounds 0 --latency-iters 0 --no-verify --other-configs 0 --curve-iters 0 --shard-iters 0"
for args in "--steps 4 --warmup 1 --unique 4" "--steps 4 --warmup 1 --unique 16" "--steps 20 --warmup 5 --unique 4" "--steps 20 --warmup 5 --unique 16" "--steps 4 --warmup 1 --unique 4"; do
  timeout -k 10 200 python bench.py $C $args > gpurun_out/b_x.log 2>&1 && python tools/probe/show.py "[$args]" gpurun_out/b_x.log | tail -1
done
