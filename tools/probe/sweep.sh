run() { timeout -k 10 200 python bench.py --steps 4 --warmup 1 --no-cpu --e2e-rounds 0 --latency-iters 0 --unique 4 --no-verify --roofline-launches 0 "$@" > gpurun_out/b_x.log 2>&1 && python tools/probe/show.py "$*" gpurun_out/b_x.log; }
run
run --subseq-bytes 256
run --subseq-bytes 64
run --batch 64
run --batch 256
run --streams 2
run --streams 8
run --streams 1 --overlap 4
run --sync-iters 2
run
