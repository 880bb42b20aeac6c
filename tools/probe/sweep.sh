run() { timeout -k 10 200 python bench.py --steps 4 --warmup 1 --no-cpu --e2e-rounds 0 --latency-iters 0 --other-configs 0 --unique 4 --no-verify --roofline-launches 0 "$@" > gpurun_out/b_x.log 2>&1 && python tools/probe/show.py "$*" gpurun_out/b_x.log; }
run --subseq-bytes 256
run --subseq-bytes 256 --streams 6 --batch 192
run --subseq-bytes 256 --streams 8 --batch 256
run --subseq-bytes 256 --streams 4 --batch 256
run --subseq-bytes 256 --streams 6 --batch 96
run --subseq-bytes 256 --streams 8 --batch 128
run --subseq-bytes 256 --streams 6 --batch 192 --sync-iters 2
run --subseq-bytes 256 --streams 6 --batch 192
