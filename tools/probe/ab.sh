# In-run A/B of builds of the library on ONE box (box-to-box variation of the same build is +-4 %):
#   python jpeggpu_amd/build.py jpeggpu_amd/lib/exp_x.so [-DFLAG ...]      # an experimental build of the working tree
#   git worktree add /tmp/wt HEAD && (cd /tmp/wt && python jpeggpu_amd/build.py) && cp /tmp/wt/jpeggpu_amd/lib/libjpeggpu.so jpeggpu_amd/lib/exp_base.so
#   gpurun -- 'bash tools/probe/ab.sh exp_base libjpeggpu'
# bench.py picks the library from JPEGGPU_LIB. Remove jpeggpu_amd/lib/exp_*.so afterwards.
libs=${@:-exp_base libjpeggpu}
for rep in 1 2; do for lib in $libs; do
JPEGGPU_LIB=$PWD/jpeggpu_amd/lib/$lib.so timeout -k 10 200 python bench.py --steps 4 --warmup 1 --no-cpu --e2e-rounds 0 --latency-iters 30 --unique 4 --no-verify --other-configs 3 --photo-steps 2 --curve-iters 0 --shard-iters 0 $BENCH_ARGS > gpurun_out/b_x.log 2>&1 && python tools/probe/show.py $lib gpurun_out/b_x.log
done; done
