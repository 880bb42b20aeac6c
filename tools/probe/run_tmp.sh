mkdir -p gpurun_out
echo "== keep flows, cfg2 256/128 at 16..64"; JPEGGPU_EXP_KEEP_FLOWS_BELOW=1000000000 timeout -k 10 300 python tools/probe/batch_curve.py --images 16,24,32,48,64 --sizes 128,256 2>&1 | grep -v amdgpu.ids
echo "== marks, cfg2"; JPEGGPU_EXP_KEEP_FLOWS_BELOW=0 timeout -k 10 300 python tools/probe/batch_curve.py --images 16,24,32,48,64 --sizes 128,256 2>&1 | grep -v amdgpu.ids
echo "== keep flows, photo"; JPEGGPU_EXP_KEEP_FLOWS_BELOW=1000000000 timeout -k 10 300 python tools/probe/batch_curve.py --workload photo --images 16,32,64 --sizes 128,256 2>&1 | grep -v amdgpu.ids
echo "== marks, photo"; JPEGGPU_EXP_KEEP_FLOWS_BELOW=0 timeout -k 10 300 python tools/probe/batch_curve.py --workload photo --images 16,32,64 --sizes 128,256 2>&1 | grep -v amdgpu.ids
