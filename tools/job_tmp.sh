set -u
mkdir -p gpurun_out/r06c
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_frontend.py tests/test_gpu_flatout.py tests/test_gpu_edges.py tests/test_gpu_branches.py tests/test_gpu_golden.py tests/test_gpu_configs.py tests/test_gpu_batch.py -x -q 2>&1 | tail -3
RVIO_HIP_LIB=r-vio_amd/librvio_dbg.so timeout 300 python tools/side_phase_clocks.py 60 > gpurun_out/r06c/side_phase.txt 2>&1
cat gpurun_out/r06c/side_phase.txt | grep -v amdgpu.ids
tools/ab_lib.sh r-vio_amd/librvio_base.so 2 2>&1 | tee gpurun_out/r06c/ab.txt
