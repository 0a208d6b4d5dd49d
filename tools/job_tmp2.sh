set -u
mkdir -p gpurun_out/r06j
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_filter.py tests/test_gpu_golden.py tests/test_gpu_configs.py tests/test_gpu_solve9.py tests/test_gpu_flatout.py tests/test_gpu_truncation.py tests/test_gpu_frontend.py tests/test_gpu_sharded_ranks.py -x -q 2>&1 | tail -3
tools/ab_lib.sh r-vio_amd/librvio_base.so 2 2>&1 | tee gpurun_out/r06j/ab_dx_role.txt
RVIO_HIP_LIB=r-vio_amd/librvio_dbg.so timeout 200 python tools/chain_clocks.py 200 2>&1 | grep -v amdgpu | head -9
