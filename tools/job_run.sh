set -u
mkdir -p gpurun_out/r06r
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_filter.py tests/test_gpu_golden.py tests/test_gpu_configs.py tests/test_gpu_flatout.py -x -q 2>&1 | tail -2
tools/ab_lib.sh r-vio_amd/librvio_base.so 3 2>&1 | tee gpurun_out/r06r/ab_joseph_scratch.txt
RVIO_HIP_LIB=r-vio_amd/librvio_dbg.so timeout 200 python tools/chain_clocks.py 200 2>&1 | grep -v amdgpu | sed -n 2,8p
