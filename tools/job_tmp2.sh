set -u
mkdir -p gpurun_out/r06h
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_filter.py tests/test_gpu_golden.py tests/test_gpu_configs.py tests/test_gpu_batch.py tests/test_gpu_solve9.py tests/test_gpu_frontend.py -x -q 2>&1 | tail -3
RVIO_HIP_LIB=r-vio_amd/librvio_dbg.so timeout 300 python tools/feat_phase_clocks.py 120 2>&1 | grep -v amdgpu | tee gpurun_out/r06h/feat_phase.txt
tools/ab_batch.sh r-vio_amd/librvio_base.so 2 2>&1 | tee gpurun_out/r06h/ab_batch.txt
echo "--- klt mul24 off (base) vs on (new)"
tools/ab_lib.sh r-vio_amd/librvio_nomul24.so 2 2>&1 | tee gpurun_out/r06h/ab_mul24.txt
