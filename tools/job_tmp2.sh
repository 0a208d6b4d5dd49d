set -u
mkdir -p gpurun_out/r06k
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_flatout.py tests/test_gpu_golden.py tests/test_gpu_configs.py tests/test_gpu_frontend.py tests/test_gpu_truncation.py tests/test_gpu_z_host.py tests/test_gpu_edges.py -x -q 2>&1 | tail -3
tools/ab_lib.sh r-vio_amd/librvio_base.so 3 2>&1 | tee gpurun_out/r06k/ab_gate.txt
