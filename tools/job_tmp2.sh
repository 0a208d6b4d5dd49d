set -u
mkdir -p gpurun_out/r06g
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
RVIO_HIP_LIB=r-vio_amd/librvio_dbg.so timeout 300 python tools/feat_phase_clocks.py 120 > gpurun_out/r06g/feat_phase.txt 2>&1
grep -v amdgpu gpurun_out/r06g/feat_phase.txt
timeout 900 python -m pytest tests/test_gpu_multi_rccl.py -x -q 2>&1 | tail -5
