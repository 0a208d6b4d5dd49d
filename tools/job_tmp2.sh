set -u
mkdir -p gpurun_out/r06l
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export RVIO_HIP_LIB=r-vio_amd/librvio_dbg.so
for G in 0 16 8; do
RVIO_FEAT_GRID=$G timeout -k 5 300 rocprofv3 --kernel-trace --output-format rocpd -d gpurun_out/r06l/kt -o k -- python bench.py --no-streams --no-cpu --no-latency --steps 5 --warmup 2 --batch 2048 --no-defined-load --batch-streams "" > /dev/null 2>&1
echo "== grid $G"; python tools/rocpd_stats.py $(find gpurun_out/r06l/kt -name "*.db" | head -1) /dev/null --grid-z 2048 | head -6; rm -rf gpurun_out/r06l/kt
done
