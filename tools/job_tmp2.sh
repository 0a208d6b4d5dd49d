set -u
mkdir -p gpurun_out/r06d
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
RVIO_HIP_LIB=r-vio_amd/librvio_dbg.so timeout 200 python tools/chain_clocks.py 200 > gpurun_out/r06d/chain_clocks.txt 2>&1
RVIO_HIP_LIB=r-vio_amd/librvio_dbg.so timeout 200 python tools/chain_clocks.py 25 > gpurun_out/r06d/chain_clocks_25.txt 2>&1
timeout -k 5 300 rocprofv3 --kernel-trace --output-format rocpd -d gpurun_out/r06d/kt -o k -- python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu --no-streams --no-latency --batch '' --batch-streams '' > gpurun_out/r06d/bench_prof.json 2>/dev/null
DB=$(find gpurun_out/r06d/kt -name "*.db" | head -1)
python tools/timeline.py $DB 0 100000 > gpurun_out/r06d/timeline_driver.txt 2>&1
python tools/rocpd_stats.py $DB gpurun_out/r06d/kernel_stats_driver.md > /dev/null
rm -rf gpurun_out/r06d/kt
cat gpurun_out/r06d/chain_clocks.txt
