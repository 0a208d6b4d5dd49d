#!/bin/bash
# Round-end measurement set (run on the GPU box through gpurun): PMC traffic, kernel-trace summaries, the bench lines.
# usage: tools/measure_all.sh <out dir under gpurun_out/>
# NOTE: rocprofv3 --pmc serialises kernels across queues; the library sees ROCPROF_COUNTER_COLLECTION and switches its device-side polls (gate,
# refill half of book-keeping) to stream-level events by itself for those passes (same kernels otherwise).
#
set -u
OUT=gpurun_out/$1
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
LEAN="--steps 20 --warmup 5 --no-cpu --no-streams --no-latency --batch '' --batch-streams ''"
for C in ${PMC_CONFIGS:-B}; do
  for P in FETCH_SIZE WRITE_SIZE; do
    eval timeout -k 5 240 rocprofv3 --pmc $P --output-format csv -d $OUT/pmc_${C}_$P -o p -- python bench.py $LEAN --config $C > /dev/null 2>&1
  done
  F=$(find $OUT/pmc_${C}_FETCH_SIZE -name "*counter_collection.csv" | head -1); W=$(find $OUT/pmc_${C}_WRITE_SIZE -name "*counter_collection.csv" | head -1)
  python tools/pmc_traffic.py $F $W --json $OUT/pmc_traffic_cfg$C.json > $OUT/pmc_traffic_cfg$C.md 2>&1
  cp $OUT/pmc_traffic_cfg$C.json profiles/r06_pmc_traffic_cfg$C.json
  rm -rf $OUT/pmc_${C}_FETCH_SIZE $OUT/pmc_${C}_WRITE_SIZE
done
# kernel trace of the driver's own command
timeout -k 5 400 rocprofv3 --kernel-trace --output-format rocpd -d $OUT/kt -o k -- python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver_profiled.json 2> /dev/null
python tools/rocpd_stats.py $(find $OUT/kt -name "*.db" | head -1) $OUT/kernel_stats_driver_cmd.md > /dev/null; rm -rf $OUT/kt
# kernel trace of the plain single-stream loop only (no secondary legs): the per-frame kernels
eval timeout -k 5 300 rocprofv3 --kernel-trace --output-format rocpd -d $OUT/kt -o k -- python bench.py --steps 200 --warmup 40 --no-cpu --no-streams --no-latency --batch "''" --batch-streams "''" > /dev/null 2>&1
python tools/rocpd_stats.py $(find $OUT/kt -name "*.db" | head -1) $OUT/kernel_stats_stream.md > /dev/null; rm -rf $OUT/kt
# the per-frame kernels of the long-window configurations (the split solve, the tile-parallel Joseph kernels, the Cholesky factor on its own queue)
for C in ${KT_CONFIGS:-}; do
  eval timeout -k 5 300 rocprofv3 --kernel-trace --output-format rocpd -d $OUT/kt -o k -- python bench.py --config $C --steps 100 --warmup 20 --no-cpu --no-streams --no-latency --batch "''" --batch-streams "''" > /dev/null 2>&1
  python tools/rocpd_stats.py $(find $OUT/kt -name "*.db" | head -1) $OUT/kernel_stats_stream_cfg$C.md > /dev/null; rm -rf $OUT/kt
done
# batched filter at B = 2048
eval timeout -k 5 300 rocprofv3 --kernel-trace --output-format rocpd -d $OUT/kt -o k -- python bench.py --no-streams --no-cpu --no-latency --steps 5 --warmup 2 --batch 2048 --no-defined-load --batch-streams "''" > /dev/null 2>&1
python tools/rocpd_stats.py $(find $OUT/kt -name "*.db" | head -1) $OUT/batched_kernel_stats.md --grid-z 2048 > /dev/null; rm -rf $OUT/kt
eval timeout -k 5 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 --output-format csv -d $OUT/pm -o m -- python bench.py --no-streams --no-cpu --no-latency --steps 5 --warmup 2 --batch 2048 --no-defined-load --batch-streams "''" > /dev/null 2>&1
python tools/pmc_table.py $(find $OUT/pm -name "*counter_collection.csv" | head -1) --min-workgroups 2048 > $OUT/batched_mfma_counters.md 2>&1
python tools/mfma_util_json.py $(find $OUT/pm -name "*counter_collection.csv" | head -1) --instances 2048 > $OUT/batched_mfma_util.json 2>/dev/null && cp $OUT/batched_mfma_util.json profiles/r06_batched_mfma_util.json
rm -rf $OUT/pm
# the bench lines
timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver.json 2> $OUT/bench_driver.err
timeout 400 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
timeout 400 python bench.py --force-sharded --batch '' --batch-streams '' > $OUT/bench_forced_sharded_world1.json 2> $OUT/bench_forced_sharded_world1.err
timeout 400 python bench.py --force-sharded --config E --steps 40 --warmup 40 --batch '' --batch-streams '' --no-streams --no-cpu > $OUT/bench_forced_sharded_cfgE.json 2> $OUT/bench_forced_sharded_cfgE.err
python tools/solve9_probe.py B A C E > $OUT/solve9_probe.txt 2>&1
for C in ${BENCH_CONFIGS:-}; do timeout 400 python bench.py --config $C --steps 60 --warmup 20 --batch '' --batch-streams '' --no-streams > $OUT/bench_cfg$C.json 2> /dev/null; done
RVIO_HIP_LIB=r-vio_amd/librvio_dbg.so timeout 200 python tools/chain_clocks.py 200 > $OUT/chain_clocks.txt 2>&1
ls -la $OUT
RVIO_HIP_LIB=r-vio_amd/librvio_dbg.so timeout 200 python tools/side_phase_clocks.py 60 > $OUT/side_phase_clocks.txt 2>&1
RVIO_HIP_LIB=r-vio_amd/librvio_dbg.so timeout 200 python tools/feat_phase_clocks.py 120 > $OUT/feat_phase_clocks.txt 2>&1
# SQ instruction counters of the batched filter (B = 2048): instructions per feature of the per-feature kernel
eval timeout -k 5 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_WAVES --output-format csv -d $OUT/pq -o q -- python bench.py --no-streams --no-cpu --no-latency --steps 5 --warmup 2 --batch 2048 --no-defined-load --batch-streams "''" > /dev/null 2>&1
python tools/pmc_table.py $(find $OUT/pq -name "*counter_collection.csv" | head -1) --min-workgroups 2048 > $OUT/batched_sq_counters.md 2>&1
rm -rf $OUT/pq
ls -la $OUT
