#!/bin/bash
# Measurement refresh on the validated final tree (no product change after tools/final_check.sh): the SQ counters of the batched camera
# streams (-> profiles/r05_streams_issue_slots.json, which bench.py's batched_streams leg then prices its live frame time with), lean
# throughput lines of the other BASELINE configurations, the forced-sharded lines, the in-situ chain clocks, the driver's line again.
# usage (on the GPU box, through gpurun): tools/final_measure.sh <out dir under gpurun_out/>
set -u
OUT=gpurun_out/$1
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
date +%s > $OUT/t0
SLEAN="--steps 20 --warmup 5 --no-cpu --no-streams --no-latency --batch '' --batch-streams 128"
eval timeout -k 5 150 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_WAVES --output-format csv -d $OUT/pm -o m -- python bench.py $SLEAN > /dev/null 2>&1
CSV=$(find $OUT/pm -name "*counter_collection.csv" | head -1)
if [ -n "$CSV" ]; then
  python tools/issue_slots_json.py $CSV --streams 128 > $OUT/streams_issue_slots.json && cp $OUT/streams_issue_slots.json profiles/r05_streams_issue_slots.json
  python tools/pmc_table.py $CSV --min-workgroups 128 > $OUT/streams_sq_counters.md 2>&1
fi
rm -rf $OUT/pm
date +%s > $OUT/t1
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver.json 2> $OUT/bench_driver.err; echo "bench rc=$?"
for C in E A C D; do
  timeout 100 python bench.py --config $C --steps 60 --warmup 20 --batch '' --batch-streams '' --no-streams --no-cpu > $OUT/bench_cfg${C}_lean.json 2> /dev/null
done
date +%s > $OUT/t2
RVIO_HIP_LIB=r-vio_amd/librvio_dbg.so timeout 100 python tools/chain_clocks.py 200 > $OUT/chain_clocks.txt 2>&1
timeout 150 python bench.py --force-sharded --batch '' --batch-streams '' --no-cpu > $OUT/bench_forced_sharded_world1.json 2> /dev/null
timeout 150 python bench.py --force-sharded --config E --steps 40 --warmup 40 --batch '' --batch-streams '' --no-streams --no-cpu > $OUT/bench_forced_sharded_cfgE.json 2> /dev/null
date +%s > $OUT/t3
ls -la $OUT
