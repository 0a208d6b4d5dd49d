#!/bin/bash
# Last seconds of the round's GPU budget, on the adopted strip parameters: the SQ counters of the batched camera streams again, the
# driver's bench line (which then prices its live frame time with them), the kernel trace of 128 streams.
set -u
OUT=gpurun_out/$1
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
SLEAN="--steps 20 --warmup 5 --no-cpu --no-streams --no-latency --batch '' --batch-streams 128"
eval timeout -k 5 60 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_WAVES --output-format csv -d $OUT/pm -o m -- python bench.py $SLEAN > /dev/null 2>&1
CSV=$(find $OUT/pm -name "*counter_collection.csv" | head -1)
if [ -n "$CSV" ]; then
  python tools/issue_slots_json.py $CSV --streams 128 > $OUT/streams_issue_slots.json && cp $OUT/streams_issue_slots.json profiles/r05_streams_issue_slots.json
  python tools/pmc_table.py $CSV --min-workgroups 128 > $OUT/streams_sq_counters.md 2>&1
fi
rm -rf $OUT/pm
timeout 100 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver.json 2> $OUT/bench_driver.err; echo "bench rc=$?"
eval timeout -k 5 60 rocprofv3 --kernel-trace --output-format rocpd -d $OUT/kt -o k -- python bench.py $SLEAN > /dev/null 2>&1
python tools/rocpd_stats.py $(find $OUT/kt -name "*.db" | head -1) $OUT/kernel_stats_streams128.md --grid-z 128 > /dev/null 2>&1; rm -rf $OUT/kt
ls -la $OUT
