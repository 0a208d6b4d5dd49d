#!/bin/bash
# PMC traffic + bench line (with parity) of the other BASELINE configurations.  usage: tools/measure_configs.sh <out dir under gpurun_out/> [configs...]
set -u
OUT=gpurun_out/$1; shift
CFGS=${@:-A C E D}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
LEAN="--steps 20 --warmup 5 --no-cpu --no-streams --no-latency --batch '' --batch-streams ''"
for C in $CFGS; do
  for P in FETCH_SIZE WRITE_SIZE; do
    eval RVIO_NO_DEVFLAG=1 timeout 200 rocprofv3 --pmc $P --output-format csv -d $OUT/pmc_${C}_$P -o p -- python bench.py $LEAN --config $C > /dev/null 2>&1
  done
  F=$(find $OUT/pmc_${C}_FETCH_SIZE -name "*counter_collection.csv" | head -1); W=$(find $OUT/pmc_${C}_WRITE_SIZE -name "*counter_collection.csv" | head -1)
  python tools/pmc_traffic.py $F $W --json $OUT/pmc_traffic_cfg$C.json > $OUT/pmc_traffic_cfg$C.md 2>&1
  cp $OUT/pmc_traffic_cfg$C.json profiles/r03_pmc_traffic_cfg$C.json
  rm -rf $OUT/pmc_${C}_FETCH_SIZE $OUT/pmc_${C}_WRITE_SIZE
  timeout 260 python bench.py --config $C --steps 60 --warmup 20 --batch '' --batch-streams '' --no-streams > $OUT/bench_cfg$C.json 2> /dev/null
  python - <<EOF
import json
try:
    d=json.loads(open("$OUT/bench_cfg$C.json").read().strip().splitlines()[-1])
    print("$C", round(d["value"]), d["p50_ekf_update_ms"], d["parity"]["max_state_delta"], d["roofline"]["kernel"][:24], round(d["roofline"]["avg_us"],1), d["roofline"]["traffic"])
except Exception as e: print("$C failed", e)
EOF
done
