#!/bin/bash
# A/B of two builds of the library on the driver's command and on the steady state (same box, alternating):  tools/ab_lib.sh <base.so> [reps]
L="--no-cpu --no-latency --no-streams --batch= --batch-streams="
BASE=$1; REPS=${2:-2}
for rep in $(seq $REPS); do
for v in base new; do
  if [ $v = base ]; then export RVIO_HIP_LIB=$BASE; else unset RVIO_HIP_LIB; fi
  python bench.py --steps 20 --warmup 5 $L 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v steps20 %.0f frames/s %.4f ms host %.4f' % (d['value'], d['ms_per_step'], d.get('host_enqueue_ms_per_step')))"
  python bench.py --steps 200 --warmup 40 $L 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v steps200 %.0f frames/s %.4f ms host %.4f' % (d['value'], d['ms_per_step'], d.get('host_enqueue_ms_per_step')))"
done; done
