#!/bin/bash
# A/B of the literal path (round 6) on the driver's command and on the steady state: RVIO_NO_LITERAL=1 is the round-5 behaviour
L="--no-cpu --no-latency --no-streams --batch= --batch-streams="
for rep in 1 2; do
for v in 0 1; do
  if [ $v = 1 ]; then export RVIO_NO_LITERAL=1; else unset RVIO_NO_LITERAL; fi
  python bench.py --steps 20 --warmup 5 $L 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('no_literal=$v steps20', d['value'], d['ms_per_step'], d.get('host_enqueue_ms_per_step'))"
  python bench.py --steps 200 --warmup 40 $L 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('no_literal=$v steps200', d['value'], d['ms_per_step'], d.get('host_enqueue_ms_per_step'))"
done; done
