#!/bin/bash
# A/B of two builds of the library on the batched legs (same box, alternating):  tools/ab_batch.sh <base.so> [reps]
BASE=$1; REPS=${2:-2}
for rep in $(seq $REPS); do
for v in base new; do
  if [ $v = base ]; then export RVIO_HIP_LIB=$BASE; else unset RVIO_HIP_LIB; fi
  python bench.py --steps 20 --warmup 5 --no-cpu --no-latency --no-streams --batch 2048 --batch-streams 128 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
bf=d['batched_filter']['sizes'][-1]; bs=d['batched_streams']['sizes'][-1]; dl=d.get('batched_filter_at_defined_load',{}).get('sizes',[{}])[-1]
print('$v filter B=2048 %.0f frames/s %.4f ms frac %.4f | defined load %s | streams 128: %.0f frames/s' % (bf['filter_frames_per_s'], bf['ms_per_batched_frame'], bf['frac_fp64_peak'], {k:(round(v['ms_per_batched_frame'],3), round(v['frac_fp64_peak'],4)) for k,v in dl.items() if isinstance(v,dict) and 'frac_fp64_peak' in v}, bs['frames_per_s']))"
done; done
