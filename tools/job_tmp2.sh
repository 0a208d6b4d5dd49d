set -u
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export RVIO_HIP_LIB=r-vio_amd/librvio_dbg.so
for T in 128 64 128 64 256; do
RVIO_FEAT_THREADS=$T python bench.py --steps 20 --warmup 5 --no-cpu --no-latency --no-streams --batch 2048 --batch-streams '' 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
bf=d['batched_filter']['sizes'][-1]; dl=d.get('batched_filter_at_defined_load',{}).get('sizes',[{}])[-1]
print('T=$T filter B=2048 %.0f frames/s %.4f ms frac %.4f | defined load %s' % (bf['filter_frames_per_s'], bf['ms_per_batched_frame'], bf['frac_fp64_peak'], {k:(round(v['ms_per_batched_frame'],3), round(v['frac_fp64_peak'],4)) for k,v in dl.items() if isinstance(v,dict) and 'frac_fp64_peak' in v}))"
done
