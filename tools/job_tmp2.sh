set -u
mkdir -p gpurun_out/r06f
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 600 python tools/at_rest_literal.py > gpurun_out/r06f/at_rest_literal.txt 2>&1
grep -v amdgpu gpurun_out/r06f/at_rest_literal.txt
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r06f/bench_driver.json 2> gpurun_out/r06f/bench_driver.err
tail -c 300 gpurun_out/r06f/bench_driver.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r06f/bench_driver.json').read().strip().splitlines()[-1])
print(list(d.keys())[:22])
print(d['value'], d['ms_per_step'])
r=d['roofline']; print(r['kernel'][:60], r['avg_us'], r['frac'])
for c in d['roofline_other']: print('  ', c['kernel'][:60], c['avg_us'], c['frac'])
print(d.get('latency_ms_p50'), d.get('p50_ekf_update_ms'))
print(d.get('host_buffers'))
print(d.get('timed_run_parity'), d.get('parity',{}).get('max_state_delta'))
PY
