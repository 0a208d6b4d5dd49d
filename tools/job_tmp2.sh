set -u
mkdir -p gpurun_out/r06i
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_filter.py tests/test_gpu_sharded_ranks.py tests/test_gpu_multi_rccl.py tests/test_gpu_configs.py tests/test_gpu_flatout.py -x -q 2>&1 | tail -4
L="--no-cpu --no-latency --no-streams --batch= --batch-streams="
for rep in 1 2; do
python bench.py --steps 200 --warmup 40 $L 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('plain   steps200 %.0f frames/s' % d['value'])"
python bench.py --steps 200 --warmup 40 $L --force-sharded 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('sharded steps200 %.0f frames/s' % d['value'], d.get('max_state_delta_sharded_vs_single_gpu'))"
done
python bench.py --config E --steps 40 --warmup 40 $L 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('plain   cfgE %.0f frames/s' % d['value'])"
python bench.py --config E --steps 40 --warmup 40 $L --force-sharded 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('sharded cfgE %.0f frames/s' % d['value'])"
