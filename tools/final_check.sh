#!/bin/bash
# The round's last GPU call, most important first (the box time left may cut it short): the GPU suite on the shipping library, smoke(), the
# driver's bench line, a kernel trace of the single-stream loop, the pyramid kernel A/B on 128 batched camera streams (instrumented
# build: RVIO_PYR_V1=1 = the 25-tap gather form), the default bench line.
# usage (on the GPU box, through gpurun): tools/final_check.sh <out dir under gpurun_out/>
set -u
OUT=gpurun_out/$1
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
date +%s > $OUT/t0
( time timeout -k 5 ${PYTEST_LIMIT:-420} python -m pytest tests -m gpu -q --maxfail=8 -p no:cacheprovider ) > $OUT/pytest.log 2>&1
echo "pytest rc=$?" >> $OUT/pytest.log; tail -6 $OUT/pytest.log
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?" >> $OUT/smoke.log; tail -2 $OUT/smoke.log
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver.json 2> $OUT/bench_driver.err; echo "bench rc=$?"
LEAN="--steps 20 --warmup 5 --no-cpu --no-streams --no-latency --batch '' --batch-streams 128"
eval RVIO_HIP_LIB=r-vio_amd/librvio_dbg.so RVIO_PYR_V1=1 timeout 150 python bench.py $LEAN > $OUT/streams128_pyr_v1.json 2> /dev/null
eval RVIO_HIP_LIB=r-vio_amd/librvio_dbg.so timeout 150 python bench.py $LEAN > $OUT/streams128_pyr_sep.json 2> /dev/null
eval timeout -k 5 200 rocprofv3 --kernel-trace --output-format rocpd -d $OUT/kt -o k -- python bench.py --steps 200 --warmup 40 --no-cpu --no-streams --no-latency --batch "''" --batch-streams "''" > /dev/null 2>&1
python tools/rocpd_stats.py $(find $OUT/kt -name "*.db" | head -1) $OUT/kernel_stats_stream.md > /dev/null 2>&1; rm -rf $OUT/kt
eval timeout -k 5 200 rocprofv3 --kernel-trace --output-format rocpd -d $OUT/kt -o k -- python bench.py $LEAN > /dev/null 2>&1
python tools/rocpd_stats.py $(find $OUT/kt -name "*.db" | head -1) $OUT/kernel_stats_streams128.md --grid-z 128 > /dev/null 2>&1; rm -rf $OUT/kt
timeout 300 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
date +%s > $OUT/t1
ls -la $OUT
