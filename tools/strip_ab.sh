#!/bin/bash
# A/B of the strip detector's strip height / list size (DET_SH, DET_SL, detector.hip: rows a wave walks, 6 rows of halo per strip; LDS list of maxima) on 128 batched camera
# streams, shipping-flag builds under r-vio_amd/variants/; the fastest variant (if >= 1.5 % over both baseline runs) becomes
# r-vio_amd/librvio_hip.so and the WHOLE GPU suite + smoke() run on exactly that file.
# usage (GPU box, through gpurun): tools/strip_ab.sh <out dir under gpurun_out/>
set -u
OUT=gpurun_out/$1
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
LEAN="--steps 20 --warmup 5 --no-cpu --no-streams --no-latency --batch '' --batch-streams 128"
for V in ${VARIANTS:-sh32_sl256 sh16_sl256 sh24_sl256 sh32_sl128 sh24_sl128 sh16_sl128 sh32_sl256}; do
  N=$OUT/ab_$V.json; [ -e $N ] && N=$OUT/ab_${V}_again.json
  eval RVIO_HIP_LIB=r-vio_amd/variants/librvio_$V.so timeout 100 python bench.py $LEAN > $N 2> /dev/null
done
python - <<EOF > $OUT/chosen.txt
import json, glob, os
r = {}
for f in sorted(glob.glob("$OUT/ab_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        r[os.path.basename(f)[3:-5]] = d["batched_streams"]["sizes"][0]["frames_per_s"]
    except Exception as e:
        r[os.path.basename(f)[3:-5]] = 0.0
base = max(r.get("sh32_sl256", 0.0), r.get("sh32_sl256_again", 0.0))
best = max((k for k in r if not k.startswith("sh32_sl256")), key=lambda k: r.get(k, 0.0))
print("adopt" if base > 0 and r.get(best, 0.0) >= 1.015 * base else "keep", best, json.dumps(r))
EOF
cat $OUT/chosen.txt
if grep -q "^adopt" $OUT/chosen.txt; then
  V=$(awk '{print $2}' $OUT/chosen.txt)
  cp r-vio_amd/variants/librvio_$V.so r-vio_amd/librvio_hip.so
  ( time timeout -k 5 230 python -m pytest tests -m gpu -q --maxfail=8 -p no:cacheprovider ) > $OUT/pytest.log 2>&1
  echo "pytest rc=$?" >> $OUT/pytest.log; tail -6 $OUT/pytest.log
  timeout 60 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?" >> $OUT/smoke.log; tail -2 $OUT/smoke.log
  md5sum r-vio_amd/librvio_hip.so r-vio_amd/variants/*.so > $OUT/md5.txt
fi
ls -la $OUT
