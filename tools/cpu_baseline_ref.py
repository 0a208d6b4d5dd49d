#!/usr/bin/env python3
"""Child process of bench.py's cpu_baseline_reference leg: the same frames through System::MonoVIO of the reference's OWN sources
(oracle/_ref/libref.so = /root/reference/src/rvio/*.cc compiled unmodified against oracle/refshim/).  A child, because the reference's code
keeps its counters in function-local statics, never frees its packets and can spin forever in Ransac::SetPointPair with 17..31 candidates
(SURVEY.md D.1): bench.py gives it a time-out.  Reads an .npz written by bench.py, prints one JSON line.
Test infrastructure / CPU baseline only (see tests/ref.py)."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import ref as R  # noqa: E402

O = R.O
if not R.available():
    print(json.dumps({"skipped": "oracle/_ref/libref.so is not here (it is built by `make -C oracle ref` where the reference's sources exist)"}))
    sys.exit(0)
d = np.load(sys.argv[1], allow_pickle=False)
cfg = O.abi.config_named(str(d["config"]), enable_equalizer=int(d["equalizer"]))
imgs, imu_arr, imu_cnt = d["imgs"], d["imu"].view(O.abi.IMU_DTYPE), d["imu_cnt"]
imu_arr = imu_arr.reshape(len(imgs), -1)
sr = R.System(cfg)
sr.set_state(*R.initialize(cfg, d["wi"], d["ai"], int(d["ni"])))
warm = int(d["warm"]) if "warm" in d.files else 0
per = []
for i in range(len(imgs)):
    t0 = time.perf_counter()
    sr.frame(imu_arr[i, : imu_cnt[i]], None, img=imgs[i])
    per.append(time.perf_counter() - t0)
per = np.array(per)[warm:]
print(json.dumps({"value": len(per) / float(per.sum()), "frame_ms_p50": float(1e3 * np.median(per)), "frame_ms_p95": float(1e3 * np.percentile(per, 95)),
                  "frames_timed": int(len(per)), "x": sr.get_state()[0].tolist()}))
