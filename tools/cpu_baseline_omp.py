#!/usr/bin/env python3
"""Child process of bench.py's cpu_baseline leg: the same frames through the MULTI-CORE build of the oracle (oracle/liborc_omp.so —
the same sources with their OpenMP loops active).  Reads an .npz written by bench.py, prints one JSON line.
Test infrastructure / CPU baseline only (see tests/oracle.py)."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("ORC_LIB", "liborc_omp.so")
import oracle as O  # noqa: E402

d = np.load(sys.argv[1], allow_pickle=False)
cfg = O.abi.config_named(str(d["config"]), enable_equalizer=int(d["equalizer"]))
imgs, imu_arr, imu_cnt = d["imgs"], d["imu"].view(O.abi.IMU_DTYPE), d["imu_cnt"]
imu_arr = imu_arr.reshape(len(imgs), -1)
s = O.System(cfg)
x0, P0 = O.initialize(cfg, d["wi"], d["ai"], int(d["ni"]))
s.set_state(x0, P0)
warm = int(d["warm"]) if "warm" in d.files else 0
per = []
for i in range(len(imgs)):
    t0 = time.perf_counter()
    s.frame(imu_arr[i, : imu_cnt[i]], None, img=imgs[i])
    per.append(time.perf_counter() - t0)
per = np.array(per)[warm:]
pts, hl = s.tracker().get_points()
print(json.dumps({"value": len(per) / float(per.sum()), "frame_ms_p50": float(1e3 * np.median(per)), "frame_ms_p95": float(1e3 * np.percentile(per, 95)),
                  "x": s.get_state()[0].tolist(), "pts": pts.tolist(), "hist_len": hl.tolist()}))
