"""The multi-core build of the oracle (oracle/liborc_omp.so: the same sources with their OpenMP loops active — bench.py's secondary
CPU baseline) must give exactly the results of the single-thread build: the parallel loops are independent."""
import json
import os
import subprocess
import sys

import numpy as np

import oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_omp_oracle_equals_single_thread(tmp_path):
    abi, rv = O.abi, O.rv
    cfg = abi.config_named("B")                 # stock: CLAHE + detector + KLT
    seq = rv.synth.SynthSequence(cfg, duration=5.0)
    n, k0 = 8, 60
    imgs = np.stack([seq.render(k0 + i) for i in range(n)])
    imus = [seq.imu_between(k0 + i) for i in range(n)]
    m = max(len(u) for u in imus)
    imu_arr = np.zeros((n, m), abi.IMU_DTYPE)
    imu_cnt = np.zeros(n, np.int32)
    for i, u in enumerate(imus):
        imu_arr[i, : len(u)] = u
        imu_cnt[i] = len(u)
    wi, ai, ni = seq.init_from_static(38)
    f = str(tmp_path / "in.npz")
    np.savez(f, config="B", equalizer=1, imgs=imgs, imu=imu_arr.view(np.uint8), imu_cnt=imu_cnt, wi=np.asarray(wi, float), ai=np.asarray(ai, float), ni=int(ni))
    env = dict(os.environ, ORC_LIB="liborc_omp.so", OMP_NUM_THREADS="4")
    out = subprocess.check_output([sys.executable, os.path.join(ROOT, "tools", "cpu_baseline_omp.py"), f], env=env, timeout=300).decode()
    r = json.loads(out.strip().splitlines()[-1])
    x_omp = np.array(r["x"])
    s = O.System(cfg)
    x0, P0 = O.initialize(cfg, wi, ai, ni)
    s.set_state(x0, P0)
    for i in range(n):
        s.frame(imu_arr[i, : imu_cnt[i]], None, img=imgs[i])
    assert np.array_equal(x_omp, s.get_state()[0])
    pts, hl = s.tracker().get_points()          # CLAHE, detector, pyramid and KLT all ran in parallel loops in the child
    assert len(pts) > 100 and np.array_equal(np.array(r["pts"], np.float32), pts) and np.array_equal(np.array(r["hist_len"], np.int32), hl)
