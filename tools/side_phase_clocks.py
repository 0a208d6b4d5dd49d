"""Phase stamps inside the side chain's kernels (klt_kernel3 workgroup 0, ransac_book_a_kernel, bookkeep_b_kernel: DBG_T slots 1..28, rvio_dev.h)
over pipelined frames of the stock configuration (instrumented build):
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DRVIO_DBG_CLOCKS r-vio_amd/csrc/rvio_hip.hip -o r-vio_amd/librvio_dbg.so
    RVIO_HIP_LIB=r-vio_amd/librvio_dbg.so python tools/side_phase_clocks.py [frames] [sync]
Prints, per stamp interval, the median over the frames in microseconds (s_memrealtime: 100 MHz, 10 ns per tick).  `sync`: every frame drained before the next
(kernel phases without the other chains' interference); without it the stamps of the LAST frame in flight are read after each call (racy but
representative)."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import torch  # noqa: E402

rv, abi = bench.rv, bench.abi
from rvio_amd import hip  # noqa: E402

cfg = abi.config_named("B", enable_equalizer=1)
K = int(sys.argv[1]) if len(sys.argv) > 1 else 60
n_frames = 1 + K
seq, imgs, imu_arr, imu_cnt, cand_arr, cand_cnt = bench.build_inputs(cfg, n_frames)
h = hip.RvioHip(cfg)
d_imgs = torch.from_numpy(imgs).cuda()
d_imu = torch.from_numpy(imu_arr.view(np.uint8).reshape(n_frames, -1)).cuda()
torch.cuda.synchronize()
h.initialize(*seq.init_from_static(bench.K0))
rows = []
for i in range(n_frames):
    h.frame_dev(d_imgs.data_ptr() + i * cfg.width * cfg.height, cfg.width, d_imu.data_ptr() + i * d_imu.shape[1], int(imu_cnt[i]), 0, 0)
    h.sync()
    out = (C.c_longlong * 64)()
    h.L.rvio_hip_debug_clocks2(h.h, out)
    rows.append(np.array(list(out), dtype=np.float64) / 100.0)
r = np.array(rows)
groups = {"klt_kernel3 (workgroup 0: feature 0)": list(range(17, 29)), "ransac": [1, 2, 3, 4, 5, 6], "bookkeep_a": [7, 8, 9], "bookkeep_b": [10, 11, 12, 13, 14, 15, 16]}
names = {17: "start", 18: "patches staged", 19: "Scharr", 20: "L3 template", 21: "L3 iterations", 22: "L2 template", 23: "L2 iterations", 24: "L1 template", 25: "L1 iterations",
         26: "L0 template", 27: "L0 iterations", 28: "end", 1: "start", 2: "loads, undistort, compaction", 3: "SetPointPair", 4: "16 models", 5: "CountInliers", 6: "winner + flags",
         7: "start (after the wait)", 8: "lost tracks", 9: "tracked features", 10: "start (after the wait)", 11: "candidates + survivors staged", 12: "cell ids", 13: "cell walk",
         14: "accepted scan", 15: "free slots", 16: "end"}
for lo, hi, tag in ((2, 8, "frames 2-7"), (8, 26, "frames 8-25 (the driver's window)"), (40, K + 1, "frames 40-")):
    if hi > len(r):
        hi = len(r)
    if lo >= hi:
        continue
    print("==", tag)
    for g, idx in groups.items():
        parts = []
        for a, b in zip(idx[:-1], idx[1:]):
            d = r[lo:hi, b] - r[lo:hi, a]
            d = d[(r[lo:hi, b] > 0) & (r[lo:hi, a] > 0) & (d >= 0)]
            if len(d):
                parts.append("%s %.1f (max %.1f)" % (names[b], np.median(d), np.max(d)))
        print("  %-40s %s" % (g, " | ".join(parts)))
h.close()
