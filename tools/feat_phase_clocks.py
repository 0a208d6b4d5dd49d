"""Phases of the per-feature stage (feat_build_body inside feat_prop_kernel) over ALL workgroups of a launch, in microseconds on the constant
100 MHz clock (DBG_P, rvio_dev.h; instrumented build): per phase the mean over the features of a frame and the LONGEST — the launch ends with
its slowest workgroup —, medians over the frames of the stock sequence.
    RVIO_HIP_LIB=r-vio_amd/librvio_dbg.so python tools/feat_phase_clocks.py [frames]"""
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
K = int(sys.argv[1]) if len(sys.argv) > 1 else 120
n_frames = 1 + K
seq, imgs, imu_arr, imu_cnt, cand_arr, cand_cnt = bench.build_inputs(cfg, n_frames)
h = hip.RvioHip(cfg)
d_imgs = torch.from_numpy(imgs).cuda()
d_imu = torch.from_numpy(imu_arr.view(np.uint8).reshape(n_frames, -1)).cuda()
torch.cuda.synchronize()
h.initialize(*seq.init_from_static(bench.K0))
names = ["loads", "barrier", "U1 pose chain", "U2 LM", "U3 Jacobians", "blocks + reflectors", "apply reflectors", "gate H Pcc", "S", "LDLt", "shares"]
means, maxs, counts, gram = [], [], [], []
out, mx = (C.c_longlong * 64)(), (C.c_longlong * 64)()
h.L.rvio_hip_debug_phases(h.h, out, mx)
for i in range(n_frames):
    h.frame_dev(d_imgs.data_ptr() + i * cfg.width * cfg.height, cfg.width, d_imu.data_ptr() + i * d_imu.shape[1], int(imu_cnt[i]), 0, 0)
    h.sync()
    h.L.rvio_hip_debug_phases(h.h, out, mx)
    s = np.array(list(out), dtype=np.float64)
    m = np.array(list(mx), dtype=np.float64)
    cnt = s[32:43]
    if i >= 60 and cnt[0] > 0:
        means.append(s[0:11] / np.maximum(cnt, 1) / 100.0)
        maxs.append(m[30:41] / 100.0)
        counts.append(cnt)
        gram.append(np.diff(m[45:50]) / 100.0)      # gram_reduce_kernel workgroup 0 (DBG_U 45..49): list, counters, decision, sums
means, maxs, counts = np.array(means), np.array(maxs), np.array(counts)
gram = np.array(gram)
print("frames with an update: %d; features per update (workgroups past the header): median %d" % (len(means), int(np.median(counts[:, 0]))))
for k, nm in enumerate(names):
    print("  %-22s mean over features %6.2f us | longest %6.2f us | reached by %3d workgroups" % (nm, np.median(means[:, k]), np.median(maxs[:, k]), int(np.median(counts[:, k]))))
print("  sum of the longest phases %.1f us; sum of the means %.1f us" % (np.sum(np.median(maxs, axis=0)), np.sum(np.median(means, axis=0))))
print("gram_reduce_kernel workgroup 0 (us, medians): accepted-feature list %.2f | counters %.2f | literal / candidate decision %.2f | sums %.2f" % tuple(np.median(gram, axis=0)))
h.close()
