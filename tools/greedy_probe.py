import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, '/root/repo')
import bench, torch
rv, abi = bench.rv, bench.abi
from rvio_amd import hip
cfg = abi.config_named("B", enable_equalizer=1)
K = 120
seq, imgs, imu_arr, imu_cnt, cand_arr, cand_cnt = bench.build_inputs(cfg, 1 + K)
h = hip.RvioHip(cfg)
d_imgs = torch.from_numpy(imgs).cuda()
d_imu = torch.from_numpy(imu_arr.view(np.uint8).reshape(1 + K, -1)).cuda()
torch.cuda.synchronize()
h.initialize(*seq.init_from_static(bench.K0))
rows = []
for i in range(1 + K):
    h.frame_dev(d_imgs.data_ptr() + i * cfg.width * cfg.height, cfg.width, d_imu.data_ptr() + i * d_imu.shape[1], int(imu_cnt[i]), 0, 0)
    h.sync()
    out = (C.c_longlong * 64)()
    h.L.rvio_hip_debug_clocks(h.h, out)
    t = list(out)
    rows.append((i, t[62], t[63], (t[57] - t[56]) / 2400.0, (t[59] - t[57]) / 2400.0))
r = np.array(rows, float)
print("n candidates: median %d max %d; rounds median %d max %d" % (np.median(r[:,1]), r[:,1].max(), np.median(r[:,2]), r[:,2].max()))
o = np.argsort(-r[:,4])[:12]
for k in o: print("frame %d n=%d rounds=%d pack %.1f us rounds %.1f us" % tuple(r[k]))
print("median pack %.1f us rounds %.1f us" % (np.median(r[:,3]), np.median(r[:,4])))
