"""The first frames of the pipelined single-stream run, chain by chain (instrumented build): where the driver's 20-step window spends its time.\nusage: RVIO_HIP_LIB=r-vio_amd/librvio_dbg.so python tools/fill_clocks.py [frames]"""
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
K = int(sys.argv[1]) if len(sys.argv) > 1 else 25
n_frames = 1 + K
seq, imgs, imu_arr, imu_cnt, cand_arr, cand_cnt = bench.build_inputs(cfg, n_frames)
h = hip.RvioHip(cfg)
d_imgs = torch.from_numpy(imgs).cuda()
d_imu = torch.from_numpy(imu_arr.view(np.uint8).reshape(n_frames, -1)).cuda()
torch.cuda.synchronize()
h.initialize(*seq.init_from_static(bench.K0))
for i in range(n_frames):
    h.frame_dev(d_imgs.data_ptr() + i * cfg.width * cfg.height, cfg.width, d_imu.data_ptr() + i * d_imu.shape[1], int(imu_cnt[i]), 0, 0)
h.sync()
out = (C.c_longlong * 512)()
fr = C.c_int(0)
h.L.rvio_hip_debug_ring(h.h, out, C.byref(fr))
t = np.array(list(out), dtype=np.int64).reshape(64, 8)
last = fr.value
out2 = (C.c_longlong * 512)()
fr2 = C.c_int(0)
h.L.rvio_hip_debug_ring2(h.h, out2, C.byref(fr2))
t2 = np.array(list(out2), dtype=np.int64).reshape(64, 8)
out3 = (C.c_longlong * 512)()
h.L.rvio_hip_debug_ring3(h.h, out3)
t3 = np.array(list(out3), dtype=np.int64).reshape(64, 8) / 100.0
side_of = lambda f: t2[(fr2.value - (n_frames - 1 - f)) & 63] / 100.0
filt_of = lambda f: t[(last - (n_frames - 1 - f)) & 63] / 100.0
print("filter frames stamped %d, side frames stamped %d, frames run %d" % (last, fr2.value, n_frames))
t0 = side_of(1)[0]
print("frame | image: clahe  subpix_end | side: klt  ransac  book  after_wait  refill  end | filter: gate  gate_end  feat_prop  augcomp_end | feat_prop period")
prev = None
for f in range(1, n_frames):
    im, sd = t3[f & 63], side_of(f)
    fl = filt_of(f) if last - (n_frames - 1 - f) >= 1 else np.zeros(8)
    r = lambda v: "%7.0f" % (v - t0) if v > 0 else "      -"
    per = (fl[0] - prev) if (prev and fl[0] > 0) else float("nan")
    print("%5d | %s %s | %s %s %s %s %s %s | %s %s %s %s | %6.1f" % (f, r(im[0]), r(im[3]), r(sd[0]), r(sd[2]), r(sd[3]), r(sd[4]), r(sd[6]), r(sd[5]), r(im[4]), r(im[5]), r(fl[0]), r(fl[6]), per))
    prev = fl[0] if fl[0] > 0 else prev
h.close()
