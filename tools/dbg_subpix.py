"""subpix_kernel phase clocks of corner 0, iteration 3 (build with -DRVIO_DBG_CLOCKS; GPU box)."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import oracle as O  # noqa: E402
from rvio_amd import hip  # noqa: E402

cfg = O.abi.config_named("B", enable_equalizer=1)
seq = O.rv.synth.SynthSequence(cfg, duration=4.0)
h = hip.RvioHip(cfg)
imu = seq.imu_between(45)
for k in (45, 46, 47):
    h.track(seq.render(k), imu, None)
    h.sync()
    c = np.zeros(64, np.int64)
    h.L.rvio_hip_debug_clocks(h.h, c.ctypes.data_as(C.c_void_p))
    print(k, "iters(corner 0)", c[48], "cycles: patch+sync %d  term %d  dpp+readlane+write %d  sync %d  sums+solve %d  sync %d  | iteration %d" % (
        c[42] - c[41], c[43] - c[42], c[44] - c[43], c[45] - c[44], c[46] - c[45], c[47] - c[46], c[47] - c[41]))
seen = O.clahe(seq.render(47))
raw = O.gftt(seen, cfg.n_features, float(np.float32(cfg.qual_lvl)), 30.0)
