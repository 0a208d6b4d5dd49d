"""greedy_kernel phase clocks per frame (build with -DRVIO_DBG_CLOCKS; GPU box)."""
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
seq = O.rv.synth.SynthSequence(cfg, duration=6.0)
w, a, ni = seq.init_from_static(38)
h = hip.RvioHip(cfg)
h.initialize(w, a, ni)
for k in range(39, 39 + 30):
    h.frame(seq.render(k), seq.imu_between(k), None)
    h.sync()
    c = np.zeros(64, np.int64)
    h.L.rvio_hip_debug_clocks(h.h, c.ctypes.data_as(C.c_void_p))
    print(k, "n", c[62], "rounds", c[63], "cycles: init+pack %d rounds %d rank+out %d total %d" % (
        c[57] - c[56], c[59] - c[57], c[60] - c[59], c[60] - c[56]),
          "| round 1: t0 list %d, decide %d, barrier %d ; round 2 total %d" % (c[11] - c[10], c[12] - c[11], c[13] - c[12], c[14] - c[13]))
