"""Phase timing of klt_kernel (feature 0) via clock64() stamps (build with RVIO_HIPCC_FLAGS=-DRVIO_DBG_CLOCKS)."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import oracle as O  # noqa: E402
from rvio_amd import hip  # noqa: E402

abi, rv = O.abi, O.rv
cfg = abi.config_named("B", enable_equalizer=0)
seq = rv.synth.SynthSequence(cfg, duration=6.0)
h = hip.RvioHip(cfg)


def clocks():
    a = np.zeros(64, np.int64)
    h.L.rvio_hip_debug_clocks(h.h, a.ctypes.data_as(C.c_void_p))
    return a


for k in range(60, 66):
    img = seq.render(k)
    xy, vis = seq.project(k, noise=False)
    cand, _ = seq.candidates(k, xy, vis)
    h.track(img, seq.imu_between(k), cand)
    c = clocks()
    if k > 60:
        parts = []
        for lv in (3, 2, 1, 0):
            parts.append("L%d: stage %d tmpl %d iters(%d) %d" % (lv, c[10 + 3 * lv] - (c[9] if lv == 3 else c[12 + 3 * (lv + 1)]),
                                                                  c[11 + 3 * lv] - c[10 + 3 * lv], c[26 + lv], c[12 + 3 * lv] - c[11 + 3 * lv]))
        print(k, " | ".join(parts), "| total", c[12] - c[9])
