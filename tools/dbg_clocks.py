"""Phase timing of single-workgroup kernels via clock64() stamps (build with RVIO_HIPCC_FLAGS=-DRVIO_DBG_CLOCKS)."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import oracle as O  # noqa: E402
import scenarios as S  # noqa: E402
from rvio_amd import hip  # noqa: E402

abi = O.abi
cfg = abi.config_named("B", enable_equalizer=0)
seq, recs = S.record_sequence(cfg, n_frames=20)
r = recs[-1]
h = hip.RvioHip(cfg)
h.set_state(r["x1"], r["P1"])


def clocks():
    a = np.zeros(64, np.int64)
    h.L.rvio_hip_debug_clocks(h.h, a.ctypes.data_as(C.c_void_p))
    return a


ty, le, me = S.worst_case_tracks(cfg, r, seq)
for rep in range(2):
    h.set_state(r["x1"], r["P1"])
    h.update(ty, le, me)
    c = clocks()
    print("feat_build f0 type=%s L=%d:" % (chr(ty[0]), le[0]), " ".join("%d" % (c[i + 1] - c[i]) for i in range(30, 40)), "total", c[40] - c[30])
    h.set_state(r["x1"], r["P1"])
    h.update(r["types"], r["lens"], r["meas"])
    c = clocks()
    print("feat_build f0 type=%s L=%d:" % (chr(r["types"][0]), r["lens"][0]), " ".join("%d" % (c[i + 1] - c[i]) for i in range(30, 40)), "total", c[40] - c[30])
    print("solve: load %d step0 %d step1 %d steps2-31 %d (avg %d) steps32-59 %d (avg %d) readout %d dx %d inject-start %d" % (
        c[41] - c[40], c[42] - c[41], c[43] - c[42], c[44] - c[43], (c[44] - c[43]) // 30, c[45] - c[44], (c[45] - c[44]) // 28, c[46] - c[45], c[47] - c[46], c[47] - c[40]))
    print("solve step30: combine %d  ipiv+prv %d  eliminate %d  publish+barrier %d | whole step31 %d" % (c[51]-c[50], c[52]-c[51], c[53]-c[52], c[54]-c[53], c[55]-c[54]))
