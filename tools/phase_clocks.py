"""Phase stamps inside the filter kernels (DBG_T(i) records clock64() of workgroup 0, thread 0; rvio_dev.h).
Build an instrumented copy of the library and run one full-load update on the GPU box:
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DRVIO_DBG_CLOCKS r-vio_amd/csrc/rvio_hip.hip -o r-vio_amd/librvio_dbg.so
    RVIO_HIP_LIB=r-vio_amd/librvio_dbg.so python tools/phase_clocks.py
Prints the differences between consecutive stamps of each kernel in shader cycles (100 MHz clock64 ticks x 24 at 2.4 GHz are NOT assumed:
the raw counter differences are printed; s_memtime counts at 100 MHz on gfx950, i.e. 10 ns per tick)."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pkgload import load_pkg  # noqa: E402

rv = load_pkg()
from rvio_amd import hip  # noqa: E402

abi = rv.abi
cfg = abi.config_named(sys.argv[1] if len(sys.argv) > 1 else "B", enable_equalizer=0)
seq = rv.synth.SynthSequence(cfg, duration=5.0)
h = hip.RvioHip(cfg)
h.initialize(*seq.init_from_static(38))
drv = rv.synth.DirectTrackDriver(seq)
for f in range(cfg.max_track_len + 8):
    inp = drv.inputs(39 + f)
    h.frame_points(inp["tracked"], inp["status"], inp["imu"], inp["cand"])
    drv.after(h.get_points()[0])
h.propagate(seq.imu_between(39 + cfg.max_track_len + 8))
x1, P1 = h.get_state()
types, lens, meas = rv.synth.worst_case_tracks(cfg, x1, mix="long")
for _ in range(3):
    h.set_state(x1, P1)
    h.update(types, lens, meas)
h.sync()
out = (C.c_longlong * 64)()
h.L.rvio_hip_debug_clocks(h.h, out)
t = np.array(list(out))
groups = {"feat_build": range(30, 41), "gram_reduce": [41, 42, 43, 44], "solve": [56, 57, 58, 59, 50, 51, 52, 53, 54, 55, 63, 49, 60, 61, 62]}
for name, idx in groups.items():
    idx = [i for i in idx if t[i] != 0]
    print(name, " ".join("%d:+%d" % (b, t[b] - t[a]) for a, b in zip(idx[:-1], idx[1:])))
h.close()
