"""Phase stamps + live time of the solve kernel on a full-load update (instrumented build): RVIO_HIP_LIB=r-vio_amd/librvio_dbg.so python tools/solve_probe.py"""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pkgload import load_pkg
rv = load_pkg()
from rvio_amd import hip
abi = rv.abi
cfg = abi.config_named("B", enable_equalizer=0)
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
types, lens, meas = rv.synth.worst_case_tracks(cfg, x1, mix="half")
for _ in range(3):
    h.set_state(x1, P1)
    h.update(types, lens, meas)
h.sync()
out = (C.c_longlong * 64)()
h.L.rvio_hip_debug_clocks(h.h, out)
t = np.array(list(out))
idx = [i for i in (56, 57, 58, 60, 63, 61, 62) if t[i] != 0]
print("solve phases (cycles):", " ".join("%d->%d:%d" % (a, b, t[b] - t[a]) for a, b in zip(idx[:-1], idx[1:])), "total", t[idx[-1]] - t[idx[0]])
print("wall clock inside the kernel (us): start -> thread 0 done %.2f, -> last thread done %.2f" % ((t[46] - t[45]) / 100.0, (t[47] - t[45]) / 100.0))
h.set_state(x1, P1); h.update_tracked(); h.sync()
print("solve kernel live avg us:", h.time_kernel(0, 50))
idx = [i for i in (20, 21, 22, 23, 24, 25) if t[i] != 0]
print("joseph_lds phases (cycles): loads %d  U %d  G %d  P1c %d  X %d" % tuple(t[b] - t[a] for a, b in zip(idx[:-1], idx[1:])) if len(idx) == 6 else "joseph_lds: not launched")
print("ug %.2f us, final %.2f us (two-launch forms); as launched %.2f us" % (h.time_kernel(4, 50), h.time_kernel(5, 50), h.time_kernel(7, 50)))
h.propagate(seq.imu_between(39 + cfg.max_track_len + 9)); h.sync()
h.L.rvio_hip_debug_clocks(h.h, out); t = np.array(list(out))
idx = [i for i in range(10, 18) if t[i] != 0]
print("propagate phases (cycles): " + " ".join("%d->%d:%d" % (a, b, t[b] - t[a]) for a, b in zip(idx[:-1], idx[1:])), "(10 start, 11 loads, 12 A trig, 13 B state chain, 14 C Phi rows, 15 suffix products, 16 apply + Q, 17 P12 + stores)")
h.close()
