"""hipGraph probe (measure, don't argue): the filter tail of one frame — per-feature build, share reduction, solve, U/G/P1, Joseph
form, augmentation + composition: 6 kernels on the handle's filter stream — captured once with hipStreamBeginCapture and replayed
with hipGraphLaunch, against the same six launches issued one by one.  Prints host microseconds per tail (enqueue only) and the
device time per tail (stream-synchronised wall clock over `n` repetitions)."""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pkgload import load_pkg  # noqa: E402

rv = load_pkg()
import torch  # noqa: E402,F401  (HIP runtime load order)
from rvio_amd import hip  # noqa: E402

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
h.set_state(x1, P1)
h.update(types, lens, meas)
h.augment_compose(True)
h.sync()

rt = C.CDLL("libamdhip64.so")
stream = C.c_void_p(h.stream())
n = 300


def tail():
    h.update_tracked()
    h.augment_compose(True)


def timed(fn):
    h.sync()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    t1 = time.perf_counter()
    h.sync()
    t2 = time.perf_counter()
    return 1e6 * (t1 - t0) / n, 1e6 * (t2 - t0) / n


for _ in range(20):
    tail()
host_a, wall_a = timed(tail)
graph, gexec = C.c_void_p(), C.c_void_p()
assert rt.hipStreamBeginCapture(stream, 0) == 0        # hipStreamCaptureModeGlobal
tail()
assert rt.hipStreamEndCapture(stream, C.byref(graph)) == 0
rc = rt.hipGraphInstantiate(C.byref(gexec), graph, None, None, 0)
assert rc == 0, rc


def replay():
    assert rt.hipGraphLaunch(gexec, stream) == 0


for _ in range(20):
    replay()
host_b, wall_b = timed(replay)
print({"launches_per_tail": 6, "stream_launches": {"host_us": host_a, "wall_us": wall_a}, "graph_replay": {"host_us": host_b, "wall_us": wall_b}})
h.close()
