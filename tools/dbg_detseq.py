"""Free-running image sequence with the device detector: per-frame state delta GPU vs oracle and discrete counters (GPU box)."""
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
n = int(sys.argv[1]) if len(sys.argv) > 1 else 120
cfg = abi.config_named("B", enable_equalizer=1)
seq = O.rv.synth.SynthSequence(cfg, duration=(38 + n + 4) / 20.0)
w, a, ni = seq.init_from_static(38)
h = hip.RvioHip(cfg)
h.initialize(w, a, ni)
s = O.System(cfg)
x0, P0 = O.initialize(cfg, w, a, ni)
s.set_state(x0, P0)
prev = 0.0
for k in range(39, 39 + n):
    img, imu = seq.render(k), seq.imu_between(k)
    oi = s.frame(imu, None, img=img)[0]
    h.frame(img, imu, None)
    h.sync()
    gi = h.frame_info()
    xa, _ = h.get_state()
    xb, _ = s.get_state()
    dlt = S.state_delta(xa, xb)
    keys = ("n_tracked_in", "n_klt_ok", "n_ransac_inliers", "n_feat_update", "n_feat_accepted", "n_rows", "updated")
    diff = [kk for kk in keys if gi[kk] != oi[kk]]
    if dlt > 3 * prev + 1e-12 or diff or k % 20 == 0:
        _, Pa = h.get_state()
        _, Pb = s.get_state()
        ev = np.linalg.eigvalsh(Pb)
        pose_true = seq.pose(k)
        print(k, "delta %.3e" % dlt, "DIFF" if diff else "", {kk: (gi[kk], oi[kk]) for kk in diff},
              "acc", gi["n_feat_accepted"], "rows", gi["n_rows"], "| P max %.2e eig[min,max] %.2e %.2e dP %.2e | v %s bg %s ba %s" % (
                  np.abs(Pb).max(), ev[0], ev[-1], np.abs(Pa - Pb).max(), np.round(xb[17:20], 3), np.round(xb[20:23], 4), np.round(xb[23:26], 3)))
    prev = max(prev, dlt)
print("final delta %.3e" % dlt)
