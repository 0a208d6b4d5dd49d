"""CPU experiment: image sequence with the oracle's own detector; per frame, the literal update (Givens QR + rank truncation,
Updater.cc:469-529) against the information-form mirror (orc_update_local/global).  Prints the frames where they differ."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import oracle as O  # noqa: E402
import scenarios as S  # noqa: E402

abi = O.abi
n = int(sys.argv[1]) if len(sys.argv) > 1 else 110
cfg = abi.config_named("B", enable_equalizer=1)
seq = O.rv.synth.SynthSequence(cfg, duration=(38 + n + 4) / 20.0)
w, a, ni = seq.init_from_static(38)
x, P = O.initialize(cfg, w, a, ni)
trk = O.Tracker(cfg)
img_count = 0
worst = 0.0
for k in range(39, 39 + n):
    imu = seq.imu_between(k)
    trk.track(seq.render(k), imu, None)
    img_count += 1
    ncl = (len(x) - 26) // 7
    x1, P1 = O.propagate(cfg, x, P, imu)
    types, lens, meas = trk.get_tracks()
    if ncl > cfg.min_track_len - 1:
        x2, P2, d = O.update(cfg, x1, P1, types, lens, meas)
        blk = O.update_local(cfg, x1, P1, types, lens, meas, 0, 1)
        xi, Pi, di = O.update_global(cfg, x1, P1, blk[None, :])
        dl = S.state_delta(x2, xi)
        dP = np.abs(P2 - Pi).max() / max(np.abs(P2).max(), 1e-300)
        worst = max(worst, dl)
        if dl > 1e-10 or d["rank"] not in (-1, 6 * ncl):
            print("frame %d n %d rows %d literal rank %d  | state delta %.2e  rel dP %.2e" % (k, ncl, d["n_rows"], d["rank"], dl, dP))
    else:
        x2, P2 = x1, P1
    x, P, _, _ = O.augment_compose(cfg, x2, P2, img_count > 1)
print("worst single-update state delta %.3e" % worst)
