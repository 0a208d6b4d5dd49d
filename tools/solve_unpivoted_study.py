"""CPU study (NumPy, test infrastructure only): may T = s2 I + A Pcc be eliminated WITHOUT a pivot search?

    python tools/solve_unpivoted_study.py

On every recorded update of the synthetic sequences (stock motion, at rest, pure rotation, straight line, one-depth scene, full load at
cfg A / B / C) a Gauss-Jordan inversion of T in natural order — scalar, and blocked with 16 x 16 tiles — is as accurate as LAPACK's pivoted
LU (<= 2e-12 relative, typically 1e-15).  But nothing guarantees it: T is not symmetric, and its symmetric part is INDEFINITE on the same
data (smallest eigenvalue of (T + T^T) / 2 printed in units of s2: -1 .. -69), so a leading principal block s2 I + A[1:k, :] Pcc[:, 1:k]
can be singular while T is perfectly conditioned.  Hence solve9.hip goes through SPD pieces only (tools/solve9_model.py)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle as O  # noqa: E402

abi = O.abi


def gj_unpivoted(T):
    a = T.copy()
    for p in range(a.shape[0]):
        d = a[p, p]
        row, col = a[p, :] / d, a[:, p].copy()
        a -= np.outer(col, row)
        a[:, p] = -col / d
        a[p, :] = row
        a[p, p] = 1 / d
    return a


def run(cfg, n, full=False, **kw):
    seq = O.rv.synth.SynthSequence(cfg, duration=(38 + n + 4) / 20.0, **kw)
    w, a, ni = seq.init_from_static(38)
    x, P = O.initialize(cfg, w, a, ni)
    trk, drv, img = O.Tracker(cfg), O.rv.synth.DirectTrackDriver(seq), 0
    s2 = float(np.float32(max(cfg.sigma_px, cfg.sigma_py))) ** 2
    worst = dict(err=0.0, condT=0.0, min_eig_sym_over_s2=9e9, n=0)
    for k in range(39, 39 + n):
        inp = drv.inputs(k)
        trk.track_points(inp["tracked"], inp["status"], inp["imu"], inp["cand"])
        drv.after(trk.get_points()[0])
        img += 1
        ncl = (len(x) - 26) // 7
        x1, P1 = O.propagate(cfg, x, P, inp["imu"])
        types, lens, meas = trk.get_tracks()
        if full and ncl == cfg.max_track_len - 1:
            types, lens, meas = O.rv.synth.worst_case_tracks(cfg, x1, seed=k)
        x2, P2 = x1, P1
        if ncl > cfg.min_track_len - 1:
            x2, P2, d = O.update(cfg, x1, P1, types, lens, meas)
            if d["updated"]:
                Hw, _, _ = O.update_stack(cfg, x1, P1, types, lens, meas)
                A, Pcc = Hw.T @ Hw, P1[24:, 24:]
                T = s2 * np.eye(len(A)) + A @ Pcc
                W0, W = np.linalg.inv(T), gj_unpivoted(T)
                worst["err"] = max(worst["err"], np.abs(W - W0).max() / np.abs(W0).max())
                worst["condT"] = max(worst["condT"], np.linalg.cond(T))
                worst["min_eig_sym_over_s2"] = min(worst["min_eig_sym_over_s2"], np.linalg.eigvalsh(.5 * (T + T.T)).min() / s2)
                worst["n"] += 1
        x, P, _, _ = O.augment_compose(cfg, x2, P2, img > 1)
    return worst


if __name__ == "__main__":
    for name, cn, n, kw in (("B stock", "B", 120, {}), ("B at rest", "B", 100, dict(motion="stationary")), ("B rotation", "B", 100, dict(motion="rotation")),
                            ("B line", "B", 100, dict(motion="line")), ("B sphere", "B", 100, dict(scene="sphere")), ("B full load", "B", 40, dict(full=True)),
                            ("A full load", "A", 40, dict(full=True)), ("C full load", "C", 45, dict(full=True))):
        print(name, {k: ("%.1e" % v if isinstance(v, float) else v) for k, v in run(abi.config_named(cn), n, **kw).items()})
