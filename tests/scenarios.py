"""Per-stage golden scenarios recorded from the CPU oracle on the synthetic
direct-track sequence (test infrastructure).  Each record holds the inputs and
oracle outputs of every stage of one frame, so the HIP path can be checked
stage by stage on identical inputs (SURVEY.md 8d "parity procedure")."""
import numpy as np

import oracle as O

abi, rv = O.abi, O.rv


def record_sequence(cfg, n_frames=40, seed=0, k0=38, duration=8.0, want=None):
    """Run the oracle stage by stage over frames k0+1 .. k0+n_frames.
    Returns (seq, records)."""
    seq = rv.synth.SynthSequence(cfg, duration=duration, seed=seed)
    w, a, n = seq.init_from_static(k0)
    x, P = O.initialize(cfg, w, a, n)
    trk = O.Tracker(cfg)
    drv = rv.synth.DirectTrackDriver(seq)
    recs = []
    img_count = 0
    for k in range(k0 + 1, k0 + 1 + n_frames):
        inp = drv.inputs(k)
        info = trk.track_points(inp["tracked"], inp["status"], inp["imu"], inp["cand"])
        pts, hl = trk.get_points()
        drv.after(pts)
        img_count += 1
        rec = dict(k=k, inp=inp, info=info, x0=x.copy(), P0=P.copy(), pts=pts, hist_len=hl)
        ncl = (len(x) - 26) // 7
        x1, P1 = O.propagate(cfg, x, P, inp["imu"])
        rec.update(x1=x1, P1=P1)
        types, lens, meas = trk.get_tracks()
        rec.update(types=types, lens=lens, meas=meas)
        if ncl > cfg.min_track_len - 1:
            x2, P2, diag = O.update(cfg, x1, P1, types, lens, meas)
            rec.update(x2=x2, P2=P2, diag=diag, did_update=True)
        else:
            x2, P2 = x1, P1
            rec.update(x2=x2, P2=P2, diag=None, did_update=False)
        x3, P3, pp, pq = O.augment_compose(cfg, x2, P2, img_count > 1)
        rec.update(x3=x3, P3=P3, pose_p=pp, pose_q=pq, do_augment=img_count > 1)
        x, P = x3, P3
        recs.append(rec)
    return seq, recs


def small_image_config():
    """a half-size camera (376 x 240, 100 features): keeps the image fixture of tests/golden/ small; every front-end stage still runs"""
    return abi.config_named("B", width=376, height=240, fx=229.327, fy=228.648, cx=183.6075, cy=124.1875, n_features=100,
                            block_x=75, block_y=60)


def qfix(x):
    """sign-normalise the quaternions of a state vector (q and -q are the same rotation)"""
    x = np.array(x, float)
    idx = [0, 10] + list(range(26, len(x), 7))
    for i in idx:
        if x[i + 3] < 0:
            x[i:i + 4] *= -1
    return x


def state_delta(xa, xb):
    return float(np.max(np.abs(qfix(xa) - qfix(xb))))


def worst_case_tracks(cfg, rec, seq, n_feat=None, seed=0, mix="half"):
    """a full update load (ceil(F/2) features) that is geometrically consistent with the clone poses of `rec`'s propagated state:
    rv.synth.worst_case_tracks (the same generator bench.py's update-at-load leg uses)"""
    return rv.synth.worst_case_tracks(cfg, rec["x1"], n_feat=n_feat, seed=seed, mix=mix)
