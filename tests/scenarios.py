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


def worst_case_tracks(cfg, rec, seq, n_feat=None, seed=0):
    """Synthesise a full update load (ceil(F/2) features: half type '2' at max length, half type '1'
    with L ~ U[3, n+1]) that is geometrically consistent with the clone poses of `rec`'s state, by
    triangulating nothing: each feature is a random 3-D point projected through the state's own
    relative-pose chain plus sigma_im noise (so the chi-square gate accepts most of them)."""
    rng = np.random.default_rng(1234 + seed)
    x = rec["x1"]
    n = (len(x) - 26) // 7
    Fu = abi.fu(cfg) if n_feat is None else n_feat
    ML = cfg.max_track_len
    T = np.array(list(cfg.T_bc)).reshape(4, 4)
    Ric, tic = T[:3, :3], T[:3, 3]
    Rci, tci = Ric.T, -Ric.T @ tic
    sig = float(max(cfg.sigma_px, cfg.sigma_py))
    types = np.zeros(Fu, np.uint8)
    lens = np.zeros(Fu, np.int32)
    meas = np.zeros((Fu, ML, 2), np.float32)
    for f in range(Fu):
        if f % 2 == 0 and n + 1 == ML:
            ty, L = ord("2"), ML
        else:
            ty, L = ord("1"), int(rng.integers(3, n + 2))
        nph = L - 1
        rel = x[26 + 7 * n - 7 * nph:] if ty == ord("1") else x[26:26 + 7 * nph]
        # camera poses w.r.t. the first camera frame (Updater.cc:125-141)
        qI = [rel[0:4]]
        tI = [-O.quat_to_rot(rel[0:4]) @ rel[4:7]]
        for i in range(1, nph):
            qi = rel[7 * i:7 * i + 4]
            tI.append(O.quat_to_rot(qi) @ (tI[-1] - rel[7 * i + 4:7 * i + 7]))
            qI.append(O.quat_mul(qi, qI[-1]))
        # a point in front of the first camera
        depth = rng.uniform(2.0, 8.0)
        pc1 = np.array([rng.uniform(-0.5, 0.5) * depth, rng.uniform(-0.35, 0.35) * depth, depth])
        obs = [pc1[:2] / pc1[2]]
        for i in range(nph):
            RI = O.quat_to_rot(qI[i])
            Rc = Rci @ RI @ Ric
            tc = Rci @ RI @ tic + Rci @ tI[i] + tci
            pc = Rc @ pc1 + tc
            obs.append(pc[:2] / pc[2])
        obs = np.array(obs) + sig * rng.standard_normal((L, 2))
        types[f], lens[f] = ty, L
        meas[f, :L] = obs.astype(np.float32)
    return types, lens, meas
