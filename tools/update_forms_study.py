"""CPU study (numpy, test infrastructure only): how far do the two formulations of U7-U10 sit from the LITERAL reference form, per update
and free-running, on the motions where the window is badly conditioned (a platform at rest above all)?

  literal   oracle/filter.cpp compress_and_apply = Updater.cc:469-619: sequential Givens QR, leading-row rank scan, S = Hn P Hn^T + s2 I,
            K = P Hn^T S^-1 (explicit inverse), Joseph form
  info      what the device computes: [A|b] = Hw^T [Hw | r], T = s2 I + A Pcc, W = T^-1 (partial-pivot LU), dx = Pc W b, G = Pc W A,
            U = Pc W, P+ = sym(P1 - P1c G^T + s2 G U^T), P1 = P - G Pc^T
  qr        measurement space without the Givens order: Householder QR of [Hw | r] (LAPACK), R's rows with norm >= 1e-4 scanned from the top,
            S = R Pcc R^T + s2 I by Cholesky, K = Pc R^T S^-1, Joseph form

    python tools/update_forms_study.py [--motion stationary|rotation|line|stock] [--frames 100]

Printed: per-update distance of info / qr from literal started from the SAME (literal) state, and the free-running distance after N frames —
next to the literal form's distance from ITSELF when its state is perturbed by one ulp before the first update (the sequence's own sensitivity)."""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle as O       # noqa: E402
import scenarios as S    # noqa: E402

abi, rv = O.abi, O.rv


def inject(x, dx):
    """Updater.cc:546-613"""
    x = x.copy()
    n = (len(x) - 26) // 7

    def dq(th):
        h = .5 * th
        nn = float(h @ h)
        if nn < 1:
            return np.array([h[0], h[1], h[2], np.sqrt(1 - nn)])
        q = np.array([h[0], h[1], h[2], 1.0])
        return q / np.sqrt(1 + nn)
    x[0:4] = O.quat_mul(dq(dx[0:3]), x[0:4])
    x[4:7] += dx[3:6]
    x[7:10] += dx[6:9]
    x[7:10] /= np.linalg.norm(x[7:10])
    x[10:14] = O.quat_mul(dq(dx[9:12]), x[10:14])
    x[14:26] += dx[12:24]
    for i in range(n):
        x[26 + 7 * i:30 + 7 * i] = O.quat_mul(dq(dx[24 + 6 * i:27 + 6 * i]), x[26 + 7 * i:30 + 7 * i])
        x[30 + 7 * i:33 + 7 * i] += dx[27 + 6 * i:30 + 6 * i]
    return x


def upd_info(cfg, x, P, Hw, r):
    s2 = float(np.float32(max(cfg.sigma_px, cfg.sigma_py))) ** 2
    c6 = Hw.shape[1]
    A, b = Hw.T @ Hw, Hw.T @ r
    Pc, Pcc = P[:, 24:], P[24:, 24:]
    T = s2 * np.eye(c6) + A @ Pcc
    W = np.linalg.inv(T)
    dx = Pc @ (W @ b)
    U = Pc @ W
    G = U @ A
    P1 = P - G @ Pc.T
    Pn = P1 - P1[:, 24:] @ G.T + s2 * G @ U.T
    return inject(x, dx), .5 * (Pn + Pn.T)


def upd_qr(cfg, x, P, Hw, r):
    s2 = float(np.float32(max(cfg.sigma_px, cfg.sigma_py))) ** 2
    M, c6 = Hw.shape
    if M > c6:
        Rz = np.linalg.qr(np.hstack([Hw, r[:, None]]), mode="r")[:c6]
        R, z = Rz[:, :c6], Rz[:, c6]
        keep = 0
        while keep < c6 and np.linalg.norm(R[keep]) >= 1e-4:     # Updater.cc:516-523 on the Householder R
            keep += 1
        R, z = R[:keep], z[:keep]
    else:
        R, z = Hw, r
    d = P.shape[0]
    H = np.zeros((len(z), d))
    H[:, 24:] = R
    Sm = H @ P @ H.T + s2 * np.eye(len(z))
    Sm = .5 * (Sm + Sm.T)
    L = np.linalg.cholesky(Sm)
    Y = np.linalg.solve(L, H @ P)                     # L^-1 H P
    K = np.linalg.solve(L.T, Y).T                     # P H^T S^-1
    dx = K @ z
    IKH = np.eye(d) - K @ H
    Pn = IKH @ P @ IKH.T + s2 * K @ K.T
    return inject(x, dx), .5 * (Pn + Pn.T)


def run(cfg, seq, n_frames, form, perturb=False):
    """free-running filter on direct tracks with the given update form; returns the states after every frame"""
    w, a, ni = seq.init_from_static(38)
    x, P = O.initialize(cfg, w, a, ni)
    trk = O.Tracker(cfg)
    drv = rv.synth.DirectTrackDriver(seq)
    out, per = [], []
    img_count = 0
    for k in range(39, 39 + n_frames):
        inp = drv.inputs(k)
        trk.track_points(inp["tracked"], inp["status"], inp["imu"], inp["cand"])
        drv.after(trk.get_points()[0])
        img_count += 1
        ncl = (len(x) - 26) // 7
        x, P = O.propagate(cfg, x, P, inp["imu"])
        types, lens, meas = trk.get_tracks()
        if ncl > cfg.min_track_len - 1 and len(lens):
            Hw, r, ng = O.update_stack(cfg, x, P, types, lens, meas)
            if ng > 2:
                if perturb and not per:
                    x = x.copy()
                    x[14] = np.nextafter(x[14], 1.0)
                xl, Pl, _ = O.update_from_stack(cfg, x, P, Hw, r, ng)
                if form == "literal":
                    xn, Pn = xl, Pl
                else:
                    xn, Pn = (upd_info if form == "info" else upd_qr)(cfg, x, P, Hw, r)
                per.append((S.state_delta(upd_info(cfg, x, P, Hw, r)[0], xl), S.state_delta(upd_qr(cfg, x, P, Hw, r)[0], xl),
                            np.linalg.cond(float(np.float32(max(cfg.sigma_px, cfg.sigma_py))) ** 2 * np.eye(Hw.shape[1]) + Hw.T @ Hw @ P[24:, 24:])) if form == "literal" else (0, 0, 0))
                x, P = xn, Pn
        x, P, _, _ = O.augment_compose(cfg, x, P, img_count > 1)
        out.append(x.copy())
    return out, np.array(per)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--motion", default="stationary")
    ap.add_argument("--frames", type=int, default=100)
    ap.add_argument("--config", default="B")
    args = ap.parse_args()
    cfg = abi.config_named(args.config, enable_equalizer=0)
    kw = {} if args.motion == "stock" else dict(motion=args.motion)
    seq = rv.synth.SynthSequence(cfg, duration=(38 + args.frames + 4) / 20.0, seed=2, **kw)
    lit, per = run(cfg, seq, args.frames, "literal")
    print("%s, %d frames, %d updates; cond(T) median %.2e max %.2e" % (args.motion, args.frames, len(per), np.median(per[:, 2]), per[:, 2].max()))
    print("per update, from the literal state:  info vs literal  median %.2e  max %.2e   |   qr vs literal  median %.2e  max %.2e"
          % (np.median(per[:, 0]), per[:, 0].max(), np.median(per[:, 1]), per[:, 1].max()))
    for form in ("info", "qr"):
        got, _ = run(cfg, seq, args.frames, form)
        d = [S.state_delta(a, b) for a, b in zip(got, lit)]
        print("free-running %-5s vs literal: after %d frames %.2e, max over the sequence %.2e" % (form, args.frames, d[-1], max(d)))
    got, _ = run(cfg, seq, args.frames, "literal", perturb=True)
    d = [S.state_delta(a, b) for a, b in zip(got, lit)]
    print("free-running literal, one state perturbed by ONE ULP before the first update: after %d frames %.2e, max %.2e" % (args.frames, d[-1], max(d)))


if __name__ == "__main__":
    main()
