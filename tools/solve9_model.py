"""NumPy model of solve9 (csrc/solve9.hip): W = (s2 I + A Pcc)^-1 through SPD pieces only (test infrastructure / design study).

    Pcc = L L^T                 blocked Cholesky, 16 x 16 tiles, positive-SEMI-definite safe (a zero-variance direction gives a zero
                                column of L, as an IMU stream that ends leaves behind: clones with exactly zero covariance)
    M   = s2 I + L^T A L        symmetric, eigenvalues >= s2: no pivoting needed, ever
    Mi  = M^-1                  blocked symmetric sweep (Gauss-Jordan on diagonal tiles in order)
    W   = (I - A L Mi L^T) / s2 (Woodbury; no inverse of L or Pcc anywhere)

Every 16 x 16 diagonal tile goes through ONE in-wave primitive: forward elimination of [Mkk | I] without pivoting -> F = Lkk^-1
(rows of skipped pivots zero).  Cholesky uses F (new row panel = F * old row panel), the sweep uses D = F^T F = Mkk^-1.
The model mirrors the kernel's tile algebra (what is multiplied with what, in which order); `python tools/solve9_model.py` checks it
against numpy.linalg.inv on recorded updates of the synthetic sequences."""
import os
import sys

import numpy as np

TB = 16


def tile_inv_factor(Mkk, ref_diag, tol=1e-12):
    """F = Lkk^-1 by forward elimination of [Mkk | I]; pivots <= tol * ref_diag are treated as zero (row of F = 0)"""
    a = Mkk.copy()
    e = np.eye(TB)
    s = np.zeros(TB)
    for p in range(TB):
        d = a[p, p]
        if d <= tol * ref_diag[p] or d <= 0.0:
            continue                      # zero direction: no elimination, F row p = 0
        m = a[p + 1:, p] / d              # = a[p, p+1:] / d by symmetry (the kernel reads row p only)
        a[p + 1:, :] -= np.outer(m, a[p, :])
        e[p + 1:, :] -= np.outer(m, e[p, :])
        s[p] = 1.0 / np.sqrt(d)
    return s[:, None] * e


def blocked_cholesky(P, ref_diag):
    """returns G = L^T (upper) with P = G^T G; row-panel form: G(k, j) = F_k * S(k, j), S(i, j) -= G(k, i)^T G(k, j)"""
    n = P.shape[0]
    nt = n // TB
    S = P.copy()
    G = np.zeros_like(P)
    for k in range(nt):
        K = slice(k * TB, (k + 1) * TB)
        F = tile_inv_factor(S[K, K], ref_diag[K])
        row = F @ S[K, k * TB:]                    # new row panel, tiles j >= k
        G[K, k * TB:] = row
        G[K, K] = np.triu(G[K, K])                 # exact zeros below the diagonal (rounding residue otherwise)
        S[(k + 1) * TB:, (k + 1) * TB:] -= row[:, TB:].T @ row[:, TB:]
    return G


def blocked_sweep_inverse(M):
    """symmetric sweep in its Cholesky form: per step Z = F old, S -= Z^T Z, new = F^T Z, S(k, k) = -F^T F.  (Forming D = F^T F
    first and multiplying with it loses the backward stability of the tile's factorisation: at cond(M) = 2e8 the residual of that
    variant is 1e-2 where this one sits at LAPACK's 2e-7.)"""
    n = M.shape[0]
    nt = n // TB
    S = M.copy()
    for k in range(nt):
        K = slice(k * TB, (k + 1) * TB)
        F = tile_inv_factor(S[K, K], np.diag(S[K, K]).copy(), tol=0.0)
        old = S[K, :].copy()                       # old row panel (= old column panel transposed)
        Z = F @ old
        new = F.T @ Z
        S -= Z.T @ Z                               # trailing: S(i, j) -= Z_i^T Z_j
        S[K, :] = new
        S[:, K] = new.T                            # = Z_i^T F
        S[K, K] = -(F.T @ F)
    return -S


def solve9(A, Pcc, s2):
    c6 = A.shape[0]
    n = (c6 + TB - 1) // TB * TB
    Ap = np.zeros((n, n))
    Ap[:c6, :c6] = A
    Pp = np.eye(n)
    Pp[:c6, :c6] = Pcc
    G = blocked_cholesky(Pp, np.diag(Pp).copy())
    L = G.T
    Q = Ap @ L
    M = s2 * np.eye(n) + L.T @ Q
    Mi = blocked_sweep_inverse(M)
    X = Mi @ G
    W = (np.eye(n) - Q @ X) / s2
    return W[:c6, :c6]


if __name__ == "__main__":
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle as O

    abi = O.abi

    def run(cfg, n, full=False, **kw):
        seq = O.rv.synth.SynthSequence(cfg, duration=(38 + n + 4) / 20.0, **kw)
        w, a, ni = seq.init_from_static(38)
        x, P = O.initialize(cfg, w, a, ni)
        trk, drv, img = O.Tracker(cfg), O.rv.synth.DirectTrackDriver(seq), 0
        s2 = float(np.float32(max(cfg.sigma_px, cfg.sigma_py))) ** 2
        worst = dict(W=0.0, dx=0.0, U=0.0, n=0)
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
                    Hw, r, _ = O.update_stack(cfg, x1, P1, types, lens, meas)
                    A, b, Pcc, Pc = Hw.T @ Hw, Hw.T @ r, P1[24:, 24:], P1[:, 24:]
                    W0 = np.linalg.inv(s2 * np.eye(len(A)) + A @ Pcc)
                    W = solve9(A, Pcc, s2)
                    worst["W"] = max(worst["W"], np.abs(W - W0).max() / np.abs(W0).max())
                    worst["dx"] = max(worst["dx"], np.abs(Pc @ (W @ b) - Pc @ (W0 @ b)).max())
                    worst["U"] = max(worst["U"], np.abs(Pc @ W - Pc @ W0).max() / np.abs(Pc @ W0).max())
                    worst["n"] += 1
            x, P, _, _ = O.augment_compose(cfg, x2, P2, img > 1)
        return worst

    for name, cn, n, kw in (("B stock", "B", 100, {}), ("B at rest", "B", 100, dict(motion="stationary")), ("B rotation", "B", 100, dict(motion="rotation")),
                            ("B full load", "B", 40, dict(full=True)), ("A full load", "A", 40, dict(full=True)), ("C full load", "C", 45, dict(full=True))):
        print(name, {k: ("%.1e" % v if isinstance(v, float) else v) for k, v in run(abi.config_named(cn), n, **kw).items()})
    # zero-variance directions (an IMU stream that ended: clones with exactly zero rows / columns) and a singular A
    rng = np.random.default_rng(0)
    c6, s2 = 60, 4.7e-6
    B = rng.normal(size=(c6, c6))
    Pcc = B @ B.T * 1e-4
    Pcc[48:, :] = 0
    Pcc[:, 48:] = 0
    H = rng.normal(size=(20, c6)) * 30
    A = H.T @ H
    W0 = np.linalg.inv(s2 * np.eye(c6) + A @ Pcc)
    print("PSD Pcc (12 zero rows), rank-20 A: rel err", "%.1e" % (np.abs(solve9(A, Pcc, s2) - W0).max() / np.abs(W0).max()))
