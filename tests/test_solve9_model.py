"""The NumPy model of solve9.hip (tools/solve9_model.py): W = (s2 I + A Pcc)^-1 through SPD pieces only — blocked Cholesky of Pcc
(semi-definite safe), blocked symmetric sweep of M = s2 I + L^T A L in its Cholesky form, Woodbury.  The kernel follows this tile algebra
step by step; here the algebra itself is held against numpy.linalg.inv on recorded updates, on a clone block with zero-variance directions
(what an IMU stream that ended leaves behind) and on a badly conditioned synthetic case where the textbook form of the sweep fails."""
import os
import sys

import numpy as np

import oracle as O
import scenarios as S

sys.path.insert(0, os.path.join(O.ROOT, "tools"))
import solve9_model as M  # noqa: E402

abi = O.abi


def _cases(name, n_frames):
    cfg = abi.config_named(name)
    seq, recs = S.record_sequence(cfg, n_frames=n_frames, duration=(38 + n_frames + 4) / 20.0)
    s2 = float(np.float32(max(cfg.sigma_px, cfg.sigma_py))) ** 2
    for r in recs:
        if not r["did_update"] or not r["diag"]["updated"]:
            continue
        Hw, rr, _ = O.update_stack(cfg, r["x1"], r["P1"], r["types"], r["lens"], r["meas"])
        yield s2, Hw.T @ Hw, Hw.T @ rr, r["P1"]


def test_model_equals_the_lu_inverse_on_recorded_updates():
    worst_w, worst_dx, n = 0.0, 0.0, 0
    for name, nf in (("B", 40), ("A", 24)):
        for s2, A, b, P1 in _cases(name, nf):
            Pcc, Pc = P1[24:, 24:], P1[:, 24:]
            W0 = np.linalg.inv(s2 * np.eye(len(A)) + A @ Pcc)
            W = M.solve9(A, Pcc, s2)
            worst_w = max(worst_w, np.abs(W - W0).max() / np.abs(W0).max())
            worst_dx = max(worst_dx, np.abs(Pc @ (W @ b) - Pc @ (W0 @ b)).max())
            n += 1
    assert n > 40 and worst_w < 1e-12 and worst_dx < 1e-14, (n, worst_w, worst_dx)


def test_tile_factor_is_the_inverse_cholesky_factor_and_skips_zero_directions():
    rng = np.random.default_rng(0)
    B = rng.normal(size=(16, 16))
    Mk = B @ B.T + 0.1 * np.eye(16)
    F = M.tile_inv_factor(Mk, np.diag(Mk).copy())
    assert np.abs(F - np.linalg.inv(np.linalg.cholesky(Mk))).max() < 1e-12
    Mz = Mk.copy()
    Mz[[5, 10], :] = 0
    Mz[:, [5, 10]] = 0
    Fz = M.tile_inv_factor(Mz, np.diag(Mz).copy())
    assert not Fz[5].any() and not Fz[10].any()
    keep = [i for i in range(16) if i not in (5, 10)]
    assert np.abs(Fz[np.ix_(keep, keep)] - np.linalg.inv(np.linalg.cholesky(Mz[np.ix_(keep, keep)]))).max() < 1e-12


def test_zero_variance_clones_and_a_singular_information_block():
    """Pcc with exactly zero rows / columns (clones of a stream whose IMU ended) and rank(A) = 20 < 6n: the pivoted LU of T has no trouble,
    neither may the SPD route (no inverse of Pcc or L anywhere)"""
    rng = np.random.default_rng(0)
    c6, s2 = 60, 4.7e-6
    B = rng.normal(size=(c6, c6))
    Pcc = B @ B.T * 1e-4
    Pcc[48:, :] = 0
    Pcc[:, 48:] = 0
    H = rng.normal(size=(20, c6)) * 30
    A = H.T @ H
    T = s2 * np.eye(c6) + A @ Pcc
    W0 = np.linalg.inv(T)
    W = M.solve9(A, Pcc, s2)
    # cond(T) = 7e8 here: LAPACK's own inverse is good to ~1e-7
    assert np.abs(W - W0).max() / np.abs(W0).max() < 5e-6


def test_cholesky_form_of_the_sweep_is_the_stable_one():
    """forming D = F^T F and multiplying with it loses the tile's backward stability once a diagonal tile mixes strong and weak directions"""
    rng = np.random.default_rng(0)
    n = 64
    H = rng.normal(size=(20, n)) * 2e3
    Mm = 4.7e-6 * np.eye(n) + H.T @ H * 1e-4
    X = M.blocked_sweep_inverse(Mm)
    assert np.abs(X @ Mm - np.eye(n)).max() < 1e-5

    def textbook(Mx):
        Sx = Mx.copy()
        for k in range(n // 16):
            K = slice(16 * k, 16 * k + 16)
            D = np.linalg.inv(Sx[K, K])
            old = Sx[K, :].copy()
            new = D @ old
            Sx -= old.T @ new
            Sx[K, :] = new
            Sx[:, K] = new.T
            Sx[K, K] = -D
        return -Sx
    assert np.abs(textbook(Mm) @ Mm - np.eye(n)).max() > 1e-4
