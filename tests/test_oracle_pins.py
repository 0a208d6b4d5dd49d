"""Pins for the CPU oracle (no GPU).  The reference ships no golden vectors, so the
oracle is pinned by (1) the one external ground truth available — glibc rand() and the reference's
chi-square table, (2) analytic identities, (3) physical consistency of the whole filter on simulated
data, (4) frozen snapshots under tests/golden/ (regression)."""
import ctypes as C
import os

import numpy as np
import pytest

import oracle as O
import scenarios as S

abi, rv = O.abi, O.rv
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def rand_unit_quat(rng):
    q = rng.standard_normal(4)
    q /= np.linalg.norm(q)
    return q if q[3] >= 0 else -q


# ---- (1) external ground truth
def test_rand_stream_equals_glibc():
    libc = C.CDLL("libc.so.6")
    libc.srand(1)
    want = [libc.rand() for _ in range(2000)]
    assert list(O.rand_stream(2000, seed=1)) == want
    libc.srand(12345)
    want = [libc.rand() for _ in range(100)]
    assert list(O.rand_stream(100, seed=12345)) == want


def test_chi2_table_is_the_095_quantile():
    from scipy.stats import chi2
    for dof in (1, 2, 3, 19, 59, 100, 500):
        assert abs(O.chi2_95(dof) - chi2.ppf(0.95, dof)) < 5.1e-7


# ---- (2) analytic identities
def test_quaternion_helpers():
    rng = np.random.default_rng(0)
    for _ in range(50):
        q1, q2 = rand_unit_quat(rng), rand_unit_quat(rng)
        R1, R2 = O.quat_to_rot(q1), O.quat_to_rot(q2)
        assert np.allclose(R1 @ R1.T, np.eye(3), atol=1e-14) and abs(np.linalg.det(R1) - 1) < 1e-14
        q12 = O.quat_mul(q1, q2)
        assert q12[3] >= 0 and abs(np.linalg.norm(q12) - 1) < 1e-15
        assert np.allclose(O.quat_to_rot(q12), R1 @ R2, atol=1e-14)      # JPL: R(q1 (x) q2) = R(q1) R(q2)
        qb = O.rot_to_quat(R1)
        assert np.allclose(O.quat_to_rot(qb), R1, atol=1e-14)
    # all four Breckenridge branches
    for ax in np.eye(3):
        for ang in (0.1, 3.0, np.pi - 1e-3):
            K = np.array([[0, -ax[2], ax[1]], [ax[2], 0, -ax[0]], [-ax[1], ax[0], 0.0]])
            R = np.eye(3) + np.sin(ang) * K + (1 - np.cos(ang)) * K @ K
            assert np.allclose(O.quat_to_rot(O.rot_to_quat(R)), R, atol=1e-12)


def test_propagate_matches_fine_step_integration():
    """closed-form per-sample integration vs the same IMU signal integrated with 50x finer steps"""
    cfg = abi.config_named("B", enable_equalizer=0)
    rng = np.random.default_rng(1)
    x = np.zeros(26)
    x[0:4] = rand_unit_quat(rng); x[7:10] = [0.05, -0.02, 1.0]; x[7:10] /= np.linalg.norm(x[7:10])
    x[13] = 1.0; x[17:20] = [0.4, -0.2, 0.1]; x[20:23] = 1e-3 * rng.standard_normal(3); x[23:26] = 1e-2 * rng.standard_normal(3)
    P = np.eye(24) * 1e-6
    w, a = np.array([0.3, -0.2, 0.5]), np.array([0.2, 0.1, 9.9])
    imu = abi.as_imu_array([w] * 10, [a] * 10, np.arange(10) * 0.005, [0.005] * 10)
    fine = abi.as_imu_array([w] * 500, [a] * 500, np.arange(500) * 1e-4, [1e-4] * 500)
    x1, _ = O.propagate(cfg, x, P.copy(), imu)
    x2, _ = O.propagate(cfg, x, P.copy(), fine[:64]) if False else (None, None)
    # integrate the fine sequence in chunks of <=64 samples is not allowed (robocentric state resets), so
    # compare against an independent numpy integration of the same closed-form model with constant input:
    xs = x.copy()
    for chunk in range(1):
        pass
    # constant (w, a): one sample of dt=0.05 must equal ten samples of dt=0.005 for the rotation part
    one = abi.as_imu_array([w], [a], [0.0], [0.05])
    x3, _ = O.propagate(cfg, x, P.copy(), one)
    assert np.allclose(x1[10:14], x3[10:14], atol=1e-12)          # rotation composes exactly
    assert np.allclose(x1[14:17], x3[14:17], atol=2e-5)           # position: O(dt^2) coupling of rotating gravity
    assert np.allclose(x1[17:20], x3[17:20], atol=5e-4)


def test_propagate_covariance_is_symmetric_psd_and_mutates_clone_cross_terms():
    cfg = abi.config_named("B", enable_equalizer=0)
    seq, recs = S.record_sequence(cfg, n_frames=16)
    r = recs[-1]
    P1 = r["P1"]
    assert np.array_equal(P1, P1.T)
    assert np.linalg.eigvalsh(P1).min() > -1e-12
    assert not np.allclose(P1[:24, 24:], r["P0"][:24, 24:])
    assert np.array_equal(P1[24:, 24:], r["P0"][24:, 24:])        # clone block untouched (PreIntegrator.cc:186-193)


def test_update_jacobian_consistency_via_information_form():
    """the QR-compressed update (reference form) and the information-form update [A|b] = Hw^T[Hw|r]
    are algebraically identical: x+ and P+ agree to rounding."""
    cfg = abi.config_named("B", enable_equalizer=0)
    seq, recs = S.record_sequence(cfg, n_frames=30)
    for r in (recs[12], recs[20], recs[29]):
        xo, Po, d = O.update(cfg, r["x1"], r["P1"], r["types"], r["lens"], r["meas"])
        blk = O.update_local(cfg, r["x1"], r["P1"], r["types"], r["lens"], r["meas"], 0, 1)
        xi, Pi, di = O.update_global(cfg, r["x1"], r["P1"], blk[None, :])
        assert di["n_good"] == d["n_good"] and di["n_rows"] == d["n_rows"]
        assert S.state_delta(xo, xi) < 1e-10
        assert np.max(np.abs(Po - Pi)) < 1e-10 * np.max(np.abs(Po)) + 1e-16
        assert np.linalg.eigvalsh(Po).min() > -1e-12


def test_update_sharded_equals_unsharded():
    cfg = abi.config_named("B", enable_equalizer=0)
    seq, recs = S.record_sequence(cfg, n_frames=30)
    r = recs[-1]
    types, lens, meas = S.worst_case_tracks(cfg, r, seq)
    one = O.update_local(cfg, r["x1"], r["P1"], types, lens, meas, 0, 1)
    x1, P1, _ = O.update_global(cfg, r["x1"], r["P1"], one[None, :])
    for world in (2, 8):
        blks = np.stack([O.update_local(cfg, r["x1"], r["P1"], types, lens, meas, rk, world) for rk in range(world)])
        xw, Pw, dw = O.update_global(cfg, r["x1"], r["P1"], blks)
        assert dw["updated"] == 1
        assert S.state_delta(x1, xw) < 1e-11 and np.max(np.abs(P1 - Pw)) < 1e-12 * np.max(np.abs(P1)) + 1e-18


def test_gate_rejects_an_outlier_and_too_few_features_pass_through():
    cfg = abi.config_named("B", enable_equalizer=0)
    seq, recs = S.record_sequence(cfg, n_frames=30)
    r = recs[-1]
    types, lens, meas = r["types"].copy(), r["lens"].copy(), r["meas"].copy()
    base = O.update(cfg, r["x1"], r["P1"], types, lens, meas)[2]
    f = int(np.flatnonzero(base["accepted"])[0])
    meas[f, lens[f] - 1] += 0.2                                     # gross outlier in the last observation
    d = O.update(cfg, r["x1"], r["P1"], types, lens, meas)[2]
    assert d["accepted"][f] == 0 and d["gamma"][f] > O.chi2_95(d["ndof"][f])
    xo, Po, d2 = O.update(cfg, r["x1"], r["P1"], types[:2], lens[:2], meas[:2])
    assert d2["updated"] == 0 and np.array_equal(xo, r["x1"]) and np.array_equal(Po, r["P1"])


def test_augment_is_a_gather_and_window_slides():
    cfg = abi.config_named("B", enable_equalizer=0)
    seq, recs = S.record_sequence(cfg, n_frames=16)
    r = recs[3]                                                     # window still growing
    n = (len(r["x2"]) - 26) // 7
    assert len(r["x3"]) == 26 + 7 * (n + 1)
    r = recs[-1]                                                    # full: slides
    assert len(r["x3"]) == len(r["x2"]) == 26 + 7 * (cfg.max_track_len - 1)
    assert np.allclose(r["x3"][10:17], [0, 0, 0, 1, 0, 0, 0])      # composition resets the relative pose
    assert np.allclose(r["P3"][9:15, :], 0)                         # ... and its covariance rows (System.cc:344-353)
    assert np.array_equal(r["P3"], r["P3"].T)


# ---- (3) physical consistency: the restated filter tracks a simulated trajectory
def test_filter_tracks_simulated_ground_truth():
    cfg = abi.config_named("B", enable_equalizer=0)
    seq = rv.synth.SynthSequence(cfg, duration=10.0)
    k0 = 38
    w, a, n = seq.init_from_static(k0)
    x, P = O.initialize(cfg, w, a, n)
    s = O.System(cfg)
    s.set_state(x, P)
    drv = rv.synth.DirectTrackDriver(seq)
    trk = s.tracker()
    R0, p0 = seq.pose(seq.frame_time(k0 + 1))
    worst = 0
    for k in range(k0 + 1, seq.n_frames()):
        inp = drv.inputs(k)
        info, tms, pp, pq = s.frame(inp["imu"], inp["cand"], tracked=inp["tracked"], status=inp["status"])
        drv.after(trk.get_points()[0])
        R, p = seq.pose(seq.frame_time(k))
        travelled = np.linalg.norm(R0.T @ (p - p0))
        worst = max(worst, abs(np.linalg.norm(pp) - travelled))
    assert info["updated"] == 1
    assert worst < 0.25, worst   # distance-from-start error stays below 25 cm over 8 s / ~5 m of travel (1 px noise)


# ---- front end
def test_undistort_inverts_the_distortion_model():
    cfg = abi.config_named("B", enable_equalizer=0)
    rng = np.random.default_rng(3)
    xn = rng.uniform(-0.6, 0.6, (200, 2))
    r2 = (xn ** 2).sum(1)
    k1, k2, p1, p2 = float(cfg.k1), float(cfg.k2), float(cfg.p1), float(cfg.p2)
    cd = 1 + (k2 * r2 + k1) * r2
    xd = xn[:, 0] * cd + 2 * p1 * xn[:, 0] * xn[:, 1] + p2 * (r2 + 2 * xn[:, 0] ** 2)
    yd = xn[:, 1] * cd + p1 * (r2 + 2 * xn[:, 1] ** 2) + 2 * p2 * xn[:, 0] * xn[:, 1]
    px = np.stack([float(cfg.fx) * xd + float(cfg.cx), float(cfg.fy) * yd + float(cfg.cy)], 1).astype(np.float32)
    un = O.undistort(cfg, px)
    assert np.max(np.abs(un - xn)) < 2e-3     # 5 fixed-point iterations (cv::undistortPoints), float32 pixels


def test_fisheye_undistort_inverts_the_equidistant_model():
    """Camera.Fisheye: 1 -> cv::fisheye::undistortPoints with D = (k1, k2, p1, p2) (Tracker.cc:116-119): project rays through
    theta_d = theta (1 + k1 theta^2 + k2 theta^4 + k3 theta^6 + k4 theta^8) and undistort the pixels again"""
    cfg = abi.config_named("B", fisheye=1, k1=-0.0137, k2=0.0207, p1=-0.0128, p2=0.0025)
    rng = np.random.default_rng(3)
    xn = rng.uniform(-0.9, 0.9, (200, 2))
    r = np.linalg.norm(xn, axis=1)
    th = np.arctan(r)
    k = [float(cfg.k1), float(cfg.k2), float(cfg.p1), float(cfg.p2)]
    thd = th * (1 + k[0] * th ** 2 + k[1] * th ** 4 + k[2] * th ** 6 + k[3] * th ** 8)
    xd = xn * (thd / r)[:, None]
    px = np.stack([float(cfg.fx) * xd[:, 0] + float(cfg.cx), float(cfg.fy) * xd[:, 1] + float(cfg.cy)], 1).astype(np.float32)
    un = O.undistort(cfg, px)
    assert np.max(np.abs(un - xn)) < 5e-5      # float32 pixels in, ten fixed-point iterations
    assert np.array_equal(O.undistort(cfg, np.float32([[cfg.cx, cfg.cy]])), np.zeros((1, 2), np.float32))   # theta_d <= 1e-8: scale 1


def test_fisheye_wide_field_points_follow_the_opencv_3_3_form():
    """The restated cv::fisheye::undistortPoints is the OpenCV 3.3 form (the version the reference's README names): ten fixed-point
    iterations, NO clamp of theta_d.  Later versions differ for wide-field points — 3.4.x clamps theta_d to [-pi/2, pi/2], 4.x iterates with
    Newton steps and flags points that do not converge — so a reference built against a newer OpenCV returns something else there.  Pinned
    here: an independent numpy write-up of the 3.3 loop agrees to the last float for rays up to 88 degrees off axis and for theta_d beyond
    pi/2, and the 3.4 clamp would change exactly the points with theta_d > pi/2 (stated, not adopted)."""
    cfg = abi.config_named("B", fisheye=1, k1=-0.0137, k2=0.0207, p1=-0.0128, p2=0.0025)
    k = [float(cfg.k1), float(cfg.k2), float(cfg.p1), float(cfg.p2)]
    rng = np.random.default_rng(11)
    thd = np.concatenate([rng.uniform(0.05, 1.53, 60), rng.uniform(np.pi / 2, 2.2, 20)])      # distorted angles, 20 of them beyond pi/2
    ang = rng.uniform(0, 2 * np.pi, len(thd))
    xd = np.stack([thd * np.cos(ang), thd * np.sin(ang)], 1)
    px = np.stack([float(cfg.fx) * xd[:, 0] + float(cfg.cx), float(cfg.fy) * xd[:, 1] + float(cfg.cy)], 1).astype(np.float32)
    un = O.undistort(cfg, px)
    want, clamped = np.zeros_like(un), np.zeros_like(un)
    for i, (u, v) in enumerate(px):
        pw = np.array([(float(u) - float(cfg.cx)) / float(cfg.fx), (float(v) - float(cfg.cy)) / float(cfg.fy)])
        for out, clamp in ((want, False), (clamped, True)):
            td = float(np.linalg.norm(pw))
            if clamp:
                td = min(max(-np.pi / 2, td), np.pi / 2)
            th = td
            for _ in range(10):
                t2 = th * th
                th = td / (1 + k[0] * t2 + k[1] * t2 ** 2 + k[2] * t2 ** 3 + k[3] * t2 ** 4)
            out[i] = (pw * (np.tan(th) / np.linalg.norm(pw))).astype(np.float32)
    assert np.array_equal(un, want)
    wide = thd > np.pi / 2
    assert np.all(np.any(clamped[wide] != want[wide], axis=1)) and np.array_equal(clamped[~wide], want[~wide])


def test_pyr_down_and_scharr_against_numpy():
    rng = np.random.default_rng(5)
    img = rng.integers(0, 256, (37, 53), dtype=np.uint8)
    k = np.array([1, 4, 6, 4, 1])
    pad = np.pad(img.astype(np.int64), 2, mode="reflect")
    full = sum(k[i] * k[j] * pad[i:i + 37, j:j + 53] for i in range(5) for j in range(5))
    want = ((full[::2, ::2] + 128) >> 8).astype(np.uint8)
    assert np.array_equal(O.pyr_down(img), want)
    p1 = np.pad(img.astype(np.int64), 1, mode="reflect")
    sm_v = 3 * p1[:-2, :] + 10 * p1[1:-1, :] + 3 * p1[2:, :]
    dx = sm_v[:, 2:] - sm_v[:, :-2]
    df_v = p1[2:, :] - p1[:-2, :]
    dy = 3 * df_v[:, :-2] + 10 * df_v[:, 1:-1] + 3 * df_v[:, 2:]
    got = O.scharr(img)
    assert np.array_equal(got[..., 0], dx) and np.array_equal(got[..., 1], dy)


def test_klt_recovers_known_subpixel_shifts():
    cfg = abi.config_named("B", enable_equalizer=0)
    seq = rv.synth.SynthSequence(cfg, duration=4.0)
    img = seq.render(60)
    xy, vis = seq.project(60, noise=False)
    pts = xy[vis][:120]
    from scipy.ndimage import shift as nd_shift
    for dxy in ((0.3, -0.4), (2.6, 1.2), (-7.5, 5.25), (11.0, -9.0)):
        img2 = np.clip(np.rint(nd_shift(img.astype(np.float64), (dxy[1], dxy[0]), order=3, mode="nearest")), 0, 255).astype(np.uint8)
        out, st = O.klt(img, img2, pts)
        ok = st > 0
        assert ok.mean() > 0.9
        err = out[ok] - pts[ok] - np.array(dxy, np.float32)
        assert np.median(np.abs(err)) < 0.05, (dxy, np.median(np.abs(err)))
    # a textureless point fails the min-eigenvalue test, a point outside the image is rejected
    flat = np.full_like(img, 100)
    out, st = O.klt(flat, flat, np.array([[300.0, 200.0]], np.float32))
    assert st[0] == 0
    out, st = O.klt(img, img, np.array([[-40.0, 100.0], pts[0]], np.float32))
    assert st[0] == 0 and st[1] == 1 and np.allclose(out[1], pts[0], atol=1e-3)


def test_ransac_flags_outliers_and_is_deterministic():
    cfg = abi.config_named("B", enable_equalizer=0)
    rng = np.random.default_rng(9)
    n = 200
    X = np.stack([rng.uniform(-2, 2, n), rng.uniform(-1.5, 1.5, n), rng.uniform(3, 8, n)], 1)
    w = np.array([0.02, -0.03, 0.01])
    imu = abi.as_imu_array([w] * 10, [[0, 0, 9.8]] * 10, np.arange(10) * 0.005, [0.005] * 10)
    T = np.array(list(cfg.T_bc)).reshape(4, 4)
    Ric = T[:3, :3]
    th = w * 0.05
    K = np.array([[0, -th[2], th[1]], [th[2], 0, -th[0]], [-th[1], th[0], 0.0]])
    Rb = np.eye(3) - K + 0.5 * K @ K            # JPL-style delta rotation of the body
    Rc = Ric.T @ Rb @ Ric
    t = np.array([0.03, 0.01, -0.02])
    X2 = X @ Rc.T + t
    p1 = X / X[:, 2:3]
    p2 = X2 / X2[:, 2:3]
    bad = rng.choice(n, 60, replace=False)
    p2[bad, :2] += rng.uniform(0.02, 0.05, (60, 2)) * rng.choice([-1, 1], (60, 2))
    flags = np.ones(n, np.uint8)
    n_in, fo, win, pairs, st = O.ransac(cfg, p1, p2, imu, flags)
    good = np.setdiff1d(np.arange(n), bad)
    assert fo[good].mean() > 0.95 and fo[bad].mean() < 0.2
    assert len(set(pairs.reshape(-1))) == 32       # 16 disjoint pairs
    n_in2, fo2, win2, pairs2, _ = O.ransac(cfg, p1, p2, imu, flags)
    assert np.array_equal(fo, fo2) and win == win2 and np.array_equal(pairs, pairs2)
    # too few candidates: flags untouched (Ransac.cc:201-205)
    few = np.zeros(n, np.uint8); few[:20] = 1
    n3, fo3, _, _, _ = O.ransac(cfg, p1, p2, imu, few)
    assert n3 == 0 and np.array_equal(fo3, few)


# ---- (4) frozen snapshots
def test_golden_snapshot_regression():
    path = os.path.join(GOLD, "cfgB_direct_seed0_frame30.npz")
    g = np.load(path)
    cfg = abi.config_named("B", enable_equalizer=0)
    seq, recs = S.record_sequence(cfg, n_frames=30)
    r = recs[-1]
    assert S.state_delta(r["x3"], g["x3"]) < 1e-9
    assert np.max(np.abs(r["P3"] - g["P3"])) < 1e-9 * np.max(np.abs(g["P3"]))
    assert np.array_equal(r["diag"]["accepted"], g["accepted"])
    assert np.array_equal(r["pts"], g["pts"])


def test_golden_image_fixture_regression():
    """tests/golden/small_images_tracker.npz (4 frames of a half-size camera: CLAHE + DetectWithSubPix + KLT + RANSAC + book-keeping):
    the oracle reproduces the committed feature lists — the same file the GPU suite checks the HIP path against"""
    import zlib
    g = np.load(os.path.join(GOLD, "small_images_tracker.npz"))
    cfg = S.small_image_config()
    t = O.Tracker(cfg)
    for i in range(4):
        t.track(g["imgs"][i], g["imu%d" % i].view(abi.IMU_DTYPE), None)
        pts, hl = t.get_points()
        assert np.array_equal(pts, g["pts%d" % i]) and np.array_equal(hl, g["hist%d" % i]), i
    assert np.array_equal(O.detect(cfg, O.clahe(g["imgs"][0]), 1), g["corners0"])
    assert zlib.crc32(O.clahe(g["imgs"][0]).tobytes()) == int(g["clahe0_crc"])


def _find_newer_py(cfg, corners, ref):
    """FeatureDetector::FindNewer/ChessGrid (FeatureDetector.cc:78-150) written out independently, with the member types of
    FeatureDetector.h:66-77: mnGridCols/Rows, mnOffsetX/Y and mnMaxFeatsPerBlock are int (their initialisers truncate),
    mnBlockSizeX/Y and mnMinDistance are float"""
    f32 = np.float32
    bx, by, md = f32(cfg.block_x), f32(cfg.block_y), f32(cfg.min_dist)
    gc, gr = int(np.floor(f32(cfg.width) / bx)), int(np.floor(f32(cfg.height) / by))
    offx, offy = int(.5 * (cfg.width - gc * float(bx))), int(.5 * (cfg.height - gr * float(by)))
    maxpb = int(f32(cfg.n_features) / f32(gc * gr))
    grid = [[] for _ in range(gc * gr)]

    def outside(p):
        return p[0] <= f32(offx) or p[1] <= f32(offy) or p[0] >= f32(cfg.width - offx) or p[1] >= f32(cfg.height - offy)

    def cell(p):
        return int(np.floor((p[0] - f32(offx)) / bx)), int(np.floor((p[1] - f32(offy)) / by))
    for p in ref:
        if not outside(p):
            c, r = cell(p)
            grid[r * gc + c].append(p)
    out = []
    for p in corners:
        if outside(p):
            continue
        c, r = cell(p)
        xl = f32(f32(c) * bx + f32(offx)); xr = f32(xl + bx); yt = f32(f32(r) * by + f32(offy)); yb = f32(yt + by)
        if abs(f32(p[0] - xl)) < md or abs(f32(p[0] - xr)) < md or abs(f32(p[1] - yt)) < md or abs(f32(p[1] - yb)) < md:
            continue
        g = grid[r * gc + c]
        if float(f32(len(g))) < .75 * maxpb:
            if all(np.sqrt(float(f32(p[0] - q[0])) ** 2 + float(f32(p[1] - q[1])) ** 2) > float(md) for q in g):
                out.append(p)
                g.append(p)
    return out


FIND_NEWER_CASES = [dict(block_x=75, block_y=75, n_features=250), dict(block_x=75.5, block_y=60.25, n_features=250),
                    dict(block_x=94, block_y=80, n_features=100)]


def find_newer_inputs(cfg, seed=3):
    """ten tracked points (too few for RANSAC: every one stays) and a dense, randomly ordered candidate list"""
    rng = np.random.default_rng(seed)
    ref = (rng.uniform([20, 20], [cfg.width - 20, cfg.height - 20], (10, 2))).astype(np.float32)
    cand = (rng.uniform([0, 0], [cfg.width, cfg.height], (cfg.n_features, 2))).astype(np.float32)
    return ref, cand


@pytest.mark.parametrize("case", FIND_NEWER_CASES)
def test_find_newer_uses_the_int_members_of_the_reference(case):
    """odd left-over border (376 - 5*75 = 1 -> mnOffsetX = 0, not 0.5), nFeatures not divisible by the block count
    (250/15 -> 16, the refill cap .75*16 = 12, not 12.5) and non-integer block sizes (float members upstream)"""
    cfg = abi.config_named("B", width=376, height=240, fx=229.327, fy=228.648, cx=183.6075, cy=124.1875, min_dist=5, enable_equalizer=0, **case)
    ref, cand = find_newer_inputs(cfg)
    t = O.Tracker(cfg)
    imu = np.zeros(0, abi.IMU_DTYPE)
    t.track_points(np.zeros((0, 2), np.float32), np.zeros(0, np.uint8), imu, ref)          # first image: the list is taken as it is
    assert np.array_equal(t.get_points()[0], ref)
    t.track_points(ref, np.ones(len(ref), np.uint8), imu, cand)                              # refill through FindNewer
    pts = t.get_points()[0]
    want = _find_newer_py(cfg, cand, ref)[: cfg.n_features - len(ref)]
    assert len(want) > 40
    assert np.array_equal(pts[: len(ref)], ref)
    assert np.array_equal(pts[len(ref):], np.array(want, np.float32))


# ---- (5) finite-difference pins of the two Jacobian builders
def _small_quat(th):
    q = np.zeros(4)
    q[:3] = .5 * np.asarray(th, float)
    q[3] = np.sqrt(1 - q[:3] @ q[:3])
    return q


def _boxplus(x, d):
    """the state injection of Updater.cc:546-613 (q+ = dq (x) q, everything else additive), i.e. the error-state convention"""
    x = np.array(x, float)
    n = (len(x) - 26) // 7
    x[0:4] = O.quat_mul(_small_quat(d[0:3]), x[0:4]); x[4:10] += d[3:9]
    x[10:14] = O.quat_mul(_small_quat(d[9:12]), x[10:14]); x[14:26] += d[12:24]
    for j in range(n):
        x[26 + 7 * j:30 + 7 * j] = O.quat_mul(_small_quat(d[24 + 6 * j:27 + 6 * j]), x[26 + 7 * j:30 + 7 * j])
        x[30 + 7 * j:33 + 7 * j] += d[27 + 6 * j:30 + 6 * j]
    return x


def _boxminus(xa, xb):
    n = (len(xa) - 26) // 7
    d = np.zeros(24 + 6 * n)

    def dth(qa, qb):
        qi = np.array(qb, float)
        qi[:3] *= -1
        return 2 * O.quat_mul(qa, qi)[:3]
    d[0:3] = dth(xa[0:4], xb[0:4]); d[3:9] = xa[4:10] - xb[4:10]
    d[9:12] = dth(xa[10:14], xb[10:14]); d[12:24] = xa[14:26] - xb[14:26]
    for j in range(n):
        d[24 + 6 * j:27 + 6 * j] = dth(xa[26 + 7 * j:30 + 7 * j], xb[26 + 7 * j:30 + 7 * j])
        d[27 + 6 * j:30 + 6 * j] = xa[30 + 7 * j:33 + 7 * j] - xb[30 + 7 * j:33 + 7 * j]
    return d


def test_update_jacobians_against_central_differences():
    """U3 (Updater.cc:271-368): Hx w.r.t. every clone's (theta, p) through the relative-pose chain and Hf w.r.t. (phi, psi, rho),
    before the nullspace projection, against central differences of the residual itself at a fixed inverse-depth triple — for
    type-'1' (newest clones, column offset 6(n-(L-1))) and type-'2' (oldest clones, first ceil(L/2) observations) features of several
    lengths.  The residual passes through float32 (cv::Point2f, Updater.cc:307-308), which limits the agreement to ~3e-4."""
    cfg = abi.config_named("B", enable_equalizer=0)
    seq, recs = S.record_sequence(cfg, n_frames=30)
    r = recs[-1]
    x = r["x1"]
    n = (len(x) - 26) // 7
    types, lens, meas = S.worst_case_tracks(cfg, r, seq)
    h = 1e-4
    seen = set()
    for f in range(len(types)):
        key = (int(types[f]), int(lens[f]))
        if key in seen or len(seen) >= 6:
            continue
        seen.add(key)
        r0, Hx, Hf, pf = O.feature_model(cfg, x, types[f], meas[f], lens[f])
        Lu = (lens[f] + 1) // 2 if types[f] == ord("2") else lens[f]
        assert len(r0) == 2 * Lu
        Hfd = np.zeros_like(Hx)
        for k in range(6 * n):
            d = np.zeros(24 + 6 * n)
            d[24 + k] = h
            rp = O.feature_model(cfg, _boxplus(x, d), types[f], meas[f], lens[f], pf)[0]
            rm = O.feature_model(cfg, _boxplus(x, -d), types[f], meas[f], lens[f], pf)[0]
            Hfd[:, k] = (rm - rp) / (2 * h)            # r = z - h(x): dr = -H dx
        assert np.abs(Hx).max() > 0.5 and np.abs(Hx - Hfd).max() < 1e-3, key
        lo = 6 * (n - (lens[f] - 1)) if types[f] == ord("1") else 0
        assert not Hx[:, :lo].any() and not Hx[:, lo + 6 * (Lu - 1):].any()        # the column range of the feature (Updater.cc:288-293)
        Hffd = np.zeros_like(Hf)
        for k in range(3):
            d = np.zeros(3)
            d[k] = h
            Hffd[:, k] = (O.feature_model(cfg, x, types[f], meas[f], lens[f], pf - d)[0] - O.feature_model(cfg, x, types[f], meas[f], lens[f], pf + d)[0]) / (2 * h)
        assert np.abs(Hf - Hffd).max() < 1e-3, key
    assert len(seen) >= 4


def test_propagate_transition_against_central_differences():
    """PreIntegrator::propagate (PreIntegrator.cc:97-193): with four clones and P[0:24,24:] = I the propagated cross block IS Psi =
    prod(I + dt F); its columns against central differences of the propagated MEAN w.r.t. the initial error state.  Blocks that
    grow linearly in time agree to 1e-3; blocks that grow quadratically (v<-bg, pk<-g, pk<-ba) carry the (m-1)/m factor of the
    reference's first-order Phi = I + dt F and agree within 15 %.  Columns theta_k, p_k are left out: composition (System.cc:344-353)
    resets that pose to identity with zero covariance, so the mean integration never sees a perturbed start there."""
    cfg = abi.config_named("B", enable_equalizer=0)
    seq, recs = S.record_sequence(cfg, n_frames=8)
    rec = [q for q in recs if (len(q["x0"]) - 26) // 7 == 4][0]
    x0, imu = rec["x0"], rec["inp"]["imu"]
    d = 48
    P = np.eye(d)
    P[:24, :24] *= 1e-6
    P[:24, 24:] = np.eye(24)
    P[24:, :24] = np.eye(24)
    x1, P1 = O.propagate(cfg, x0, P, imu)
    Psi = P1[:24, 24:]
    h = 1e-5
    Fd = np.zeros((24, 24))
    for k in range(24):
        dd = np.zeros(d)
        dd[k] = h
        xp, _ = O.propagate(cfg, _boxplus(x0, dd), P, imu)
        xm, _ = O.propagate(cfg, _boxplus(x0, -dd), P, imu)
        Fd[:, k] = (_boxminus(xp, x1) - _boxminus(xm, x1))[:24] / (2 * h)
    blk = dict(thG=0, pG=3, g=6, thk=9, pk=12, v=15, bg=18, ba=21)

    def B(M, a, b):
        return M[blk[a]:blk[a] + 3, blk[b]:blk[b] + 3]
    first_order = [("thG", "thG"), ("pG", "pG"), ("g", "g"), ("thk", "bg"), ("pk", "v"), ("v", "g"), ("v", "v"), ("v", "ba"), ("bg", "bg"), ("ba", "ba")]
    for a, b in first_order:
        m = np.abs(B(Psi, a, b)).max()
        assert m > 1e-3 and np.abs(B(Psi, a, b) - B(Fd, a, b)).max() < 1e-3 * max(m, 1.0) + 3e-6, (a, b)
    for a, b in [("v", "bg"), ("pk", "g"), ("pk", "ba")]:
        m = np.abs(B(Fd, a, b)).max()
        assert m > 1e-4 and np.abs(B(Psi, a, b) - B(Fd, a, b)).max() < 0.15 * m, (a, b)
    # nothing else: every remaining block outside the theta_k / p_k columns is zero in both
    for a in blk:
        for b in blk:
            if b in ("thk", "pk") or (a, b) in first_order or (a, b) in [("v", "bg"), ("pk", "g"), ("pk", "ba"), ("pk", "bg")]:
                continue
            assert np.abs(B(Psi, a, b)).max() < 1e-12 and np.abs(B(Fd, a, b)).max() < 1e-6, (a, b)


def _lk_numpy(I_pyr, D_pyr, J_pyr, pt):
    """cv::calcOpticalFlowPyrLK for one point, written from SURVEY.md appendix B.2 with NumPy window arithmetic (int64 sums
    converted once to float32, as oracle/frontend.cpp states for its accumulators): returns (x, y, status)"""
    f32 = np.float32
    W = 15

    def win(img, x0, y0, border):
        """17x17 samples img[y0-? ..]: rows y0..y0+15, cols x0..x0+15 with the padded-image semantics"""
        Hh, Ww = img.shape[:2]
        ys, xs = np.arange(y0, y0 + W + 1), np.arange(x0, x0 + W + 1)
        if border == "reflect":
            def rf(v, nmax):
                v = np.where(v < 0, -v, v)
                return np.where(v >= nmax, 2 * nmax - 2 - v, v)
            return img[np.ix_(rf(ys, Hh), rf(xs, Ww))].astype(np.int64)
        out = np.zeros((W + 1, W + 1) + img.shape[2:], np.int64)
        oky, okx = (ys >= 0) & (ys < Hh), (xs >= 0) & (xs < Ww)
        sub = img[np.ix_(ys[oky], xs[okx])].astype(np.int64)
        out[np.ix_(np.nonzero(oky)[0], np.nonzero(okx)[0])] = sub
        return out

    def weights(a, b):
        w00 = int(np.rint(f32(f32(f32(1) - a) * f32(f32(1) - b)) * f32(1 << 14)))
        w01 = int(np.rint(f32(a * f32(f32(1) - b)) * f32(1 << 14)))
        w10 = int(np.rint(f32(f32(f32(1) - a) * b) * f32(1 << 14)))
        return w00, w01, w10, (1 << 14) - w00 - w01 - w10

    def interp(p, w, shift):
        s = p[:-1, :-1] * w[0] + p[:-1, 1:] * w[1] + p[1:, :-1] * w[2] + p[1:, 1:] * w[3]
        return (s + (1 << (shift - 1))) >> shift

    st = 1
    nx = ny = f32(0)
    top = len(I_pyr) - 1
    scale = f32(1.0 / (1 << 20))
    for lv in range(top, -1, -1):
        I, D, J = I_pyr[lv], D_pyr[lv], J_pyr[lv]
        sc = f32(1.0 / (1 << lv))
        ppx, ppy = f32(pt[0]) * sc, f32(pt[1]) * sc
        if lv == top:
            nx, ny = ppx, ppy
        else:
            nx, ny = f32(nx * f32(2)), f32(ny * f32(2))
        ppx, ppy = f32(ppx - f32(7)), f32(ppy - f32(7))
        ix, iy = int(np.floor(ppx)), int(np.floor(ppy))
        if ix < -W or ix >= I.shape[1] or iy < -W or iy >= I.shape[0]:
            if lv == 0:
                st = 0
            continue
        w = weights(f32(ppx - f32(ix)), f32(ppy - f32(iy)))
        Iw = interp(win(I, ix, iy, "reflect"), w, 14 - 5)
        Dw = win(D, ix, iy, "zero")
        Ix, Iy = interp(Dw[..., 0], w, 14), interp(Dw[..., 1], w, 14)
        A11, A12, A22 = f32(f32(int((Ix * Ix).sum())) * scale), f32(f32(int((Ix * Iy).sum())) * scale), f32(f32(int((Iy * Iy).sum())) * scale)
        Dd = f32(f32(A11 * A22) - f32(A12 * A12))
        disc = f32(f32(f32(A11 - A22) * f32(A11 - A22)) + f32(f32(f32(4) * A12) * A12))
        min_eig = f32(f32(f32(A22 + A11) - np.sqrt(disc)) / f32(2 * W * W))
        if min_eig < f32(1e-3) or Dd < f32(1.1920929e-07):
            if lv == 0:
                st = 0
            continue
        Dd = f32(f32(1) / Dd)
        npx, npy = f32(nx - f32(7)), f32(ny - f32(7))
        pdx = pdy = f32(0)
        for j in range(30):
            jx, jy = int(np.floor(npx)), int(np.floor(npy))
            if jx < -W or jx >= J.shape[1] or jy < -W or jy >= J.shape[0]:
                if lv == 0:
                    st = 0
                break
            w = weights(f32(npx - f32(jx)), f32(npy - f32(jy)))
            diff = interp(win(J, jx, jy, "reflect"), w, 14 - 5) - Iw
            b1, b2 = f32(f32(int((diff * Ix).sum())) * scale), f32(f32(int((diff * Iy).sum())) * scale)
            dx = f32(f32(f32(A12 * b2) - f32(A22 * b1)) * Dd)
            dy = f32(f32(f32(A12 * b1) - f32(A11 * b2)) * Dd)
            npx, npy = f32(npx + dx), f32(npy + dy)
            nx, ny = f32(npx + f32(7)), f32(npy + f32(7))
            if float(dx) * float(dx) + float(dy) * float(dy) <= 1e-4:
                break
            if j > 0 and abs(float(f32(dx + pdx))) < 0.01 and abs(float(f32(dy + pdy))) < 0.01:
                nx, ny = f32(nx - f32(dx * f32(0.5))), f32(ny - f32(dy * f32(0.5)))
                break
            pdx, pdy = dx, dy
        if st and lv == 0:
            rx, ry = int(np.rint(f32(nx - f32(7)))), int(np.rint(f32(ny - f32(7))))
            if rx < -W or rx >= J.shape[1] or ry < -W or ry >= J.shape[0]:
                st = 0
    return float(nx), float(ny), st


def test_klt_against_an_independent_numpy_lucas_kanade():
    """pyramidal LK on single points, NumPy write-up of OpenCV's LKTrackerInvoker against orc_klt: to the last bit"""
    cfg = abi.config_named("B", enable_equalizer=0)
    seq = rv.synth.SynthSequence(cfg, duration=4.0)
    a, b = seq.render(60), seq.render(61)
    xy, vis = seq.project(60, noise=False)
    pts = np.concatenate([xy[vis][:10], np.array([[3.5, 200.2], [748.9, 10.1], [300.0, 478.5]], np.float32)]).astype(np.float32)

    def pyr(img):
        lv = [img]
        for _ in range(3):
            lv.append(O.pyr_down(lv[-1]))
        return lv
    Ip, Jp = pyr(a), pyr(b)          # (pyrDown and Scharr themselves are pinned by test_pyr_down_and_scharr_against_numpy)
    Dp = [O.scharr(l) for l in Ip]
    want, st = O.klt(a, b, pts)
    assert st.sum() >= 8
    for i, p in enumerate(pts):
        x, y, s = _lk_numpy(Ip, Dp, Jp, p)
        assert s == st[i], i
        assert np.float32(x) == want[i, 0] and np.float32(y) == want[i, 1], (i, x, y, want[i])


def test_triangulation_reaches_the_minimum_of_the_reprojection_cost():
    """U2 (Updater.cc:143-269): the inverse-depth LM estimate (phi, psi, rho) of a type-'1' feature (every observation enters both the
    triangulation and the residual) is a stationary point of the reprojection cost — Hf^T r = 0 to the noise floor of the float32
    residual — and scipy's trust-region least squares, started from a perturbed triple and using only the residual function, ends at the
    same point.  Independent of the oracle's own normal equations, damping schedule and stopping rule.  (Tracks: the hand-over of a
    simulated sequence, i.e. geometrically consistent observations with pixel noise.)"""
    from scipy.optimize import least_squares
    cfg = abi.config_named("B", enable_equalizer=0)
    seq, recs = S.record_sequence(cfg, n_frames=30)
    done = 0
    for rec in recs[-6:]:
        if rec.get("types") is None:
            continue
        x, types, lens, meas = rec["x1"], rec["types"], rec["lens"], rec["meas"]
        for f in range(len(types)):
            if types[f] != ord("1") or lens[f] < 5 or done >= 12:
                continue
            r0, _, Hf, pf = O.feature_model(cfg, x, types[f], meas[f], lens[f])
            g = Hf.T @ r0
            assert np.linalg.norm(g) <= 1e-4 * np.linalg.norm(Hf) * np.linalg.norm(r0), (f, g)
            if done < 4:
                fun = lambda p: O.feature_model(cfg, x, types[f], meas[f], lens[f], p)[0]            # noqa: E731
                sol = least_squares(fun, pf * np.array([1.02, 0.98, 1.3]) + np.array([1e-3, -1e-3, 0.0]), method="trf", diff_step=1e-5,
                                    xtol=1e-12, ftol=1e-12, gtol=1e-12)
                assert np.abs(sol.x[:2] - pf[:2]).max() < 2e-4 and abs(sol.x[2] - pf[2]) < 2e-3 * max(abs(pf[2]), 1e-3), (f, sol.x, pf)
                assert np.linalg.norm(fun(sol.x)) >= np.linalg.norm(r0) * (1 - 1e-4)           # scipy finds nothing better
            done += 1
    assert done >= 8
