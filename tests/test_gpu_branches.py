"""Reference branches the stock settings never take, each through the C-ABI against the oracle:

  * Tracker.UseSampson: 0 — the algebraic epipolar error |p2^T E p1| instead of the Sampson distance (Ransac.cc:261-266 instead of :250-258);
  * Camera.k3 present and non-zero — the five-coefficient radial-tangential model of cv::undistortPoints (Tracker.cc:56-61,116-119);
  * the number of RANSAC candidates around the model's 16 iterations (Ransac.cc:201-205): <= 16 returns 0 upstream and leaves the flags
    alone; 17..31 makes SetPointPair (Ransac.cc:50-83) look for 16 DISJOINT index pairs among fewer than 32 indices — it never returns.
    The device and the oracle treat < 32 as "too few" (deliberate deviation); 32 is the first count that runs.
"""
import numpy as np
import pytest

import oracle as O
import scenarios as S
from test_gpu_edges import _same_tracker

abi, rv = O.abi, O.rv
pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("thr", [1e-5, 2e-3])
def test_algebraic_error_ransac(gpu_required, thr):
    """Tracker.UseSampson: 0 on rendered frames (KLT outliers included): winner, inlier count, flags, track tables bit-exact; with the stock
    threshold (tuned for the Sampson distance) and with one at which the algebraic error separates inliers from outliers"""
    from rvio_amd import hip
    cfg = abi.config_named("B", enable_equalizer=0, use_sampson=0, inlier_thr=thr)
    seq = rv.synth.SynthSequence(cfg, duration=8.0)
    h, t = hip.RvioHip(cfg), O.Tracker(cfg)
    inl, lost = [], 0
    for k in range(60, 72):
        img, imu = seq.render(k), seq.imu_between(k)
        xy, vis = seq.project(k, noise=False)
        cand, _ = seq.candidates(k, xy, vis)
        oi = t.track(img, imu, cand)
        h.track(img, imu, cand)
        gi = h.frame_info()
        for key in ("n_tracked_in", "n_klt_ok", "n_ransac_inliers", "ransac_winner", "n_tracked_out", "n_feat_update"):
            assert gi[key] == oi[key], (k, key, gi, oi)
        _same_tracker(h, t, k)
        if k > 60:
            inl.append(gi["n_ransac_inliers"])
            lost += gi["n_klt_ok"] - gi["n_ransac_inliers"]
    h.close()
    assert max(inl) > 32 and lost > 0, (inl, lost)       # RANSAC ran and rejected something: the branch decided, not a pass-through


def test_algebraic_error_differs_from_sampson(gpu_required):
    """the two error forms pick different inlier sets on the same frames (so the test above cannot pass by running the Sampson branch)"""
    from rvio_amd import hip
    res = []
    for us in (0, 1):
        cfg = abi.config_named("B", enable_equalizer=0, use_sampson=us, inlier_thr=2e-4)
        seq = rv.synth.SynthSequence(cfg, duration=8.0)
        h = hip.RvioHip(cfg)
        counts = []
        for k in range(60, 68):
            xy, vis = seq.project(k, noise=False)
            cand, _ = seq.candidates(k, xy, vis)
            h.track(seq.render(k), seq.imu_between(k), cand)
            counts.append(h.frame_info()["n_ransac_inliers"])
        h.close()
        res.append(counts)
    assert res[0] != res[1], res


@pytest.mark.parametrize("k3", [0.05, -0.2])
def test_k3_radial_term(gpu_required, k3):
    """Camera.k3 != 0 (Tracker.cc:56-61): normalised coordinates bit-exact (no library call in the rad-tan iteration), tracker tables
    bit-exact, filter states within 1e-6 over a direct-track sequence; and different from the k3 = 0 camera"""
    from rvio_amd import hip
    cfg = abi.config_named("B", enable_equalizer=0, k3=k3)
    seq, recs = S.record_sequence(cfg, n_frames=24)
    h = hip.RvioHip(cfg)
    h.initialize(*seq.init_from_static(38))
    worst, moved = 0.0, 0.0
    cfg0 = abi.config_named("B", enable_equalizer=0)
    for r in recs:
        inp = r["inp"]
        h.frame_points(inp["tracked"], inp["status"], inp["imu"], inp["cand"])
        x, _ = h.get_state()
        worst = max(worst, S.state_delta(x, r["x3"]))
        pts, hl = h.get_points()
        assert np.array_equal(pts, r["pts"]) and np.array_equal(hl, r["hist_len"])
        if len(inp["tracked"]):
            _, un = h.debug_tracked(len(inp["tracked"]))
            assert np.array_equal(un, O.undistort(cfg, inp["tracked"]))
            moved = max(moved, float(np.max(np.abs(un - O.undistort(cfg0, inp["tracked"])))))
    h.close()
    assert recs[-1]["did_update"] and worst <= 1e-6, worst
    assert moved > 1e-4, moved                            # the k3 term reached the result


@pytest.mark.parametrize("n_cand", [16, 17, 25, 31, 32, 33])
def test_candidate_counts_around_the_models_iterations(gpu_required, n_cand):
    """n_cand tracked points, 20 % of them gross outliers: below 32 candidates RANSAC returns 0 and touches no flag (every point survives,
    outliers included); from 32 on it runs and the outliers go.  Device and oracle agree on every count, flag and list."""
    from rvio_amd import hip
    cfg = abi.config_named("B", enable_equalizer=0)
    seq = rv.synth.SynthSequence(cfg, duration=8.0)
    imu0 = np.zeros(0, abi.IMU_DTYPE)
    xy0, vis0 = seq.project(60, noise=False)
    xy1, vis1 = seq.project(61, noise=False)
    ids = np.flatnonzero(vis0 & vis1)[:n_cand]
    assert len(ids) == n_cand
    p0 = np.ascontiguousarray(xy0[ids], np.float32)
    p1 = np.ascontiguousarray(xy1[ids], np.float32).copy()
    bad = np.arange(n_cand) % 5 == 2
    p1[bad] += np.array([37.0, -23.0], np.float32)        # gross outliers to the epipolar geometry
    h, t = hip.RvioHip(cfg), O.Tracker(cfg)
    for tr in (h, t):
        tr.track_points(np.zeros((0, 2), np.float32), np.zeros(0, np.uint8), imu0, p0)           # first image: seed the slots with p0
        tr.track_points(p1, np.ones(n_cand, np.uint8), seq.imu_between(61), np.zeros((0, 2), np.float32))
    gi = h.frame_info()
    n_pts, _ = _same_tracker(h, t, n_cand)
    h.close()
    if n_cand < 32:
        assert gi["n_ransac_inliers"] == 0 and n_pts == n_cand, (gi, n_pts)                      # "too few": flags untouched, nothing dropped
    else:
        # (an offset that happens to lie along a point's epipolar line survives: "at most one of the outliers" is what the oracle finds here)
        assert n_cand - int(bad.sum()) <= gi["n_ransac_inliers"] < n_cand - int(bad.sum()) + 2 and n_pts == gi["n_ransac_inliers"], (gi, n_pts)
