"""The two families of golden fixtures agree: tests/golden/ref_*.npz (expected outputs computed by the reference's OWN compiled sources,
tests/golden/make_golden_ref.py) against tests/golden/*.npz (the oracle's, make_golden.py) on the same stored inputs.

This runs anywhere (the files are committed).  Where /root/reference exists it also recomputes the reference's outputs and holds the
committed files against them, so a stale fixture cannot survive.  The GPU suite compares the HIP path with BOTH families
(tests/test_gpu_golden.py): against the reference-written one the device is held to numbers no code of this repository's oracle produced."""
import os
import sys

import numpy as np
import pytest

import oracle as O
import scenarios as S

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _same(fresh, stored, what):
    """the same program on the same inputs: integers and float32 tables identical; doubles to 1e-12 of the array's scale (another host's libm
    may round a sine differently in the last bit — the container this runs in is not always the one that wrote the files)"""
    assert fresh.shape == stored.shape, what
    if fresh.dtype.kind == "f" and fresh.dtype.itemsize == 8:
        assert np.max(np.abs(fresh - stored), initial=0.0) <= 1e-12 * max(1e-300, np.max(np.abs(stored), initial=0.0)), what
    else:
        assert np.array_equal(fresh, stored), what


def _rel(a, b):
    return float(np.max(np.abs(a - b)) / max(1e-300, np.max(np.abs(b))))


def test_filter_snapshot_reference_vs_oracle():
    g, r = np.load(os.path.join(GOLD, "cfgB_direct_seed0_frame30.npz")), np.load(os.path.join(GOLD, "ref_cfgB_direct_seed0_frame30.npz"))
    for k in ("1", "2", "3"):
        assert S.state_delta(r["x" + k], g["x" + k]) <= 1e-12, k
        assert _rel(r["P" + k], g["P" + k]) <= 1e-12, k
    assert int(r["updated"]) == 1
    assert int(r["n_cloud"]) == int(np.count_nonzero(g["accepted"]))                       # the chi-square gate: same number of accepted features
    assert int(r["gate_rejects"]) + int(r["invalid"]) == len(g["types"]) - int(r["n_cloud"])


def test_tracker_tables_reference_vs_oracle():
    g, r = np.load(os.path.join(GOLD, "small_images_tracker.npz")), np.load(os.path.join(GOLD, "ref_small_images_tracker.npz"))
    for i in range(4):
        assert np.array_equal(r["pts%d" % i], g["pts%d" % i]) and np.array_equal(r["hist%d" % i], g["hist%d" % i]), i   # bit-identical tables
    assert sum(len(r["types%d" % i]) for i in range(4)) >= 0


def test_committed_reference_fixtures_are_current():
    try:
        import ref as R
        have = R.available()
    except Exception as e:   # a broken build must fail loudly where the sources exist
        if "failed to build" in str(e):
            raise
        have = False
    if not have:
        pytest.skip("oracle/_ref/libref.so needs the reference's sources (/root/reference)")
    sys.path.insert(0, GOLD)
    import make_golden_ref as M
    for name, fresh in (("ref_cfgB_direct_seed0_frame30.npz", M.filter_outputs()), ("ref_small_images_tracker.npz", M.tracker_outputs())):
        stored = np.load(os.path.join(GOLD, name))
        assert set(stored.files) == set(fresh), name
        for k, v in fresh.items():
            _same(np.asarray(v), stored[k], (name, k))


def _golden_ref_module():
    sys.path.insert(0, GOLD)
    import make_golden_ref as M
    return M


def _golden_io():
    sys.path.insert(0, GOLD)
    import golden_io as M
    return M


@pytest.mark.parametrize("name", ["A", "B", "C", "E"])
def test_full_load_digest_reference_vs_oracle(name):
    """the worst-case update loads at the 14- / 10- / 20- / 30-clone windows: the oracle's update on the stored inputs against
    the digest the reference's own Updater::update left (state, diag P, P V on fixed probe vectors, size of the accepted set)"""
    M = _golden_io()
    g, r = np.load(os.path.join(GOLD, "full_load_inputs.npz")), np.load(os.path.join(GOLD, "ref_full_load_outputs.npz"))
    cfg, x1, P1, types, lens, meas = M.load_full_load_case(g, name)
    assert np.array_equal(P1, P1.T) and len(types) == len(lens) == len(meas)
    x2, P2, dg = O.update(cfg, x1, P1, types, lens, meas)
    assert dg["updated"] and dg["n_rows"] > 6 * (cfg.max_track_len - 1)                   # a tall stack: compression and rank scan ran
    scale = float(r[name + "_maxP2"])
    assert S.state_delta(x2, r[name + "_x2"]) <= 1e-11
    assert np.max(np.abs(np.diag(P2) - r[name + "_diagP2"])) <= 1e-11 * scale
    assert np.max(np.abs(P2 @ M.probes(P2.shape[0]) - r[name + "_P2V"])) <= 1e-10 * scale
    assert dg["n_good"] == int(r[name + "_n_cloud"])


def test_committed_full_load_digest_is_current():
    try:
        import ref as R
        have = R.available()
    except Exception as e:
        if "failed to build" in str(e):
            raise
        have = False
    if not have:
        pytest.skip("oracle/_ref/libref.so needs the reference's sources (/root/reference)")
    M = _golden_ref_module()
    fresh, stored = M.full_load_outputs(np.load(os.path.join(GOLD, "full_load_inputs.npz"))), np.load(os.path.join(GOLD, "ref_full_load_outputs.npz"))
    assert set(stored.files) == set(fresh)
    for k, v in fresh.items():
        _same(np.asarray(v), stored[k], k)


def _replay_free_run(make_system, per_frame):
    """replays tests/golden/ref_free_run_30_frames.npz through `make_system(cfg, g)` -> object with frame(i, tracked, status, imu, cand),
    state() and points(); returns the worst state delta against the reference's per-frame states"""
    M = _golden_io()
    g = np.load(os.path.join(GOLD, "ref_free_run_30_frames.npz"))
    cfg = O.abi.config_named("B", enable_equalizer=0)
    s = make_system(cfg, g)
    n, worst = len(g["ref_xlen"]), 0.0
    for i in range(n):
        s.frame(g["tracked%d" % i], g["status%d" % i], g["imu%d" % i].view(O.abi.IMU_DTYPE), g["cand%d" % i])
        x, P = s.state()
        assert len(x) == int(g["ref_xlen"][i]), i                                         # the window grew / slid when the reference's did
        worst = max(worst, S.state_delta(x, g["ref_x"][i, : len(x)]))
        if ("pts%d" % i) in g.files:
            pts, hl = s.points()
            assert np.array_equal(pts, g["pts%d" % i]) and np.array_equal(hl, g["hist%d" % i]), i   # the reference's own feature table
        per_frame(s, i, g)
    scale = float(g["ref_maxP"])
    assert np.max(np.abs(np.diag(P) - g["ref_diagP"])) <= 1e-6 * scale
    assert np.max(np.abs(P @ M.probes(P.shape[0]) - g["ref_PV"])) <= 1e-6 * scale
    return worst


def test_free_run_oracle_replays_the_reference_states():
    """the oracle's System fed the recorded inputs of the reference's free run: every frame's state, the feature tables, the final covariance"""
    class Sys:
        def __init__(self, cfg, g):
            self.s = O.System(cfg)
            self.s.set_state(g["x0"], g["P0"])
        def frame(self, tracked, status, imu, cand):
            self.info = self.s.frame(imu, cand, tracked=tracked, status=status)[0]
        def state(self):
            return self.s.get_state()
        def points(self):
            return self.s.tracker().get_points()

    def per_frame(s, i, g):
        if s.info["updated"]:   # (the reference reports an update only through the landmark cloud it publishes: its size is the accepted set)
            assert int(s.info["n_feat_accepted"]) == int(g["ref_n_cloud"][i]), i
            counted[0] += 1
    counted = [0]
    assert _replay_free_run(Sys, per_frame) <= 1e-11
    assert counted[0] >= 20
