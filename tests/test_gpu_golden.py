"""GPU parity against the COMMITTED golden fixtures — no oracle call at test time, the files carry the inputs and the expected outputs.
Two families of expected outputs on the same stored inputs: "oracle" (tests/golden/*.npz, written by make_golden.py from the CPU
oracle) and "reference" (tests/golden/ref_*.npz, written by make_golden_ref.py from the reference's OWN compiled sources,
oracle/_ref/libref.so — numbers no code of this repository's oracle produced; tests/test_golden_ref.py holds the two families
against each other on the CPU)."""
import os
import zlib

import numpy as np
import pytest

import oracle as O          # only for the shared config / dtype helpers (O.abi); no oracle function is called here
import scenarios as S

abi = O.abi
pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _fixture(name, family):
    """the stored inputs + the expected outputs of one family (the reference-written file carries outputs only)"""
    g = dict(np.load(os.path.join(GOLD, name)))
    if family == "reference":
        g.update(dict(np.load(os.path.join(GOLD, "ref_" + name))))
    return g


@pytest.mark.parametrize("family", ["oracle", "reference"])
def test_filter_stages_against_the_golden_snapshot(gpu_required, family):
    """frame 30 of the direct-track sequence, cfg B: propagate -> update -> augment/compose from the stored (x0, P0, IMU, tracks)"""
    from rvio_amd import hip
    g = _fixture("cfgB_direct_seed0_frame30.npz", family)
    cfg = abi.config_named("B", enable_equalizer=0)
    h = hip.RvioHip(cfg)
    h.set_state(g["x0"], g["P0"])
    h.propagate(g["imu"].view(abi.IMU_DTYPE))
    x, P = h.get_state()
    assert S.state_delta(x, g["x1"]) <= 1e-9 and np.max(np.abs(P - g["P1"])) <= 1e-9 * np.max(np.abs(g["P1"]))
    h.update(g["types"], g["lens"], g["meas"])
    x, P = h.get_state()
    assert S.state_delta(x, g["x2"]) <= 1e-9 and np.max(np.abs(P - g["P2"])) <= 1e-9 * np.max(np.abs(g["P2"]))
    diag = h.update_diag()
    if family == "reference":   # the reference reports its accepted set only as the size of the landmark cloud it publishes (Updater.cc:430-448)
        assert int(np.count_nonzero(diag["accepted"])) == int(g["n_cloud"])
    else:
        assert np.array_equal(diag["accepted"], g["accepted"])                  # the chi-square gate took the same decisions
        assert np.allclose(diag["gamma"], g["gamma"], rtol=1e-7, atol=1e-9)
    h.augment_compose(bool(g["do_augment"]))
    x, P = h.get_state()
    assert S.state_delta(x, g["x3"]) <= 1e-9 and np.max(np.abs(P - g["P3"])) <= 1e-9 * np.max(np.abs(g["P3"]))
    h.close()


@pytest.mark.parametrize("family", ["oracle", "reference"])
def test_tracker_against_the_golden_image_fixture(gpu_required, family):
    """4 frames of the half-size camera through CLAHE, the device detector, KLT, RANSAC and book-keeping: bit-exact feature lists
    (family "reference": the tables RVIO::Tracker itself — Tracker.cc, FeatureDetector.cc, Ransac.cc compiled unmodified — ended with)"""
    from rvio_amd import hip
    g = _fixture("small_images_tracker.npz", family)
    cfg = S.small_image_config()
    h = hip.RvioHip(cfg)
    for i in range(4):
        h.track(g["imgs"][i], g["imu%d" % i].view(abi.IMU_DTYPE), None)
        if i == 0:
            xy, raw = h.get_corners()
            assert np.array_equal(xy, g["corners0"])
            eq, _ = h.debug_pyramid(0)                       # level 0 of the pyramid = the equalised image
            assert zlib.crc32(np.ascontiguousarray(eq).tobytes()) == int(g["clahe0_crc"])
        pts, hl = h.get_points()
        assert np.array_equal(pts, g["pts%d" % i]) and np.array_equal(hl, g["hist%d" % i]), i
    h.close()


@pytest.mark.parametrize("name", ["A", "B", "C", "E"])
def test_full_load_update_against_the_reference_written_digest(gpu_required, name):
    """the worst-case update load at the 14- / 10- / 20- / 30-clone windows — every form of the solve (all-LDS, one workgroup,
    split) and of the Joseph stage — from stored inputs, against what the reference's own Updater::update left for them
    (tests/golden/ref_full_load_outputs.npz: state, diag P, P V on fixed probe vectors, the size of the accepted set)"""
    import sys
    from rvio_amd import hip
    sys.path.insert(0, GOLD)
    import golden_io as M
    g, r = np.load(os.path.join(GOLD, "full_load_inputs.npz")), np.load(os.path.join(GOLD, "ref_full_load_outputs.npz"))
    cfg, x1, P1, types, lens, meas = M.load_full_load_case(g, name)
    h = hip.RvioHip(cfg)
    h.set_state(x1, P1)
    h.update(types, lens, meas)
    x2, P2 = h.get_state()
    diag = h.update_diag()
    h.close()
    scale = float(r[name + "_maxP2"])
    assert S.state_delta(x2, r[name + "_x2"]) <= 1e-9
    assert np.max(np.abs(np.diag(P2) - r[name + "_diagP2"])) <= 1e-9 * scale
    assert np.max(np.abs(P2 @ M.probes(P2.shape[0]) - r[name + "_P2V"])) <= 1e-8 * scale
    assert int(np.count_nonzero(diag["accepted"])) == int(r[name + "_n_cloud"])


def test_free_run_replays_the_reference_states(gpu_required):
    """30 free-running frames from System::initialize (direct-track mode: window filling, the first type-'2' features, the window sliding)
    on the inputs recorded while the reference's OWN System::MonoVIO ran them (tests/golden/ref_free_run_30_frames.npz): the device must
    end every frame where the reference did — state <= 1e-6 (observed ~1e-12), the accepted count of every update, the reference's feature
    tables bit for bit, the final covariance's digest."""
    import sys
    from rvio_amd import hip
    sys.path.insert(0, GOLD)
    import golden_io as M
    g = np.load(os.path.join(GOLD, "ref_free_run_30_frames.npz"))
    cfg = abi.config_named("B", enable_equalizer=0)
    h = hip.RvioHip(cfg)
    h.initialize(g["init_w"], g["init_a"], int(g["init_n"]))
    x, P = h.get_state()
    assert S.state_delta(x, g["x0"]) <= 1e-12 and np.max(np.abs(P - g["P0"])) <= 1e-12 * np.max(np.abs(g["P0"]))   # System::initialize: the reference's own x0, P0
    worst, n_upd = 0.0, 0
    for i in range(len(g["ref_xlen"])):
        h.frame_points(g["tracked%d" % i], g["status%d" % i], g["imu%d" % i].view(abi.IMU_DTYPE), g["cand%d" % i])
        x, P = h.get_state()
        info = h.frame_info()
        assert len(x) == int(g["ref_xlen"][i]), i
        worst = max(worst, S.state_delta(x, g["ref_x"][i, : len(x)]))
        if info["updated"]:
            assert info["n_feat_accepted"] == int(g["ref_n_cloud"][i]), i
            n_upd += 1
        if ("pts%d" % i) in g.files:
            pts, hl = h.get_points()
            assert np.array_equal(pts, g["pts%d" % i]) and np.array_equal(hl, g["hist%d" % i]), i
    h.close()
    scale = float(g["ref_maxP"])
    assert n_upd >= 20 and worst <= 1e-6, (n_upd, worst)
    assert np.max(np.abs(np.diag(P) - g["ref_diagP"])) <= 1e-6 * scale
    assert np.max(np.abs(P @ M.probes(P.shape[0]) - g["ref_PV"])) <= 1e-6 * scale
