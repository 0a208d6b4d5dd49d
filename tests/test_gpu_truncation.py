"""The device's structural form of the reference's rank truncation (Updater.cc:516-529; filter_kernels.hip trunc_finish) on the motions it
was NOT derived for, free-running on the GPU against the LITERAL oracle (sequential Givens QR + leading-row scan): a platform at rest,
pure rotation about the camera centre, a constant-velocity straight line, a scene of one common depth — on rendered images (stock
configuration: CLAHE + device detector) and on direct tracks with many lost features.  tests/test_truncation.py holds the CPU mirror
of the same rule against the same literal code on the same sequences."""
import numpy as np
import pytest

import oracle as O
import scenarios as S

abi, rv = O.abi, O.rv
pytestmark = pytest.mark.gpu

# Free-running bar: 1e-6 per state (north star).  The platform AT REST is the exception and gets 1.5e-5 (2 x the largest device figure
# measured, 7.1e-6 on images / 2.9e-6 on direct tracks): there position and velocity are
# unobservable, the sequence itself amplifies any difference ~1e7..1e9-fold within 80-100 frames (the LITERAL oracle started ONE ULP away
# from itself ends 1e-9 .. 2e-7 away; 8e-14 on the stock motion: tests/test_truncation.py::test_the_reference_itself_is_ill_conditioned_at_rest),
# and the device's per-update difference of ~1e-14 grows to a few 1e-6 in the integrated position.  What the device owes — and what the
# structural truncation rule owes — is agreement PER UPDATE: both tests below run a second handle that is re-seeded with the literal
# state before every frame and hold it to 1e-9 (1e-8 at rest; measured: 3e-14 typical, 1.5e-9 worst at rest).
# Round 5 (tests/test_ref_pins.py::test_monovio_at_rest_direct): the reference's OWN sources, compiled, against their restatement — two
# programs that differ only in the order of a few sums — end 3.4e-7 (state) / 2e-5 (P, relative) apart on the stationary sequence, growing
# ~3x per frame from 3e-16.  A 1e-6 free-running bar at rest is not a property of any implementation of this filter.
def bar(kw):
    return 1.5e-5 if kw.get("motion") == "stationary" else 1e-6


def bar1(kw):
    """ONE frame from the literal state: 1e-9, the stage-parity bar; 1e-8 at rest (zero parallax: the worst-conditioned T = s2 I + A Pcc; measured 1.5e-9)"""
    return 1e-8 if kw.get("motion") == "stationary" else 1e-9


MOTIONS = [dict(motion="stationary"), dict(motion="rotation"), dict(motion="line"), dict(scene="sphere")]
IDS = ["stationary", "rotation", "line", "sphere"]
COUNTERS = ("n_tracked_in", "n_klt_ok", "n_ransac_inliers", "n_feat_update", "n_feat_accepted", "n_rows", "updated")


@pytest.mark.parametrize("kw", MOTIONS, ids=IDS)
def test_degenerate_motion_on_images_tracks_the_literal_oracle(gpu_required, kw):
    from rvio_amd import hip
    cfg = abi.config_named("B", enable_equalizer=1)
    n = 80
    seq = rv.synth.SynthSequence(cfg, duration=(38 + n + 4) / 20.0, seed=2, **kw)
    w, a, ni = seq.init_from_static(38)
    x0, P0 = O.initialize(cfg, w, a, ni)
    h = hip.RvioHip(cfg)
    h.initialize(w, a, ni)
    h1 = hip.RvioHip(cfg)          # re-seeded with the literal state before every frame (the tracker does not read the filter state)
    h1.initialize(w, a, ni)
    lit = O.System(cfg)
    lit.set_state(x0, P0)
    worst, worst1, updates, early = 0.0, 0.0, 0, 0
    for k in range(39, 39 + n):
        img, imu = seq.render(k), seq.imu_between(k)
        if k > 39:
            h1.set_state(*lit.get_state())
        oi = lit.frame(imu, None, img=img)[0]
        h.frame(img, imu, None)
        h.sync()
        h1.frame(img, imu, None)
        worst1 = max(worst1, S.state_delta(h1.get_state()[0], lit.get_state()[0]))
        assert worst1 <= bar1(kw), (k, worst1)
        gi = h.frame_info()
        for key in COUNTERS:
            assert gi[key] == oi[key], (k, key, gi[key], oi[key])
        assert gi["device_error"] == 0, (k, gi["device_error"])
        xa, Pa = h.get_state()
        xl, Pl = lit.get_state()
        worst = max(worst, S.state_delta(xa, xl))
        assert worst <= bar(kw), (k, worst)
        # (covariance: 2e-5 of its largest entry, 1e-3 at rest — zero-parallax windows are the badly conditioned ones; the stock motion holds
        # 1e-6, tests/test_gpu_detector.py)
        cov_bar = 2e-3 if kw.get("motion") == "stationary" else 2e-5       # (at rest the unobservable block of P itself grows frame by frame)
        assert np.max(np.abs(Pa - Pl)) <= cov_bar * np.max(np.abs(Pl)), (k, float(np.max(np.abs(Pa - Pl))), float(np.max(np.abs(Pl))))
        if gi["updated"]:
            updates += 1
            c6 = 6 * min(k - 39, cfg.max_track_len - 1)
            early += 0 <= lit.last_rank() < min(gi["n_rows"], c6)
        if gi["rank_truncated_at"] >= 0:
            assert gi["rank_truncated_at"] == lit.last_rank(), k
    h.close()
    h1.close()
    print("MEASURED images %s: free-running max state delta %.3e (bar %.0e), one-update max %.3e" % (kw, worst, bar(kw), worst1))
    assert updates >= 15 and early >= updates // 2, (updates, early)     # the literal scan does stop early in these windows


@pytest.mark.parametrize("kw", MOTIONS, ids=IDS)
def test_degenerate_motion_on_direct_tracks(gpu_required, kw):
    """direct-track mode with 15 % random drops: a dozen type-'1' features of every length per frame beside the type-'2' ones"""
    from rvio_amd import hip
    cfg = abi.config_named("B", enable_equalizer=0)
    n = 100
    seq = rv.synth.SynthSequence(cfg, duration=(38 + n + 4) / 20.0, seed=4, drop_prob=0.15, **kw)
    w, a, ni = seq.init_from_static(38)
    x0, P0 = O.initialize(cfg, w, a, ni)
    h = hip.RvioHip(cfg)
    h.initialize(w, a, ni)
    h1 = hip.RvioHip(cfg)          # re-seeded with the literal state before every frame: ONE update's worth of difference
    h1.initialize(w, a, ni)
    lit = O.System(cfg)
    lit.set_state(x0, P0)
    drv = rv.synth.DirectTrackDriver(seq)
    worst, worst1, updates = 0.0, 0.0, 0
    for k in range(39, 39 + n):
        inp = drv.inputs(k)
        h1.set_state(*lit.get_state())
        oi = lit.frame(inp["imu"], inp["cand"], tracked=inp["tracked"], status=inp["status"])[0]
        h.frame_points(inp["tracked"], inp["status"], inp["imu"], inp["cand"])
        h1.frame_points(inp["tracked"], inp["status"], inp["imu"], inp["cand"])
        worst1 = max(worst1, S.state_delta(h1.get_state()[0], lit.get_state()[0]))
        assert worst1 <= bar1(kw), (k, worst1)
        pts = h.get_points()[0]
        assert np.array_equal(pts, lit.tracker().get_points()[0]), k
        drv.after(pts)
        gi = h.frame_info()
        for key in ("n_feat_update", "n_feat_accepted", "n_rows", "updated"):
            assert gi[key] == oi[key], (k, key, gi[key], oi[key])
        xa, _ = h.get_state()
        xl, _ = lit.get_state()
        worst = max(worst, S.state_delta(xa, xl))
        assert worst <= bar(kw), (k, worst)
        updates += gi["updated"]
        if gi["rank_truncated_at"] >= 0:
            assert gi["rank_truncated_at"] == lit.last_rank(), k
    h.close()
    h1.close()
    print("MEASURED direct tracks %s: free-running max state delta %.3e (bar %.0e), one-update max %.3e" % (kw, worst, bar(kw), worst1))
    assert updates >= 80


@pytest.mark.parametrize("name,scale", [("B", 0.1), ("A", 0.1), ("A", 0.03), ("C", 0.1)])
def test_information_form_under_small_image_noise_and_long_windows(gpu_required, name, scale):
    """The device forms A = Hw^T Hw (condition number squared) and inverts T = s2 I + A Pcc.  Stressed where that hurts most: the image
    noise sigma_im 10x / 30x smaller than EuRoC's (s2 100x / 1000x smaller against the same A), on the 14- and 20-clone windows, direct
    tracks with 10 % drops — free-running against the LITERAL oracle (QR compression, S = H P H^T + R in measurement space), 70 frames."""
    from rvio_amd import hip
    cfg = abi.config_named(name, enable_equalizer=0, sigma_px=0.002180293 * scale, sigma_py=0.002186767 * scale)
    n = 70
    seq = rv.synth.SynthSequence(cfg, duration=(38 + n + 4) / 20.0, seed=6, drop_prob=0.1)
    w, a, ni = seq.init_from_static(38)
    h = hip.RvioHip(cfg)
    h.initialize(w, a, ni)
    lit = O.System(cfg)
    lit.set_state(*O.initialize(cfg, w, a, ni))
    drv = rv.synth.DirectTrackDriver(seq)
    worst, accepted = 0.0, 0
    for k in range(39, 39 + n):
        inp = drv.inputs(k)
        oi = lit.frame(inp["imu"], inp["cand"], tracked=inp["tracked"], status=inp["status"])[0]
        h.frame_points(inp["tracked"], inp["status"], inp["imu"], inp["cand"])
        pts = h.get_points()[0]
        drv.after(pts)
        gi = h.frame_info()
        for key in ("n_feat_update", "n_feat_accepted", "n_rows", "updated"):
            assert gi[key] == oi[key], (k, key, gi[key], oi[key])
        assert gi["device_error"] == 0, (k, gi["device_error"])
        worst = max(worst, S.state_delta(h.get_state()[0], lit.get_state()[0]))
        assert worst <= 1e-6, (k, worst)
        accepted += gi["n_feat_accepted"]
    h.close()
    assert accepted > 500
