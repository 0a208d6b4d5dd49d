"""GPU parity of the device detector (T7: goodFeaturesToTrack + cornerSubPix, FeatureDetector.cc:55-75) against the oracle:
min-eigenvalue map, selected corners and refined corners BIT-EXACT, for s = 1 (first image) and s = 2 (refill), and the
tracker / whole frame driven by the device detector (NULL corner list) against the oracle running its own detector."""
import numpy as np
import pytest

import oracle as O

abi, rv = O.abi, O.rv
pytestmark = pytest.mark.gpu
f32 = np.float32


def _imgs():
    rng = np.random.default_rng(5)
    cfg = abi.config_named("B", enable_equalizer=0)
    seq = rv.synth.SynthSequence(cfg, duration=4.0)
    yy, xx = np.mgrid[0:480, 0:752]
    checker = (((yy // 24) + (xx // 24)) % 2 * 150 + 50 + rng.integers(0, 6, (480, 752))).astype(np.uint8)
    return {"synth": (seq.render(50), seq.render(51)),
            "noise": (rng.integers(0, 256, (480, 752), dtype=np.uint8), rng.integers(0, 256, (480, 752), dtype=np.uint8)),
            "checker": (checker, np.roll(checker, 3, axis=1)),
            "flat": (np.full((480, 752), 77, np.uint8), np.full((480, 752), 78, np.uint8))}


@pytest.mark.parametrize("throughput", [0, 1], ids=["latency-forms", "throughput-forms"])
@pytest.mark.parametrize("name", ["synth", "noise", "checker", "flat"])
@pytest.mark.parametrize("eq", [0, 1])
def test_detector_bit_exact(gpu_required, name, eq, throughput):
    """(throughput = 1: the forms a batch handle launches — 4 pixels per thread, the map through HBM, four corners per wave — on this one stream)"""
    from rvio_amd import hip
    cfg = abi.config_named("B", enable_equalizer=eq)
    im0, im1 = _imgs()[name]
    h = hip.RvioHip(cfg)
    h.kernel_forms(throughput)
    imu = np.zeros(2, abi.IMU_DTYPE)
    imu["dt"] = 0.005
    for s, im in ((1, im0), (2, im1)):       # first image: s = 1; afterwards the refill factor 2 (unless nothing was found)
        h.track(im, imu, None)
        seen = O.clahe(im) if eq else im
        xy, raw, eig = h.get_corners(want_eig=True)
        want_eig = O.min_eig(seen)
        assert np.array_equal(eig, want_eig), (name, s, float(np.abs(eig - want_eig).max()))
        s_eff = 1 if (s == 2 and name == "flat") else s          # no corner found on the first image: it stays "the first image"
        want_raw = O.gftt(seen, cfg.n_features, float(f32(cfg.qual_lvl)), float(f32(s_eff) * f32(cfg.min_dist)))
        assert raw.shape == want_raw.shape and np.array_equal(raw, want_raw), (name, s, len(raw), len(want_raw))
        want = O.detect(cfg, seen, s_eff)
        assert np.array_equal(xy, want), (name, s, float(np.abs(xy - want).max()) if len(xy) else 0)
    h.close()


def test_detector_bit_exact_1080p(gpu_required):
    """cfg D (1920x1080, 800 features): more grid cells and candidates than the in-LDS fast path takes -> the general
    (global-memory) neighbour / selection path, s = 1 and s = 2"""
    from rvio_amd import hip
    cfg = abi.config_named("D", enable_equalizer=1)
    seq = rv.synth.SynthSequence(cfg, duration=3.0)
    h = hip.RvioHip(cfg)
    imu = np.zeros(2, abi.IMU_DTYPE)
    imu["dt"] = 0.005
    for s, k in ((1, 50), (2, 51)):
        im = seq.render(k)
        h.track(im, imu, None)
        seen = O.clahe(im)
        xy, raw, eig = h.get_corners(want_eig=True)
        assert np.array_equal(eig, O.min_eig(seen)), s
        want_raw = O.gftt(seen, cfg.n_features, float(f32(cfg.qual_lvl)), float(f32(s) * f32(cfg.min_dist)))
        assert raw.shape == want_raw.shape and np.array_equal(raw, want_raw), (s, len(raw), len(want_raw))
        assert np.array_equal(xy, O.detect(cfg, seen, s)), s
    h.close()


@pytest.mark.parametrize("throughput", [0, 1], ids=["latency-forms", "throughput-forms"])
def test_tracker_sequence_with_device_detector_bit_exact(gpu_required, throughput):
    from rvio_amd import hip
    cfg = abi.config_named("B", enable_equalizer=1)
    seq = rv.synth.SynthSequence(cfg, duration=8.0)
    h = hip.RvioHip(cfg)
    h.kernel_forms(throughput)
    t = O.Tracker(cfg)
    n_upd = 0
    for k in range(60, 72):
        img = seq.render(k)
        imu = seq.imu_between(k)
        oi = t.track(img, imu, None)
        h.track(img, imu, None)
        gi = h.frame_info()
        for key in ("n_tracked_in", "n_klt_ok", "n_ransac_inliers", "ransac_winner", "n_tracked_out", "n_feat_update"):
            assert gi[key] == oi[key], (k, key, gi, oi)
        pa, ha = h.get_points()
        pb, hb = t.get_points()
        assert np.array_equal(pa, pb) and np.array_equal(ha, hb), k
        ta, la, ma = h.get_tracks()
        tb, lb, mb = t.get_tracks()
        assert np.array_equal(ta, tb) and np.array_equal(la, lb), k
        n_upd += len(la)
    assert n_upd > 0
    h.close()


def test_long_image_sequence_tracks_the_literal_oracle(gpu_required):
    """130 free-running frames of the stock workload against the LITERAL oracle (sequential Givens QR + leading-row rank scan,
    Updater.cc:469-529): states within 1e-6 throughout, every counter equal, and in the frames where the reference's scan cuts
    informative rows off (90, 108, 110 of this sequence) the device reports the same nRank (tests/test_truncation.py)."""
    from rvio_amd import hip
    import scenarios as S
    cfg = abi.config_named("B", enable_equalizer=1)
    n = 130
    seq = rv.synth.SynthSequence(cfg, duration=(38 + n + 4) / 20.0)
    w, a, ni = seq.init_from_static(38)
    x0, P0 = O.initialize(cfg, w, a, ni)
    h = hip.RvioHip(cfg)
    h.initialize(w, a, ni)
    lit = O.System(cfg)
    lit.set_state(x0, P0)
    worst, cuts = 0.0, []
    for k in range(39, 39 + n):
        img, imu = seq.render(k), seq.imu_between(k)
        oi = lit.frame(imu, None, img=img)[0]
        h.frame(img, imu, None)
        h.sync()
        gi = h.frame_info()
        for key in ("n_tracked_in", "n_klt_ok", "n_ransac_inliers", "n_feat_update", "n_feat_accepted", "n_rows", "updated"):
            assert gi[key] == oi[key], (k, key)
        assert gi["device_error"] == 0, (k, gi["device_error"])    # no singular pivot, no dropped track, no device-side counter timeout
        xa, Pa = h.get_state()
        xl, Pl = lit.get_state()
        worst = max(worst, S.state_delta(xa, xl))
        assert worst <= 1e-6, (k, worst)
        assert np.max(np.abs(Pa - Pl)) <= 1e-9 * max(1.0, np.max(np.abs(Pl))), k
        if gi["rank_truncated_at"] >= 0:
            assert gi["rank_truncated_at"] == lit.last_rank(), k
            cuts.append(k)
    h.close()
    assert len(cuts) >= 3, cuts


def test_whole_frame_with_device_detector(gpu_required):
    """rvio_hip_frame with no corner list: CLAHE + detector + KLT + RANSAC + filter, all on the device, vs the oracle's
    System::MonoVIO body running its own detector; states within 1e-6."""
    from rvio_amd import hip
    import scenarios as S
    cfg = abi.config_named("B", enable_equalizer=1)
    seq = rv.synth.SynthSequence(cfg, duration=8.0)
    w, a, n = seq.init_from_static(38)
    h = hip.RvioHip(cfg)
    h.initialize(w, a, n)
    s = O.System(cfg)
    x0, P0 = O.initialize(cfg, w, a, n)
    s.set_state(x0, P0)
    for k in range(39, 39 + 16):
        img, imu = seq.render(k), seq.imu_between(k)
        s.frame(imu, None, img=img)
        h.frame(img, imu, None)
    h.sync()
    xa, _ = h.get_state()
    xb, _ = s.get_state()
    info = h.frame_info()
    h.close()
    assert info["updated"] == 1 and info["device_error"] == 0
    assert S.state_delta(xa, xb) <= 1e-6


@pytest.mark.parametrize("min_dist", [6.0, 10.0, 12.5, 20.0, 31.0, 32.0, 45.0, 64.5, 100.0])
def test_detector_other_subpix_windows(gpu_required, min_dist):
    """cornerSubPix's window is floor(nMinDist / 2) (FeatureDetector.cc:68): 3, 5, 6 (the 16 x 16 summation grid), 10 and 15 (32 x 32), 16, 22
    (64 x 64: subpix_wide_kernel), 32 and 50 (128 x 128) beside the stock 7 — goodFeaturesToTrack corners and refined corners bit-exact for s = 1 and s = 2, and a short free-running whole-frame run."""
    from rvio_amd import hip
    import scenarios as S
    cfg = abi.config_named("B", enable_equalizer=1, min_dist=min_dist)
    seq = rv.synth.SynthSequence(cfg, duration=4.0)
    h = hip.RvioHip(cfg)
    imu = np.zeros(2, abi.IMU_DTYPE)
    imu["dt"] = 0.005
    for s, k in ((1, 50), (2, 51)):
        im = seq.render(k)
        h.track(im, imu, None)
        seen = O.clahe(im)
        xy, raw = h.get_corners()
        want_raw = O.gftt(seen, cfg.n_features, float(f32(cfg.qual_lvl)), float(f32(s) * f32(cfg.min_dist)))
        assert raw.shape == want_raw.shape and np.array_equal(raw, want_raw), (min_dist, s)
        want = O.detect(cfg, seen, s)
        assert len(xy) > (20 if min_dist < 32 else 3) and np.array_equal(xy, want), (min_dist, s, float(np.abs(xy - want).max()))
        assert np.any(xy != raw)                                   # the refinement did move corners
    h.close()
    w, a, n = seq.init_from_static(38)
    h = hip.RvioHip(cfg)
    h.initialize(w, a, n)
    lit = O.System(cfg)
    lit.set_state(*O.initialize(cfg, w, a, n))
    for k in range(39, 39 + 16):
        img, imu = seq.render(k), seq.imu_between(k)
        lit.frame(imu, None, img=img)
        h.frame(img, imu, None)
    h.sync()
    assert np.array_equal(h.get_points()[0], lit.tracker().get_points()[0])
    assert S.state_delta(h.get_state()[0], lit.get_state()[0]) <= 1e-6
    h.close()


def test_detector_refuses_what_it_cannot_do(gpu_required):
    from rvio_amd import hip
    cfg = abi.config_named("B", min_dist=128.0)
    h = hip.RvioHip(cfg)
    imu = np.zeros(2, abi.IMU_DTYPE)
    with pytest.raises(hip.RvioHipError, match="half-windows"):
        h.track(np.zeros((480, 752), np.uint8), imu, None)
    h.close()


@pytest.mark.parametrize("throughput", [0, 1], ids=["latency-forms", "throughput-forms"])
@pytest.mark.parametrize("W,H,eq", [(200, 136, 0), (757, 483, 1), (1000, 562, 0)], ids=["200x136", "757x483-clahe", "1000x562"])
def test_image_sizes_that_fit_no_tile(gpu_required, W, H, eq, throughput):
    """widths / heights that are no multiple of anything the kernels tile by (the strip detector's 60 x 32 strips, 4-pixel words, the 64-lane
    tiles) on noise images — candidates up against every border: corners,
    refined corners and the KLT result (features near the borders: reflected staging, early-outs) bit-exact, both kernel families"""
    from rvio_amd import hip
    cfg = abi.config_named("B", width=W, height=H, enable_equalizer=eq)
    rng = np.random.default_rng(W * 1000 + H)
    base = rng.integers(0, 256, (H + 8, W + 8), dtype=np.uint8)
    k = np.ones(3) / 3.0                                     # a little smoothing: corners that KLT can hold on to
    sm = np.apply_along_axis(lambda v: np.convolve(v, k, mode="same"), 0, base.astype(np.float64))
    sm = np.apply_along_axis(lambda v: np.convolve(v, k, mode="same"), 1, sm)
    sm = np.clip((sm - sm.mean()) * 3.0 + 128.0, 0, 255).astype(np.uint8)
    im0 = np.ascontiguousarray(sm[4:4 + H, 4:4 + W])
    im1 = np.ascontiguousarray(sm[3:3 + H, 2:2 + W])         # the scene moved by (+2, +1) px
    h = hip.RvioHip(cfg)
    h.kernel_forms(throughput)
    imu = np.zeros(2, abi.IMU_DTYPE)
    imu["dt"] = 0.005
    seen = []
    for s, im in ((1, im0), (2, im1)):
        h.track(im, imu, None)
        if s == 2:
            got, _ = h.debug_tracked(len(pts0))
        sn = O.clahe(im) if eq else im
        seen.append(sn)
        xy, raw = h.get_corners()
        want_raw = O.gftt(sn, cfg.n_features, float(f32(cfg.qual_lvl)), float(f32(s) * f32(cfg.min_dist)))
        assert raw.shape == want_raw.shape and np.array_equal(raw, want_raw), (s, len(raw), len(want_raw))
        assert np.array_equal(xy, O.detect(cfg, sn, s)), s
        if s == 1:
            pts0, _ = h.get_points()
            assert len(pts0) >= min(20, cfg.n_features) and np.array_equal(pts0, xy)
    want, st = O.klt(seen[0], seen[1], pts0)
    assert st.sum() >= len(pts0) // 2, (int(st.sum()), len(pts0))
    assert np.array_equal(got, want)
    assert h.frame_info()["n_klt_ok"] == int(st.sum())
    h.close()
