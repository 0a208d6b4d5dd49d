"""GPU parity at the edges of the path: images without a single corner (start-up and mid-sequence), an empty IMU batch, a row-strided
image, an empty caller-side corner list.  The reference handles all of these with silent early-outs; the device must
take the same ones: tracker tables bit-exact, filter states within 1e-6 of the oracle."""
import numpy as np
import pytest

import oracle as O
import scenarios as S

abi, rv = O.abi, O.rv
pytestmark = pytest.mark.gpu


def _same_tracker(h, t, tag):
    pa, ha = h.get_points()
    pb, hb = t.get_points()
    assert np.array_equal(pa, pb) and np.array_equal(ha, hb), tag
    ta, la, ma = h.get_tracks()
    tb, lb, mb = t.get_tracks()
    assert np.array_equal(ta, tb) and np.array_equal(la, lb), tag
    for f in range(len(la)):
        assert np.array_equal(ma[f, : la[f]], mb[f, : lb[f]]), (tag, f)
    return len(pa), len(la)


@pytest.fixture(scope="module")
def seqB():
    cfg = abi.config_named("B")                          # stock settings: CLAHE on
    seq = rv.synth.SynthSequence(cfg, duration=8.0)
    ks = list(range(60, 72))
    return cfg, seq, ks, [seq.render(k) for k in ks]


@pytest.mark.parametrize("blank_at", [(0, 1), (5,), (4, 5, 6)])
def test_cornerless_images(gpu_required, seqB, blank_at):
    """uniform images: the detector finds nothing, KLT loses every point, RANSAC sees no pair, the update is skipped — at the first
    image (the tracker must start later) and mid-sequence (every track ends at once, then the list refills)"""
    from rvio_amd import hip
    cfg, seq, ks, imgs = seqB
    w, a, n = seq.init_from_static(38)
    h = hip.RvioHip(cfg)
    h.initialize(w, a, n)
    s = O.System(cfg)
    x0, P0 = O.initialize(cfg, w, a, n)
    s.set_state(x0, P0)
    t = s.tracker()
    blank = np.full_like(imgs[0], 97)
    fewest = 10 ** 9
    worst = 0.0
    for i, (k, img) in enumerate(zip(ks, imgs)):
        im = blank if i in blank_at else img
        imu = seq.imu_between(k)
        s.frame(imu, None, img=im)
        h.frame(im, imu)                                  # host buffers, device detector
        h.sync()
        n_pts, n_upd = _same_tracker(h, t, (blank_at, i))
        fewest = min(fewest, n_pts)
        xa, _ = h.get_state()
        xb, _ = s.get_state()
        assert np.all(np.isfinite(xa))
        worst = max(worst, S.state_delta(xa, xb))
    h.close()
    assert fewest <= 6          # (a handful of points can survive ONE uniform image: flat patches match anywhere)
    assert worst <= 1e-6, worst


def test_empty_imu_batch(gpu_required, seqB):
    """a frame that arrives with no IMU sample between it and the previous one (m = 0): propagate integrates nothing,
    RANSAC's gyro rotation is the identity"""
    from rvio_amd import hip
    cfg, seq, ks, imgs = seqB
    w, a, n = seq.init_from_static(38)
    h = hip.RvioHip(cfg)
    h.initialize(w, a, n)
    s = O.System(cfg)
    x0, P0 = O.initialize(cfg, w, a, n)
    s.set_state(x0, P0)
    t = s.tracker()
    worst = 0.0
    for i, (k, img) in enumerate(zip(ks, imgs)):
        imu = seq.imu_between(k)
        if i in (3, 7):
            imu = imu[:0]
        s.frame(imu, None, img=img)
        h.frame(img, imu)
        h.sync()
        _same_tracker(h, t, i)
        xa, Pa = h.get_state()
        xb, Pb = s.get_state()
        assert np.all(np.isfinite(xa)) and np.all(np.isfinite(Pa))
        worst = max(worst, S.state_delta(xa, xb))
    h.close()
    assert worst <= 1e-6, worst


def test_row_strided_image_and_empty_corner_list(gpu_required, seqB):
    """cv::Mat::step > width (an ROI of a wider buffer) through rvio_hip_track / rvio_hip_frame; and a caller that passes a corner
    list with no entry (n = 0, non-NULL): the feature list simply is not refilled"""
    from rvio_amd import hip
    cfg, seq, ks, imgs = seqB
    wide = np.zeros((cfg.height, cfg.width + 40), np.uint8)
    h, h2 = hip.RvioHip(cfg), hip.RvioHip(cfg)
    t = O.Tracker(cfg)
    none = np.zeros((0, 2), np.float32)
    for i, (k, img) in enumerate(zip(ks[:6], imgs)):
        imu = seq.imu_between(k)
        wide[:] = 255 - (i * 37) % 200                   # the padding must never be read
        view = wide[:, 13:13 + cfg.width]
        view[:] = img
        cand = None if i < 3 else none                   # three frames with the device detector, then no refill
        t.track(img, imu, cand)
        h.track(view, imu, cand)
        h2.track(img, imu, cand)
        _same_tracker(h, t, i)
        _same_tracker(h2, t, i)
    h.close()
    h2.close()


def test_find_newer_int_members_on_the_device(gpu_required):
    """FeatureDetector's grid members are int upstream (FeatureDetector.h:66-77): odd left-over border, nFeatures not divisible by the
    block count, non-integer block sizes — the device refill against the oracle (itself pinned by tests/test_oracle_pins.py)"""
    from rvio_amd import hip
    from test_oracle_pins import FIND_NEWER_CASES, find_newer_inputs
    imu = np.zeros(0, abi.IMU_DTYPE)
    for case in FIND_NEWER_CASES:
        cfg = abi.config_named("B", width=376, height=240, fx=229.327, fy=228.648, cx=183.6075, cy=124.1875, min_dist=5, enable_equalizer=0, **case)
        ref, cand = find_newer_inputs(cfg)
        h, t = hip.RvioHip(cfg), O.Tracker(cfg)
        for tr in (h, t):
            tr.track_points(np.zeros((0, 2), np.float32), np.zeros(0, np.uint8), imu, ref)
            tr.track_points(ref, np.ones(len(ref), np.uint8), imu, cand)
        n_pts, _ = _same_tracker(h, t, case)
        assert n_pts > 50
        h.close()


def test_fisheye_camera(gpu_required):
    """Camera.Fisheye: 1 (Tracker.cc:116-119): the device undistortion against the oracle's on a direct-track sequence — normalised
    coordinates equal to within one float32 ulp (tan() is the only library call in the path), filter states within 1e-6"""
    from rvio_amd import hip
    cfg = abi.config_named("B", enable_equalizer=0, fisheye=1, k1=-0.0137, k2=0.0207, p1=-0.0128, p2=0.0025)
    seq, recs = S.record_sequence(cfg, n_frames=24)
    h = hip.RvioHip(cfg)
    w, a, n = seq.init_from_static(38)
    h.initialize(w, a, n)
    worst = 0.0
    for r in recs:
        inp = r["inp"]
        h.frame_points(inp["tracked"], inp["status"], inp["imu"], inp["cand"])
        x, P = h.get_state()
        worst = max(worst, S.state_delta(x, r["x3"]))
        pts, hl = h.get_points()
        assert np.array_equal(pts, r["pts"]) and np.array_equal(hl, r["hist_len"])
        if len(inp["tracked"]):
            _, un = h.debug_tracked(len(inp["tracked"]))
            want = O.undistort(cfg, inp["tracked"])
            assert np.all(np.abs(un - want) <= np.spacing(np.abs(want)).astype(np.float32))
    h.close()
    assert recs[-1]["did_update"] and worst <= 1e-6, worst


def test_imu_batches_longer_than_the_preallocated_staging(gpu_required, seqB):
    """PreIntegrator::propagate iterates any list (PreIntegrator.cc:96-97): dropped images make it long.  Frames 5 and 8 of the sequence
    are dropped, then a 1.5 s camera outage (300 samples > RVIO_HIP_MAX_IMU = 192): the IMU batch of the next frame is the union of the
    gaps.  Through rvio_hip_frame (host buffers: the staging grows once), rvio_hip_propagate and the _dev entry points; RANSAC's gyro
    prior (Ransac.cc:120-155) walks all of them too.  Tracker tables bit-exact, states within 1e-6 of the oracle."""
    from rvio_amd import hip
    import torch
    cfg, seq, ks, imgs = seqB
    w, a, n = seq.init_from_static(38)
    x0, P0 = O.initialize(cfg, w, a, n)
    # direct check of propagate alone: 500 samples in one call
    imu_all = seq.imu_all()
    long_imu = np.ascontiguousarray(imu_all[400:900])
    h = hip.RvioHip(cfg)
    h.set_state(x0, P0)
    h.propagate(long_imu)
    xa, Pa = h.get_state()
    xb, Pb = O.propagate(cfg, x0, P0, long_imu)
    assert S.state_delta(xa, xb) <= 1e-9 and np.max(np.abs(Pa - Pb)) <= 1e-9 * max(1.0, np.max(np.abs(Pb)))
    d_imu = torch.from_numpy(long_imu.view(np.uint8)).cuda()
    torch.cuda.synchronize()
    h.set_state(x0, P0)
    h.propagate_dev(d_imu.data_ptr(), len(long_imu))
    xc, _ = h.get_state()
    assert np.array_equal(xa, xc)
    h.close()
    # the whole frame with gaps
    frames = [60, 61, 62, 63, 64, 66, 67, 69, 70, 71, 72, 73, 104, 105, 106]      # 65 and 68 dropped; 74..103 = a 1.5 s outage
    h = hip.RvioHip(cfg)
    h.initialize(w, a, n)
    s = O.System(cfg)
    s.set_state(x0, P0)
    t = s.tracker()
    prev, longest = None, 0
    for k in frames:
        imu = seq.imu_between(k) if prev is None else np.concatenate([seq.imu_between(j) for j in range(prev + 1, k + 1)])
        prev = k
        longest = max(longest, len(imu))
        img = seq.render(k)
        s.frame(imu, None, img=img)
        h.frame(img, imu)
        h.sync()
        _same_tracker(h, t, k)
        xa, _ = h.get_state()
        xb, _ = s.get_state()
        assert S.state_delta(xa, xb) <= 1e-6, k
        assert h.frame_info()["device_error"] == 0
    h.close()
    assert longest > 192


def test_imu_stream_that_ends_mid_sequence(gpu_required):
    """The IMU stream stops while images keep coming (m = 0 for every later frame): the newer clones' relative poses are EXACTLY the identity,
    the inverse-depth column of the triangulation's normal equations is exactly zero for tracks inside that stretch, and the reference's
    rank-revealing QR (colPivHouseholderQr, Updater.cc:239) leaves that component of the step at 0 — some of those features then pass
    the gate.  (Found by a test whose sequence was shorter than its frame loop: a plain reciprocal of the zero pivot turned the triangulation
    into NaN and rejected what the reference accepts.)  Accept sets, counters and tracker tables equal, states within 1e-6."""
    from rvio_amd import hip
    cfg = abi.config_named("B")
    seq = rv.synth.SynthSequence(cfg, duration=3.0)          # 60 frames of IMU; the loop runs to frame 84
    w, a, n = seq.init_from_static(38)
    h = hip.RvioHip(cfg)
    h.initialize(w, a, n)
    x, P = O.initialize(cfg, w, a, n)
    trk = O.Tracker(cfg)
    img_count, zero_rho_accepts, empties = 0, 0, 0
    for k in range(39, 85):
        img, imu = seq.render(k), seq.imu_between(k)
        empties += len(imu) == 0
        trk.track(img, imu, None)
        img_count += 1
        ncl = (len(x) - 26) // 7
        x1, P1 = O.propagate(cfg, x, P, imu)
        types, lens, meas = trk.get_tracks()
        d = None
        if ncl > cfg.min_track_len - 1:
            x, P, d = O.update(cfg, x1, P1, types, lens, meas)
        else:
            x, P = x1, P1
        x, P, _, _ = O.augment_compose(cfg, x, P, img_count > 1)
        h.frame(img, imu)
        h.sync()
        _same_tracker(h, trk, k)
        if d is not None:
            dg = h.update_diag()
            assert np.array_equal(dg["accepted"], d["accepted"]), (k, dg["accepted"], d["accepted"])
            assert np.array_equal(np.isfinite(dg["pfinv"]), np.isfinite(d["pfinv"])), k
            zero_rho_accepts += int(np.sum((d["pfinv"][:, 2] == 0) & (d["accepted"] == 1)))
        xa, _ = h.get_state()
        assert np.all(np.isfinite(xa)) and S.state_delta(xa, x) <= 1e-6, k
    h.close()
    assert empties >= 20 and zero_rho_accepts >= 1, (empties, zero_rho_accepts)
