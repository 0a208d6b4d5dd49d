"""GPU parity of the visual front end (pyramid, Scharr, KLT, undistort, RANSAC, book-keeping)
against the CPU oracle on rendered synthetic frames.  Integer/fixed-point + float32 arithmetic
with order-free sums: the bar is BIT-EXACT positions, flags and track tables."""
import numpy as np
import pytest

import oracle as O

abi, rv = O.abi, O.rv
pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def frames():
    cfg = abi.config_named("B", enable_equalizer=0)
    seq = rv.synth.SynthSequence(cfg, duration=8.0)
    ks = list(range(60, 60 + 14))
    imgs = [seq.render(k) for k in ks]
    return cfg, seq, ks, imgs


def test_pyramid_and_scharr_bit_exact(gpu_required, frames):
    from rvio_amd import hip
    cfg, seq, ks, imgs = frames
    h = hip.RvioHip(cfg)
    xy, vis = seq.project(ks[0])
    cand, _ = seq.candidates(ks[0], xy, vis)
    h.track(imgs[0], seq.imu_between(ks[0]), cand)
    ref = imgs[0]
    for lv in range(4):
        img, dxy = h.debug_pyramid(lv)
        assert np.array_equal(img, ref), lv
        assert np.array_equal(dxy, O.scharr(ref)), lv
        ref = O.pyr_down(ref)
    h.close()


FORMS = pytest.mark.parametrize("throughput", [0, 1], ids=["latency-forms", "throughput-forms"])


@FORMS
def test_klt_bit_exact(gpu_required, frames, throughput):
    """(throughput = 1: klt_kernel16, the four-features-per-wave form a batch handle launches, on this one stream)"""
    from rvio_amd import hip
    cfg, seq, ks, imgs = frames
    h = hip.RvioHip(cfg)
    h.kernel_forms(throughput)
    xy, vis = seq.project(ks[0], noise=False)
    cand, _ = seq.candidates(ks[0], xy, vis)
    h.track(imgs[0], seq.imu_between(ks[0]), cand)
    pts0, _ = h.get_points()
    assert len(pts0) == cfg.n_features
    h.track(imgs[1], seq.imu_between(ks[1]), cand)
    got, _ = h.debug_tracked(len(pts0))
    want, st = O.klt(imgs[0], imgs[1], pts0)
    assert st.sum() > 150
    assert np.array_equal(got, want)
    h.close()


@FORMS
def test_klt_early_outs(gpu_required, frames, throughput):
    """The early-outs of cv::calcOpticalFlowPyrLK, each with its own point and an explicit status:
    a window that starts outside the image at the coarsest level / a start position outside the image (status 0), a textureless
    patch (min eigenvalue below minEigThreshold: status 0), a window that leaves the image mid-iteration (status 0), and ordinary
    blobs beside them (status 1) — positions and flags bit-exact against the oracle."""
    from rvio_amd import hip
    cfg, seq, ks, imgs = frames
    H, W = cfg.height, cfg.width
    a = np.full((H, W), 90, np.uint8)
    b = np.full((H, W), 90, np.uint8)
    yy, xx = np.mgrid[0:H, 0:W]

    def blob(img, cx, cy, amp=120.0, sig=2.5):
        g = amp * np.exp(-((xx - cx) ** 2 + (yy - cy) ** 2) / (2 * sig * sig))
        np.copyto(img, np.clip(img.astype(np.float64) + g, 0, 255).astype(np.uint8))
    centres = [(200.0, 150.0), (400.0, 300.0), (600.0, 100.0)]
    for cx, cy in centres:
        blob(a, cx, cy)
        blob(b, cx + 2.3, cy - 1.6)                      # ordinary sub-pixel motion: tracked
    blob(a, 6.0, 240.0)                                  # a blob at the left border that moves out of the image
    blob(b, -9.0, 240.0)
    blob(a, 300.0, 474.0)
    blob(b, 300.0, 492.0)                                # ... and one that leaves through the bottom
    pts = np.array(list(centres) + [(6.0, 240.0), (300.0, 474.0),
                                    (500.0, 400.0),      # textureless patch (uniform 90)
                                    (0.25, 0.25), (751.0, 479.0),    # corners of the image
                                    (-30.0, 100.0), (900.0, 600.0)], np.float32)   # outside the image
    want, st = O.klt(a, b, pts)
    assert list(st[:3]) == [1, 1, 1]
    assert np.all(np.abs(want[:3] - (pts[:3] + np.float32([2.3, -1.6]))) < 0.15)
    assert st[5] == 0, "textureless patch must fail the min-eigenvalue test"
    assert st[8] == 0 and st[9] == 0, "start positions outside the image"
    assert st[3] == 0 or st[4] == 0, "a window that leaves the image"
    h = hip.RvioHip(cfg)
    h.kernel_forms(throughput)
    imu = np.zeros(0, abi.IMU_DTYPE)
    h.track(a, imu, pts)                                 # first image: the list is taken as it is
    assert np.array_equal(h.get_points()[0], pts)
    h.track(b, imu, np.zeros((0, 2), np.float32))
    got, _ = h.debug_tracked(len(pts))
    info = h.frame_info()
    assert np.array_equal(got, want)
    assert info["n_klt_ok"] == int(st.sum())
    h.close()


def test_tracker_sequence_bit_exact(gpu_required, frames):
    """Tracker::track over 14 rendered frames: identical feature lists, histories and update tracks."""
    from rvio_amd import hip
    cfg, seq, ks, imgs = frames
    h = hip.RvioHip(cfg)
    t = O.Tracker(cfg)
    n_upd = 0
    for k, img in zip(ks, imgs):
        xy, vis = seq.project(k, noise=False)
        cand, _ = seq.candidates(k, xy, vis)
        imu = seq.imu_between(k)
        oi = t.track(img, imu, cand)
        h.track(img, imu, cand)
        gi = h.frame_info()
        for key in ("n_tracked_in", "n_klt_ok", "n_ransac_inliers", "ransac_winner", "n_tracked_out", "n_feat_update"):
            assert gi[key] == oi[key], (k, key, gi, oi)
        pa, ha = h.get_points()
        pb, hb = t.get_points()
        assert np.array_equal(pa, pb) and np.array_equal(ha, hb), k
        ta, la, ma = h.get_tracks()
        tb, lb, mb = t.get_tracks()
        assert np.array_equal(ta, tb) and np.array_equal(la, lb), k
        for f in range(len(la)):
            assert np.array_equal(ma[f, : la[f]], mb[f, : lb[f]]), (k, f)
        n_upd += len(la)
    assert n_upd > 0
    h.close()


def test_whole_frame_from_host_buffers(gpu_required, frames):
    """rvio_hip_frame (host image / IMU / corners, copies on the tracker stream, no sync between frames) gives the same
    states as the oracle's System::MonoVIO body — the pipelined staging must not let frame k+1's copies disturb frame k."""
    from rvio_amd import hip
    import scenarios as S
    cfg, seq, ks, imgs = frames
    w, a, n = seq.init_from_static(38)
    h = hip.RvioHip(cfg)
    h.initialize(w, a, n)
    s = O.System(cfg)
    x0, P0 = O.initialize(cfg, w, a, n)
    s.set_state(x0, P0)
    for k, img in zip(ks, imgs):
        xy, vis = seq.project(k, noise=False)
        cand, _ = seq.candidates(k, xy, vis)
        imu = seq.imu_between(k)
        s.frame(imu, cand, img=img)
        h.frame(img.copy(), imu.copy(), cand.copy())     # temporaries: the call must have consumed them on return
    h.sync()
    xa, Pa = h.get_state()
    xb, Pb = s.get_state()
    h.close()
    assert S.state_delta(xa, xb) <= 1e-6


def test_frame_begin_end_split_equals_frame_dev(gpu_required, frames):
    """rvio_hip_frame_begin_dev / frame_plan / update / augment_compose / frame_end (what the sharded updater uses) gives the
    same states, bit for bit, as rvio_hip_frame_dev; no host synchronisation between frames"""
    from rvio_amd import hip
    import torch
    cfg, seq, ks, imgs = frames
    w, a, n = seq.init_from_static(38)
    res = []
    for split in (False, True):
        h = hip.RvioHip(cfg)
        h.initialize(w, a, n)
        keep = []
        for k, img in zip(ks, imgs):
            xy, vis = seq.project(k, noise=False)
            cand, _ = seq.candidates(k, xy, vis)
            imu = seq.imu_between(k)
            d_img = torch.from_numpy(img).cuda()
            d_imu = torch.from_numpy(imu.view(np.uint8)).cuda()
            d_cand = torch.from_numpy(cand).cuda()
            keep += [d_img, d_imu, d_cand]
            torch.cuda.synchronize()
            args = (d_img.data_ptr(), img.shape[1], d_imu.data_ptr(), len(imu), d_cand.data_ptr(), len(cand))
            if split:
                h.frame_begin_dev(*args)
                do_update, do_augment = h.frame_plan()
                if do_update:
                    h.update_tracked()
                h.augment_compose(do_augment)
                h.frame_end()
            else:
                h.frame_dev(*args)
        h.sync()
        res.append(h.get_state())
        h.close()
    assert np.array_equal(res[0][0], res[1][0]) and np.array_equal(res[0][1], res[1][1])


def test_pose_is_final_before_the_refill(gpu_required, frames):
    """rvio_hip_get_pose waits for the filter stream only; in run-ahead mode (device detector) the filter of a frame starts behind the
    hand-over half of book-keeping, before the detector and the refill half have finished.  The pose read right after the call must be the
    pose read after a full synchronisation, and the run must end in the same state as one that synchronises after every frame."""
    from rvio_amd import hip
    import torch
    cfg, seq, ks, imgs = frames
    w, a, n = seq.init_from_static(38)
    res = []
    for sync_each in (False, True):
        h = hip.RvioHip(cfg)
        h.initialize(w, a, n)
        keep = []
        for k, img in zip(ks, imgs):
            imu = seq.imu_between(k)
            d_img = torch.from_numpy(img).cuda()
            d_imu = torch.from_numpy(imu.view(np.uint8)).cuda()
            keep += [d_img, d_imu]
            torch.cuda.synchronize()
            h.frame_dev(d_img.data_ptr(), img.shape[1], d_imu.data_ptr(), len(imu), 0, 0)     # NULL corner list: device detector, run-ahead
            p0, q0 = h.pose()
            if sync_each:
                h.sync()
                p1, q1 = h.pose()
                assert np.array_equal(p0, p1) and np.array_equal(q0, q1), k
        h.sync()
        info = h.frame_info()
        res.append(h.get_state())
        h.close()
        assert info["device_error"] == 0
    assert np.array_equal(res[0][0], res[1][0]) and np.array_equal(res[0][1], res[1][1])


def test_whole_frame_with_images(gpu_required, frames):
    """System::MonoVIO body on images: HIP vs oracle states within 1e-6 over the sequence."""
    from rvio_amd import hip
    import scenarios as S
    cfg, seq, ks, imgs = frames
    w, a, n = seq.init_from_static(38)
    h = hip.RvioHip(cfg)
    h.initialize(w, a, n)
    s = O.System(cfg)
    x0, P0 = O.initialize(cfg, w, a, n)
    s.set_state(x0, P0)
    import torch
    worst = 0.0
    for k, img in zip(ks, imgs):
        xy, vis = seq.project(k, noise=False)
        cand, _ = seq.candidates(k, xy, vis)
        imu = seq.imu_between(k)
        s.frame(imu, cand, img=img)
        d_img = torch.from_numpy(img).cuda()
        d_imu = torch.from_numpy(imu.view(np.uint8)).cuda()
        d_cand = torch.from_numpy(cand).cuda()
        torch.cuda.synchronize()
        h.frame_dev(d_img.data_ptr(), img.shape[1], d_imu.data_ptr(), len(imu), d_cand.data_ptr(), len(cand))
        h.sync()
        xa, Pa = h.get_state()
        xb, Pb = s.get_state()
        worst = max(worst, S.state_delta(xa, xb))
    h.close()
    assert worst <= 1e-6, worst


def test_runahead_without_equalizer_long_sequence(gpu_required):
    """Run-ahead mode (device detector, no host synchronisation between frames) with Tracker.EnableEqualizer: 0: the image chain of frame k
    (here: the detector alone) rewrites corner list k % 3, which the refill half of book-keeping(k-3) reads — it has to start behind that
    book-keeping with or without CLAHE (round-2 advisor finding: the ring wait sat inside the equaliser branch).  60 free-running frames
    from device-resident images against the oracle: bit-exact feature lists at the end, states within 1e-6, twice with the same result."""
    from rvio_amd import hip
    import scenarios as S
    import torch
    cfg = abi.config_named("B", enable_equalizer=0)
    seq = rv.synth.SynthSequence(cfg, duration=8.0)
    ks = list(range(39, 39 + 60))
    imgs = np.stack([seq.render(k) for k in ks])
    imus = [seq.imu_between(k) for k in ks]
    w, a, n = seq.init_from_static(38)
    s = O.System(cfg)
    x0, P0 = O.initialize(cfg, w, a, n)
    s.set_state(x0, P0)
    for img, imu in zip(imgs, imus):
        s.frame(imu, None, img=img)
    xb, _ = s.get_state()
    pb, hb = s.tracker().get_points()
    d_imgs = torch.from_numpy(imgs).cuda()
    d_imus = [torch.from_numpy(i.view(np.uint8)).cuda() for i in imus]
    torch.cuda.synchronize()
    res = []
    for rep in range(2):
        h = hip.RvioHip(cfg)
        h.initialize(w, a, n)
        for i in range(len(ks)):
            h.frame_dev(d_imgs[i].data_ptr(), cfg.width, d_imus[i].data_ptr(), len(imus[i]), 0, 0)
        h.sync()
        xa, Pa = h.get_state()
        pa, ha = h.get_points()
        info = h.frame_info()
        h.close()
        assert info["device_error"] == 0 and info["updated"] == 1
        assert np.array_equal(pa, pb) and np.array_equal(ha, hb), rep
        assert S.state_delta(xa, xb) <= 1e-6, rep
        res.append((xa, Pa))
    assert np.array_equal(res[0][0], res[1][0]) and np.array_equal(res[0][1], res[1][1])
