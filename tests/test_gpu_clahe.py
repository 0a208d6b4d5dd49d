"""GPU parity of the CLAHE stage (enable_equalizer = 1, Tracker.cc:198-202): the equalized level-0 image, the pyramid
built from it and the tracker driven by it are BIT-EXACT against the oracle."""
import numpy as np
import pytest

import oracle as O
from test_clahe import test_images

abi, rv = O.abi, O.rv
pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("idx", range(12))
def test_equalized_pyramid_bit_exact(gpu_required, idx):
    from rvio_amd import hip
    img = test_images()[idx]
    hh, ww = img.shape
    cfg = abi.config_named("B", enable_equalizer=1, width=ww, height=hh, block_x=min(150, ww // 4), block_y=min(120, hh // 4))
    h = hip.RvioHip(cfg)
    imu = np.zeros(1, abi.IMU_DTYPE)
    imu["dt"] = 0.005
    cand = np.array([[ww / 2, hh / 2], [ww / 3, hh / 3]], np.float32)
    h.track(img, imu, cand)
    ref = O.clahe(img)
    levels, lw, lh = 1, ww, hh
    for _ in range(3):
        lw, lh = (lw + 1) // 2, (lh + 1) // 2
        if lw <= 15 or lh <= 15:
            break
        levels += 1
    for lv in range(levels):
        got, dxy = h.debug_pyramid(lv)
        assert np.array_equal(got, ref), (img.shape, lv, int(np.abs(got.astype(int) - ref.astype(int)).max()))
        assert np.array_equal(dxy, O.scharr(ref)), lv
        ref = O.pyr_down(ref)
    h.close()


def test_tracker_sequence_with_equalizer_bit_exact(gpu_required):
    from rvio_amd import hip
    cfg = abi.config_named("B", enable_equalizer=1)
    seq = rv.synth.SynthSequence(cfg, duration=8.0)
    h = hip.RvioHip(cfg)
    t = O.Tracker(cfg)
    n_upd = 0
    for k in range(60, 70):
        img = seq.render(k)
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
