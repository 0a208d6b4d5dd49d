"""GPU parity against the COMMITTED golden fixtures (tests/golden/*.npz, written by tests/golden/make_golden.py from the CPU oracle):
no oracle call at test time — the files carry the inputs and the expected outputs."""
import os
import zlib

import numpy as np
import pytest

import oracle as O          # only for the shared config / dtype helpers (O.abi); no oracle function is called here
import scenarios as S

abi = O.abi
pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_filter_stages_against_the_golden_snapshot(gpu_required):
    """frame 30 of the direct-track sequence, cfg B: propagate -> update -> augment/compose from the stored (x0, P0, IMU, tracks)"""
    from rvio_amd import hip
    g = np.load(os.path.join(GOLD, "cfgB_direct_seed0_frame30.npz"))
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
    assert np.array_equal(diag["accepted"], g["accepted"])                      # the chi-square gate took the same decisions
    assert np.allclose(diag["gamma"], g["gamma"], rtol=1e-7, atol=1e-9)
    h.augment_compose(bool(g["do_augment"]))
    x, P = h.get_state()
    assert S.state_delta(x, g["x3"]) <= 1e-9 and np.max(np.abs(P - g["P3"])) <= 1e-9 * np.max(np.abs(g["P3"]))
    h.close()


def test_tracker_against_the_golden_image_fixture(gpu_required):
    """4 frames of the half-size camera through CLAHE, the device detector, KLT, RANSAC and book-keeping: bit-exact feature lists"""
    from rvio_amd import hip
    g = np.load(os.path.join(GOLD, "small_images_tracker.npz"))
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
