"""The GPU golden tests' OWN code, run on the CPU with the oracle standing in for the device.

tests/test_gpu_golden.py holds the HIP path against committed fixtures on the GPU box; GPU minutes are scarce, so its test functions are
also walked here with a stand-in `RvioHip` that forwards every call to the oracle: a typo in a key, a wrong view of a stored array or a
fixture that no longer matches its reader fails in the CPU suite already.  (What it cannot show is the device's numbers: that is the
`-m gpu` run.)"""
import numpy as np
import pytest

import oracle as O
import test_gpu_golden as T
from rvio_amd import hip


class OracleAsDevice:
    def __init__(self, cfg, **kw):
        self.cfg, self.trk, self.first, self.sys = cfg, O.Tracker(cfg), None, None

    # stage-wise calls
    def set_state(self, x, P): self.x, self.P = np.array(x), np.array(P)
    def get_state(self): return self.sys.get_state() if self.sys is not None else (self.x, self.P)
    def propagate(self, imu): self.x, self.P = O.propagate(self.cfg, self.x, self.P, imu)
    def update(self, types, lens, meas): self.x, self.P, self.diag = O.update(self.cfg, self.x, self.P, types, lens, meas)
    def update_diag(self): return self.diag
    def augment_compose(self, aug): self.x, self.P, _, _ = O.augment_compose(self.cfg, self.x, self.P, aug)

    # front end on images
    def track(self, img, imu, cand):
        if self.first is None:
            self.first = img
        self.trk.track(img, imu, cand)

    def get_corners(self): return O.detect(self.cfg, O.clahe(self.first), 1), None
    def debug_pyramid(self, lv): return O.clahe(self.first), None
    def get_points(self): return self.sys.tracker().get_points() if self.sys is not None else self.trk.get_points()

    # whole frames
    def initialize(self, w, a, n):
        self.sys = O.System(self.cfg)
        self.sys.set_state(*O.initialize(self.cfg, w, a, n))

    def frame_points(self, tracked, status, imu, cand): self.info = self.sys.frame(imu, cand, tracked=tracked, status=status)[0]
    def frame_info(self): return self.info
    def close(self): pass


@pytest.fixture()
def oracle_device(monkeypatch):
    monkeypatch.setattr(hip, "RvioHip", OracleAsDevice)


@pytest.mark.parametrize("family", ["oracle", "reference"])
def test_stage_and_tracker_golden_tests_run(oracle_device, family):
    T.test_filter_stages_against_the_golden_snapshot(None, family)
    T.test_tracker_against_the_golden_image_fixture(None, family)


@pytest.mark.parametrize("name", ["A", "B", "C", "E"])
def test_full_load_golden_test_runs(oracle_device, name):
    T.test_full_load_update_against_the_reference_written_digest(None, name)


def test_free_run_golden_test_runs(oracle_device):
    T.test_free_run_replays_the_reference_states(None)
