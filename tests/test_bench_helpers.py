"""bench.py's host logic, without a GPU: the algorithmic-work model (W_filter) the batched figures are priced with, and —
after round 2's driver run died of a negative frame index turned into a device pointer — every leg's index arithmetic, walked with a
fake handle that refuses any pointer outside the resident sequence."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402  (importing bench does not touch the GPU or the oracle)


def test_filter_flops_matches_the_survey_figures():
    abi = bench.abi
    cfg = abi.config_named("B")
    Fu = abi.fu(cfg)
    # worst case of SURVEY.md 8d: every one of the ceil(F/2) features at maximum length -> "cfg B 51 MFLOP/frame"
    # (gate 18.9 + compression 21.3 incl. the 2.3 of the nullspace step, which the W_filter formula itself does not carry, + EKF 7.4 + propagate 0.9)
    w = bench.filter_flops(cfg, 10, [11] * Fu, [ord("1")] * Fu, 10) / 1e6
    assert 46.0 < w < 51.5, w
    gate = sum(2 * 19 * 60 ** 2 + 2 * 19 ** 2 * 60 + (4 / 3) * 19 ** 3 for _ in range(Fu)) / 1e6
    assert abs(gate - 18.9) < 0.1
    # no update while the window is too short: propagation only (m * 6 * 24^3)
    assert bench.filter_flops(cfg, 1, [3] * 5, [ord("1")] * 5, 10) == 10 * 6.0 * 24 ** 3
    # a type-'2' feature contributes its first ceil(L/2) observations only
    a = bench.filter_flops(cfg, 10, [11], [ord("2")], 0)
    b = bench.filter_flops(cfg, 10, [6], [ord("1")], 0)
    assert a == b


class FakeHandle:
    """stands in for rvio_amd.hip.RvioHip: records the frames it is handed and refuses a pointer that is not frame i of the FrameSet"""
    fs = None
    seen = None

    def __init__(self, cfg, device=0, **kw):
        self.cfg = cfg

    def _ck(self, p_img, stride, p_imu, m, p_cand, n_cand):
        fs = FakeHandle.fs
        off = p_img - fs.p_img
        assert off >= 0 and off % fs.isb == 0 and off // fs.isb < fs.n, "image pointer outside the resident sequence"
        i = off // fs.isb
        assert p_imu == fs.p_imu + i * fs.msb and 0 <= m <= fs.imu_arr.shape[1] and p_cand == 0 and n_cand == 0
        FakeHandle.seen.append(int(i))

    def frame_dev(self, *a):
        self._ck(*a)

    def track_dev(self, *a):
        self._ck(*a)

    def frame(self, img, imu, cand):
        FakeHandle.seen.append(-1)

    def initialize(self, *a):
        pass

    def sync(self):
        pass

    def close(self):
        pass

    def pose(self):
        return np.zeros(3), np.zeros(4)

    def frame_info(self):
        return {"updated": 1}


@pytest.mark.parametrize("steps,warmup", [(20, 5), (1, 0), (200, 40), (8, 0), (60, 2)])
@pytest.mark.parametrize("name", ["B", "E"])
def test_every_leg_stays_inside_the_resident_sequence(monkeypatch, steps, warmup, name):
    from rvio_amd import hip
    monkeypatch.setattr(hip, "RvioHip", FakeHandle)
    cfg = bench.abi.config_named(name)
    n_frames = 1 + warmup + steps
    fs = bench.FrameSet.fake(cfg, n_frames)
    FakeHandle.fs, FakeHandle.seen = fs, []
    # the checked accessor itself
    for bad in (-1, -14, n_frames, n_frames + 3):
        with pytest.raises(IndexError):
            fs.args(bad)
        with pytest.raises(IndexError):
            fs.host(bad)
    assert fs.args(n_frames - 1)[0] == fs.p_img + (n_frames - 1) * fs.isb
    # pose latency: a plan exists only when the window can be filled first, and it never leaves [0, n)
    plan = bench.pose_latency_plan(n_frames, cfg.max_track_len)
    if plan is None:
        assert n_frames < 3 * cfg.max_track_len + 10 + 8
        long_plan = bench.pose_latency_plan(bench.PARITY_FRAMES, cfg.max_track_len)   # the leg then runs on the parity sequence
        assert long_plan is not None and long_plan[0] + long_plan[1] <= bench.PARITY_FRAMES
    else:
        n_warm, n_timed = plan
        assert 0 <= n_warm and n_timed >= 8 and n_warm + n_timed <= n_frames
        FakeHandle.seen = []
        r = bench.pose_latency_leg(cfg, fs, plan, None, None, 0, 0)
        assert FakeHandle.seen == list(range(n_warm + n_timed)) and r["frames"] == n_timed
    # host buffers / multi stream: warm-up clamped, every frame exactly once per handle
    FakeHandle.seen = []
    r = bench.host_buffer_leg(cfg, fs, None, None, 0, 1 + warmup)
    assert len(FakeHandle.seen) == n_frames and r["value"] > 0
    FakeHandle.seen = []
    r = bench.multi_stream(cfg, fs, None, None, 0, 1 + warmup, streams=3)
    assert sorted(FakeHandle.seen) == sorted(list(range(n_frames)) * 3)
    assert r["frames_per_stream"] == n_frames - min(1 + warmup, n_frames - 1)


def test_a_failing_secondary_leg_becomes_an_error_object():
    out = {"value": 1.0}

    def boom():
        raise RuntimeError("leg died")
    bench.safe_leg(out, "pose_latency_unpipelined", boom)
    assert out["value"] == 1.0 and "leg died" in out["pose_latency_unpipelined"]["error"]


def test_batch_size_entries():
    """--batch entries: plain counts and "instances x handles" """
    assert bench.parse_batch_size("2048") == (2048, 1) and bench.parse_batch_size(256) == (256, 1)
    assert bench.parse_batch_size("2048x2") == (2048, 2) and bench.parse_batch_size("4096x4") == (4096, 4)
    for bad in ("0", "10x3", "8x0", "x2"):
        with pytest.raises(ValueError):
            bench.parse_batch_size(bad)


def test_rocpd_stats_reports_the_median(tmp_path):
    """tools/rocpd_stats.py on a hand-made rocpd database: one slow first launch must not hide in the figure quoted (the median)"""
    import sqlite3
    import subprocess
    import sys
    db = tmp_path / "k.db"
    con = sqlite3.connect(str(db))
    con.execute("create table kernels (name text, start integer, end integer, grid_size_z integer, workgroup_size_z integer)")
    rows = [("joseph(int)", 0, 11000000, 2048, 1)] + [("joseph(int)", 0, 340000 + 1000 * i, 2048, 1) for i in range(9)] + [("other(int)", 0, 5000, 1, 1)]
    con.executemany("insert into kernels values (?, ?, ?, ?, ?)", rows)
    con.commit()
    con.close()
    tool = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "rocpd_stats.py")
    out = subprocess.run([sys.executable, tool, str(db), "--grid-z", "2048"], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    line = [l for l in out.stdout.splitlines() if l.startswith("| joseph")]
    assert len(line) == 1 and "other" not in out.stdout
    cells = [c.strip() for c in line[0].strip("|").split("|")]
    assert int(cells[1]) == 10 and abs(float(cells[4]) - 345.0) < 1e-6 and float(cells[3]) > 1400.0     # median 345 us, mean dragged up by the 11 ms launch


def test_issue_slots_of_the_batched_streams(tmp_path):
    """tools/issue_slots_json.py (rocprofv3 --pmc csv -> per-frame SQ figures) and bench.issue_slots (those figures over the issue slots of
    a live frame time): only launches that cover every stream count, torch's harness kernels do not, a frame = one pyramid launch"""
    import json
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import issue_slots_json as T
    rows = ["Kernel_Name,Grid_Size,Workgroup_Size,Counter_Name,Counter_Value"]
    for f in range(3):                                   # three batched frames of 128 streams
        rows += ['"pyramid_kernel(unsigned char const*, int)",%d,256,SQ_ACTIVE_INST_ANY,%d' % (96 * 128 * 256, 100 + f),
                 '"pyramid_kernel(unsigned char const*, int)",%d,256,SQ_ACTIVE_INST_VALU,60' % (96 * 128 * 256),
                 '"void klt_kernel16(PyrDev, PyrDev)",%d,64,SQ_ACTIVE_INST_ANY,50' % (50 * 128 * 64),
                 '"void klt_kernel16(PyrDev, PyrDev)",%d,64,SQ_ACTIVE_INST_VALU,40' % (50 * 128 * 64),
                 '"void at::native::vectorized_gather_kernel<16, long>(char*)",%d,128,SQ_ACTIVE_INST_ANY,1000' % (4096 * 128),
                 '"stage_gate_kernel(unsigned long long const*)",64,64,SQ_ACTIVE_INST_ANY,7']
    p = tmp_path / "c.csv"
    p.write_text("\n".join(rows) + "\n")
    cm = T.summarise(T.collect([str(p)], 128), 128)
    assert cm["frames_profiled"] == 3 and set(cm["per_kernel"]) == {"pyramid_kernel", "klt_kernel16"}
    assert cm["per_batched_frame"]["SQ_ACTIVE_INST_ANY"] == pytest.approx(101 + 50)
    assert cm["per_batched_frame"]["SQ_ACTIVE_INST_VALU"] == pytest.approx(100)
    j = tmp_path / "s.json"
    j.write_text(json.dumps(cm))
    o = bench.issue_slots(128, 1.0, path=str(j))["issue_slots"]
    slots = 1e-3 * 2.4e9 * 1024 / 4
    assert o["slots_per_batched_frame"] == pytest.approx(slots)
    assert o["frac_valu"] == pytest.approx(100 / slots) and o["frac_any"] == pytest.approx(151 / slots)
    assert bench.issue_slots(16, 1.0, path=str(j)) == {}                   # committed counters are for another batch size: nothing is claimed
    assert bench.issue_slots(128, 1.0, path=str(tmp_path / "none.json")) == {}
