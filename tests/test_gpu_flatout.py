"""The pipelined whole-frame path of the STOCK configuration (Tracker.EnableEqualizer: 1, corners from the device detector) with NO host
synchronisation between frames — what bench.py times and what host/rvio_replay drives (System.cc:253-367 per image, rvio_mono.cc:54-79):

  * flat out and with only rvio_hip_get_pose between the frames, through rvio_hip_frame (host buffers) and rvio_hip_frame_dev (resident
    frames): 130 frames against the LITERAL oracle (sequential Givens QR + rank scan) <= 1e-6, bit-exact feature lists at the end;
  * every dependency between the handle's streams has to hold at ANY pacing of the caller and of the queues: a sweep of host-side delays
    between the calls, and runs with sleeping kernels sprinkled over the four streams (rvio_hip_debug_stall), must reproduce the synchronised
    run BIT FOR BIT — the stalls make chains overtake each other on any box, fast or slow (round 3: one `corners` counter for two image
    chains; the driver's box then produced a trajectory 5.7e-3 off while every synchronised test passed);
  * RVIO_PARANOID=1 (every hand-off a default-flag stream event on plain streams) gives the same bits.
"""
import os
import subprocess
import sys
import time

import numpy as np
import pytest

import oracle as O
import scenarios as S

abi, rv = O.abi, O.rv
pytestmark = pytest.mark.gpu

K0 = 38


def stock(cfg_name="B", n=130, duration=9.0):
    cfg = abi.config_named(cfg_name, enable_equalizer=1)
    seq = rv.synth.SynthSequence(cfg, duration=duration)
    ks = list(range(K0 + 1, K0 + 1 + n))
    imgs = np.stack([seq.render(k) for k in ks])
    imus = [seq.imu_between(k) for k in ks]
    return cfg, seq, imgs, imus


@pytest.fixture(scope="module")
def stock_b():
    cfg, seq, imgs, imus = stock("B", 130)
    init = seq.init_from_static(K0)
    s = O.System(cfg)                       # literal Givens QR + rank scan (Updater.cc:469-536)
    s.set_state(*O.initialize(cfg, *init))
    poses = []
    for img, imu in zip(imgs, imus):
        _, _, pp, pq = s.frame(imu, None, img=img)
        poses.append(np.concatenate((pp, pq)))
    x, P = s.get_state()
    pts, hl = s.tracker().get_points()
    return dict(cfg=cfg, init=init, imgs=imgs, imus=imus, x=x, P=P, pts=pts, hl=hl, poses=np.array(poses))


def pose_delta(a, b):
    a, b = np.array(a), np.array(b)
    qa = a[:, 3:] * np.sign(a[:, 6:7] + (a[:, 6:7] == 0))
    qb = b[:, 3:] * np.sign(b[:, 6:7] + (b[:, 6:7] == 0))
    return max(float(np.abs(a[:, :3] - b[:, :3]).max()), float(np.abs(qa - qb).max()))


def run_hip(d, mode, pose_between, stalls=None, delay_us=0, sync_every=False, n=None, poison=0, noise=None):
    """mode 'host': rvio_hip_frame on host buffers; 'dev': rvio_hip_frame_dev on resident frames.  Returns poses (if read), final state, lists"""
    from rvio_amd import hip
    cfg, imgs, imus = d["cfg"], d["imgs"], d["imus"]
    n = len(imgs) if n is None else n
    if mode == "dev":
        import torch
        d_imgs = torch.from_numpy(imgs[:n]).cuda()
        d_imus = [torch.from_numpy(i.view(np.uint8)).cuda() for i in imus[:n]]
        torch.cuda.synchronize()
    h = hip.RvioHip(cfg)
    h.initialize(*d["init"])
    rng = np.random.default_rng(stalls) if stalls is not None else None
    poses = []
    for i in range(n):
        if rng is not None:
            for _ in range(int(rng.integers(0, 3))):     # 0..2 stalls in front of this frame, any stream, 30..900 us
                h.stall(int(rng.integers(0, 4)), int(rng.integers(30, 900)))
        if noise is not None and i % noise[2] == 0:
            h.noise(noise[0], noise[1])
        if mode == "dev":
            h.frame_dev(d_imgs[i].data_ptr(), cfg.width, d_imus[i].data_ptr(), len(imus[i]), 0, 0)
        else:
            h.frame(imgs[i].copy(), imus[i].copy(), None)   # temporaries: the call must have consumed them on return
        if sync_every:
            h.sync()
            if poison:
                h.poison(poison)
        if pose_between:
            p, q = h.pose()                                 # waits for the filter stream only
            poses.append(np.concatenate((p, q)))
        if delay_us:
            t_end = time.perf_counter() + 1e-6 * delay_us
            while time.perf_counter() < t_end:
                pass
    h.sync()
    x, P = h.get_state()
    pts, hl = h.get_points()
    info = h.frame_info()
    h.close()
    assert info["device_error"] == 0, info
    return dict(poses=np.array(poses), x=x, P=P, pts=pts, hl=hl, info=info)


@pytest.mark.parametrize("mode", ["host", "dev"])
@pytest.mark.parametrize("pose_between", [False, True])
def test_stock_frames_flat_out_track_the_literal_oracle(gpu_required, stock_b, mode, pose_between):
    r = run_hip(stock_b, mode, pose_between)
    assert r["info"]["updated"] == 1
    assert np.array_equal(r["pts"], stock_b["pts"]) and np.array_equal(r["hl"], stock_b["hl"])     # feature lists bit-exact after 130 frames
    assert S.state_delta(r["x"], stock_b["x"]) <= 1e-6
    scale = np.sqrt(np.abs(np.outer(np.diag(stock_b["P"]), np.diag(stock_b["P"])))) + 1e-300
    assert float(np.max(np.abs(r["P"] - stock_b["P"]) / scale)) <= 1e-3
    if pose_between:
        assert pose_delta(r["poses"], stock_b["poses"]) <= 1e-6


@pytest.fixture(scope="module")
def sync_ref(stock_b):
    """the synchronised run of the first 60 frames (rvio_hip_sync behind every frame): the bits every other pacing has to reproduce"""
    return {m: run_hip(stock_b, m, True, sync_every=True, n=60) for m in ("host", "dev")}


def same_bits(a, b):
    return (np.array_equal(a["poses"], b["poses"]) and np.array_equal(a["x"], b["x"]) and np.array_equal(a["P"], b["P"])
            and np.array_equal(a["pts"], b["pts"]) and np.array_equal(a["hl"], b["hl"]))


def first_diff(a, b):
    d = np.abs(a["poses"] - b["poses"]).max(axis=1)
    return "first differing frame %d, max |pose diff| %.3e" % (int(np.argmax(d > 0)) if np.any(d > 0) else -1, float(d.max()))


@pytest.mark.parametrize("mode", ["host", "dev"])
def test_pacing_sweep_is_bit_identical(gpu_required, stock_b, sync_ref, mode):
    """host-side delays between the calls (a caller that decodes an image in between): 13 pacings, poses and final state bit for bit"""
    for d_us in (0, 50, 100, 150, 200, 250, 300, 400, 500, 700, 1000, 1500, 2500):
        r = run_hip(stock_b, mode, True, delay_us=d_us, n=60)
        assert same_bits(r, sync_ref[mode]), (d_us, first_diff(r, sync_ref[mode]))


@pytest.mark.parametrize("mode", ["host", "dev"])
@pytest.mark.parametrize("seed", [1, 2, 3, 4, 5, 6])
def test_stalled_queues_do_not_change_a_bit(gpu_required, stock_b, sync_ref, mode, seed):
    """sleeping kernels on random streams in front of random frames: image chains overtake each other, the filter runs late or early, the
    side stream starves — flat out, with and without the pose read-back; the results are those of the synchronised run"""
    r = run_hip(stock_b, mode, seed % 2 == 0, stalls=seed, n=60)
    ref = sync_ref[mode]
    if seed % 2 == 0:
        assert same_bits(r, ref), first_diff(r, ref)
    else:
        assert np.array_equal(r["x"], ref["x"]) and np.array_equal(r["P"], ref["P"]) and np.array_equal(r["pts"], ref["pts"])


@pytest.mark.parametrize("mode", ["host", "dev"])
@pytest.mark.parametrize("noise", [(64, 400, 2), (256, 1500, 6), (1024, 3000, 10)], ids=["64wg", "256wg", "1024wg"])
def test_a_loaded_chip_does_not_change_a_bit(gpu_required, stock_b, sync_ref, mode, noise):
    """64 / 256 / 1024 workgroups of HBM + L2 + LDS traffic beside the pipeline (rvio_hip_debug_noise: the pipeline's waves share SIMDs, LDS
    ports and L2 slices with them): the hand-overs INSIDE kernels — LDS flags of the solve, last-block patterns, atomics of the detector —
    hold at any relative pace of the waves; flat out, and with stalled queues on top"""
    for stalls in (None, 11):
        r = run_hip(stock_b, mode, False, stalls=stalls, n=60, noise=noise)
        ref = sync_ref[mode]
        assert np.array_equal(r["pts"], ref["pts"]) and np.array_equal(r["hl"], ref["hl"]), (stalls, "feature lists")
        assert np.array_equal(r["x"], ref["x"]) and np.array_equal(r["P"], ref["P"]), (stalls, S.state_delta(r["x"], ref["x"]))


@pytest.fixture(scope="module")
def stock_a():
    """cfg A = the stock settings file (14-clone window, 6n = 84: the 12-wave, two-rows-per-lane form of the solve; what host/rvio_replay's
    tests run) — 70 frames, literal oracle end state, synchronised device reference"""
    cfg, seq, imgs, imus = stock("A", 70)
    init = seq.init_from_static(K0)
    s = O.System(cfg)
    s.set_state(*O.initialize(cfg, *init))
    for img, imu in zip(imgs, imus):
        s.frame(imu, None, img=img)
    d = dict(cfg=cfg, init=init, imgs=imgs, imus=imus, x=s.get_state()[0], pts=s.tracker().get_points()[0])
    d["ref"] = run_hip(d, "host", True, sync_every=True)
    return d


@pytest.mark.parametrize("kind", ["flat", "stall1", "stall2", "stall3", "noise64", "noise256", "noise1024+stall"])
def test_stock_settings_file_window_under_every_pacing(gpu_required, stock_a, kind):
    """the same bars for cfg A: the synchronised run tracks the literal oracle (<= 1e-6, bit-exact feature list), and flat out / stalled /
    loaded runs reproduce the synchronised run bit for bit"""
    ref = stock_a["ref"]
    assert np.array_equal(ref["pts"], stock_a["pts"]) and S.state_delta(ref["x"], stock_a["x"]) <= 1e-6
    kw = {"flat": {}, "stall1": dict(stalls=1), "stall2": dict(stalls=2), "stall3": dict(stalls=3), "noise64": dict(noise=(64, 400, 2)),
          "noise256": dict(noise=(256, 1500, 6)), "noise1024+stall": dict(noise=(1024, 3000, 10), stalls=4)}[kind]
    r = run_hip(stock_a, "host", True, **kw)
    assert same_bits(r, ref), first_diff(r, ref)


@pytest.fixture(scope="module")
def stock_c():
    """cfg C (400 features, 20-clone window, 6n = 120): the LONG-window form of the frame — the solve in its split form (six launches), the
    Cholesky factor of the next update's clone block on the queue the second image chain holds at the short windows (ordered by two events:
    augment / compose -> factor -> the solve's first product), the Joseph stage one wave per tile — 60 frames, window full after 21"""
    cfg, seq, imgs, imus = stock("C", 60)
    init = seq.init_from_static(K0)
    s = O.System(cfg)
    s.set_state(*O.initialize(cfg, *init))
    for img, imu in zip(imgs, imus):
        s.frame(imu, None, img=img)
    d = dict(cfg=cfg, init=init, imgs=imgs, imus=imus, x=s.get_state()[0], pts=s.tracker().get_points()[0])
    d["ref"] = run_hip(d, "host", True, sync_every=True)
    return d


@pytest.mark.parametrize("kind", ["flat", "stall1", "stall2", "stall3", "stall4", "noise256", "noise1024+stall"])
def test_long_window_under_every_pacing(gpu_required, stock_c, kind):
    """cfg C: the synchronised run tracks the literal oracle (<= 1e-6, bit-exact feature list); flat out, with stalled queues (the factor's queue
    among them: the solve then waits for its event, a late augment / compose holds the factor back) and on a loaded chip the results are
    the synchronised run's, bit for bit"""
    ref = stock_c["ref"]
    assert np.array_equal(ref["pts"], stock_c["pts"]) and S.state_delta(ref["x"], stock_c["x"]) <= 1e-6
    kw = {"flat": {}, "stall1": dict(stalls=1), "stall2": dict(stalls=2), "stall3": dict(stalls=3), "stall4": dict(stalls=7), "noise256": dict(noise=(256, 1500, 6)),
          "noise1024+stall": dict(noise=(1024, 3000, 10), stalls=4)}[kind]
    r = run_hip(stock_c, "host", True, **kw)
    assert same_bits(r, ref), first_diff(r, ref)


@pytest.mark.parametrize("mode", ["host", "dev"])
def test_queues_descheduled_for_milliseconds(gpu_required, stock_b, sync_ref, mode):
    """An oversubscribed GPU (other tenants hold hardware queue slots) time-slices the handle's four queues with a quantum of milliseconds: one
    chain stands still while the others run to completion.  Emulated with 4-8 ms stalls, one stream at a time, every few frames — the pacing at
    which round 3's shared `corners` counter let the next frame's image chain announce corners the stalled chain had not written yet."""
    from rvio_amd import hip
    cfg, imgs, imus = stock_b["cfg"], stock_b["imgs"], stock_b["imus"]
    if mode == "dev":
        import torch
        d_imgs = torch.from_numpy(imgs[:60]).cuda()
        d_imus = [torch.from_numpy(i.view(np.uint8)).cuda() for i in imus[:60]]
        torch.cuda.synchronize()
    h = hip.RvioHip(cfg)
    h.initialize(*stock_b["init"])
    for i in range(60):
        if i % 5 == 2:
            h.stall((i // 5) % 4, 4000 + 1000 * ((i // 5) % 5))
        if mode == "dev":
            h.frame_dev(d_imgs[i].data_ptr(), cfg.width, d_imus[i].data_ptr(), len(imus[i]), 0, 0)
        else:
            h.frame(imgs[i], imus[i], None)
        if i % 3 == 0:
            h.pose()
    h.sync()
    x, P = h.get_state()
    pts, hl = h.get_points()
    info = h.frame_info()
    h.close()
    ref = sync_ref[mode]
    assert info["device_error"] == 0
    assert np.array_equal(pts, ref["pts"]) and np.array_equal(hl, ref["hl"])
    assert np.array_equal(x, ref["x"]) and np.array_equal(P, ref["P"])


def test_a_delayed_image_chain_cannot_be_overtaken(gpu_required, stock_b, sync_ref):
    """the `corners` hand-off: the image chain of every other frame is held back by 600 us, so the NEXT frame's chain (other queue) finishes
    first — the refill half of book-keeping must still see its own frame's corner list"""
    from rvio_amd import hip
    cfg, imgs, imus = stock_b["cfg"], stock_b["imgs"], stock_b["imus"]
    for which in (1, 3):
        h = hip.RvioHip(cfg)
        h.initialize(*stock_b["init"])
        poses = []
        for i in range(60):
            if i >= 2 and i % 4 == (0 if which == 1 else 1):
                h.stall(which, 600)
            h.frame(imgs[i], imus[i], None)
            if i % 8 == 7:
                p, q = h.pose()
        h.sync()
        x, P = h.get_state()
        pts, hl = h.get_points()
        info = h.frame_info()
        h.close()
        ref = sync_ref["host"]
        assert info["device_error"] == 0
        assert np.array_equal(pts, ref["pts"]) and np.array_equal(hl, ref["hl"]), which
        assert np.array_equal(x, ref["x"]) and np.array_equal(P, ref["P"]), which


def test_left_over_state_is_never_read(gpu_required, stock_b, sync_ref):
    """NaN in every scratch buffer, in the spare state / covariance buffer, in the hand-over tables beyond their counts, in the tracker's
    per-frame scratch and in the LDS of the whole chip between the frames (rvio_hip_debug_poison): not one bit of any result moves"""
    r = run_hip(stock_b, "host", True, sync_every=True, n=60, poison=7)
    assert same_bits(r, sync_ref["host"]), first_diff(r, sync_ref["host"])


def test_left_over_state_is_never_read_on_the_long_window(gpu_required, stock_c):
    """the same at cfg C (6n = 120): there the Cholesky factor of the NEXT update's clone block is handed from one frame to the next through the
    solve's slab (its own queue, an event in front of the solve) — the poison fills that slab too, so the hand-over is dropped with it and the
    solve factors the clone block itself: the same bits as the synchronised run (round 5's advisor finding: the flags used to survive the fill)"""
    r = run_hip(stock_c, "host", True, sync_every=True, poison=7)
    assert same_bits(r, stock_c["ref"]), first_diff(r, stock_c["ref"])


def test_paranoid_mode_gives_the_same_bits(gpu_required, stock_b, sync_ref, tmp_path):
    """RVIO_PARANOID=1 (read when the library is loaded: a child process): default-flag events, no device-side polls, plain streams, host
    waits behind the staging copies, every frame drained — the A/B of every non-default mechanism in one switch"""
    np.savez(str(tmp_path / "ref.npz"), poses=sync_ref["host"]["poses"], x=sync_ref["host"]["x"], P=sync_ref["host"]["P"])
    code = ("import sys, numpy as np; sys.path.insert(0, %r); import test_gpu_flatout as T\n"
            "cfg, seq, imgs, imus = T.stock('B', 60)\n"
            "d = dict(cfg=cfg, init=seq.init_from_static(T.K0), imgs=imgs, imus=imus)\n"
            "r = T.run_hip(d, 'host', True)\n"
            "ref = np.load(%r)\n"
            "assert np.array_equal(r['poses'], ref['poses']) and np.array_equal(r['x'], ref['x']) and np.array_equal(r['P'], ref['P'])\n"
            "print('PARANOID_OK')\n") % (os.path.dirname(os.path.abspath(__file__)), str(tmp_path / "ref.npz"))
    env = dict(os.environ, RVIO_PARANOID="1")
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env)
    assert out.returncode == 0 and "PARANOID_OK" in out.stdout, out.stderr[-2000:]


def test_recovery_after_a_stage_counter_time_out(gpu_required, stock_b):
    """error bit 4 (a device-side stage counter timed out) makes rvio_hip_sync fail with RVIO_ERR_STATE; rvio_hip_initialize is the way out —
    it drains with plain waits, resets counters and targets, clears the flag — and the handle then runs like a fresh one"""
    from rvio_amd import hip
    cfg, imgs, imus = stock_b["cfg"], stock_b["imgs"], stock_b["imus"]
    h = hip.RvioHip(cfg)
    h.initialize(*stock_b["init"])
    for i in range(8):
        h.frame(imgs[i], imus[i], None)
    h.poison(8)
    with pytest.raises(hip.RvioHipError):
        h.sync()
    h.initialize(*stock_b["init"])
    h.sync()
    for i in range(12):
        h.frame(imgs[i], imus[i], None)
    h.sync()
    xa, Pa = h.get_state()
    h.close()
    h2 = hip.RvioHip(cfg)
    h2.initialize(*stock_b["init"])
    for i in range(12):
        h2.frame(imgs[i], imus[i], None)
    h2.sync()
    xb, Pb = h2.get_state()
    h2.close()
    assert np.array_equal(xa, xb) and np.array_equal(Pa, Pb)
