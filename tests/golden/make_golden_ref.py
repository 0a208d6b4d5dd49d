#!/usr/bin/env python3
"""Writes tests/golden/ref_*.npz: the EXPECTED OUTPUTS of the two golden fixtures as computed by the reference's OWN sources
(oracle/_ref/libref.so = /root/reference/src/rvio/*.cc compiled unmodified against oracle/refshim/, `make -C oracle ref`) on the inputs the
oracle-written fixtures carry (cfgB_direct_seed0_frame30.npz, small_images_tracker.npz: their inputs are reused, nothing is stored twice).

Runs only where /root/reference exists (this container).  The GPU box has neither the reference nor a way to build it: its tests read
the committed files (tests/test_gpu_golden.py, both fixture families), and tests/test_golden_ref.py checks on the CPU that the two
families agree — and, where the reference is present, that the committed files are what this script writes today.

Filter stages: PreIntegrator::propagate -> Updater::update -> the System.cc:279-365 block, chained on the reference's own intermediate
results.  Front end: RVIO::Tracker::track on the four images (its OpenCV IMAGE calls forward to the oracle's restatements — the shim has no
other implementation — so what the reference contributes here is Tracker.cc / FeatureDetector.cc / Ransac.cc: undistortion, RANSAC, the
book-keeping, FindNewer / ChessGrid)."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import oracle as O  # noqa: E402
import scenarios as S  # noqa: E402
import ref as R  # noqa: E402

abi = O.abi


def filter_outputs():
    g = np.load(os.path.join(HERE, "cfgB_direct_seed0_frame30.npz"))
    cfg = abi.config_named("B", enable_equalizer=0)
    x1, P1 = R.propagate(cfg, g["x0"], g["P0"], g["imu"].view(abi.IMU_DTYPE))
    x2, P2, d = R.update(cfg, x1, P1, g["types"], g["lens"], g["meas"])
    x3, P3, pp, pq = R.augment_compose(cfg, x2, P2, bool(g["do_augment"]))
    return dict(x1=x1, P1=P1, x2=x2, P2=P2, x3=x3, P3=P3, pose_p=pp, pose_q=pq, n_cloud=np.int32(d["n_cloud"]),
                gate_rejects=np.int32(d["gate_rejects"]), invalid=np.int32(d["invalid"]), updated=np.int32(d["updated"]), cloud=d["cloud"])


def tracker_outputs():
    g = np.load(os.path.join(HERE, "small_images_tracker.npz"))
    cfg = S.small_image_config()
    t = R.Tracker(cfg)
    out = {}
    for i in range(4):
        t.track(g["imgs"][i], g["imu%d" % i].view(abi.IMU_DTYPE), None)
        out["pts%d" % i], out["hist%d" % i] = t.get_points()
        ty, ln, me = t.get_tracks()
        out["types%d" % i], out["lens%d" % i], out["meas%d" % i] = ty, ln, me
    return out


from golden_io import FULL_LOAD, load_full_load_case, probes  # noqa: E402


def full_load_inputs():
    """the worst-case update load of SURVEY.md 8(d) at the 14- / 10- / 20- / 30-clone windows (written ONCE, by the oracle's sequence
    recorder; stored because another host's libm may round a sine differently and the GPU box must see exactly these inputs)"""
    out = {}
    for name, mix, nf in FULL_LOAD:
        cfg = abi.config_named(name)
        seq, recs = S.record_sequence(cfg, n_frames=cfg.max_track_len + 4, duration=(38 + cfg.max_track_len + 8) / 20.0)
        r = recs[-1]
        types, lens, meas = S.worst_case_tracks(cfg, r, seq, n_feat=nf, mix=mix)
        d = r["P1"].shape[0]
        out.update({name + "_x1": r["x1"], name + "_P1u": r["P1"][np.triu_indices(d)], name + "_types": types, name + "_lens": lens,
                    name + "_meas": meas[:, : int(lens.max())].copy()})
    return out


def full_load_outputs(g):
    """Updater::update of the reference's own sources on the stored loads: state, diag P, P V, the size of the accepted set"""
    out = {}
    for name, _, _ in FULL_LOAD:
        cfg, x1, P1, types, lens, meas = load_full_load_case(g, name)
        x2, P2, d = R.update(cfg, x1, P1, types, lens, meas)
        assert d["updated"] == 1
        out.update({name + "_x2": x2, name + "_diagP2": np.diag(P2).copy(), name + "_P2V": P2 @ probes(P2.shape[0]), name + "_maxP2": np.float64(np.max(np.abs(P2))),
                    name + "_n_cloud": np.int32(d["n_cloud"])})
    return out


FREE_RUN_FRAMES = 30
FREE_RUN_TABLES = (9, 19, 29)      # frames whose feature tables are stored


def free_run():
    """System::MonoVIO of the reference's own sources, free-running for 30 direct-track frames from System::initialize (window filling,
    the first type-'2' features, the window sliding): the inputs of every frame (the simulated KLT result depends on the tracker's own
    feature table, so they are recorded while the reference runs), the reference's state after every frame, a digest of its final
    covariance, its feature tables at three frames.  A device that replays the inputs must end every frame where the reference did."""
    cfg = abi.config_named("B", enable_equalizer=0)
    n = FREE_RUN_FRAMES
    seq = O.rv.synth.SynthSequence(cfg, duration=(38 + n + 4) / 20.0, seed=0)
    w, a, ni = seq.init_from_static(38)
    x0, P0 = R.initialize(cfg, w, a, ni)
    sr = R.System(cfg)
    sr.set_state(x0, P0)
    drv = O.rv.synth.DirectTrackDriver(seq)
    out = dict(init_w=np.asarray(w, float), init_a=np.asarray(a, float), init_n=np.int32(ni), x0=x0, P0=P0)
    xs = np.zeros((n, 26 + 7 * (cfg.max_track_len - 1)))
    xlen, updated, ncloud = np.zeros(n, np.int32), np.zeros(n, np.int32), np.zeros(n, np.int32)
    for i, k in enumerate(range(39, 39 + n)):
        inp = drv.inputs(k)
        info, pp, pq = sr.frame(inp["imu"], inp["cand"], tracked=inp["tracked"], status=inp["status"])
        pts, hl = sr.get_points()
        drv.after(pts)
        x, P = sr.get_state()
        xs[i, : len(x)] = x
        xlen[i], updated[i], ncloud[i] = len(x), info["updated"], info["n_cloud"]
        out.update({"tracked%d" % i: np.ascontiguousarray(inp["tracked"], np.float32), "status%d" % i: np.ascontiguousarray(inp["status"], np.uint8),
                    "imu%d" % i: np.ascontiguousarray(inp["imu"]).view(np.uint8), "cand%d" % i: np.ascontiguousarray(inp["cand"], np.float32)})
        if i in FREE_RUN_TABLES:
            out.update({"pts%d" % i: pts, "hist%d" % i: hl})
    out.update(ref_x=xs, ref_xlen=xlen, ref_updated=updated, ref_n_cloud=ncloud, ref_diagP=np.diag(P).copy(), ref_PV=P @ probes(P.shape[0]),
               ref_maxP=np.float64(np.max(np.abs(P))))
    return out


if __name__ == "__main__":
    assert R.available(), "needs /root/reference (make -C oracle ref)"
    np.savez_compressed(os.path.join(HERE, "ref_cfgB_direct_seed0_frame30.npz"), **filter_outputs())
    np.savez_compressed(os.path.join(HERE, "ref_small_images_tracker.npz"), **tracker_outputs())
    fl = os.path.join(HERE, "full_load_inputs.npz")
    if not os.path.exists(fl) or "--inputs" in sys.argv:
        np.savez_compressed(fl, **full_load_inputs())
    np.savez_compressed(os.path.join(HERE, "ref_full_load_outputs.npz"), **full_load_outputs(np.load(fl)))
    np.savez_compressed(os.path.join(HERE, "ref_free_run_30_frames.npz"), **free_run())
    for f in ("ref_cfgB_direct_seed0_frame30.npz", "ref_small_images_tracker.npz", "full_load_inputs.npz", "ref_full_load_outputs.npz", "ref_free_run_30_frames.npz"):
        print("written", f, os.path.getsize(os.path.join(HERE, f)), "bytes")
