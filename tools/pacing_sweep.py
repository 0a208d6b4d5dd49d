"""Pacing sweep of the pipelined host-buffer frame (rvio_hip_frame + rvio_hip_get_pose per image: what host/rvio_replay does).

    python tools/pacing_sweep.py [--config A] [--frames 40] [--delays 0,50,100,...] [--repeat 3]

Background.  The tests drive the pipelined frame either fully synchronised (h.sync() behind every frame) or flat out (bench.py); a host that
reads images from disk sits in between: the next call arrives some hundreds of microseconds after the previous one, while the image chain
and the refill half of book-keeping of the previous frame may still be in flight.  Every dependency between the streams of a handle has to
hold at EVERY pacing, so the poses of a paced run must be those of the synchronised run bit for bit.  (Round 3: one full `pytest -m gpu` run
out of six had the three host-binary tests off by 5e-3 on one box — the piped runs only, the staged runs of the same binary were right —
and it did not reproduce on four other boxes; this tool is the reproducer to start from.)

For each delay d the sequence is replayed from the same initial state with a busy-wait of d microseconds between the calls; printed: the
largest |pose - pose_sync| over the sequence (0 expected), and the first frame that differs."""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

rv, abi = bench.rv, bench.abi
from rvio_amd import hip  # noqa: E402


def replay(cfg, init, frames, delay_us, sync_every_frame):
    h = hip.RvioHip(cfg)
    h.initialize(*init)
    poses = []
    for img, imu in frames:
        h.frame(img, imu, None)
        if sync_every_frame:
            h.sync()
        p, q = h.pose()                      # waits for the filter stream only
        poses.append(np.concatenate((p, q)))
        if delay_us > 0:
            t_end = time.perf_counter() + 1e-6 * delay_us
            while time.perf_counter() < t_end:
                pass
    h.sync()
    info = h.frame_info()
    h.close()
    return np.array(poses), info


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="A")
    ap.add_argument("--frames", type=int, default=40)
    ap.add_argument("--delays", default="0,50,100,150,200,250,300,400,500,700,1000,1500,2500")
    ap.add_argument("--repeat", type=int, default=3)
    ap.add_argument("--no-equalizer", action="store_true")
    args = ap.parse_args()
    cfg = abi.config_named(args.config, enable_equalizer=0 if args.no_equalizer else 1)
    k0 = bench.K0
    seq = rv.synth.SynthSequence(cfg, duration=(k0 + args.frames + 4) / 20.0, seed=0)
    init = seq.init_from_static(k0)
    frames = [(seq.render(k), seq.imu_between(k)) for k in range(k0 + 1, k0 + 1 + args.frames)]
    ref, info = replay(cfg, init, frames, 0, True)
    print("synchronised run: %d frames, device_error %s" % (len(ref), info.get("device_error")))
    bad = 0
    for d in [int(v) for v in args.delays.split(",") if v]:
        for r in range(args.repeat):
            got, info = replay(cfg, init, frames, d, False)
            diff = np.abs(got - ref).max(axis=1)
            first = int(np.argmax(diff > 0)) if np.any(diff > 0) else -1
            flag = "" if first < 0 else "   <-- differs from frame %d on" % first
            bad += first >= 0
            print("delay %5d us  run %d: max |pose - pose_sync| = %.3e  device_error %s%s" % (d, r, float(diff.max()), info.get("device_error"), flag))
    print("paced runs that differ from the synchronised run: %d" % bad)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
