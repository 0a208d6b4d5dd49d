#!/usr/bin/env python3
"""Regenerates tests/golden/*.npz from the CPU oracle (the reference ships no golden vectors and cannot be
built here — SURVEY.md 8c — so these snapshots pin the oracle against itself: a regression guard)."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import oracle as O  # noqa: E402
import scenarios as S  # noqa: E402

cfg = O.abi.config_named("B", enable_equalizer=0)
seq, recs = S.record_sequence(cfg, n_frames=30)
r = recs[-1]
imu = np.ascontiguousarray(r["inp"]["imu"])
np.savez_compressed(os.path.join(HERE, "cfgB_direct_seed0_frame30.npz"), x3=r["x3"], P3=r["P3"], x1=r["x1"], P1=r["P1"],
                    x2=r["x2"], P2=r["P2"], types=r["types"], lens=r["lens"], meas=r["meas"],
                    accepted=r["diag"]["accepted"], gamma=r["diag"]["gamma"], pts=r["pts"],
                    x0=r["x0"], P0=r["P0"], imu=imu.view(np.uint8), do_augment=int(r["do_augment"]))   # inputs: the GPU test replays the stages from the file alone


import zlib  # noqa: E402
cfg2 = S.small_image_config()
seq2 = O.rv.synth.SynthSequence(cfg2, duration=5.0, n_landmarks=1500)
t = O.Tracker(cfg2)
imgs, imus, pts, hls = [], [], [], []
for k in (60, 61, 62, 63):
    img, u = seq2.render(k), np.ascontiguousarray(seq2.imu_between(k))
    t.track(img, u, None)                       # CLAHE + DetectWithSubPix + KLT + RANSAC + book-keeping
    p_, h_ = t.get_points()
    imgs.append(img); imus.append(u.view(np.uint8)); pts.append(p_); hls.append(h_)
corners0 = O.detect(cfg2, O.clahe(imgs[0]), 1)
out = dict(imgs=np.stack(imgs), corners0=corners0, clahe0_crc=np.uint32(zlib.crc32(O.clahe(imgs[0]).tobytes())))
for i in range(4):
    out["imu%d" % i], out["pts%d" % i], out["hist%d" % i] = imus[i], pts[i], hls[i]
np.savez_compressed(os.path.join(HERE, "small_images_tracker.npz"), **out)
print("written", os.path.getsize(os.path.join(HERE, "small_images_tracker.npz")), "bytes for the image fixture;", [len(p_) for p_ in pts], "points")
