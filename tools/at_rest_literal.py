"""the platform at rest, free-running on the GPU against the LITERAL oracle — once with the device's own choice of
compression (information form + structural rank rule; the literal sweep only where lit_decide asks for it) and once with the literal sweep
FORCED for every update it can take (<= 24 features handed over, > 2 accepted, tall stack).  Reports the largest state difference and how
many updates ran the literal sweep, on images (80 frames) and on direct tracks (100 frames), the sequences of tests/test_gpu_truncation.py."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle as O      # noqa: E402
import scenarios as S   # noqa: E402
import torch            # noqa: E402,F401

abi, rv = O.abi, O.rv
from rvio_amd import hip  # noqa: E402


def run(kind, force):
    if kind == "images":
        cfg = abi.config_named("B", enable_equalizer=1)
        n = 80
        seq = rv.synth.SynthSequence(cfg, duration=(38 + n + 4) / 20.0, seed=2, motion="stationary")
    else:
        cfg = abi.config_named("B", enable_equalizer=0)
        n = 100
        seq = rv.synth.SynthSequence(cfg, duration=(38 + n + 4) / 20.0, seed=4, drop_prob=0.15, motion="stationary")
    w, a, ni = seq.init_from_static(38)
    x0, P0 = O.initialize(cfg, w, a, ni)
    h = hip.RvioHip(cfg)
    h.L.rvio_hip_debug_literal_force(h.h, 1 if force else 0)
    h.initialize(w, a, ni)
    lit = O.System(cfg)
    lit.set_state(x0, P0)
    drv = rv.synth.DirectTrackDriver(seq) if kind != "images" else None
    worst, updates, literal, feasible = 0.0, 0, 0, 0
    for k in range(39, 39 + n):
        if kind == "images":
            img, imu = seq.render(k), seq.imu_between(k)
            lit.frame(imu, None, img=img)
            h.frame(img, imu, None)
        else:
            inp = drv.inputs(k)
            lit.frame(inp["imu"], inp["cand"], tracked=inp["tracked"], status=inp["status"])
            h.frame_points(inp["tracked"], inp["status"], inp["imu"], inp["cand"])
            drv.after(h.get_points()[0])
        h.sync()
        gi = h.frame_info()
        updates += gi["updated"]
        literal += 1 if gi.get("literal_rank", -1) >= 0 else 0
        feasible += 1 if (gi["updated"] and gi["n_feat_update"] <= 24 and gi["n_feat_accepted"] > 2) else 0
        worst = max(worst, S.state_delta(h.get_state()[0], lit.get_state()[0]))
    h.L.rvio_hip_debug_literal_force(h.h, 0)
    h.close()
    return worst, updates, literal, feasible


for kind in ("images", "direct tracks"):
    for force in (False, True):
        worst, updates, literal, feasible = run(kind, force)
        print("at rest, %-13s literal sweep %-6s: free-running max state delta vs the literal oracle %.3e | %d updates, %d of them through the literal sweep "
              "(%d handed <= 24 features with > 2 accepted)" % (kind, "FORCED" if force else "as is", worst, updates, literal, feasible))
