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
np.savez_compressed(os.path.join(HERE, "cfgB_direct_seed0_frame30.npz"), x3=r["x3"], P3=r["P3"], x1=r["x1"], P1=r["P1"],
                    x2=r["x2"], P2=r["P2"], types=r["types"], lens=r["lens"], meas=r["meas"],
                    accepted=r["diag"]["accepted"], gamma=r["diag"]["gamma"], pts=r["pts"])
print("written")
