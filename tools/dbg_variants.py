"""A/B timing of the solve kernel generations on the worst-case update of config B (GPU box only)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = r'''
import sys, os
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tests"))
import numpy as np, oracle as O, scenarios as S
from rvio_amd import hip
cfg = O.abi.config_named(os.environ.get("CFG", "B"), enable_equalizer=0)
seq, recs = S.record_sequence(cfg, n_frames=20)
r = recs[-1]
h = hip.RvioHip(cfg); h.set_state(r["x1"], r["P1"])
ty, le, me = S.worst_case_tracks(cfg, r, seq)
h.update(ty, le, me); h.sync()
print(os.environ.get("TAG"), "solve us:", round(h.time_kernel(0, 30), 2))
''' % (ROOT, ROOT)
for tag, extra in (("solve6 (8 waves)", {}), ("solve4", {"RVIO_SOLVE4": "1"})):   # the 4- and 16-wave solve6 variants lost (49 vs 57 / 66 us) and were removed
    subprocess.run([sys.executable, "-c", code], env=dict(os.environ, TAG=tag, **extra))
