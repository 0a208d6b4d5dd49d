import os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = r'''
import sys, os
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tests"))
import numpy as np, oracle as O, scenarios as S
from rvio_amd import hip
cfg = O.abi.config_named("B", enable_equalizer=0)
seq, recs = S.record_sequence(cfg, n_frames=20)
r = recs[-1]
h = hip.RvioHip(cfg); h.set_state(r["x1"], r["P1"])
ty, le, me = S.worst_case_tracks(cfg, r, seq)
h.update(ty, le, me); h.sync()
print(os.environ.get("RVIO_HIP_LIB","").split("/")[-1], "solve us:", round(h.time_kernel(0, 30), 2))
''' % (ROOT, ROOT)
for v in sorted(os.listdir(os.path.join(ROOT, "r-vio_amd", "variants"))):
    env = dict(os.environ, RVIO_HIP_LIB=os.path.join(ROOT, "r-vio_amd", "variants", v))
    subprocess.run([sys.executable, "-c", code], env=env)
