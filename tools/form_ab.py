"""A/B of two kernel forms on one full-load update: dumps the updated state / covariance so that two runs (e.g. the instrumented build with and without a
switch such as RVIO_NO_UG_TILE=1) can be compared bit for bit.

    RVIO_HIP_LIB=r-vio_amd/librvio_dbg.so [SWITCH=1] python tools/form_ab.py out.npz [A C E]; python tools/form_ab.py --compare a.npz b.npz"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
if sys.argv[1] == "--compare":
    a, b = np.load(sys.argv[2]), np.load(sys.argv[3])
    ok = True
    for k in a.files:
        same = np.array_equal(a[k], b[k])
        ok &= same
        print("%s: %s (max abs diff %.3e)" % (k, "identical" if same else "DIFFERENT", float(np.max(np.abs(a[k] - b[k])))))
    sys.exit(0 if ok else 1)
import oracle as O       # noqa: E402  (only for the scenario helpers' configuration; nothing of the oracle is measured here)
import scenarios as S    # noqa: E402
from rvio_amd import hip  # noqa: E402

out = {}
for name in sys.argv[2:] or ["A", "C", "E"]:
    cfg = O.abi.config_named(name, enable_equalizer=0)
    nfr = cfg.max_track_len + 6
    seq, recs = S.record_sequence(cfg, n_frames=nfr, duration=(38 + nfr + 4) / 20.0)
    r = recs[-1]
    types, lens, meas = S.worst_case_tracks(cfg, r, seq, n_feat=None, mix="half")
    h = hip.RvioHip(cfg)
    h.set_state(r["x1"], r["P1"])
    h.update(types, lens, meas)
    x, P = h.get_state()
    out["x_" + name], out["P_" + name] = x, P
    h.close()
np.savez(sys.argv[1], **out)
