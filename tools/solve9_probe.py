"""Parity + live time of the solve kernel on a full-load update at the BASELINE windows (cfg B 6n = 60, A 84, C 120, E 180).

    python tools/solve9_probe.py [B A C E]          # RVIO_SOLVE7=1 in the environment selects the register-tableau elimination (solve7.hip)

Per config: the full-load update of SURVEY.md 8(d) (ceil(F/2) features, half type '2') on the device against the oracle (state, covariance),
then HIP-event timing of the solve kernel alone on those inputs (rvio_hip_debug_time_kernel(0))."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle as O       # noqa: E402  (the checker; this is a measurement tool, not product code)
import scenarios as S    # noqa: E402
from rvio_amd import hip  # noqa: E402

abi, rv = O.abi, O.rv
names = sys.argv[1:] or ["B", "A", "C", "E"]
for name in names:
    cfg = abi.config_named(name, enable_equalizer=0)
    nfr = cfg.max_track_len + 6
    seq, recs = S.record_sequence(cfg, n_frames=nfr, duration=(38 + nfr + 4) / 20.0)
    r = recs[-1]
    n_feat = None
    types, lens, meas = S.worst_case_tracks(cfg, r, seq, n_feat=n_feat, mix="half")
    t0 = time.time()
    xo, Po, dg = O.update(cfg, r["x1"], r["P1"], types, lens, meas)
    t_or = time.time() - t0
    h = hip.RvioHip(cfg)
    h.set_state(r["x1"], r["P1"])
    h.update(types, lens, meas)
    x, P = h.get_state()
    info = h.frame_info()
    dxs, dP = S.state_delta(x, xo), float(np.max(np.abs(P - Po)))
    us = h.time_kernel(0, 50)
    print("cfg %s 6n=%d: accepted %d/%d rows %d | state delta %.2e  P delta %.2e (|P| %.1e) err=%d | solve kernel %.1f us (form: %s) | oracle update %.0f ms"
          % (name, 6 * (cfg.max_track_len - 1), info["n_feat_accepted"], len(types), info["n_rows"], dxs, dP, float(np.max(np.abs(Po))), info["reserved"][0] if "reserved" in info else -1,
             us, "solve7" if os.environ.get("RVIO_SOLVE7") else "solve9", 1e3 * t_or), flush=True)
    if hasattr(h.L, "rvio_hip_debug_clocks") and os.environ.get("RVIO_HIP_LIB", "").endswith("dbg.so"):
        import ctypes as C
        if not os.environ.get("RVIO_PROBE_CHAIN"):   # default: the stamps of the update's own solve launch (the full kernel); RVIO_PROBE_CHAIN=1: of the launches just timed (the chain's form)
            h.set_state(r["x1"], r["P1"]); h.update(types, lens, meas); h.sync()
        out = (C.c_longlong * 64)()
        h.L.rvio_hip_debug_clocks(h.h, out)
        t = np.array(list(out))
        ph = [30, 31, 32, 33, 34, 35, 36, 37, 38]
        nm = ["P0 load", "P1 chol", "P2 Q", "P3 M", "P4 sweep", "P5 X", "P6 W+y", "dx+inject"]
        print("   phases (cycles): " + "  ".join("%s %d" % (nm[i], t[ph[i + 1]] - t[ph[i]]) for i in range(8)) + "  total %d" % (t[38] - t[30]))
        print("   first Cholesky step: factor %d, panel + trailing %d" % (t[41] - t[40], t[42] - t[41]))
        print("   second sweep step: factor + barrier %d, Z + trailing %d; sweep launch (34 -> 35) %d; dx launch (37 -> 38) %d   [the split form stamps per launch: only differences inside one launch mean anything]"
              % (t[44] - t[43], t[45] - t[44], t[35] - t[34], t[38] - t[37]))
        print("   sweep steps (start to start): " + " ".join("%d" % (t[47 + k] - t[46 + k]) for k in range(11) if t[47 + k] > 0) + "; launch start -> first step %d" % (t[46] - t[34]))
    h.close()
