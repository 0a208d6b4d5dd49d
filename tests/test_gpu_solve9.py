"""solve9.hip — the solve as blocked SPD factorisations (Cholesky of the clone block, symmetric sweep of M = s2 I + L^T A L, Woodbury).

The solve of an update runs in one of three forms on a plain handle:
  * the generic kernel with its own Cholesky phase             (an update that no propagate / fused per-feature launch precedes: set_state -> update),
  * the generic kernel behind the Cholesky ROLE workgroup      (6n in 65..96; propagate -> update: the role rides in propagate's launch),
  * solve9_small_kernel behind the role (6n <= 64, all in LDS) (the same sequence at the headline window; W^T form of the Woodbury product).
Every form must give the oracle's update (stage bar 1e-9, measured 1e-14), and the forms must agree with each other: bit for bit where only
the place of the Cholesky differs (same tiles, same order of operations per tile), to rounding where the Woodbury association differs."""
import numpy as np
import pytest

import oracle as O
import scenarios as S

abi, rv = O.abi, O.rv
pytestmark = pytest.mark.gpu

WINDOWS = {"B": 34, "A": 34, "C": 46, "E-shaped": 66}


def _cfg(name):
    return abi.config_named("E", enable_equalizer=0, n_features=400) if name == "E-shaped" else abi.config_named(name, enable_equalizer=0)


@pytest.fixture(scope="module", params=list(WINDOWS))
def case(request, gpu_required):
    from rvio_amd import hip
    name = request.param
    cfg = _cfg(name)
    seq, recs = S.record_sequence(cfg, n_frames=WINDOWS[name], duration=6.0)
    h = hip.RvioHip(cfg)
    yield name, cfg, seq, recs, h
    h.close()


def test_update_behind_the_cholesky_role_equals_the_update_with_its_own_cholesky(case):
    """propagate -> update (the role workgroup of propagate's launch factors the clone block, the solve starts at Q = A L) against
    set_state(propagated state) -> update (the solve kernel factors it itself) — and both against the oracle, at full load too"""
    name, cfg, seq, recs, h = case
    r = recs[-1]
    assert (len(r["x1"]) - 26) // 7 == cfg.max_track_len - 1
    loads = [(r["types"], r["lens"], r["meas"])]
    if name != "E-shaped":
        loads.append(S.worst_case_tracks(cfg, r, seq, mix="half"))
    for types, lens, meas in loads:
        xo, Po, dg = O.update(cfg, r["x1"], r["P1"], types, lens, meas)
        # own Cholesky
        h.set_state(r["x1"], r["P1"])
        h.update(types, lens, meas)
        xa, Pa = h.get_state()
        # behind the role: the device's own propagate, then the update on ITS state (so compare against the same state through the other form)
        h.set_state(r["x0"], r["P0"])
        h.propagate(r["inp"]["imu"])
        x1d, P1d = h.get_state()
        h.update(types, lens, meas)
        xb, Pb = h.get_state()
        h.set_state(x1d, P1d)
        h.update(types, lens, meas)
        xc, Pc = h.get_state()
        assert h.frame_info()["device_error"] == 0, name      # (bit 1: a pivot of the Cholesky / the sweep was not positive where it had to be)
        assert S.state_delta(xa, xo) <= 1e-9 and np.max(np.abs(Pa - Po)) <= 1e-9 * np.max(np.abs(Po)), name
        if 6 * (cfg.max_track_len - 1) > 64:
            # only the place of the Cholesky differs (6n <= 96), or nothing at all (longer windows: no role): the same bits
            assert np.array_equal(xb, xc) and np.array_equal(Pb, Pc), name
        else:
            # 6n <= 64: the all-LDS kernel behind the role associates the Woodbury product the other way round
            assert S.state_delta(xb, xc) <= 1e-12 and np.max(np.abs(Pb - Pc)) <= 1e-12 * np.max(np.abs(Pc)), name


def test_window_filling_and_zero_variance_clones(gpu_required):
    """every window size on the way to a full window (6n = 18, 24, ... pads the 16 x 16 tiles differently each time), free-running against
    the oracle; then clones with exactly zero covariance (set by hand: rows / columns of the two newest clones zeroed) — a semi-definite
    clone block, the case the Cholesky's zero-direction rule exists for"""
    from rvio_amd import hip
    cfg = abi.config_named("B", enable_equalizer=0)
    seq, recs = S.record_sequence(cfg, n_frames=30)
    h = hip.RvioHip(cfg)
    sizes = set()
    for r in recs:
        if not r["did_update"]:
            continue
        h.set_state(r["x0"], r["P0"])
        h.propagate(r["inp"]["imu"])
        h.update(r["types"], r["lens"], r["meas"])
        x, P = h.get_state()
        assert S.state_delta(x, r["x2"]) <= 1e-9 and np.max(np.abs(P - r["P2"])) <= 1e-9 * np.max(np.abs(r["P2"])), r["k"]
        sizes.add((len(x) - 26) // 7)
    assert len(sizes) >= 6, sizes
    r = recs[-1]
    P1 = r["P1"].copy()
    P1[-12:, :] = 0
    P1[:, -12:] = 0
    xo, Po, dg = O.update(cfg, r["x1"], P1, r["types"], r["lens"], r["meas"])
    h.set_state(r["x1"], P1)
    h.update(r["types"], r["lens"], r["meas"])
    x, P = h.get_state()
    assert dg["updated"] and h.frame_info()["updated"] == 1       # (a full load of tracks on the last record: the update is applied on both sides)
    assert h.frame_info()["device_error"] == 0                    # the zero directions are not errors
    assert np.all(np.isfinite(x)) and np.all(np.isfinite(P))
    assert S.state_delta(x, xo) <= 1e-9 and np.max(np.abs(P - Po)) <= 1e-9 * np.max(np.abs(Po))
    h.close()
