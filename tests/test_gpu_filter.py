"""GPU parity: the HIP filter stages (propagate, update, augment+compose) against the CPU
oracle on identical inputs, through the C-ABI.  Tolerance: per-state |delta| <= 1e-6 is the
north-star bar (BASELINE.json); the tests assert 1e-9 on states and 1e-12 + 1e-9*|P| on P."""
import numpy as np
import pytest

import oracle as O
import scenarios as S

abi, rv = O.abi, O.rv
pytestmark = pytest.mark.gpu

X_TOL = 1e-9


def p_close(Pa, Pb):
    scale = np.max(np.abs(Pb))
    return float(np.max(np.abs(Pa - Pb))) <= 1e-9 * scale + 1e-15


@pytest.fixture(scope="module")
def recB():
    cfg = abi.config_named("B", enable_equalizer=0)
    seq, recs = S.record_sequence(cfg, n_frames=30)
    return cfg, seq, recs


@pytest.fixture(scope="module")
def hipB(gpu_required, recB):
    from rvio_amd import hip
    h = hip.RvioHip(recB[0])
    yield h
    h.close()


def test_set_get_state_roundtrip(hipB, recB):
    r = recB[2][-1]
    hipB.set_state(r["x0"], r["P0"])
    x, P = hipB.get_state()
    assert np.array_equal(x, r["x0"]) and np.array_equal(P, r["P0"])


@pytest.mark.parametrize("fi", [0, 1, 3, 9, 15, 29])
def test_propagate_parity(hipB, recB, fi):
    r = recB[2][fi]
    hipB.set_state(r["x0"], r["P0"])
    hipB.propagate(r["inp"]["imu"])
    x, P = hipB.get_state()
    assert S.state_delta(x, r["x1"]) <= X_TOL
    assert p_close(P, r["P1"])
    assert np.allclose(P, P.T, rtol=0, atol=1e-18 + 1e-14 * np.max(np.abs(P)))


@pytest.mark.parametrize("fi", [5, 9, 12, 15, 20, 29])
def test_update_parity(hipB, recB, fi):
    r = recB[2][fi]
    assert r["did_update"]
    hipB.set_state(r["x1"], r["P1"])
    hipB.update(r["types"], r["lens"], r["meas"])
    x, P = hipB.get_state()
    dg = hipB.update_diag()
    od = r["diag"]
    assert np.array_equal(dg["accepted"], od["accepted"]), (dg["gamma"], od["gamma"])
    assert np.array_equal(dg["ndof"][od["accepted"] > 0], od["ndof"][od["accepted"] > 0])
    assert np.allclose(dg["gamma"], od["gamma"], rtol=1e-7, atol=1e-9)
    assert np.allclose(dg["pfinv"], od["pfinv"], rtol=1e-8, atol=1e-10)
    assert S.state_delta(x, r["x2"]) <= X_TOL
    assert p_close(P, r["P2"])


@pytest.mark.parametrize("fi", [0, 1, 2, 5, 12, 20])
def test_augment_compose_parity(hipB, recB, fi):
    r = recB[2][fi]
    hipB.set_state(r["x2"], r["P2"])
    hipB.augment_compose(r["do_augment"])
    x, P = hipB.get_state()
    assert len(x) == len(r["x3"])
    assert S.state_delta(x, r["x3"]) <= 1e-12
    assert p_close(P, r["P3"])
    p, q = hipB.pose()
    assert np.max(np.abs(p - r["pose_p"])) <= 1e-12


def test_update_full_load(hipB, recB):
    """ceil(F/2)=100 features, half at max track length: the BASELINE '200 feat / 10 clone' load."""
    cfg, seq, recs = recB
    r = recs[-1]
    types, lens, meas = S.worst_case_tracks(cfg, r, seq)
    xo, Po, od = O.update(cfg, r["x1"], r["P1"], types, lens, meas)
    assert od["n_good"] > 50
    hipB.set_state(r["x1"], r["P1"])
    hipB.update(types, lens, meas)
    x, P = hipB.get_state()
    dg = hipB.update_diag()
    assert np.array_equal(dg["accepted"], od["accepted"])
    assert S.state_delta(x, xo) <= X_TOL
    assert p_close(P, Po)


def test_update_passthrough_when_too_few(hipB, recB):
    cfg, seq, recs = recB
    r = recs[-1]
    types, lens, meas = r["types"][:2], r["lens"][:2], r["meas"][:2]
    hipB.set_state(r["x1"], r["P1"])
    hipB.update(types, lens, meas)
    x, P = hipB.get_state()
    assert np.array_equal(x, r["x1"]) and np.array_equal(P, r["P1"])
    assert hipB.frame_info()["updated"] == 0


def test_sequence_free_running(hipB, recB):
    """whole MonoVIO body (direct-track mode) for 30 frames without resets."""
    cfg, seq, recs = recB
    from rvio_amd import hip
    h = hip.RvioHip(cfg)
    w, a, n = seq.init_from_static(38)
    h.initialize(w, a, n)
    worst = 0.0
    for r in recs:
        inp = r["inp"]
        h.frame_points(inp["tracked"], inp["status"], inp["imu"], inp["cand"])
        x, P = h.get_state()
        info = h.frame_info()
        assert info["n_clones"] == (len(r["x3"]) - 26) // 7
        worst = max(worst, S.state_delta(x, r["x3"]))
        pts, hl = h.get_points()
        assert np.array_equal(pts, r["pts"]) and np.array_equal(hl, r["hist_len"])
    h.close()
    assert worst <= 1e-6, worst


@pytest.mark.parametrize("world", [2, 8])
def test_sharded_updater_fake_shards_on_one_gpu(hipB, recB, world):
    """SURVEY.md 8e 'fake shard' mode: the `world` shards run one after the other on one GPU, their [A|b]
    blocks are concatenated (what the all-gather would deliver) and the global stage is applied."""
    import torch
    cfg, seq, recs = recB
    r = recs[-1]
    types, lens, meas = S.worst_case_tracks(cfg, r, seq)
    xo, Po, od = O.update(cfg, r["x1"], r["P1"], types, lens, meas)
    hipB.set_state(r["x1"], r["P1"])

    class DA:
        def __init__(self, ptr, n):
            self.__cuda_array_interface__ = {"shape": (n,), "typestr": "<f8", "data": (ptr, False), "version": 2}
    blocks = []
    for rk in range(world):
        ptr, n = hipB.update_local(types, lens, meas, rk, world)
        hipB.sync()
        blocks.append(torch.as_tensor(DA(ptr, n), device="cuda").clone())
    allb = torch.cat(blocks).contiguous()
    torch.cuda.synchronize()
    # block of rank k equals the oracle's block: type-'2' part, type-'1' part, counters (information form is order-insensitive up to rounding)
    nb = blocks[0].numel()
    ob = O.update_local(cfg, r["x1"], r["P1"], types, lens, meas, 1, world)
    gb = blocks[1].cpu().numpy()
    ldh = 6 * (cfg.max_track_len - 1) + 1
    part = ldh * (ldh - 1)
    c6 = 6 * ((len(r["x1"]) - 26) // 7)
    # the payload is the WIRE FORMAT of csrc/rvio_dev.h shard_layout (abi.shard_pack is its NumPy mirror): 8 counters, then the 16 x 16 tiles of the
    # type-'2' part and of the type-'1' part that can be non-zero, tiles on and above the diagonal only (A is symmetric: mirrored after the sum)
    assert nb == abi.shard_payload_doubles(c6, cfg.max_track_len) and len(ob) == 2 * part + 8
    assert nb * 8 <= 0.55 * 2 * ldh * ldh * 8          # (rounds 2-5 gathered 2 (6n + 1)^2 doubles)
    parts = np.stack([ob[pt * part: (pt + 1) * part].reshape(ldh - 1, ldh) for pt in range(2)])
    want, live = abi.shard_pack(parts, ob[2 * part: 2 * part + 8], c6, cfg.max_track_len)
    assert np.allclose(gb[live][8:], want[live][8:], rtol=1e-9, atol=1e-9 * np.max(np.abs(ob[: 2 * part])))
    assert np.array_equal(gb[:5], want[:5])
    # what the wire format leaves out of the type-'2' part is zero in the oracle's block too
    back, _ = abi.shard_unpack(want, c6, cfg.max_track_len, ldh)
    assert np.allclose(back[0][:c6, : c6 + 1], parts[0][:c6, : c6 + 1], atol=1e-12 * np.max(np.abs(parts)))
    hipB.update_global(allb.data_ptr(), world)
    x, P = hipB.get_state()
    assert S.state_delta(x, xo) <= X_TOL
    assert p_close(P, Po)
