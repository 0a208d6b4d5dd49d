"""The reference's rank truncation after the measurement-compression QR (Updater.cc:516-529) and its structural equivalent.

After its Givens QR the reference keeps only the LEADING rows of R whose norm is >= 1e-4 and stops at the first smaller row.
The device (and oracle/filter.cpp:orc_update_local/global, its CPU mirror) compresses in information form
[A|b] = Hw^T [Hw | r], i.e. with all rows, and reproduces the scan's effect from the structure of the stack:

  * the sweep treats exact zeros specially (makeGivens(0,q) swaps the rows, makeGivens(p,0) is the identity), so rows of
    type-'2' features (columns 0..e2, e2 = 6(ceil(L/2)-1)-1) and rows of type-'1' features that start behind e2 are never
    mixed while columns 0..e2 are swept;
  * the type-'2' block has the scale gauge of a monocular window as null direction, its column e2 is dependent: a left-over
    row of rounding residue (norm ~1e-15) arrives at position e2, the scan stops there (nRank = e2) and every type-'1' row —
    information included — is discarded;
  * in every other constellation the scan only drops rows that are zero to rounding.

These tests hold the structural rule against the LITERAL restatement (orc_update: sequential Givens + scan) on the stock image
workload, and show that the literal decision itself is stable (not decided by rounding noise): +-1 ulp on every entry of the
stacked Hw and random feature orders leave nRank and the updated state where they were."""
import os
import sys

import numpy as np
import pytest

import oracle as O
import scenarios as S

abi = O.abi


def _run(cfg, n, image=True, seed=0, keep=None, **kw):
    """free-running oracle sequence; yields one record per applied update: literal result, mirror result, inputs"""
    seq = O.rv.synth.SynthSequence(cfg, duration=(38 + n + 4) / 20.0, seed=seed, **kw)
    w, a, ni = seq.init_from_static(38)
    x, P = O.initialize(cfg, w, a, ni)
    trk = O.Tracker(cfg)
    drv = None if image else O.rv.synth.DirectTrackDriver(seq)
    img_count, out = 0, []
    for k in range(39, 39 + n):
        if image:
            imu = seq.imu_between(k)
            trk.track(seq.render(k), imu, None)
        else:
            inp = drv.inputs(k)
            imu = inp["imu"]
            trk.track_points(inp["tracked"], inp["status"], imu, inp["cand"])
            drv.after(trk.get_points()[0])
        img_count += 1
        ncl = (len(x) - 26) // 7
        x1, P1 = O.propagate(cfg, x, P, imu)
        types, lens, meas = trk.get_tracks()
        if ncl > cfg.min_track_len - 1:
            x2, P2, d = O.update(cfg, x1, P1, types, lens, meas)
            if d["updated"]:
                blk = O.update_local(cfg, x1, P1, types, lens, meas, 0, 1)
                xi, Pi, di = O.update_global(cfg, x1, P1, blk[None, :])
                out.append(dict(k=k, x1=x1, P1=P1, types=types, lens=lens, meas=meas, x2=x2, P2=P2, d=d, xi=xi, Pi=Pi, di=di, blk=blk))
        else:
            x2, P2 = x1, P1
        x, P, _, _ = O.augment_compose(cfg, x2, P2, img_count > 1)
    return out


def _without_any_rank_decision(cfg, r):
    """the update of record r in information form with EVERY row kept (no literal path, no structural rule): what round 1 shipped"""
    old = os.environ.get("ORC_LIT_FEATS")
    os.environ["ORC_LIT_FEATS"] = "0"
    try:
        blk = O.update_local(cfg, r["x1"], r["P1"], r["types"], r["lens"], r["meas"], 0, 1)
    finally:
        if old is None:
            del os.environ["ORC_LIT_FEATS"]
        else:
            os.environ["ORC_LIT_FEATS"] = old
    blk[-8 + 4] = 0          # pretend a type-'1' feature starts at column 0: the structural rule does not apply
    xa, _, da = O.update_global(cfg, r["x1"], r["P1"], blk[None, :])
    assert da["truncated_at"] == -1
    return xa


def _informative_cuts(cfg, recs):
    """the updates in which the reference's scan cut INFORMATION off (not just rounding residue): keeping every row moves the state by > 1e-8 (residue: < 1e-17)"""
    return [r for r in recs if r["di"]["truncated_at"] >= 0 and S.state_delta(_without_any_rank_decision(cfg, r), r["x2"]) > 1e-8]



@pytest.fixture(scope="module")
def stock_b():
    return _run(abi.config_named("B", enable_equalizer=1), 130)


def test_structural_rule_equals_the_literal_scan_on_the_stock_workload(stock_b):
    """every update of 130 stock frames (CLAHE + detector + KLT on rendered images): mirror == literal to rounding, and the
    mirror reports exactly the literal nRank whenever the scan cut informative rows off (frames 90, 108, 110 of this sequence)"""
    cfg = abi.config_named("B", enable_equalizer=1)
    for r in stock_b:
        assert S.state_delta(r["x2"], r["xi"]) < 1e-10, r["k"]
        assert np.max(np.abs(r["P2"] - r["Pi"])) < 1e-12, r["k"]
        if r["di"]["truncated_at"] >= 0:
            assert r["di"]["truncated_at"] == r["d"]["rank"], r["k"]
    # without any rank decision (every row kept) those updates differ by up to 1e-4 per state: the deviation round 1 shipped
    ev = _informative_cuts(cfg, stock_b)
    assert len(stock_b) > 100 and [r["k"] for r in ev] == [90, 108, 110], [r["k"] for r in ev]
    assert all(int(r["blk"][-8 + 5]) != 1 for r in ev)      # all three by the structural rule (the gap right behind a lone type-'2' block is its own constellation)
    worst = max(S.state_delta(_without_any_rank_decision(cfg, r), r["x2"]) for r in ev)
    assert 1e-6 < worst < 1e-3, worst


def test_structural_rule_on_the_stock_yaml_window_and_on_direct_tracks():
    """cfg A (14-clone window: e2 = 41) on images, cfg B / C direct-track sequences with many lost features"""
    ra = _run(abi.config_named("A", enable_equalizer=1), 120)
    assert max(S.state_delta(r["x2"], r["xi"]) for r in ra) < 1e-10
    assert all(r["di"]["truncated_at"] == r["d"]["rank"] for r in ra if r["di"]["truncated_at"] >= 0)
    ev = [r for r in ra if r["di"]["truncated_at"] >= 0 and r["blk"][-8 + 5] != 1]     # (!= 1: the structural rule, not the literal path of round 6)
    assert ev and all(r["d"]["rank"] == 41 for r in ev)
    for name, dp in (("B", 0.3), ("C", 0.1)):
        rd = _run(abi.config_named(name, enable_equalizer=0), 70, image=False, seed=1, drop_prob=dp)
        assert len(rd) > 40
        assert max(S.state_delta(r["x2"], r["xi"]) for r in rd) < 1e-10, name


def test_shards_agree_with_the_literal_scan(stock_b):
    """the rule is applied to the gathered whole: 3 shards (features f mod 3) give the literal result in the truncating frames too"""
    cfg = abi.config_named("B", enable_equalizer=1)
    for r in _informative_cuts(cfg, stock_b) + stock_b[60:62]:
        blks = np.stack([O.update_local(cfg, r["x1"], r["P1"], r["types"], r["lens"], r["meas"], rk, 3) for rk in range(3)])
        xw, Pw, dw = O.update_global(cfg, r["x1"], r["P1"], blks)
        assert dw["truncated_at"] == r["di"]["truncated_at"]
        assert S.state_delta(xw, r["x2"]) < 1e-10


def test_literal_rank_decision_is_not_decided_by_rounding_noise(stock_b):
    """Is the reference's decision reproducible at all?  Perturb every non-zero entry of the stacked Hw by +-1 ulp (12 draws) and
    permute the feature order (8 draws), in the three truncating frames and in five ordinary ones, and run the LITERAL sweep +
    scan on the perturbed stack.  Where the scan cuts information off, nRank and the state stay put (the short row is rounding
    residue, 1e-15 against a threshold of 1e-4): the decision is structural.  Elsewhere nRank may move with the feature order
    (how many mixture rows a dependent column leaves depends on the order) but the state moves by < 1e-6: nothing informative
    is dropped either way."""
    cfg = abi.config_named("B", enable_equalizer=1)
    rng = np.random.default_rng(0)
    ev = _informative_cuts(cfg, stock_b)
    plain = [r for r in stock_b if r["di"]["truncated_at"] < 0 and r["d"]["rank"] >= 0][20:60:8]
    assert len(ev) == 3 and len(plain) == 5
    for r in ev + plain:
        Hw, rr, ng = O.update_stack(cfg, r["x1"], r["P1"], r["types"], r["lens"], r["meas"])
        x0, _, d0 = O.update_from_stack(cfg, r["x1"], r["P1"], Hw, rr, ng)
        assert np.array_equal(x0, r["x2"]) and d0["rank"] == r["d"]["rank"]
        for _ in range(12):
            up = rng.integers(0, 2, Hw.shape) > 0
            Hp = np.where(Hw != 0, np.nextafter(Hw, np.where(up, np.inf, -np.inf)), 0.0)
            xp, _, dp = O.update_from_stack(cfg, r["x1"], r["P1"], Hp, rr, ng)
            assert dp["rank"] == d0["rank"], r["k"]
            assert S.state_delta(xp, x0) < 1e-12, r["k"]
        for _ in range(8):
            p = rng.permutation(len(r["types"]))
            xp, _, dp = O.update(cfg, r["x1"], r["P1"], r["types"][p], r["lens"][p], r["meas"][p])
            if r in ev:
                assert dp["rank"] == d0["rank"] and S.state_delta(xp, x0) < 1e-12, r["k"]
            else:
                assert S.state_delta(xp, x0) < 1e-6, r["k"]


# ---------------------------------------------------------------- windows the structural rule was NOT derived for (round-2 verdict)
# The rule assumes one null direction of the type-'2' block (the monocular scale) sitting at column e2.  These sequences take that
# assumption away — a platform at rest (zero parallax, every relative translation ~ IMU noise), pure rotation about the camera centre,
# a constant-velocity straight line without rotation, a scene in which every landmark has the same depth — and hold the mirror against the
# LITERAL sweep + scan (Updater.cc:494-529) in every update.
DEGENERATE = [dict(motion="stationary"), dict(motion="rotation"), dict(motion="line"), dict(scene="sphere"), dict(motion="line", scene="sphere")]


def _check(recs, tol):
    worst, cuts, low = 0.0, 0, 0
    for r in recs:
        worst = max(worst, S.state_delta(r["x2"], r["xi"]))
        assert worst <= tol, (r["k"], worst, r["d"]["rank"], r["di"]["truncated_at"])
        assert np.max(np.abs(r["P2"] - r["Pi"])) <= tol * max(1.0, np.max(np.abs(r["P2"]))), r["k"]
        if r["di"]["truncated_at"] >= 0:      # informative rows cut: the literal scan must have stopped at the same row
            assert r["di"]["truncated_at"] == r["d"]["rank"], r["k"]
            cuts += 1
        c6 = 6 * ((len(r["x1"]) - 26) // 7)
        low += r["d"]["rank"] < min(r["d"]["n_rows"], c6)
    return worst, cuts, low


@pytest.mark.parametrize("kw", DEGENERATE, ids=lambda kw: "-".join("%s" % v for v in kw.values()))
def test_structural_rule_on_degenerate_motions_direct_tracks(kw):
    """direct-track sequences (noisy projections, random drops => many type-'1' features of every length): 1e-10, like the stock motion"""
    for name, n in (("B", 110), ("A", 90)):
        recs = _run(abi.config_named(name, enable_equalizer=0), n, image=False, seed=2, **kw)
        assert len(recs) > n - 30
        worst, cuts, low = _check(recs, 1e-10)
        assert low > len(recs) // 2          # the literal scan does stop early in most of these updates (at 6n-1 or at e2): the cases exist


@pytest.mark.parametrize("kw", [dict(motion="rotation"), dict(motion="line")], ids=["rotation", "line"])
def test_structural_rule_on_degenerate_motions_images(kw):
    """the same on rendered images (CLAHE + detector + KLT): hundreds of stacked rows per update, the literal scan stopping at e2 = 29,
    41..59 or not at all.  Bar 1e-8: on the straight line the literal scan once drops two rows of norm just under its 1e-4 threshold that
    are not rounding residue (5e-10 per state — the threshold's own granularity, three orders below the 1e-6 bar)."""
    recs = _run(abi.config_named("B", enable_equalizer=1), 70, image=True, seed=2, **kw)
    assert len(recs) > 20
    _check(recs, 1e-8)


def test_structural_rule_on_randomised_stacks():
    """Random type / length mixes on a full window whose clone poses are made degenerate by hand: runs of IDENTICAL relative poses
    (duplicated clones: the platform at rest inside the window), zero relative translations (pure rotation), identical non-zero
    relative poses (constant velocity).  400 stacks of 3..100 features, tracks geometrically consistent with the modified window."""
    synth = O.rv.synth
    cfg = abi.config_named("B", enable_equalizer=0)
    base = [r for r in _run(cfg, 60, image=False, seed=3) if (len(r["x1"]) - 26) // 7 == 10][-1]
    rng = np.random.default_rng(5)
    n, Fu = 10, abi.fu(cfg)
    worst, applied, low = 0.0, 0, 0
    for trial in range(400):
        x, P = base["x1"].copy(), base["P1"]
        mode = trial % 4
        if mode == 1:
            a = int(rng.integers(0, n - 2))
            for c in range(a, int(rng.integers(a + 1, n))):
                x[26 + 7 * c: 33 + 7 * c] = [0, 0, 0, 1, 0, 0, 0]
        elif mode == 2:
            for c in range(n):
                x[30 + 7 * c: 33 + 7 * c] = 0
        elif mode == 3:
            for c in range(n):
                x[26 + 7 * c: 33 + 7 * c] = [0, 0, 0, 1, 0.02, 0.01, 0.0]
        nf = int(rng.integers(3, Fu + 1))
        ty, ln, me = synth.worst_case_tracks(cfg, x, n_feat=nf, seed=int(rng.integers(1 << 30)), mix="half")
        for f in range(nf):       # shorten a third of the type-'1' tracks at random (they keep their NEWEST observations)
            if ty[f] == ord("1") and rng.uniform() < 0.35:
                L = int(rng.integers(2, ln[f] + 1))
                me[f, :L] = me[f, ln[f] - L: ln[f]].copy()
                ln[f] = L
        x2, P2, d = O.update(cfg, x, P, ty, ln, me)
        if not d["updated"]:
            continue
        applied += 1
        xi, Pi, di = O.update_global(cfg, x, P, O.update_local(cfg, x, P, ty, ln, me, 0, 1)[None, :])
        worst = max(worst, S.state_delta(x2, xi))
        assert worst <= 1e-10, (trial, mode, worst, d["rank"], di["truncated_at"])
        if di["truncated_at"] >= 0:
            assert di["truncated_at"] == d["rank"], trial
        low += d["rank"] < min(d["n_rows"], 60)
    assert applied > 300 and low > 100


def test_the_reference_itself_is_ill_conditioned_at_rest():
    """Why the free-running bar is looser for a platform at rest (tests/test_gpu_truncation.py): the LITERAL oracle started one ulp away
    from itself (x0 nudged to the next double, P0 by +-1 ulp per entry) ends > 1e-8 away after 100 stationary frames and < 1e-11 away on the
    stock motion.  At rest the window is unobservable in scale and the sequence amplifies rounding ~1e9-fold: any two correct implementations
    (two compilers, two summation orders) differ by that much there, the reference's own builds included."""
    cfg = abi.config_named("B", enable_equalizer=0)
    n = 100
    res = {}
    for motion in ("stationary", "sinus"):
        seq = O.rv.synth.SynthSequence(cfg, duration=(38 + n + 4) / 20.0, seed=4, drop_prob=0.15, motion=motion)
        w, a, ni = seq.init_from_static(38)
        x0, P0 = O.initialize(cfg, w, a, ni)
        lit, per = O.System(cfg), O.System(cfg)
        lit.set_state(x0, P0)
        P0p = P0 * (1 + 2.2e-16 * np.random.default_rng(0).integers(-1, 2, P0.shape))
        per.set_state(np.nextafter(x0, np.inf), 0.5 * (P0p + P0p.T))
        drv = O.rv.synth.DirectTrackDriver(seq)
        worst = 0.0
        for k in range(39, 39 + n):
            inp = drv.inputs(k)
            lit.frame(inp["imu"], inp["cand"], tracked=inp["tracked"], status=inp["status"])
            per.frame(inp["imu"], inp["cand"], tracked=inp["tracked"], status=inp["status"])
            drv.after(lit.tracker().get_points()[0])
            worst = max(worst, S.state_delta(per.get_state()[0], lit.get_state()[0]))
        res[motion] = worst
    assert res["stationary"] > 1e-8 and res["sinus"] < 1e-11, res


def test_last_bit_noise_per_frame_moves_the_reference_by_1e_7_at_rest():
    """The floor under any free-running comparison at rest (tools/at_rest_sensitivity.py; VERDICT round 3, "the at-rest sequence needs a
    relaxed bar"): the LITERAL oracle against itself with +-1 ulp of noise on every state / covariance entry after every frame — less than
    what two correct implementations of any stage differ by — ends 2e-7 .. 2e-6 apart after 100 stationary frames (five seeds measured; two
    here), while the information-form mirror of the device's U7-U10, every other stage shared bit for bit, ends 3e-7 from the literal form:
    inside that band.  So the measurement-space form would not buy the device a 1e-6 free-running bar at rest; what the forms owe each other
    is agreement per update (tests/test_gpu_truncation.py: 1e-9 asked, 6.5e-14 measured on direct tracks)."""
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import at_rest_sensitivity as A
    noise, mirror = A.study("stationary", frames=100, seeds=2)
    assert max(noise) > 1e-7, noise          # last-bit noise alone reaches 1e-7 (measured 2e-7 .. 2e-6)
    assert mirror < 20 * max(noise) and mirror < 5e-6, (mirror, noise)


def _gap_stop(n, types, lens):
    """greedy fill of the leading rows of R by the accepted features' row counts in the order of their first columns (type '2': columns
    0..e2, 2 ceil(L/2) - 3 rows; type '1': columns 6(n-L+1)..6n-1, 2L - 3 rows): the position of the first row no feature can fill while
    later features still wait — a COLUMN GAP —, or -1"""
    groups = {}
    for t, L in zip(types, lens):
        if t == ord("2"):
            Lu = (int(L) + 1) // 2
            s, e, rho = 0, 6 * (Lu - 1) - 1, 2 * Lu - 3
        else:
            s, e, rho = 6 * (n - (int(L) - 1)), 6 * n - 1, 2 * int(L) - 3
        if rho > 0:
            g = groups.setdefault(s, [0, e])
            g[0] += rho
            g[1] = max(g[1], e)
    p = 0
    for s in sorted(groups):
        if p < s:
            return p
        p = min(p + groups[s][0], groups[s][1] + 1)
    return -1


def sweep_wider(trials=1500):
    """randomised stacks on hand-degenerate windows (cfg B, 10 clones) with ALL length mixes (half / all type '2' / all type '1'), 3..Fu features.
    mode 0: the recorded window; 1: a run of clones set to the identity pose; 2: every relative translation zero; 3: every relative pose the
    same (exactly repeated relative poses).  Yields (trial, mode, cfg, n, x, P, types, lens, meas)."""
    synth = O.rv.synth
    cfg = abi.config_named("B", enable_equalizer=0)
    n, Fu = 10, abi.fu(cfg)
    base = [r for r in _run(cfg, 60, image=False, seed=3) if (len(r["x1"]) - 26) // 7 == n][-1]
    rng = np.random.default_rng(11)
    for trial in range(trials):
        x, P = base["x1"].copy(), base["P1"]
        mode = trial % 4
        if mode == 1:
            a = int(rng.integers(0, n - 2))
            for c in range(a, int(rng.integers(a + 1, n))):
                x[26 + 7 * c: 33 + 7 * c] = [0, 0, 0, 1, 0, 0, 0]
        elif mode == 2:
            for c in range(n):
                x[30 + 7 * c: 33 + 7 * c] = 0
        elif mode == 3:
            for c in range(n):
                x[26 + 7 * c: 33 + 7 * c] = [0, 0, 0, 1, 0.02, 0.01, 0.0]
        nf = int(rng.integers(3, Fu + 1))
        mix = ("half", "all2", "all1")[int(rng.integers(0, 3))]
        ty, ln, me = synth.worst_case_tracks(cfg, x, n_feat=nf, seed=int(rng.integers(1 << 30)), mix=mix)
        for f in range(nf):
            if ty[f] == ord("1") and rng.uniform() < 0.35:
                L = int(rng.integers(2, ln[f] + 1))
                me[f, :L] = me[f, ln[f] - L: ln[f]].copy()
                ln[f] = L
        yield trial, mode, cfg, n, x, P, ty, ln, me


def sweep_few(trials=1500):
    """the stock motion at the 14-clone window (cfg A), 3..15 features per update with random type-'1' lengths — a scene with little texture.
    Yields (trial, 0, cfg, n, x, P, types, lens, meas)."""
    synth = O.rv.synth
    cfg = abi.config_named("A", enable_equalizer=0)
    n = cfg.max_track_len - 1
    recs = [r for r in _run(cfg, 4 * n + 30, image=False, seed=3) if (len(r["x1"]) - 26) // 7 == n]
    rng = np.random.default_rng(103)
    for trial in range(trials):
        base = recs[int(rng.integers(0, len(recs)))]
        x, P = base["x1"].copy(), base["P1"]
        nf = int(rng.integers(3, 16))
        mix = ("half", "all2", "all1")[int(rng.integers(0, 3))]
        ty, ln, me = synth.worst_case_tracks(cfg, x, n_feat=nf, seed=int(rng.integers(1 << 30)), mix=mix)
        for f in range(nf):
            if ty[f] == ord("1") and rng.uniform() < 0.5:
                L = int(rng.integers(2, ln[f] + 1))
                me[f, :L] = me[f, ln[f] - L: ln[f]].copy()
                ln[f] = L
        yield trial, 0, cfg, n, x, P, ty, ln, me


def _mirror_vs_literal(sweep):
    """(tall updates, [(trial, mode, n_good, n_rows, literal nRank, literal path taken?, delta)] of the updates where the device's formulation
    (oracle mirror: information form + structural rule, or the literal sweep where literal.h's decision asks for it) and the reference's
    literal sweep + scan differ by more than 1e-9 per state)"""
    tall, exceptions, taken = 0, [], 0
    for trial, mode, cfg, n, x, P, ty, ln, me in sweep:
        x2, P2, d = O.update(cfg, x, P, ty, ln, me)
        if not d["updated"] or d["n_rows"] <= 6 * n:
            continue
        tall += 1
        blk = O.update_local(cfg, x, P, ty, ln, me, 0, 1)
        xi, Pi, di = O.update_global(cfg, x, P, blk[None, :])
        lit = blk[-8 + 5] == 1
        taken += lit
        delta = S.state_delta(x2, xi)
        if di["truncated_at"] >= 0:
            assert di["truncated_at"] == d["rank"], trial          # where the device's form cuts, it cuts where the reference does
        if lit:
            assert blk[-8 + 6] == d["rank"] and delta < 1e-12, (trial, delta)      # the literal path IS the reference's sequence
        if delta > 1e-9:
            exceptions.append((trial, mode, d["n_good"], d["n_rows"], d["rank"], bool(lit), delta))
    return tall, taken, exceptions


def _reference_noise(cfg, x, P, ty, ln, me, draws=12):
    """how far the reference's literal result moves when every non-zero entry of the stacked Hw moves by +-1 ulp (less than two correct
    implementations of U1-U4 differ by): (max state delta over the draws, the set of nRank values seen)"""
    Hw, r, ng = O.update_stack(cfg, x, P, ty, ln, me)
    x0, _, d0 = O.update_from_stack(cfg, x, P, Hw, r, ng)
    rng = np.random.default_rng(1)
    worst, ranks = 0.0, {d0["rank"]}
    for _ in range(draws):
        up = rng.integers(0, 2, Hw.shape) > 0
        Hp = np.where(Hw != 0, np.nextafter(Hw, np.where(up, np.inf, -np.inf)), 0.0)
        xp, _, dp = O.update_from_stack(cfg, x, P, Hp, r, ng)
        worst = max(worst, S.state_delta(xp, x0))
        ranks.add(dp["rank"])
    return worst, ranks


def test_wider_random_sweep_only_the_repeated_pose_class_is_left():
    """1500 randomised stacks on hand-degenerate windows.  Round 5 found two classes of small stacks the structural rule does not cover
    (5 of 1422 tall updates, 1e-8 .. 2e-4): (i) a row whose norm lies just under the scan's 1e-4 threshold WITHOUT being rounding residue,
    (ii) a column gap behind an over-determined block in a barely tall stack of ~8 features (residue rows are compacted into the gap, the
    scan stops there and discards every later feature).  Round 6: literal.h's decision — at most 24 features handed in and a column gap
    behind an over-determined group (other than the one right behind a lone type-'2' block, which is the structural rule's own, proven
    constellation) — sends class (ii) through the reference's own sequence of rotations.  What is left is class (i) on windows of exactly
    repeated relative poses (mode 3; nRank 58 of 60; 1.7e-7 and 3.9e-8, inside the 1e-6 bar) — the class on which the reference disagrees
    with ITSELF (its compiled sources against their restatement: tests/test_ref_pins.py::test_update_on_random_small_and_degenerate_stacks),
    where "parity" is not defined."""
    tall, taken, exceptions = _mirror_vs_literal(sweep_wider(1500))
    assert tall > 1300 and taken >= 5, (tall, taken)
    assert len(exceptions) <= 3, exceptions                          # measured: 2
    for trial, mode, n_good, n_rows, rank, lit, delta in exceptions:
        assert mode == 3 and rank >= 6 * 10 - 2 and delta < 1e-6, (trial, mode, n_good, n_rows, rank, delta)


def test_few_features_on_the_stock_motion_only_noise_decided_updates_are_left():
    """The stock motion at the 14-clone window, 3..15 features per update (a scene with little texture): round 5 measured 5 of ~900 tall
    updates off the literal scan, two of them (column gaps: five type-'2' features fill columns 0..40, one type-'1' feature of 9
    observations carries the rows to position 57, the next feature starts at column 60) by 1.6e-4 and 3.8e-4 — outside the bar.  With
    the literal path of round 6 those are gone.  What is left (measured: one update, 5.4e-9) is decided by the reference's OWN rounding:
    the stack's Gram matrix is singular before its last column (the scale gauge), the rows of R behind the dependent column are a mixture
    whose angle is makeGivens(residue, residue), and +-1 ulp on the entries of the stacked Hw moves the reference's nRank by one and its
    state by more than the difference in question (_reference_noise)."""
    sweep = list(sweep_few(1500))
    tall, taken, exceptions = _mirror_vs_literal(iter(sweep))
    assert tall > 800 and taken > 20, (tall, taken)
    assert len(exceptions) <= 2, exceptions
    for trial, mode, n_good, n_rows, rank, lit, delta in exceptions:
        _, _, cfg, n, x, P, ty, ln, me = sweep[trial]
        noise, ranks = _reference_noise(cfg, x, P, ty, ln, me)
        assert delta < 1e-7 and delta <= 4 * noise and len(ranks) > 1, (trial, delta, noise, ranks)
