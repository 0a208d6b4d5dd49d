"""How far is "bit-exact to the oracle" from the library the reference actually calls?  (round-2 verdict, item 8)

The oracle fixes a definition wherever OpenCV's own arithmetic depends on the build: exact integer sums in the LK tracker where OpenCV's
scalar path accumulates in float, a canonical summation tree in cornerSubPix where OpenCV adds in one row-major chain, fresh three-term box
sums in the min-eigenvalue map where cv::boxFilter keeps running sums.  OpenCV is not installed and cannot be (no network), so the distance is
measured against INDEPENDENT write-ups of those three OpenCV orders (oracle/frontend.cpp lk_point float_acc, oracle/detector.cpp cv_order /
row_major): per 1000 features, how many status flags flip and how far positions move.  The numbers asserted here are quoted in DESIGN.md
section 5; they bound what a real OpenCV build can differ by THROUGH THESE ORDERS — not what it may differ by for other reasons (a
different OpenCV version's algorithmic changes stay unpinned)."""
import numpy as np
import pytest

import oracle as O

abi, rv = O.abi, O.rv


@pytest.fixture(scope="module")
def frames():
    cfg = abi.config_named("B", enable_equalizer=1)
    seq = rv.synth.SynthSequence(cfg, duration=6.0)
    ks = [45, 60, 75, 90, 105]                       # ramp-up, fast and slow stretches of the trajectory
    return cfg, [(O.clahe(seq.render(k)), O.clahe(seq.render(k + 1))) for k in ks]


def test_lk_exact_sums_vs_opencv_scalar_float_accumulators(frames):
    cfg, pairs = frames
    n, flips, moved = 0, 0, []
    for a, b in pairs:
        pts = O.detect(cfg, a, 1)
        xe, se = O.klt(a, b, pts)
        xf, sf = O.klt_float(a, b, pts)
        n += len(pts)
        flips += int(np.sum(se != sf))
        both = (se == 1) & (sf == 1)
        moved.append(np.abs(xe[both] - xf[both]).max(axis=1))
    moved = np.concatenate(moved)
    per1000 = 1000.0 * flips / n
    print("LK: %d features, status flips per 1000 = %.2f, position |delta| px: median %.2e  p99 %.2e  max %.2e, identical %.1f %%"
          % (n, per1000, np.median(moved), np.percentile(moved, 99), moved.max(), 100 * np.mean(moved == 0)))
    assert n >= 700
    assert per1000 <= 3.0                            # at most a few flags per thousand features
    assert np.percentile(moved, 99) <= 2e-3 and np.median(moved) <= 1e-4     # sub-milli-pixel for all but a handful


def test_corner_subpix_tree_vs_row_major_chain(frames):
    cfg, pairs = frames
    n, same, worst = 0, 0, 0.0
    for a, _ in pairs:
        raw = O.gftt(a, cfg.n_features, float(np.float32(cfg.qual_lvl)), float(np.float32(cfg.min_dist)))
        t = O.corner_subpix(a, raw)
        r = O.corner_subpix_rowmajor(a, raw)
        n += len(raw)
        same += int(np.sum(np.all(t == r, axis=1)))
        worst = max(worst, float(np.abs(t - r).max()))
    print("cornerSubPix: %d corners, bit-identical %.2f %%, max |delta| %.2e px" % (n, 100.0 * same / n, worst))
    assert n >= 700 and same / n >= 0.97 and worst <= 1e-4        # float positions: the double sums differ by O(1e-16) relative


def test_min_eig_fresh_sums_vs_boxfilter_running_sums(frames):
    cfg, pairs = frames
    diff_px, tot_px, lists_equal = 0, 0, 0
    for a, _ in pairs:
        e0, e1 = O.min_eig(a), O.min_eig_cvorder(a)
        diff_px += int(np.sum(e0 != e1))
        tot_px += e0.size
        assert np.abs(e0 - e1).max() <= 1e-6 * max(1e-30, float(e0.max()))
    print("min-eigenvalue map: %d of %d pixels differ in the last bit(s) (%.3f %%)" % (diff_px, tot_px, 100.0 * diff_px / tot_px))
    assert diff_px / tot_px <= 0.02
