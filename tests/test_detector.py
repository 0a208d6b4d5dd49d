"""FeatureDetector::DetectWithSubPix (FeatureDetector.cc:55-75) = cv::goodFeaturesToTrack + cv::cornerSubPix.
OpenCV is not vendored in the reference and not installed here, so the oracle's C restatement (oracle/detector.cpp) is
cross-checked on the CPU against independent numpy / pure-Python write-ups of the published algorithms and against
constructed images with known answers; the HIP kernels are held bit-exact to the oracle in tests/test_gpu_detector.py."""
import numpy as np
import pytest

import oracle as O

abi = O.abi
f32 = np.float32


def min_eig_np(img):
    """cornerMinEigenVal(blockSize 3, ksize 3) in numpy: scaled Sobel (float), products (float), 3x3 box in double
    (vertical sums first, then the three columns), lambda_min (float)."""
    s = np.pad(img.astype(f32), 1, mode="reflect")
    scale = 1.0 / (4.0 * 3.0 * 255.0)
    k1, k0 = f32(scale), f32(2.0 * scale)
    c = s[1:-1, 1:-1]
    r = s[:, 2:] - s[:, :-2]                                    # R[y][x] = s[y][x+1] - s[y][x-1], rows -1..h
    dx = k0 * r[1:-1] + k1 * (r[:-2] + r[2:])
    q = k0 * s[:, 1:-1] + k1 * (s[:, :-2] + s[:, 2:])           # Q[y][x], rows -1..h
    dy = q[2:] - q[:-2]
    assert dx.dtype == f32 and dy.dtype == f32 and dx.shape == c.shape
    out = []
    for p in (dx * dx, dx * dy, dy * dy):
        pp = np.pad(p, 1, mode="reflect").astype(np.float64)
        col = (pp[:-2] + pp[1:-1]) + pp[2:]
        out.append(((col[:, :-2] + col[:, 1:-1]) + col[:, 2:]).astype(f32))
    a, b, cc = out[0] * f32(0.5), out[1], out[2] * f32(0.5)
    return (a + cc) - np.sqrt((a - cc) * (a - cc) + b * b)


def gftt_py(eig, max_corners, quality, min_distance):
    """goodFeaturesToTrack on a given eigenvalue map, sequential reference (featureselect.cpp)"""
    h, w = eig.shape
    thr = f32(float(eig.max()) * quality)
    e = np.where(eig > thr, eig, f32(0))
    cands = []
    for y in range(1, h - 1):
        row = e[y]
        for x in np.nonzero(row[1:-1])[0] + 1:
            if row[x] == e[y - 1:y + 2, x - 1:x + 2].max():
                cands.append((float(row[x]), y * w + x))
    cands.sort(key=lambda t: (-t[0], -t[1]))
    cell = int(np.rint(min_distance))
    gw, gh = (w + cell - 1) // cell, (h + cell - 1) // cell
    grid = [[] for _ in range(gw * gh)]
    out = []
    for _, idx in cands:
        y, x = divmod(idx, w)
        xc, yc = x // cell, y // cell
        good = True
        for yy in range(max(0, yc - 1), min(gh - 1, yc + 1) + 1):
            for xx in range(max(0, xc - 1), min(gw - 1, xc + 1) + 1):
                for (ox, oy) in grid[yy * gw + xx]:
                    if (x - ox) ** 2 + (y - oy) ** 2 < min_distance * min_distance:
                        good = False
        if good:
            grid[yc * gw + xc].append((x, y))
            out.append((x, y))
            if len(out) == max_corners:
                break
    return np.array(out, f32).reshape(-1, 2)


def images():
    rng = np.random.default_rng(3)
    cfg = abi.config_named("B", enable_equalizer=0)
    seq = O.rv.synth.SynthSequence(cfg, duration=4.0)
    a = seq.render(50)
    b = O.clahe(seq.render(61))
    noise = rng.integers(0, 256, (120, 200), dtype=np.uint8)
    yy, xx = np.mgrid[0:96, 0:128]
    checker = (((yy // 16) + (xx // 16)) % 2 * 160 + 40).astype(np.uint8)
    flat = np.full((64, 80), 93, np.uint8)
    return {"synth": a, "synth_eq": b, "noise": noise, "checker": checker, "flat": flat}


@pytest.mark.parametrize("name", ["synth", "synth_eq", "noise", "checker", "flat"])
def test_min_eig_matches_numpy_restatement(name):
    img = images()[name]
    got, want = O.min_eig(img), min_eig_np(img)
    assert np.array_equal(got, want), float(np.abs(got - want).max())


@pytest.mark.parametrize("name,md", [("synth", 15.0), ("synth_eq", 30.0), ("noise", 15.0), ("checker", 15.0), ("flat", 15.0)])
def test_gftt_matches_sequential_python(name, md):
    img = images()[name]
    got = O.gftt(img, 200, float(f32(0.01)), md)
    want = gftt_py(O.min_eig(img), 200, float(f32(0.01)), md)
    assert got.shape == want.shape and np.array_equal(got, want)
    if name == "flat":
        assert len(got) == 0
    if len(got) > 1:      # the defining property: pairwise distance >= minDistance
        d = np.linalg.norm(got[:, None] - got[None], axis=2) + np.eye(len(got)) * 1e9
        assert d.min() >= md


def test_gftt_checkerboard_corners_are_found():
    img = images()["checker"]
    got = O.gftt(img, 100, 0.01, 10.0)
    # every interior checker corner (multiples of 16) has a detection within 1.5 px
    corners = np.array([(x, y) for y in range(16, 96, 16) for x in range(16, 128, 16)], f32)
    d = np.linalg.norm(corners[:, None] - got[None], axis=2).min(axis=1)
    assert d.max() <= 1.5


def test_corner_subpix_recovers_a_known_subpixel_corner():
    """a blurred ideal corner at a known sub-pixel location: cornerSubPix converges to it from 2 px away"""
    def render(cx, cy, n=64, ss=8):
        yy, xx = np.mgrid[0:n * ss, 0:n * ss]
        hi = (((xx + 0.5) / ss - 0.5 > cx) ^ ((yy + 0.5) / ss - 0.5 > cy)).astype(np.float64)
        return (40 + 170 * hi.reshape(n, ss, n, ss).mean(axis=(1, 3))).round().astype(np.uint8)
    for (cx, cy) in ((31.3, 30.6), (28.75, 33.1)):
        img = render(cx, cy)
        start = np.array([[cx + 1.7, cy - 1.4]], f32)
        out = O.corner_subpix(img, start, win=7)
        assert np.hypot(out[0, 0] - cx, out[0, 1] - cy) < 0.2, (out, cx, cy)     # started 2.2 px away; 8-bit, box-filtered corner


def test_corner_subpix_keeps_points_that_diverge():
    """flat patch: singular system -> the point is returned unchanged; near the border the replicated patch still works"""
    img = np.full((48, 48), 120, np.uint8)
    pts = np.array([[20.25, 21.5], [2.0, 3.0]], f32)
    out = O.corner_subpix(img, pts, win=7)
    assert np.array_equal(out, pts)


def test_detect_with_subpix_contract():
    cfg = abi.config_named("B", enable_equalizer=0)
    img = images()["synth"]
    c1, c2 = O.detect(cfg, img, 1), O.detect(cfg, img, 2)
    assert 0 < len(c2) <= len(c1) <= cfg.n_features
    raw = O.gftt(img, cfg.n_features, float(f32(0.01)), 15.0)
    assert len(raw) == len(c1) and np.abs(raw - c1).max() <= 7.0     # refinement never leaves the 7-px half window
    assert np.array_equal(O.corner_subpix(img, raw, win=7), c1)


def test_tracker_runs_its_own_detector_when_no_corners_are_given():
    cfg = abi.config_named("B", enable_equalizer=1)
    seq = O.rv.synth.SynthSequence(cfg, duration=4.0)
    t = O.Tracker(cfg)
    info = t.track(seq.render(45), seq.imu_between(45))
    assert info["n_tracked_out"] == len(O.detect(cfg, O.clahe(seq.render(45)), 1)) > 50
    info = t.track(seq.render(46), seq.imu_between(46))
    assert info["n_klt_ok"] > 50
