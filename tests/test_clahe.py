"""CLAHE (Tracker.cc:198-202: createCLAHE(3.0, Size(5,5))->apply).  OpenCV is a third-party dependency that is not
vendored in the reference and not installed here, so the oracle's C restatement is cross-checked on the CPU against
an independent numpy write-up of the published algorithm and against a closed form; the HIP kernels are then held
bit-exact to the oracle (`-m gpu`)."""
import numpy as np
import pytest

import oracle as O

abi = O.abi


def clahe_np(img, clip=3.0, tx=5, ty=5):
    """numpy restatement of OpenCV's clahe.cpp (8-bit path), written independently of oracle/frontend.cpp"""
    h, w = img.shape
    ext = img
    if w % tx or h % ty:
        ext = np.pad(img, ((0, ty - h % ty), (0, tx - w % tx)), mode="reflect")   # numpy 'reflect' == BORDER_REFLECT_101
    th, tw = ext.shape[0] // ty, ext.shape[1] // tx
    area = tw * th
    scale = np.float32(255.0) / np.float32(area)
    cl = max(int(clip * area / 256), 1)
    lut = np.zeros((ty, tx, 256), np.uint8)
    for j in range(ty):
        for i in range(tx):
            hist = np.bincount(ext[j * th:(j + 1) * th, i * tw:(i + 1) * tw].ravel(), minlength=256).astype(np.int64)
            clipped = int(np.maximum(hist - cl, 0).sum())
            hist = np.minimum(hist, cl)
            batch = clipped // 256
            residual = clipped - batch * 256
            hist += batch
            if residual:
                hist[np.arange(0, 256, max(256 // residual, 1))[:residual]] += 1
            lut[j, i] = np.clip(np.rint(np.cumsum(hist).astype(np.float32) * scale), 0, 255).astype(np.uint8)
    one, half = np.float32(1), np.float32(0.5)

    def axis(n, tile, tiles):
        f = np.arange(n, dtype=np.float32) * (one / np.float32(tile)) - half
        t1 = np.floor(f).astype(np.int64)
        a = (f - t1.astype(np.float32)).astype(np.float32)
        return np.maximum(t1, 0), np.minimum(t1 + 1, tiles - 1), a, (one - a).astype(np.float32)

    tx1, tx2, xa, xa1 = axis(w, tw, tx)
    ty1, ty2, ya, ya1 = axis(h, th, ty)
    v = img.astype(np.int64)
    L = lut.astype(np.float32)
    p11, p12 = L[ty1[:, None], tx1[None, :], v], L[ty1[:, None], tx2[None, :], v]
    p21, p22 = L[ty2[:, None], tx1[None, :], v], L[ty2[:, None], tx2[None, :], v]
    res = (p11 * xa1 + p12 * xa) * ya1[:, None] + (p21 * xa1 + p22 * xa) * ya[:, None]
    assert res.dtype == np.float32
    return np.clip(np.rint(res), 0, 255).astype(np.uint8)


def test_images(seed=0):
    rng = np.random.default_rng(seed)
    out = []
    for (h, w) in ((480, 752), (60, 100), (243, 321), (1080, 1920)):
        yy, xx = np.mgrid[0:h, 0:w]
        smooth = (96 + 60 * np.sin(xx / 37.0) * np.cos(yy / 23.0) + rng.normal(0, 6, (h, w))).clip(0, 255).astype(np.uint8)
        noise = rng.integers(0, 256, (h, w), dtype=np.uint8)
        dark = (rng.integers(0, 24, (h, w)) + (xx > w // 2) * 40).astype(np.uint8)
        out += [smooth, noise, dark]
    return out


test_images.__test__ = False


@pytest.mark.parametrize("idx", range(9))
def test_oracle_clahe_matches_numpy_restatement(idx):
    img = test_images()[idx]
    got, want = O.clahe(img), clahe_np(img)
    assert np.array_equal(got, want), (img.shape, int(np.abs(got.astype(int) - want.astype(int)).max()))


def test_clahe_uniform_image_closed_form():
    """constant image v: every tile LUT is identical, so the output is the constant lut[v] of the clipped histogram"""
    h, w, v = 480, 752, 77
    img = np.full((h, w), v, np.uint8)
    tw, th = (w + 3) // 5, (h + 5) // 5          # 752 -> 755, 480 -> 485 (a dimension that divides is extended by 5 too)
    area = tw * th
    cl = max(int(3.0 * area / 256), 1)
    clipped = area - cl
    batch, residual = clipped // 256, clipped % 256
    step = max(256 // residual, 1) if residual else 1
    bonus = min(len(range(0, v + 1, step)), residual) if residual else 0
    cum = (v + 1) * batch + bonus + cl
    want = int(np.clip(np.rint(np.float32(cum) * (np.float32(255.0) / np.float32(area))), 0, 255))
    out = O.clahe(img)
    assert out.min() == out.max() == want


def test_tracker_with_equalizer_differs_and_runs():
    """enable_equalizer=1 reaches the tracker: KLT runs on the equalized image"""
    cfg1 = abi.config_named("B", enable_equalizer=1)
    cfg0 = abi.config_named("B", enable_equalizer=0)
    seq = O.rv.synth.SynthSequence(cfg0, duration=4.0)
    res = []
    for cfg in (cfg0, cfg1):
        t = O.Tracker(cfg)
        for k in (40, 41, 42):
            xy, vis = seq.project(k, noise=False)
            cand, _ = seq.candidates(k, xy, vis)
            info = t.track(seq.render(k), seq.imu_between(k), cand)
        res.append((info["n_klt_ok"], t.get_points()[0].copy()))
    assert res[0][0] > 100 and res[1][0] > 100
    assert not np.array_equal(res[0][1], res[1][1])
