"""The C++ host side above the C-ABI (host/: settings reader, InputBuffer, System::MonoVIO, EuRoC replay — SURVEY.md 8f rank 4).
CPU part: the OpenCV-YAML settings reader, the ASL dataset reader and the PNG/PGM decoders through rvio_replay's --check-*
modes (no GPU needed).  The replay itself is covered by tests/test_gpu_host.py."""
import json
import os
import struct
import subprocess
import zlib

import numpy as np
import pytest

import oracle as O

abi = O.abi
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "host", "rvio_replay")

# the stock settings (values of config/rvio_euroc.yaml), typed here so that the test does not read the reference tree
EUROC_YAML = """%YAML:1.0

#--------------------------------------------------------------------------------------------
# IMU Parameters (fixed).
#--------------------------------------------------------------------------------------------
IMU.dps: 200
IMU.sigma_g: 1.6968e-04
IMU.sigma_wg: 1.9393e-05
IMU.sigma_a: 2.0000e-3
IMU.sigma_wa: 3.0000e-3
IMU.nG: 9.8082
IMU.nSmallAngle: 0.001745329
Camera.fps: 20
Camera.RGB: 0
Camera.Fisheye: 0
Camera.width: 752
Camera.height: 480
Camera.fx: 458.654
Camera.fy: 457.296
Camera.cx: 367.215
Camera.cy: 248.375
Camera.k1: -0.28340811
Camera.k2: 0.07395907
Camera.p1: 0.00019359
Camera.p2: 1.76187114e-05
Camera.sigma_px: 0.002180293
Camera.sigma_py: 0.002186767
# Camera extrinsics [B:IMU,C0:cam0]
Camera.T_BC0: !!opencv-matrix
    rows: 4
    cols: 4
    dt: d
    data: [ 0.0148655429818, -0.999880929698, 0.00414029679422, -0.0216401454975,
            0.999557249008, 0.0149672133247, 0.025715529948, -0.064676986768,
           -0.0257744366974, 0.00375618835797, 0.999660727178, 0.00981073058949,
            0.0, 0.0, 0.0, 1.0]
Camera.nTimeOffset: 0
Tracker.nFeatures: 200
Tracker.nMaxTrackingLength: 15
Tracker.nMinTrackingLength: 3
Tracker.nMinDist: 15
Tracker.nQualLvl: 0.01
Tracker.nBlockSizeX: 150
Tracker.nBlockSizeY: 120
Tracker.EnableEqualizer: 1
Tracker.UseSampson: 1
Tracker.nInlierThrd: 1e-5
INI.nThresholdAngle: 0.005 # 0.01 (for MH_*)
INI.nThresholdDispl: 0.01
INI.EnableAlignment: 1
INI.RecordOutputs: 0
Landmark.nScale: 0.03
Landmark.nPubRate: 5
"""


def ensure_bin():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "host")], stdout=subprocess.DEVNULL)
    return BIN


def write_png_gray(path, img, filters=(0, 1, 2, 3, 4)):
    """8-bit PNG (gray [h,w], RGB [h,w,3] or RGBA [h,w,4]) with the row filters cycling through `filters` (exercises every un-filter branch)"""
    h, w = img.shape[:2]
    bpp = 1 if img.ndim == 2 else img.shape[2]
    ctype = {1: 0, 3: 2, 4: 6}[bpp]
    img = img.reshape(h, w * bpp)
    w = w * bpp
    raw = bytearray()
    prev = np.zeros(w, np.int32)
    for y in range(h):
        ft = filters[y % len(filters)]
        cur = img[y].astype(np.int32)
        a = np.concatenate((np.zeros(bpp, np.int32), cur[:-bpp]))
        b = prev
        c = np.concatenate((np.zeros(bpp, np.int32), prev[:-bpp]))
        if ft == 0:
            pred = np.zeros(w, np.int32)
        elif ft == 1:
            pred = a
        elif ft == 2:
            pred = b
        elif ft == 3:
            pred = (a + b) >> 1
        else:
            p = a + b - c
            pa, pb, pc = np.abs(p - a), np.abs(p - b), np.abs(p - c)
            pred = np.where((pa <= pb) & (pa <= pc), a, np.where(pb <= pc, b, c))
        raw.append(ft)
        raw += ((cur - pred) & 255).astype(np.uint8).tobytes()
        prev = cur

    def chunk(t, d):
        return struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d) & 0xffffffff)
    z = zlib.compress(bytes(raw), 6)
    half = len(z) // 2                       # two IDAT chunks
    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w // bpp, h, 8, ctype, 0, 0, 0)) +
                chunk(b"IDAT", z[:half]) + chunk(b"IDAT", z[half:]) + chunk(b"IEND", b""))


def write_pgm(path, img):
    h, w = img.shape
    with open(path, "wb") as f:
        f.write(b"P5\n# synthetic\n%d %d\n255\n" % (w, h) + img.tobytes())


def write_asl(root, seq, frames, as_png=False):
    """mav0/{cam0,imu0} in the EuRoC ASL layout: integer-nanosecond stamps, images named <stamp>.png|.pgm"""
    cam = os.path.join(root, "mav0", "cam0", "data")
    os.makedirs(cam, exist_ok=True)
    os.makedirs(os.path.join(root, "mav0", "imu0"), exist_ok=True)
    t0 = 1403636579_000000000
    with open(os.path.join(root, "mav0", "cam0", "data.csv"), "w") as f:
        f.write("#timestamp [ns],filename\n")
        for k in frames:
            ns = t0 + int(round(seq.frame_time(k) * 1e9))
            name = "%d.%s" % (ns, "png" if as_png else "pgm")
            f.write("%d,%s\n" % (ns, name))
            (write_png_gray if as_png else write_pgm)(os.path.join(cam, name), seq.render(k))
    imu = seq.imu_all()
    with open(os.path.join(root, "mav0", "imu0", "data.csv"), "w") as f:
        f.write("#timestamp [ns],w_RS_S_x [rad s^-1],w_RS_S_y [rad s^-1],w_RS_S_z [rad s^-1],a_RS_S_x [m s^-2],a_RS_S_y [m s^-2],a_RS_S_z [m s^-2]\n")
        for s in imu:
            ns = t0 + int(round(float(s["t"]) * 1e9))
            f.write("%d,%s\n" % (ns, ",".join(repr(float(v)) for v in list(s["w"]) + list(s["a"]))))
    return t0


def test_settings_reader_matches_the_c_abi_defaults(tmp_path):
    p = tmp_path / "rvio_euroc.yaml"
    p.write_text(EUROC_YAML)
    got = json.loads(subprocess.check_output([ensure_bin(), "--check-settings", str(p)]))
    want = abi.config_euroc()
    for k in ("imu_rate", "sigma_g", "sigma_wg", "sigma_a", "sigma_wa", "gravity", "small_angle", "width", "height", "fisheye", "n_features",
              "max_track_len", "min_track_len", "block_x", "block_y", "enable_equalizer", "use_sampson", "inlier_thr", "ini_thr_angle",
              "ini_thr_displ", "ini_enable_alignment"):
        assert got[k] == getattr(want, k), k
    for k in ("fx", "fy", "cx", "cy", "k1", "k2", "p1", "p2", "k3", "sigma_px", "sigma_py", "min_dist", "qual_lvl"):
        assert np.float32(got[k]) == np.float32(getattr(want, k)), k          # float32 members
    assert got["T_bc"] == list(want.T_bc)
    assert got["cam_time_offset"] == 0 and got["record_outputs"] == 0


def test_settings_reader_overrides_and_errors(tmp_path):
    p = tmp_path / "s.yaml"
    p.write_text(EUROC_YAML.replace("Tracker.nFeatures: 200", "Tracker.nFeatures: 400   # more")
                 .replace("Camera.nTimeOffset: 0", "Camera.nTimeOffset: -0.0125").replace("INI.RecordOutputs: 0", "INI.RecordOutputs: 1"))
    got = json.loads(subprocess.check_output([ensure_bin(), "--check-settings", str(p)]))
    assert got["n_features"] == 400 and got["cam_time_offset"] == -0.0125 and got["record_outputs"] == 1
    r = subprocess.run([BIN, "--check-settings", str(tmp_path / "missing.yaml")], capture_output=True, text=True)
    assert r.returncode != 0 and "Failed to open settings file" in r.stderr      # System.cc:54-58


@pytest.mark.parametrize("shape", [(480, 752), (37, 53)])
def test_png_and_pgm_decoders(tmp_path, shape):
    rng = np.random.default_rng(1)
    yy, xx = np.mgrid[0:shape[0], 0:shape[1]]
    img = ((xx * 3 + yy * 5) % 256 ^ rng.integers(0, 32, shape)).astype(np.uint8)
    idx = np.arange(img.size) % 251 + 1
    want = {"width": shape[1], "height": shape[0], "channels": 1, "sum": int(img.sum(dtype=np.int64)), "wsum": int((img.ravel().astype(np.int64) * idx).sum())}
    write_png_gray(str(tmp_path / "a.png"), img)
    write_pgm(str(tmp_path / "a.pgm"), img)
    for name in ("a.png", "a.pgm"):
        assert json.loads(subprocess.check_output([ensure_bin(), "--check-image", str(tmp_path / name)])) == want, name


def test_asl_reader(tmp_path):
    cfg = abi.config_named("B", enable_equalizer=1)
    seq = O.rv.synth.SynthSequence(cfg, duration=1.0)
    t0 = write_asl(str(tmp_path), seq, range(0, 4))
    got = json.loads(subprocess.check_output([ensure_bin(), "--check-dataset", str(tmp_path)]))
    imu = seq.imu_all()
    assert got["images"] == 4 and got["imu"] == len(imu)
    assert got["first_image"].endswith("mav0/cam0/data/%d.pgm" % t0)
    assert abs(got["dt1"] - 0.005) < 1e-6        # stamps are doubles of ~1.4e9 s (ros::Time::toSec): 2.4e-7 s resolution
    assert got["w1"] == [float(v) for v in imu["w"][1]]


def test_settings_reader_reports_missing_keys(tmp_path):
    """a key the reference reads but the file does not hold keeps the EuRoC default AND is reported (upstream's cv::FileStorage would
    silently yield 0); non-integer block sizes survive (float members upstream, FeatureDetector.h:72-73)"""
    p = tmp_path / "s.yaml"
    lines = [ln for ln in EUROC_YAML.splitlines() if not ln.startswith("Tracker.nQualLvl")]
    p.write_text("\n".join("Tracker.nBlockSizeX: 150.5" if ln.startswith("Tracker.nBlockSizeX") else ln for ln in lines) + "\n")
    r = subprocess.run([ensure_bin(), "--check-settings", str(p)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    cfg = json.loads(r.stdout)
    assert cfg["block_x"] == 150.5 and np.float32(cfg["qual_lvl"]) == np.float32(0.01)
    assert "Tracker.nQualLvl is missing" in r.stderr and "nBlockSize" not in r.stderr


def test_colour_images_are_converted_like_cvtColor(tmp_path):
    """Tracker.cc:182-196: 3- and 4-channel input goes through cvtColor (RGB2GRAY / BGR2GRAY by Camera.RGB) before anything else.  The host
    applies OpenCV's 8-bit fixed-point form (R 4899 + G 9617 + B 1868 + 8192) >> 14 — checked against its float definition
    0.299 R + 0.587 G + 0.114 B (never more than one grey level away, equal on > 95 % of random pixels) and bit-exact against the integers."""
    rng = np.random.default_rng(3)
    h, w = 37, 53
    rgb = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    rgba = np.concatenate([rgb, rng.integers(0, 256, (h, w, 1), dtype=np.uint8)], axis=2)
    idx = np.arange(h * w) % 251 + 1

    def gray(img, is_rgb):
        r, g, b = (img[..., 0], img[..., 1], img[..., 2]) if is_rgb else (img[..., 2], img[..., 1], img[..., 0])
        return ((r.astype(np.int64) * 4899 + g.astype(np.int64) * 9617 + b.astype(np.int64) * 1868 + 8192) >> 14).astype(np.uint8)
    fl = np.rint(0.299 * rgb[..., 0] + 0.587 * rgb[..., 1] + 0.114 * rgb[..., 2])
    d = np.abs(gray(rgb, True).astype(np.int64) - fl)
    assert d.max() <= 1 and (d == 0).mean() > 0.95
    write_png_gray(str(tmp_path / "c.png"), rgb)
    write_png_gray(str(tmp_path / "d.png"), rgba)
    with open(tmp_path / "e.ppm", "wb") as f:
        f.write(b"P6\n%d %d\n255\n" % (w, h) + rgb.tobytes())
    for name, img, ch in (("c.png", rgb, 3), ("d.png", rgba, 4), ("e.ppm", rgb, 3)):
        for flag, is_rgb in (([], True), (["--bgr"], False)):
            g = gray(img, is_rgb)
            want = {"width": w, "height": h, "channels": ch, "sum": int(g.sum(dtype=np.int64)), "wsum": int((g.ravel().astype(np.int64) * idx).sum())}
            assert json.loads(subprocess.check_output([ensure_bin(), "--check-image", str(tmp_path / name)] + flag)) == want, (name, flag)
    # the settings key
    p = tmp_path / "s.yaml"
    p.write_text(EUROC_YAML + "\nCamera.RGB: 1\n")
    assert json.loads(subprocess.check_output([ensure_bin(), "--check-settings", str(p)]))["is_rgb"] == 1
