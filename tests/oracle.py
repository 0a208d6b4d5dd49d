"""ctypes binding of oracle/liborc.so — the CPU oracle (TEST INFRASTRUCTURE).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import
this module.  The product package never does.
"""
import ctypes as C
import os
import subprocess
import sys
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pkgload import load_pkg  # noqa: E402

rv = load_pkg()
abi = rv.abi

_LIB = None
dp = C.POINTER(C.c_double)
fp = C.POINTER(C.c_float)
ip = C.POINTER(C.c_int32)
up = C.POINTER(C.c_ubyte)


def _p(a, t):
    return a.ctypes.data_as(t) if a is not None else None


def lib():
    global _LIB
    if _LIB is None:
        # ORC_LIB=liborc_omp.so selects the multi-core build of the same sources (bench.py's secondary CPU baseline)
        path = os.path.join(ROOT, "oracle", os.environ.get("ORC_LIB", "liborc.so"))
        if not os.path.exists(path):
            subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")])
        L = C.CDLL(path)
        L.orc_chi2_95.restype = C.c_double
        L.orc_tracker_create.restype = C.c_void_p
        L.orc_system_create.restype = C.c_void_p
        L.orc_system_tracker.restype = C.c_void_p
        L.orc_rand.restype = C.c_int
        L.orc_ransac.restype = C.c_int
        _LIB = L
    return _LIB


def quat_mul(q1, q2):
    o = np.zeros(4)
    lib().orc_quat_mul(_p(np.ascontiguousarray(q1, float), dp), _p(np.ascontiguousarray(q2, float), dp), _p(o, dp))
    return o


def quat_to_rot(q):
    o = np.zeros(9)
    lib().orc_quat_to_rot(_p(np.ascontiguousarray(q, float), dp), _p(o, dp))
    return o.reshape(3, 3)


def rot_to_quat(R):
    o = np.zeros(4)
    lib().orc_rot_to_quat(_p(np.ascontiguousarray(R, float).reshape(-1), dp), _p(o, dp))
    return o


def chi2_95(dof):
    return lib().orc_chi2_95(int(dof))


def initialize(cfg, w, a, n_imu):
    x, P = np.zeros(26), np.zeros((24, 24))
    lib().orc_initialize(C.byref(cfg), _p(np.ascontiguousarray(w, float), dp), _p(np.ascontiguousarray(a, float), dp),
                         int(n_imu), _p(x, dp), _p(P, dp))
    return x, P


def propagate(cfg, x, P, imu):
    """returns (x_out, P_out); P is col-major == row-major for the symmetric in/out, but we keep
    explicit Fortran order to be exact."""
    x = np.ascontiguousarray(x, float)
    d = P.shape[0]
    Pf = np.asfortranarray(P, dtype=float).copy(order="F")
    xo = np.zeros_like(x)
    imu = np.ascontiguousarray(imu)
    lib().orc_propagate(C.byref(cfg), _p(x, dp), len(x), Pf.ctypes.data_as(dp), d,
                        imu.ctypes.data_as(C.POINTER(abi.rvio_imu)), len(imu), _p(xo, dp))
    return xo, np.array(Pf)


def update(cfg, x, P, types, lens, meas):
    x = np.ascontiguousarray(x, float)
    d = P.shape[0]
    Pf = np.asfortranarray(P, dtype=float)
    tr = abi.make_tracks(types, lens, meas)
    nf = tr.n_feat
    xo, Po = np.zeros_like(x), np.zeros((d, d), order="F")
    acc, gam, ndof, pf = np.zeros(nf, np.int32), np.zeros(nf), np.zeros(nf, np.int32), np.zeros((nf, 3))
    info = np.zeros(4, np.int32)
    lib().orc_update(C.byref(cfg), _p(x, dp), len(x), Pf.ctypes.data_as(dp), d, C.byref(tr), _p(xo, dp),
                     Po.ctypes.data_as(dp), _p(acc, ip), _p(gam, dp), _p(ndof, ip), _p(pf, dp), _p(info, ip))
    return xo, np.array(Po), dict(accepted=acc, gamma=gam, ndof=ndof, pfinv=pf, n_good=int(info[0]),
                                  n_rows=int(info[1]), rank=int(info[2]), updated=int(info[3]))


def feature_model(cfg, x, ftype, meas, L, pf=None):
    """U1..U3 of one feature at pf = (phi, psi, rho) (None: the LM estimate): (r [2Lu], Hx [2Lu, 6n], Hf [2Lu, 3], pf)"""
    x = np.ascontiguousarray(x, float)
    meas = np.ascontiguousarray(meas, np.float32)
    nc6 = 6 * ((len(x) - 26) // 7)
    r, Hx, Hf, pfo = np.zeros(2 * L), np.zeros((2 * L, nc6)), np.zeros((2 * L, 3)), np.zeros(3)
    pfa = None if pf is None else np.ascontiguousarray(pf, float)
    L_ = lib()
    L_.orc_feature_model.restype = C.c_int
    M = L_.orc_feature_model(C.byref(cfg), _p(x, dp), len(x), C.c_ubyte(int(ftype)), _p(meas, fp), int(L), _p(pfa, dp), _p(r, dp), _p(Hx, dp), _p(Hf, dp), _p(pfo, dp))
    return r[:M].copy(), Hx[:M].copy(), Hf[:M].copy(), pfo


def update_stack(cfg, x, P, types, lens, meas):
    """U1..U6: the stacked (Hw [M, 6n], r [M]) of the accepted features and nGoodFeatCount"""
    x = np.ascontiguousarray(x, float)
    d = P.shape[0]
    Pf = np.asfortranarray(P, dtype=float)
    tr = abi.make_tracks(types, lens, meas)
    nc6 = 6 * ((len(x) - 26) // 7)
    rows = int(2 * np.sum(lens))
    Hw, r = np.zeros((max(rows, 1), nc6)), np.zeros(max(rows, 1))
    ng = C.c_int32(0)
    L = lib()
    L.orc_update_stack.restype = C.c_int
    M = L.orc_update_stack(C.byref(cfg), _p(x, dp), len(x), Pf.ctypes.data_as(dp), d, C.byref(tr), _p(Hw, dp), _p(r, dp), C.byref(ng))
    return Hw[:M].copy(), r[:M].copy(), ng.value


def update_from_stack(cfg, x, P, Hw, r, n_good):
    """U7..U10 (literal Givens QR + rank scan) on a given stacked pair; diag carries the row norms after the sweep"""
    x = np.ascontiguousarray(x, float)
    d = P.shape[0]
    Pf = np.asfortranarray(P, dtype=float)
    Hw = np.ascontiguousarray(Hw, float)
    r = np.ascontiguousarray(r, float)
    M, nc6 = Hw.shape
    xo, Po = np.zeros_like(x), np.zeros((d, d), order="F")
    info = np.zeros(4, np.int32)
    norms = np.zeros(max(min(M, 2 * nc6), 1))
    lib().orc_update_from_stack(C.byref(cfg), _p(x, dp), len(x), Pf.ctypes.data_as(dp), d, _p(Hw, dp), _p(r, dp), M, int(n_good),
                                _p(xo, dp), Po.ctypes.data_as(dp), _p(info, ip), _p(norms, dp))
    return xo, np.array(Po), dict(n_good=int(info[0]), n_rows=int(info[1]), rank=int(info[2]), updated=int(info[3]), row_norms=norms)


def block_len(n_clones):
    """per-shard payload of the oracle's mirror: type-'2' part, type-'1' part, 8 counters (oracle/filter.cpp:orc_update_local)"""
    return 2 * 6 * n_clones * (6 * n_clones + 1) + 8


def update_local(cfg, x, P, types, lens, meas, rank, world):
    x = np.ascontiguousarray(x, float)
    d = P.shape[0]
    Pf = np.asfortranarray(P, dtype=float)
    tr = abi.make_tracks(types, lens, meas)
    blk = np.zeros(block_len((len(x) - 26) // 7))
    lib().orc_update_local(C.byref(cfg), _p(x, dp), len(x), Pf.ctypes.data_as(dp), d, C.byref(tr), rank, world, _p(blk, dp))
    return blk


def update_global(cfg, x, P, blocks):
    x = np.ascontiguousarray(x, float)
    d = P.shape[0]
    Pf = np.asfortranarray(P, dtype=float)
    blocks = np.ascontiguousarray(blocks, float)
    world = blocks.shape[0]
    xo, Po = np.zeros_like(x), np.zeros((d, d), order="F")
    info = np.zeros(4, np.int32)
    lib().orc_update_global(C.byref(cfg), _p(x, dp), len(x), Pf.ctypes.data_as(dp), d, _p(blocks, dp), world,
                            _p(xo, dp), Po.ctypes.data_as(dp), _p(info, ip))
    return xo, np.array(Po), dict(n_good=int(info[0]), n_rows=int(info[1]), truncated_at=int(info[2]), updated=int(info[3]))


def augment_compose(cfg, x, P, do_augment=True):
    nmax = cfg.max_track_len - 1
    xb = np.zeros(26 + 7 * (nmax + 1))
    xb[: len(x)] = x
    d = P.shape[0]
    Pb = np.zeros((24 + 6 * (nmax + 1)) ** 2)
    Pb[: d * d] = np.asfortranarray(P, dtype=float).reshape(-1, order="F")
    xdim, dd = C.c_int(len(x)), C.c_int(d)
    pp, pq = np.zeros(3), np.zeros(4)
    lib().orc_augment_compose(C.byref(cfg), _p(xb, dp), C.byref(xdim), _p(Pb, dp), C.byref(dd), int(do_augment), _p(pp, dp), _p(pq, dp))
    d2 = dd.value
    return xb[: xdim.value].copy(), Pb[: d2 * d2].reshape(d2, d2, order="F").copy(), pp, pq


def undistort(cfg, xy):
    xy = np.ascontiguousarray(xy, np.float32)
    out = np.zeros_like(xy)
    lib().orc_undistort(C.byref(cfg), _p(xy, fp), len(xy), _p(out, fp))
    return out


def rand_stream(n, seed=1):
    st = np.zeros(35, np.int32)
    lib().orc_srand(_p(st, ip), seed)
    return np.array([lib().orc_rand(_p(st, ip)) for _ in range(n)])


def ransac(cfg, p1, p2, imu, flags, rng_state=None):
    """p1/p2: [n,3] arrays (rows = homogeneous points). returns (n_inl, flags_out, winner, pairs, rng_state)."""
    p1 = np.ascontiguousarray(p1, float)
    p2 = np.ascontiguousarray(p2, float)
    flags = np.ascontiguousarray(flags, np.uint8).copy()
    st = np.zeros(35, np.int32) if rng_state is None else rng_state.copy()
    win = C.c_int(0)
    pairs = np.zeros(32, np.int32)
    imu = np.ascontiguousarray(imu)
    n = lib().orc_ransac(C.byref(cfg), _p(p1, dp), _p(p2, dp), len(p1), imu.ctypes.data_as(C.POINTER(abi.rvio_imu)), len(imu),
                         _p(flags, up), _p(st, ip), C.byref(win), _p(pairs, ip))
    return n, flags, win.value, pairs.reshape(16, 2), st


def pyr_down(img):
    img = np.ascontiguousarray(img, np.uint8)
    h, w = img.shape
    out = np.zeros(((h + 1) // 2, (w + 1) // 2), np.uint8)
    lib().orc_pyr_down(_p(img, up), w, h, w, _p(out, up))
    return out


def clahe(img):
    """cv::createCLAHE(3.0, Size(5,5))->apply restated (Tracker.cc:198-202)"""
    img = np.ascontiguousarray(img, np.uint8)
    h, w = img.shape
    out = np.zeros((h, w), np.uint8)
    lib().orc_clahe(_p(img, up), w, h, w, _p(out, up))
    return out


def min_eig(img):
    """cv::cornerMinEigenVal(blockSize 3, ksize 3) restated"""
    img = np.ascontiguousarray(img, np.uint8)
    h, w = img.shape
    out = np.zeros((h, w), np.float32)
    lib().orc_min_eig(_p(img, up), w, h, w, _p(out, fp))
    return out


def gftt(img, max_corners, quality, min_distance):
    img = np.ascontiguousarray(img, np.uint8)
    h, w = img.shape
    out = np.zeros((max(max_corners, 1), 2), np.float32)
    n = lib().orc_gftt(_p(img, up), w, h, w, int(max_corners), C.c_double(quality), C.c_double(min_distance), _p(out, fp))
    return out[:n].copy()


def corner_subpix(img, pts, win=7):
    img = np.ascontiguousarray(img, np.uint8)
    h, w = img.shape
    pts = np.ascontiguousarray(pts, np.float32).copy()
    lib().orc_corner_subpix(_p(img, up), w, h, w, _p(pts, fp), len(pts), int(win))
    return pts


def detect(cfg, img, s):
    """FeatureDetector::DetectWithSubPix (FeatureDetector.cc:55-75)"""
    img = np.ascontiguousarray(img, np.uint8)
    out = np.zeros((cfg.n_features, 2), np.float32)
    n = lib().orc_detect(C.byref(cfg), _p(img, up), img.shape[1], int(s), _p(out, fp))
    return out[:n].copy()


def scharr(img):
    img = np.ascontiguousarray(img, np.uint8)
    h, w = img.shape
    out = np.zeros((h, w, 2), np.int16)
    lib().orc_scharr(_p(img, up), w, h, w, out.ctypes.data_as(C.POINTER(C.c_int16)))
    return out


def klt(prev, nxt, pts):
    prev = np.ascontiguousarray(prev, np.uint8)
    nxt = np.ascontiguousarray(nxt, np.uint8)
    pts = np.ascontiguousarray(pts, np.float32)
    h, w = prev.shape
    out = np.zeros_like(pts)
    st = np.zeros(len(pts), np.uint8)
    lib().orc_klt(_p(prev, up), _p(nxt, up), w, h, w, _p(pts, fp), len(pts), _p(out, fp), _p(st, up))
    return out, st


def klt_float(prev, nxt, pts):
    """measurement only: the tracker with OpenCV's scalar-path float accumulators (row-major)"""
    prev = np.ascontiguousarray(prev, np.uint8)
    nxt = np.ascontiguousarray(nxt, np.uint8)
    pts = np.ascontiguousarray(pts, np.float32)
    h, w = prev.shape
    out = np.zeros_like(pts)
    st = np.zeros(len(pts), np.uint8)
    lib().orc_klt_float(_p(prev, up), _p(nxt, up), w, h, w, _p(pts, fp), len(pts), _p(out, fp), _p(st, up))
    return out, st


def min_eig_cvorder(img):
    """measurement only: 3x3 box sums as cv::boxFilter's running sums"""
    img = np.ascontiguousarray(img, np.uint8)
    h, w = img.shape
    out = np.zeros((h, w), np.float32)
    lib().orc_min_eig_cvorder(_p(img, up), w, h, w, _p(out, fp))
    return out


def corner_subpix_rowmajor(img, pts, win=7):
    """measurement only: cornerSubPix with its five sums as row-major chains"""
    img = np.ascontiguousarray(img, np.uint8)
    pts = np.array(pts, np.float32, copy=True)
    h, w = img.shape
    lib().orc_corner_subpix_rowmajor(_p(img, up), w, h, w, _p(pts, fp), len(pts), int(win))
    return pts


class Tracker:
    def __init__(self, cfg):
        self.cfg = cfg
        self.h = C.c_void_p(lib().orc_tracker_create(C.byref(cfg)))
        self.Fu = abi.fu(cfg)

    def __del__(self):
        if getattr(self, "h", None) and getattr(self, "_owned", True):
            lib().orc_tracker_destroy(self.h)
            self.h = None

    def track(self, img, imu, cand=None):
        """cand=None: the oracle runs FeatureDetector::DetectWithSubPix itself (as the reference does)"""
        img = np.ascontiguousarray(img, np.uint8)
        imu = np.ascontiguousarray(imu)
        info = abi.rvio_frame_info()
        if cand is None:
            cp, cn = None, 0
        else:
            cand = np.ascontiguousarray(cand, np.float32)
            cp, cn = _p(cand, fp), len(cand)
        lib().orc_tracker_track(self.h, _p(img, up), img.shape[1], imu.ctypes.data_as(C.POINTER(abi.rvio_imu)), len(imu),
                                cp, cn, C.byref(info))
        return info.asdict()

    def track_points(self, tracked, status, imu, cand):
        tracked = np.ascontiguousarray(tracked, np.float32)
        status = np.ascontiguousarray(status, np.uint8)
        imu = np.ascontiguousarray(imu)
        cand = np.ascontiguousarray(cand, np.float32)
        info = abi.rvio_frame_info()
        lib().orc_tracker_track_points(self.h, _p(tracked, fp), _p(status, up), imu.ctypes.data_as(C.POINTER(abi.rvio_imu)), len(imu),
                                       _p(cand, fp), len(cand), C.byref(info))
        return info.asdict()

    def get_tracks(self):
        ML = self.cfg.max_track_len
        types, lens, meas = np.zeros(self.Fu, np.uint8), np.zeros(self.Fu, np.int32), np.zeros((self.Fu, ML, 2), np.float32)
        n = C.c_int32(0)
        lib().orc_tracker_get_tracks(self.h, C.byref(n), _p(types, up), _p(lens, ip), _p(meas, fp))
        return types[: n.value].copy(), lens[: n.value].copy(), meas[: n.value].copy()

    def get_points(self):
        F = self.cfg.n_features
        xy, hl = np.zeros((F, 2), np.float32), np.zeros(F, np.int32)
        n = C.c_int32(0)
        lib().orc_tracker_get_points(self.h, C.byref(n), _p(xy, fp), _p(hl, ip))
        return xy[: n.value].copy(), hl[: n.value].copy()


class System:
    """oracle mirror of the timed body of System::MonoVIO."""

    def __init__(self, cfg, information_form=False):
        """information_form=True: diagnostic mode — the update runs in the device's formulation ([A|b] = Hw^T[Hw|r] with the
        structural form of the reference's rank truncation, oracle/filter.cpp:orc_update_local/global) instead of the literal
        Givens QR + leading-row scan (Updater.cc:469-529)."""
        self.cfg = cfg
        self.h = C.c_void_p(lib().orc_system_create(C.byref(cfg)))
        self.nmax = cfg.max_track_len - 1
        if information_form:
            lib().orc_system_set_information_form(self.h, 1)

    def last_rank(self):
        return int(lib().orc_system_last_rank(self.h))

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_system_destroy(self.h)
            self.h = None

    def set_state(self, x, P):
        x = np.ascontiguousarray(x, float)
        Pf = np.asfortranarray(P, dtype=float)
        lib().orc_system_set_state(self.h, _p(x, dp), len(x), Pf.ctypes.data_as(dp), P.shape[0])

    def get_state(self):
        xb = np.zeros(26 + 7 * (self.nmax + 1))
        Pb = np.zeros((24 + 6 * (self.nmax + 1)) ** 2)
        xd, d = C.c_int(0), C.c_int(0)
        lib().orc_system_get_state(self.h, _p(xb, dp), C.byref(xd), _p(Pb, dp), C.byref(d))
        return xb[: xd.value].copy(), Pb[: d.value ** 2].reshape(d.value, d.value, order="F").copy()

    def frame(self, imu, cand, img=None, tracked=None, status=None):
        imu = np.ascontiguousarray(imu)
        if cand is None:
            cand = np.zeros((0, 2), np.float32)
            cand_p = None
        else:
            cand = np.ascontiguousarray(cand, np.float32)
            cand_p = _p(cand, fp)
        info = abi.rvio_frame_info()
        tms, pp, pq = np.zeros(4), np.zeros(3), np.zeros(4)
        if img is not None:
            img = np.ascontiguousarray(img, np.uint8)
            ia, st, tx, ss = _p(img, up), img.shape[1], None, None
        else:
            tracked = np.ascontiguousarray(tracked, np.float32)
            status = np.ascontiguousarray(status, np.uint8)
            ia, st, tx, ss = None, 0, _p(tracked, fp), _p(status, up)
        lib().orc_system_frame(self.h, ia, st, tx, ss, imu.ctypes.data_as(C.POINTER(abi.rvio_imu)), len(imu),
                               cand_p, len(cand), C.byref(info), _p(tms, dp), _p(pp, dp), _p(pq, dp))
        return info.asdict(), tms, pp, pq

    def tracker(self):
        t = Tracker.__new__(Tracker)
        t.cfg, t.Fu = self.cfg, abi.fu(self.cfg)
        t._owned = False  # borrowed: never destroyed through this view
        t.h = C.c_void_p(lib().orc_system_tracker(self.h))
        return t
