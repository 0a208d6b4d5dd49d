"""ctypes binding of oracle/_ref/libref.so — the reference's OWN sources compiled against oracle/refshim/ (TEST INFRASTRUCTURE).

Only tests/test_ref_pins.py imports this module; it exists to pin the CPU oracle (oracle/*.cpp) to the code it restates.
libref.so can only be built where /root/reference is present (`make -C oracle ref`); `available()` says whether it is there.
"""
import ctypes as C
import os
import shutil
import subprocess
import tempfile

import numpy as np

import oracle as O

abi = O.abi
ROOT = O.ROOT
REF_SRC = os.environ.get("RVIO_REFERENCE", "/root/reference")
LIB = os.path.join(ROOT, "oracle", "_ref", "libref.so")
dp, fp, ip, up = O.dp, O.fp, O.ip, O.up
_p = O._p
_LIB = None


def available():
    """build (when the reference's sources are here) or find libref.so"""
    if os.path.isdir(os.path.join(REF_SRC, "src", "rvio")):
        r = subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "ref", "REF=" + REF_SRC], capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("oracle/_ref failed to build from %s:\n%s" % (REF_SRC, r.stderr[-3000:]))
    return os.path.exists(LIB)


def _sig(L):
    L.ref_chi2_95.restype = C.c_double
    L.ref_ransac.restype = C.c_int
    L.ref_tracker_create.restype = C.c_void_p
    L.ref_system_create.restype = C.c_void_p
    L.ref_system_frame.restype = C.c_int
    L.ref_system_set_state.restype = C.c_int
    return L


def lib():
    global _LIB
    if _LIB is None:
        O.lib()  # liborc.so first: libref.so forwards the OpenCV image algorithms to it
        _LIB = _sig(C.CDLL(LIB))
    return _LIB


_COPIES = 0


def private_copy():
    """a separately loaded copy of libref.so: System::MonoVIO keeps its counters in function-local statics (System.cc:175-176),
    so every free-running sequence needs its own image of the library.  The copy stays in oracle/_ref/ (its $ORIGIN/.. rpath
    finds liborc.so) and is removed once loaded."""
    global _COPIES
    _COPIES += 1
    O.lib()
    fd, path = tempfile.mkstemp(prefix="libref_copy_%d_%d_" % (os.getpid(), _COPIES), suffix=".so", dir=os.path.dirname(LIB))
    os.close(fd)
    shutil.copy(LIB, path)
    try:
        return _sig(C.CDLL(path))
    finally:
        os.unlink(path)


def quat_mul(q1, q2):
    o = np.zeros(4)
    lib().ref_quat_mul(_p(np.ascontiguousarray(q1, float), dp), _p(np.ascontiguousarray(q2, float), dp), _p(o, dp))
    return o


def quat_to_rot(q):
    o = np.zeros(9)
    lib().ref_quat_to_rot(_p(np.ascontiguousarray(q, float), dp), _p(o, dp))
    return o.reshape(3, 3)


def rot_to_quat(R):
    o = np.zeros(4)
    lib().ref_rot_to_quat(_p(np.ascontiguousarray(R, float).reshape(-1), dp), _p(o, dp))
    return o


def chi2_95(dof):
    return lib().ref_chi2_95(int(dof))


def initialize(cfg, w, a, n_imu):
    x, P = np.zeros(26), np.zeros((24, 24), order="F")
    lib().ref_initialize(C.byref(cfg), _p(np.ascontiguousarray(w, float), dp), _p(np.ascontiguousarray(a, float), dp), int(n_imu),
                         _p(x, dp), P.ctypes.data_as(dp))
    return x, np.array(P)


def propagate(cfg, x, P, imu):
    x = np.ascontiguousarray(x, float)
    d = P.shape[0]
    Pf = np.asfortranarray(P, dtype=float).copy(order="F")
    xo = np.zeros_like(x)
    imu = np.ascontiguousarray(imu)
    lib().ref_propagate(C.byref(cfg), _p(x, dp), len(x), Pf.ctypes.data_as(dp), d, imu.ctypes.data_as(C.POINTER(abi.rvio_imu)), len(imu),
                        _p(xo, dp))
    return xo, np.array(Pf)


def update(cfg, x, P, types, lens, meas):
    x = np.ascontiguousarray(x, float)
    d = P.shape[0]
    Pf = np.asfortranarray(P, dtype=float)
    tr = abi.make_tracks(types, lens, meas)
    xo, Po = np.zeros_like(x), np.zeros((d, d), order="F")
    info = np.zeros(6, np.int32)
    cloud = np.zeros((max(tr.n_feat, 1), 3))
    lib().ref_update(C.byref(cfg), _p(x, dp), len(x), Pf.ctypes.data_as(dp), d, C.byref(tr), _p(xo, dp), Po.ctypes.data_as(dp),
                     _p(info, ip), _p(cloud, dp))
    return xo, np.array(Po), dict(n_cloud=int(info[0]), gate_rejects=int(info[1]), invalid=int(info[2]), updated=int(info[3]),
                                  hf_deficient=int(info[4]), hw_deficient=int(info[5]), cloud=cloud[: info[0]].copy())


def augment_compose(cfg, x, P, do_augment):
    nmax = cfg.max_track_len - 1
    xb = np.zeros(26 + 7 * (nmax + 1))
    D = 24 + 6 * (nmax + 1)
    Pb = np.zeros(D * D)
    d = P.shape[0]
    xb[: len(x)] = x
    Pb[: d * d] = np.asfortranarray(P, dtype=float).reshape(-1, order="F")
    xd, dd = C.c_int(len(x)), C.c_int(d)
    pp, pq = np.zeros(3), np.zeros(4)
    lib().ref_augment_compose(C.byref(cfg), _p(xb, dp), C.byref(xd), _p(Pb, dp), C.byref(dd), int(bool(do_augment)), _p(pp, dp), _p(pq, dp))
    return xb[: xd.value].copy(), Pb[: dd.value ** 2].reshape(dd.value, dd.value, order="F").copy(), pp, pq


def ransac(cfg, p1, p2, imu, flags, seed=1):
    """p1/p2: [n, 3]; returns (n_inliers, flags, pairs [16, 2], votes [16])"""
    p1 = np.ascontiguousarray(p1, float)
    p2 = np.ascontiguousarray(p2, float)
    fl = np.ascontiguousarray(flags, np.uint8).copy()
    imu = np.ascontiguousarray(imu)
    pairs, votes = np.zeros(32, np.int32), np.zeros(16, np.int32)
    n = lib().ref_ransac(C.byref(cfg), _p(p1, dp), _p(p2, dp), len(fl), imu.ctypes.data_as(C.POINTER(abi.rvio_imu)), len(imu), _p(fl, up),
                         C.c_uint(seed), _p(pairs, ip), _p(votes, ip))
    return int(n), fl, pairs.reshape(16, 2), votes


class Tracker:
    """RVIO::Tracker (Tracker.cc) — same call shapes as oracle.Tracker"""

    def __init__(self, cfg, L=None):
        self.cfg, self.L = cfg, (L or lib())
        self.Fu = abi.fu(cfg)
        self.h = C.c_void_p(self.L.ref_tracker_create(C.byref(cfg)))

    def __del__(self):
        if getattr(self, "h", None):
            self.L.ref_tracker_destroy(self.h)
            self.h = None

    def _track(self, img, tracked, status, imu, cand):
        imu = np.ascontiguousarray(imu)
        cand_a = None if cand is None else np.ascontiguousarray(cand, np.float32)
        ia = st = tx = ss = None
        if img is not None:
            img = np.ascontiguousarray(img, np.uint8)
            ia, st = _p(img, up), img.shape[1]
        else:
            tracked = np.ascontiguousarray(tracked, np.float32)
            status = np.ascontiguousarray(status, np.uint8)
            tx, ss = _p(tracked, fp), _p(status, up)
        self.L.ref_tracker_track(self.h, ia, st or 0, tx, ss, imu.ctypes.data_as(C.POINTER(abi.rvio_imu)), len(imu),
                                 _p(cand_a, fp) if cand_a is not None else None, 0 if cand_a is None else len(cand_a))

    def track(self, img, imu, cand):
        self._track(img, None, None, imu, cand)

    def track_points(self, tracked, status, imu, cand):
        self._track(None, tracked, status, imu, cand)

    def get_tracks(self):
        return _tracks(self.cfg, lambda *a: self.L.ref_tracker_get_tracks(self.h, *a))

    def get_points(self):
        return _points(self.cfg, lambda *a: self.L.ref_tracker_get_points(self.h, *a))


def _tracks(cfg, call):
    Fu, ML = abi.fu(cfg), cfg.max_track_len
    types, lens, meas = np.zeros(Fu, np.uint8), np.zeros(Fu, np.int32), np.zeros((Fu, ML, 2), np.float32)
    n = C.c_int32(0)
    call(C.byref(n), _p(types, up), _p(lens, ip), _p(meas, fp))
    return types[: n.value].copy(), lens[: n.value].copy(), meas[: n.value].copy()


def _points(cfg, call):
    F = cfg.n_features
    xy, hl = np.zeros((F, 2), np.float32), np.zeros(F, np.int32)
    n = C.c_int32(0)
    call(C.byref(n), _p(xy, fp), _p(hl, ip))
    return xy[: n.value].copy(), hl[: n.value].copy()


class System:
    """RVIO::System driven through PushImuData / PushImageData / MonoVIO on a private copy of libref.so"""

    def __init__(self, cfg):
        self.cfg, self.L = cfg, private_copy()
        self.nmax = cfg.max_track_len - 1
        self.h = C.c_void_p(self.L.ref_system_create(C.byref(cfg)))
        assert self.h

    def set_state(self, x, P):
        x = np.ascontiguousarray(x, float)
        Pf = np.asfortranarray(P, dtype=float)
        assert self.L.ref_system_set_state(self.h, _p(x, dp), len(x), Pf.ctypes.data_as(dp), P.shape[0]) == 0

    def get_state(self):
        xb = np.zeros(26 + 7 * (self.nmax + 1))
        Pb = np.zeros((24 + 6 * (self.nmax + 1)) ** 2)
        xd, d = C.c_int(0), C.c_int(0)
        self.L.ref_system_get_state(self.h, _p(xb, dp), C.byref(xd), _p(Pb, dp), C.byref(d))
        return xb[: xd.value].copy(), Pb[: d.value ** 2].reshape(d.value, d.value, order="F").copy()

    def frame(self, imu, cand, img=None, tracked=None, status=None):
        imu = np.ascontiguousarray(imu)
        cand_a = None if cand is None else np.ascontiguousarray(cand, np.float32)
        ia = tx = ss = None
        st = 0
        if img is not None:
            img = np.ascontiguousarray(img, np.uint8)
            ia, st = _p(img, up), img.shape[1]
        else:
            tracked = np.ascontiguousarray(tracked, np.float32)
            status = np.ascontiguousarray(status, np.uint8)
            tx, ss = _p(tracked, fp), _p(status, up)
        info, pp, pq = np.zeros(5, np.int32), np.zeros(3), np.zeros(4)
        ok = self.L.ref_system_frame(self.h, ia, st, tx, ss, imu.ctypes.data_as(C.POINTER(abi.rvio_imu)), len(imu),
                                     _p(cand_a, fp) if cand_a is not None else None, 0 if cand_a is None else len(cand_a),
                                     _p(info, ip), _p(pp, dp), _p(pq, dp))
        assert ok == 1, "MonoVIO did not consume the frame"
        return dict(n_cloud=int(info[0]), gate_rejects=int(info[1]), invalid=int(info[2]), updated=int(info[3]), n_tracked_out=int(info[4])), pp, pq

    def get_tracks(self):
        return _tracks(self.cfg, lambda *a: self.L.ref_system_get_tracks(self.h, *a))

    def get_points(self):
        return _points(self.cfg, lambda *a: self.L.ref_system_get_points(self.h, *a))
