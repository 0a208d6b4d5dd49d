"""Loader + thin ctypes wrapper of librvio_hip.so (the C-ABI of include/rvio_hip.h).

There is NO CPU fallback: if the HIP library is missing, cannot be loaded, or no
GPU is present, every entry point raises.
"""
import ctypes as C
import os
import numpy as np

from . import abi

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RVIO_HIP_LIB", os.path.join(HERE, "librvio_hip.so"))   # override: A/B kernel experiments

# every symbol include/rvio_hip.h declares (tests/test_abi.py checks the export list)
SYMBOLS = [
    "rvio_config_euroc", "rvio_hip_create", "rvio_hip_destroy", "rvio_hip_last_error", "rvio_hip_abi_version",
    "rvio_hip_stream", "rvio_hip_sync", "rvio_hip_set_state", "rvio_hip_get_state", "rvio_hip_initialize",
    "rvio_hip_propagate", "rvio_hip_update", "rvio_hip_augment_compose", "rvio_hip_track", "rvio_hip_track_dev",
    "rvio_hip_track_points", "rvio_hip_get_tracks", "rvio_hip_get_tracker_points", "rvio_hip_update_tracked",
    "rvio_hip_frame", "rvio_hip_frame_dev", "rvio_hip_frame_points", "rvio_hip_get_frame_info", "rvio_hip_get_pose",
    "rvio_hip_update_local", "rvio_hip_update_global", "rvio_hip_get_update_diag",
    "rvio_hip_debug_pyramid", "rvio_hip_debug_tracked", "rvio_hip_frame_plan", "rvio_hip_propagate_dev",
    "rvio_hip_debug_time_kernel", "rvio_hip_get_corners", "rvio_hip_frame_begin_dev", "rvio_hip_frame_end",
    "rvio_hip_create_batch", "rvio_hip_batch_size", "rvio_hip_set_state_at", "rvio_hip_get_state_at", "rvio_hip_frame_tracks_dev",
    "rvio_hip_frame_batch_dev", "rvio_hip_get_tracker_points_at", "rvio_hip_frame_sharded_dev", "rvio_hip_debug_poison", "rvio_hip_debug_stall", "rvio_hip_debug_noise", "rvio_hip_debug_kernel_forms",
]

_LIB = None
dp, fp, ip, up = C.POINTER(C.c_double), C.POINTER(C.c_float), C.POINTER(C.c_int32), C.POINTER(C.c_ubyte)


class RvioHipError(RuntimeError):
    pass


def load():
    """dlopen the in-tree HIP library; fail loudly when it is absent."""
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise RvioHipError("librvio_hip.so not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                               "(no CPU fallback exists for the product path)")
        try:
            import torch  # noqa: F401  when torch is used in the same process its bundled HIP runtime has to be loaded first
        except ImportError:
            pass
        L = C.CDLL(LIB_PATH)
        # the version first: a library built from an older header must fail HERE, with a sentence, not at the first missing symbol
        L.rvio_hip_abi_version.restype = C.c_int
        if L.rvio_hip_abi_version() != abi.ABI_VERSION:
            raise RvioHipError("%s speaks ABI %d, this binding expects %d (include/rvio_hip.h): rebuild it — python -c 'import __graft_entry__ as g; g.build()'"
                               % (LIB_PATH, L.rvio_hip_abi_version(), abi.ABI_VERSION))
        L.rvio_hip_last_error.restype = C.c_char_p
        L.rvio_hip_last_error.argtypes = [C.c_void_p]
        L.rvio_hip_stream.restype = C.c_void_p
        L.rvio_hip_stream.argtypes = [C.c_void_p]
        L.rvio_hip_destroy.argtypes = [C.c_void_p]
        L.rvio_hip_destroy.restype = None
        _LIB = L
    return _LIB


def _p(a, t):
    return a.ctypes.data_as(t) if a is not None else None


class RvioHip:
    """One filter instance on one GPU (mirrors the System-owned stage objects, System.h:89-92)."""

    def __init__(self, cfg, device=0, batch=None, front_end=False):
        """batch=B: B independent instances behind one handle (rvio_hip_create_batch); front_end: with the tracker, else filter only"""
        self.L = load()
        self.cfg = cfg
        self.h = C.c_void_p()
        self.batch = 1 if batch is None else int(batch)
        if batch is None:
            rc = self.L.rvio_hip_create(C.byref(cfg), int(device), C.byref(self.h))
        else:
            rc = self.L.rvio_hip_create_batch(C.byref(cfg), int(device), int(batch), int(bool(front_end)), C.byref(self.h))
        if rc != 0:
            msg = self.L.rvio_hip_last_error(self.h).decode() if self.h else ""
            if self.h:
                self.L.rvio_hip_destroy(self.h)
                self.h = C.c_void_p()
            raise RvioHipError("rvio_hip_create failed: rc=%d %s" % (rc, msg))
        self.nmax = cfg.max_track_len - 1
        self.Fu = abi.fu(cfg)

    def close(self):
        if getattr(self, "h", None):
            self.L.rvio_hip_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ck(self, rc, what):
        if rc != 0:
            raise RvioHipError("%s failed: rc=%d %s" % (what, rc, self.L.rvio_hip_last_error(self.h).decode()))

    def sync(self):
        self._ck(self.L.rvio_hip_sync(self.h), "sync")

    def stream(self):
        return self.L.rvio_hip_stream(self.h)

    # ---- state
    def set_state(self, x, P):
        x = np.ascontiguousarray(x, float)
        Pf = np.asfortranarray(P, dtype=float)
        self._ck(self.L.rvio_hip_set_state(self.h, _p(x, dp), len(x), Pf.ctypes.data_as(dp), P.shape[0]), "set_state")

    def get_state(self):
        xb = np.zeros(26 + 7 * (self.nmax + 1))
        Pb = np.zeros((24 + 6 * (self.nmax + 1)) ** 2)
        xd, d = C.c_int(0), C.c_int(0)
        self._ck(self.L.rvio_hip_get_state(self.h, _p(xb, dp), C.byref(xd), _p(Pb, dp), C.byref(d)), "get_state")
        return xb[: xd.value].copy(), Pb[: d.value ** 2].reshape(d.value, d.value, order="F").copy()

    def set_state_at(self, i, x, P):
        x = np.ascontiguousarray(x, float)
        Pf = np.asfortranarray(P, dtype=float)
        self._ck(self.L.rvio_hip_set_state_at(self.h, int(i), _p(x, dp), len(x), Pf.ctypes.data_as(dp), P.shape[0]), "set_state_at")

    def get_state_at(self, i):
        xb = np.zeros(26 + 7 * (self.nmax + 1))
        Pb = np.zeros((24 + 6 * (self.nmax + 1)) ** 2)
        xd, d = C.c_int(0), C.c_int(0)
        self._ck(self.L.rvio_hip_get_state_at(self.h, int(i), _p(xb, dp), C.byref(xd), _p(Pb, dp), C.byref(d)), "get_state_at")
        return xb[: xd.value].copy(), Pb[: d.value ** 2].reshape(d.value, d.value, order="F").copy()

    def frame_tracks_dev(self, d_imu_ptr, imu_stride, m, d_n_feat_ptr, d_types_ptr, d_len_ptr, d_meas_ptr):
        """MonoVIO body after the tracker on device-resident hand-over tables, all instances in one launch per stage"""
        self._ck(self.L.rvio_hip_frame_tracks_dev(self.h, C.c_void_p(d_imu_ptr), int(imu_stride), int(m), C.c_void_p(d_n_feat_ptr),
                                                  C.c_void_p(d_types_ptr), C.c_void_p(d_len_ptr), C.c_void_p(d_meas_ptr)), "frame_tracks_dev")

    def frame_batch_dev(self, d_imgs_ptr, stride, img_stride, d_imu_ptr, imu_stride, m):
        """one camera frame of every instance of a batch handle with front end (device detector)"""
        self._ck(self.L.rvio_hip_frame_batch_dev(self.h, C.c_void_p(d_imgs_ptr), int(stride), C.c_size_t(int(img_stride)), C.c_void_p(d_imu_ptr),
                                                 int(imu_stride), int(m)), "frame_batch_dev")

    def initialize(self, w, a, n_imu):
        w = np.ascontiguousarray(w, float)
        a = np.ascontiguousarray(a, float)
        self._ck(self.L.rvio_hip_initialize(self.h, _p(w, dp), _p(a, dp), int(n_imu)), "initialize")

    # ---- stages
    def propagate(self, imu):
        imu = np.ascontiguousarray(imu)
        self._ck(self.L.rvio_hip_propagate(self.h, imu.ctypes.data_as(C.POINTER(abi.rvio_imu)), len(imu)), "propagate")

    def update(self, types, lens, meas):
        tr = abi.make_tracks(types, lens, meas)
        self._ck(self.L.rvio_hip_update(self.h, C.byref(tr)), "update")

    def update_tracked(self):
        self._ck(self.L.rvio_hip_update_tracked(self.h), "update_tracked")

    def update_local(self, types, lens, meas, rank, world):
        """returns (device pointer, n_doubles) of this rank's [A|b] block"""
        tr = abi.make_tracks(types, lens, meas)
        ptr, n = dp(), C.c_int(0)
        self._ck(self.L.rvio_hip_update_local(self.h, C.byref(tr), rank, world, C.byref(ptr), C.byref(n)), "update_local")
        return C.cast(ptr, C.c_void_p).value, n.value

    def update_global(self, d_blocks_ptr, world):
        self._ck(self.L.rvio_hip_update_global(self.h, C.c_void_p(d_blocks_ptr), world), "update_global")

    def update_diag(self):
        nf = C.c_int32(0)
        acc, gam, ndof, pf = np.zeros(self.Fu, np.int32), np.zeros(self.Fu), np.zeros(self.Fu, np.int32), np.zeros((self.Fu, 3))
        self._ck(self.L.rvio_hip_get_update_diag(self.h, C.byref(nf), _p(acc, ip), _p(gam, dp), _p(ndof, ip), _p(pf, dp)), "update_diag")
        n = nf.value
        return dict(accepted=acc[:n], gamma=gam[:n], ndof=ndof[:n], pfinv=pf[:n])

    def augment_compose(self, do_augment=True):
        self._ck(self.L.rvio_hip_augment_compose(self.h, int(do_augment)), "augment_compose")

    # ---- front end
    @staticmethod
    def _cand(cand):
        """(pointer, count) of a corner list; None selects the device detector"""
        if cand is None:
            return None, 0, None
        cand = np.ascontiguousarray(cand, np.float32)
        return _p(cand, fp), len(cand), cand

    @staticmethod
    def _img(img):
        """(array, row stride in bytes): a row-strided uint8 view is handed over as it is (cv::Mat::step), anything else is packed"""
        img = np.asarray(img)
        if img.dtype != np.uint8 or img.ndim != 2 or img.strides[1] != 1 or img.strides[0] < img.shape[1]:
            img = np.ascontiguousarray(img, np.uint8)
        return img, img.strides[0]

    def track(self, img, imu, cand=None):
        img, stride = self._img(img)
        imu = np.ascontiguousarray(imu)
        cp, cn, _keep = self._cand(cand)
        self._ck(self.L.rvio_hip_track(self.h, C.cast(img.ctypes.data, up), stride, imu.ctypes.data_as(C.POINTER(abi.rvio_imu)), len(imu),
                                       cp, cn), "track")

    def frame(self, img, imu, cand=None):
        """whole MonoVIO body from host buffers (System.cc:253-367); cand=None: device detector"""
        img, stride = self._img(img)
        imu = np.ascontiguousarray(imu)
        cp, cn, _keep = self._cand(cand)
        self._ck(self.L.rvio_hip_frame(self.h, C.cast(img.ctypes.data, up), stride, imu.ctypes.data_as(C.POINTER(abi.rvio_imu)), len(imu),
                                       cp, cn), "frame")

    def get_corners(self, want_eig=False):
        """last result of the device detector: (refined corners, goodFeaturesToTrack corners[, min-eigenvalue map])"""
        F = self.cfg.n_features
        n = C.c_int32(0)
        xy, raw = np.zeros((F, 2), np.float32), np.zeros((F, 2), np.float32)
        eig = np.zeros((self.cfg.height, self.cfg.width), np.float32) if want_eig else None
        self._ck(self.L.rvio_hip_get_corners(self.h, C.byref(n), _p(xy, fp), _p(raw, fp), _p(eig, fp) if want_eig else None), "get_corners")
        return (xy[: n.value].copy(), raw[: n.value].copy()) + ((eig,) if want_eig else ())

    def track_points(self, tracked, status, imu, cand):
        tracked = np.ascontiguousarray(tracked, np.float32)
        status = np.ascontiguousarray(status, np.uint8)
        imu = np.ascontiguousarray(imu)
        cand = np.ascontiguousarray(cand, np.float32)
        self._ck(self.L.rvio_hip_track_points(self.h, _p(tracked, fp), _p(status, up), len(status),
                                              imu.ctypes.data_as(C.POINTER(abi.rvio_imu)), len(imu), _p(cand, fp), len(cand)), "track_points")

    def frame_points(self, tracked, status, imu, cand):
        tracked = np.ascontiguousarray(tracked, np.float32)
        status = np.ascontiguousarray(status, np.uint8)
        imu = np.ascontiguousarray(imu)
        cand = np.ascontiguousarray(cand, np.float32)
        self._ck(self.L.rvio_hip_frame_points(self.h, _p(tracked, fp), _p(status, up), len(status),
                                              imu.ctypes.data_as(C.POINTER(abi.rvio_imu)), len(imu), _p(cand, fp), len(cand)), "frame_points")

    def frame_dev(self, d_img_ptr, stride, d_imu_ptr, m, d_cand_ptr, n_cand):
        self._ck(self.L.rvio_hip_frame_dev(self.h, C.c_void_p(d_img_ptr), int(stride), C.c_void_p(d_imu_ptr), int(m),
                                           C.c_void_p(d_cand_ptr), int(n_cand)), "frame_dev")

    def track_dev(self, d_img_ptr, stride, d_imu_ptr, m, d_cand_ptr, n_cand):
        self._ck(self.L.rvio_hip_track_dev(self.h, C.c_void_p(d_img_ptr), int(stride), C.c_void_p(d_imu_ptr), int(m),
                                           C.c_void_p(d_cand_ptr), int(n_cand)), "track_dev")

    def frame_plan(self):
        du, da = C.c_int(0), C.c_int(0)
        self._ck(self.L.rvio_hip_frame_plan(self.h, C.byref(du), C.byref(da)), "frame_plan")
        return bool(du.value), bool(da.value)

    def propagate_dev(self, d_imu_ptr, m):
        self._ck(self.L.rvio_hip_propagate_dev(self.h, C.c_void_p(d_imu_ptr), int(m)), "propagate_dev")

    def update_local_tracked(self, rank, world):
        """stage A of the sharded updater on the tracker's device-resident tracks"""
        ptr, n = dp(), C.c_int(0)
        self._ck(self.L.rvio_hip_update_local(self.h, None, rank, world, C.byref(ptr), C.byref(n)), "update_local")
        return C.cast(ptr, C.c_void_p).value, n.value

    def frame_tail_staged(self, d_imu_ptr, m, evs, stream, torch):
        """propagate -> [update] -> augment/compose with events evs[2..4] recorded between the stages"""
        do_update, do_augment = self.frame_plan()
        self.propagate_dev(d_imu_ptr, m)
        with torch.cuda.stream(stream):
            evs[2].record()
        if do_update:
            self.update_tracked()
        with torch.cuda.stream(stream):
            evs[3].record()
        self.augment_compose(do_augment)
        with torch.cuda.stream(stream):
            evs[4].record()
        return do_update

    def frame_begin_dev(self, d_img_ptr, stride, d_imu_ptr, m, d_cand_ptr, n_cand):
        self._ck(self.L.rvio_hip_frame_begin_dev(self.h, C.c_void_p(d_img_ptr), int(stride), C.c_void_p(d_imu_ptr), int(m),
                                                 C.c_void_p(d_cand_ptr), int(n_cand)), "frame_begin_dev")

    def frame_end(self):
        self._ck(self.L.rvio_hip_frame_end(self.h), "frame_end")

    def frame_sharded_dev(self, d_img_ptr, stride, d_imu_ptr, m, d_cand_ptr, n_cand, rank, world, comm=None, allgather=None):
        """one pipelined frame with the feature-sharded updater behind ONE C-ABI call; comm: ncclComm_t (int / c_void_p), None with world 1"""
        cp = comm.value if isinstance(comm, C.c_void_p) else comm
        self._ck(self.L.rvio_hip_frame_sharded_dev(self.h, C.c_void_p(d_img_ptr), int(stride), C.c_void_p(d_imu_ptr), int(m), C.c_void_p(d_cand_ptr), int(n_cand),
                                                   int(rank), int(world), C.c_void_p(cp), C.c_void_p(allgather)), "frame_sharded_dev")

    def frame_sharded_piped(self, d_img_ptr, stride, d_imu_ptr, m, d_cand_ptr, n_cand, rank, world, gathered, dist, DeviceArray, torch, stream,
                            force_collective=False, comm=None):
        """One pipelined frame with the feature-sharded updater, no host synchronisation: the front end of this
        frame overlaps the previous frame's filter work; local [A|b] block -> ONE all-gather, enqueued behind the block kernel
        because the handle's filter stream is torch's current stream -> replicated global update."""
        with torch.cuda.stream(stream):
            self.frame_begin_dev(d_img_ptr, stride, d_imu_ptr, m, d_cand_ptr, n_cand)
            do_update, do_augment = self.frame_plan()
            if do_update:
                if world > 1 or force_collective:
                    ptr, n = self.update_local_tracked(rank, world)
                    if comm is not None:       # RCCL directly on the filter stream (rccl.py): plain stream order, nothing else
                        comm.all_gather_f64(ptr, gathered.data_ptr(), n, self.stream())
                    else:
                        if getattr(self, "_local_key", None) != (ptr, n):  # the block lives in one fixed device buffer: wrap it once
                            self._local_key, self._local = (ptr, n), torch.as_tensor(DeviceArray(ptr, n), device="cuda")
                        dist.all_gather_into_tensor(gathered[: world * n], self._local)     # (the payload grows with the window: n doubles per rank, rank-major)
                    self.update_global(gathered.data_ptr(), world)
                else:
                    self._ck(self.L.rvio_hip_update_tracked(self.h), "update_tracked")
            self.augment_compose(do_augment)
            self.frame_end()

    def frame_tail_sharded(self, d_imu_ptr, m, rank, world, gathered, dist, DeviceArray, torch):
        """Feature-sharded frame tail: local [A|b] block -> ONE all-gather -> replicated EKF update."""
        do_update, do_augment = self.frame_plan()
        self.propagate_dev(d_imu_ptr, m)
        if do_update:
            ptr, n = self.update_local_tracked(rank, world)
            self.sync()                                   # block ready (handle stream) before the collective's stream reads it
            local = torch.as_tensor(DeviceArray(ptr, n), device="cuda")
            dist.all_gather_into_tensor(gathered[: world * n], local)
            torch.cuda.current_stream().synchronize()     # gathered blocks visible before the handle stream consumes them
            self.update_global(gathered.data_ptr(), world)
        self.augment_compose(do_augment)

    def get_tracks(self):
        ML = self.cfg.max_track_len
        types, lens, meas = np.zeros(self.Fu, np.uint8), np.zeros(self.Fu, np.int32), np.zeros((self.Fu, ML, 2), np.float32)
        n = C.c_int32(0)
        self._ck(self.L.rvio_hip_get_tracks(self.h, C.byref(n), _p(types, up), _p(lens, ip), _p(meas, fp)), "get_tracks")
        return types[: n.value].copy(), lens[: n.value].copy(), meas[: n.value].copy()

    def get_points_at(self, i):
        F = self.cfg.n_features
        xy, hl = np.zeros((F, 2), np.float32), np.zeros(F, np.int32)
        n = C.c_int32(0)
        self._ck(self.L.rvio_hip_get_tracker_points_at(self.h, int(i), C.byref(n), _p(xy, fp), _p(hl, ip)), "get_tracker_points_at")
        return xy[: n.value].copy(), hl[: n.value].copy()

    def get_points(self):
        F = self.cfg.n_features
        xy, hl = np.zeros((F, 2), np.float32), np.zeros(F, np.int32)
        n = C.c_int32(0)
        self._ck(self.L.rvio_hip_get_tracker_points(self.h, C.byref(n), _p(xy, fp), _p(hl, ip)), "get_tracker_points")
        return xy[: n.value].copy(), hl[: n.value].copy()

    def debug_pyramid(self, level):
        w, hh = C.c_int32(0), C.c_int32(0)
        self._ck(self.L.rvio_hip_debug_pyramid(self.h, level, C.byref(w), C.byref(hh), None, None), "debug_pyramid")
        img = np.zeros((hh.value, w.value), np.uint8)
        dxy = np.zeros((hh.value, w.value, 2), np.int16)
        self._ck(self.L.rvio_hip_debug_pyramid(self.h, level, C.byref(w), C.byref(hh), _p(img, up),
                                               dxy.ctypes.data_as(C.POINTER(C.c_int16))), "debug_pyramid")
        return img, dxy

    def debug_tracked(self, n):
        xy, un = np.zeros((n, 2), np.float32), np.zeros((n, 2), np.float32)
        self._ck(self.L.rvio_hip_debug_tracked(self.h, n, _p(xy, fp), _p(un, fp)), "debug_tracked")
        return xy, un

    def time_kernel(self, which, iters=20):
        """average device time (us) of one hot kernel: 0 solve, 1 KLT, 2 per-feature build, 3 share reduction, 4 U/G/P1, 5 Joseph form, 6 cornerSubPix, 7 U/G/P1 + Joseph form as launched, 8 feat_prop_kernel as the pipelined frame launches it (state restored), 9 the detector's greedy selection (HIP events, handle stream)"""
        us = C.c_float(0)
        self._ck(self.L.rvio_hip_debug_time_kernel(self.h, int(which), int(iters), C.byref(us)), "debug_time_kernel")
        return float(us.value)

    def poison(self, what=7):
        """drain the handle, then overwrite left-over state: 1 filter scratch, 2 LDS of the chip, 4 hand-over tables + tracker scratch, 8 set error bit 4"""
        self._ck(self.L.rvio_hip_debug_poison(self.h, int(what)), "debug_poison")

    def stall(self, which, usec):
        """occupy one of the handle's streams (0 filter, 1 tracker / image chain 0, 2 side, 3 image chain 1) for usec microseconds"""
        self._ck(self.L.rvio_hip_debug_stall(self.h, int(which), int(usec)), "debug_stall")

    def noise(self, wgs, usec):
        """`wgs` workgroups of HBM / L2 / LDS traffic on a stream of their own for usec microseconds (a box under load)"""
        self._ck(self.L.rvio_hip_debug_noise(self.h, int(wgs), int(usec)), "debug_noise")

    def kernel_forms(self, throughput):
        """select the throughput (batch) or latency (one stream) forms of the image kernels on this handle: same bits either way"""
        self._ck(self.L.rvio_hip_debug_kernel_forms(self.h, int(throughput)), "debug_kernel_forms")

    def frame_info(self):
        info = abi.rvio_frame_info()
        self._ck(self.L.rvio_hip_get_frame_info(self.h, C.byref(info)), "get_frame_info")
        return info.asdict()

    def pose(self):
        p, q = np.zeros(3), np.zeros(4)
        self._ck(self.L.rvio_hip_get_pose(self.h, _p(p, dp), _p(q, dp)), "get_pose")
        return p, q
