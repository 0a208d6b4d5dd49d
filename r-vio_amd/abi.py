"""ctypes mirror of include/rvio_hip.h (POD structs + function prototypes).

The C header is the source of truth; tests/test_abi.py checks sizeof/offsets
against the compiled library and that every declared symbol is exported.
"""
import ctypes as C
import math
import numpy as np

ABI_VERSION = 5


class rvio_config(C.Structure):
    _fields_ = [
        ("imu_rate", C.c_double), ("sigma_g", C.c_double), ("sigma_wg", C.c_double),
        ("sigma_a", C.c_double), ("sigma_wa", C.c_double), ("gravity", C.c_double),
        ("small_angle", C.c_double),
        ("width", C.c_int32), ("height", C.c_int32),
        ("fx", C.c_float), ("fy", C.c_float), ("cx", C.c_float), ("cy", C.c_float),
        ("k1", C.c_float), ("k2", C.c_float), ("p1", C.c_float), ("p2", C.c_float), ("k3", C.c_float),
        ("sigma_px", C.c_float), ("sigma_py", C.c_float),
        ("T_bc", C.c_double * 16),
        ("fisheye", C.c_int32),
        ("n_features", C.c_int32), ("max_track_len", C.c_int32), ("min_track_len", C.c_int32),
        ("min_dist", C.c_float), ("qual_lvl", C.c_float),
        ("block_x", C.c_float), ("block_y", C.c_float),
        ("enable_equalizer", C.c_int32), ("use_sampson", C.c_int32),
        ("inlier_thr", C.c_double),
        ("ini_thr_angle", C.c_double), ("ini_thr_displ", C.c_double),
        ("ini_enable_alignment", C.c_int32), ("reserved0", C.c_int32),
    ]


class rvio_imu(C.Structure):
    _fields_ = [("w", C.c_double * 3), ("a", C.c_double * 3), ("t", C.c_double), ("dt", C.c_double)]


IMU_DTYPE = np.dtype([("w", "f8", 3), ("a", "f8", 3), ("t", "f8"), ("dt", "f8")])
assert IMU_DTYPE.itemsize == C.sizeof(rvio_imu) == 64


class rvio_tracks(C.Structure):
    _fields_ = [("n_feat", C.c_int32), ("max_len", C.c_int32),
                ("types", C.POINTER(C.c_ubyte)), ("len", C.POINTER(C.c_int32)), ("meas", C.POINTER(C.c_float))]


class rvio_frame_info(C.Structure):
    _fields_ = [(k, C.c_int32) for k in
                ("n_clones", "n_tracked_in", "n_klt_ok", "n_ransac_inliers", "n_feat_update",
                 "n_feat_accepted", "n_rows", "updated", "n_tracked_out", "ransac_winner")] + \
               [("reserved", C.c_int32 * 5), ("rank_truncated_at", C.c_int32)]

    def asdict(self):
        d = {k: getattr(self, k) for k, _ in self._fields_ if k != "reserved"}
        if hasattr(self, "reserved"):
            d["device_error"] = int(self.reserved[0])   # sticky: 1 singular pivot, 2 track the window cannot hold, 4 a device-side stage counter timed out, 8 indefinite gate matrix
            d["literal_rank"] = int(self.reserved[1])   # nRank of the reference's literal Givens sweep + scan when the device ran it for this update (csrc/literal.h), -1 otherwise
        return d


# Camera.T_BC0 of config/rvio_euroc.yaml:55-62 (row-major)
_T_BC0 = [0.0148655429818, -0.999880929698, 0.00414029679422, -0.0216401454975,
          0.999557249008, 0.0149672133247, 0.025715529948, -0.064676986768,
          -0.0257744366974, 0.00375618835797, 0.999660727178, 0.00981073058949,
          0.0, 0.0, 0.0, 1.0]


def config_euroc(**over):
    """Stock values of config/rvio_euroc.yaml:8-111 (same as C rvio_config_euroc)."""
    c = rvio_config()
    c.imu_rate = 200
    c.sigma_g, c.sigma_wg, c.sigma_a, c.sigma_wa = 1.6968e-04, 1.9393e-05, 2.0e-3, 3.0e-3
    c.gravity = 9.8082
    c.small_angle = 0.001745329
    c.width, c.height = 752, 480
    c.fx, c.fy, c.cx, c.cy = 458.654, 457.296, 367.215, 248.375
    c.k1, c.k2, c.p1, c.p2, c.k3 = -0.28340811, 0.07395907, 0.00019359, 1.76187114e-05, 0.0
    c.sigma_px, c.sigma_py = 0.002180293, 0.002186767
    for i, v in enumerate(_T_BC0):
        c.T_bc[i] = v
    c.fisheye = 0
    c.n_features, c.max_track_len, c.min_track_len = 200, 15, 3
    c.min_dist, c.qual_lvl = 15, 0.01
    c.block_x, c.block_y = 150, 120
    c.enable_equalizer, c.use_sampson = 1, 1
    c.inlier_thr = 1e-5
    c.ini_thr_angle, c.ini_thr_displ = 0.005, 0.01
    c.ini_enable_alignment = 1
    for k, v in over.items():
        if not hasattr(c, k):
            raise AttributeError(k)
        setattr(c, k, v)
    return c


# BASELINE.json configs (SURVEY.md section 8 size table): name -> overrides
BASELINE_CONFIGS = {
    "A": dict(n_features=200, max_track_len=15),                       # stock EuRoC V1_01
    "B": dict(n_features=200, max_track_len=11),                       # MH_01 200 f / 10 clones (headline)
    "C": dict(n_features=400, max_track_len=21),                       # 400 f / 20 clones
    "D": dict(n_features=800, max_track_len=16, width=1920, height=1080,
              fx=1171.0, fy=1171.0, cx=960.0, cy=540.0, block_x=384, block_y=270),  # 1080p
    "E": dict(n_features=1600, max_track_len=31),                      # 8-GPU 1600 f / 30 clones
}


def config_named(name, **over):
    kw = dict(BASELINE_CONFIGS[name])
    kw.update(over)
    return config_euroc(**kw)


def dims(cfg, n_clones=None):
    """(n_max, d, xdim) for the configured window (or for n_clones)."""
    n = cfg.max_track_len - 1 if n_clones is None else n_clones
    return n, 24 + 6 * n, 26 + 7 * n


def fu(cfg):
    return int(math.ceil(0.5 * cfg.n_features))


def make_tracks(types, lens, meas):
    """Build an rvio_tracks view over numpy arrays (kept alive by the caller)."""
    types = np.ascontiguousarray(types, dtype=np.uint8)
    lens = np.ascontiguousarray(lens, dtype=np.int32)
    meas = np.ascontiguousarray(meas, dtype=np.float32)
    assert meas.ndim == 3 and meas.shape[2] == 2 and meas.shape[0] >= len(types)
    t = rvio_tracks(len(types), meas.shape[1],
                    types.ctypes.data_as(C.POINTER(C.c_ubyte)),
                    lens.ctypes.data_as(C.POINTER(C.c_int32)),
                    meas.ctypes.data_as(C.POINTER(C.c_float)))
    t._keep = (types, lens, meas)
    return t


def as_imu_array(w, a, t, dt):
    arr = np.zeros(len(t), dtype=IMU_DTYPE)
    arr["w"], arr["a"], arr["t"], arr["dt"] = w, a, t, dt
    return arr


# ---- wire format of a shard's share of the information block (csrc/rvio_dev.h shard_layout: the all-gather payload of rvio_hip_update_local /
# rvio_hip_frame_sharded_dev): [8 counters | S2 tiles | S1 tiles], 16 x 16 tiles of 256 doubles, upper triangle only, S2 only where a type-'2'
# feature can reach.  NumPy mirror for tests and host-side collectives.
def shard_layout(c6, max_len):
    ntq, ntp = (c6 >> 4) + 1, (c6 + 15) >> 4
    hi2 = min(6 * ((max_len + 1) // 2 - 1), c6)
    t2 = min((hi2 - 1) >> 4 if hi2 > 0 else -1, ntp - 1)
    x2 = 1 if ntq - 1 > t2 else 0
    tiles2 = 0 if t2 < 0 else (t2 + 1) * (t2 + 2) // 2 + x2 * (t2 + 1)
    tiles1 = ntp * ntq - ntp * (ntp - 1) // 2
    return dict(ntq=ntq, ntp=ntp, t2=t2, x2=x2, tiles2=tiles2, tiles1=tiles1)


def shard_payload_doubles(c6, max_len):
    L = shard_layout(c6, max_len)
    return 8 + 256 * (L["tiles2"] + L["tiles1"])


def _shard_tiles(L):
    """[(part, pt, qt, offset)] of every carried tile (part 0 = S2, 1 = S1)"""
    out = []
    for pt in range(L["t2"] + 1):
        row = pt * (L["t2"] + 1 + L["x2"]) - pt * (pt - 1) // 2
        for qt in range(pt, L["t2"] + 1):
            out.append((0, pt, qt, 8 + 256 * (row + qt - pt)))
        if L["x2"]:
            out.append((0, pt, L["ntq"] - 1, 8 + 256 * (row + L["t2"] + 1 - pt)))
    for pt in range(L["ntp"]):
        for qt in range(pt, L["ntq"]):
            out.append((1, pt, qt, 8 + 256 * (L["tiles2"] + pt * L["ntq"] - pt * (pt - 1) // 2 + qt - pt)))
    return out


def shard_pack(parts, counters, c6, max_len):
    """parts: (2, rows >= c6, cols >= c6 + 1) full-layout S2, S1 (row p, column q; column c6 = the residual); counters: 8 numbers.
    Returns (payload, live): the wire-format vector and the mask of its entries that carry matrix elements (tile padding excluded)."""
    L = shard_layout(c6, max_len)
    pay = np.zeros(shard_payload_doubles(c6, max_len))
    live = np.zeros(len(pay), bool)
    pay[:8] = counters
    live[:8] = True
    for part, pt, qt, off in _shard_tiles(L):
        r1, c1 = min(16 * pt + 16, c6), min(16 * qt + 16, c6 + 1)
        tile = np.zeros((16, 16))
        mask = np.zeros((16, 16), bool)
        tile[: r1 - 16 * pt, : c1 - 16 * qt] = parts[part][16 * pt: r1, 16 * qt: c1]
        mask[: r1 - 16 * pt, : c1 - 16 * qt] = True
        pay[off: off + 256] = tile.reshape(-1)
        live[off: off + 256] = mask.reshape(-1)
    return pay, live


def shard_unpack(pay, c6, max_len, ld):
    """inverse of shard_pack into full-layout (2, ld, ld) matrices with the lower triangle mirrored from the upper one (the residual column has
    no mirror image), and the 8 counters"""
    L = shard_layout(c6, max_len)
    parts = np.zeros((2, ld, ld))
    for part, pt, qt, off in _shard_tiles(L):
        r1, c1 = min(16 * pt + 16, c6), min(16 * qt + 16, c6 + 1)
        parts[part][16 * pt: r1, 16 * qt: c1] = np.asarray(pay[off: off + 256]).reshape(16, 16)[: r1 - 16 * pt, : c1 - 16 * qt]
    for part in range(2):
        a = parts[part][:c6, :c6]
        iu = np.triu_indices(c6, 1)
        # tiles strictly above the diagonal tile row carry both triangles of nothing: only elements whose TILE lies on or above the diagonal are valid
        valid = (iu[1] >> 4) >= (iu[0] >> 4)
        a[(iu[1][valid], iu[0][valid])] = a[(iu[0][valid], iu[1][valid])]
    return parts, np.array(pay[:8])
