"""Deterministic synthetic EuRoC-shaped input generator.

No dataset ships with the reference and none is reachable, so every parity
test and the benchmark run on this generator: a smooth 6-DoF sinusoidal
trajectory with a 2 s stationary start, 200 Hz IMU derived from it (white noise
+ bias walk with the YAML sigmas), 4000 landmarks on a room shell, and either
  * rendered u8 frames (Gaussian blobs on a low-frequency background) for the
    KLT path, or
  * "direct-track" projections (+ pixel noise) that stand in for the KLT output
    so the filter load is exact and repeatable.
The landmark projections double as the corner detector's output (the detector
is outside the hot path: SURVEY.md 8(f)#3).
"""
import math
import numpy as np

from . import abi


def _skew(v):
    return np.array([[0, -v[2], v[1]], [v[2], 0, -v[0]], [-v[1], v[0], 0.0]])


def _expm_so3(th):
    a = np.linalg.norm(th)
    K = _skew(th)
    if a < 1e-9:
        return np.eye(3) + K + 0.5 * K @ K
    return np.eye(3) + (math.sin(a) / a) * K + ((1 - math.cos(a)) / a ** 2) * K @ K


def _ramp(t, t0=2.0, t1=4.0):
    """C^2 smoothstep 0 -> 1 on [t0, t1]: stationary before t0."""
    s = np.clip((t - t0) / (t1 - t0), 0.0, 1.0)
    return s * s * s * (s * (6 * s - 15) + 10)


class SynthSequence:
    def __init__(self, cfg, n_landmarks=4000, cam_hz=20.0, duration=20.0, seed=0,
                 pixel_noise=None, drop_prob=0.02, motion_scale=1.0, motion="sinus", scene="room"):
        """motion: "sinus" (the default 6-DoF sinusoids), "stationary" (the platform never moves: zero parallax), "rotation" (pure rotation
        about the CAMERA centre: zero parallax with changing views), "line" (constant-velocity straight line without rotation once the
        ramp has ended: the motion a monocular window cannot scale).  scene: "room" (landmarks on the shell of a room) or "sphere" (every
        landmark 4 m from the start position: one common depth).  The non-default modes exist to stress the rank truncation of the
        measurement compression (tests/test_truncation.py) with windows its structural form was not derived for."""
        assert motion in ("sinus", "stationary", "rotation", "line") and scene in ("room", "sphere")
        self.motion, self.scene = motion, scene
        self.cfg = cfg
        self.cam_hz = cam_hz
        self.imu_hz = float(cfg.imu_rate)
        self.duration = duration
        self.seed = seed
        self.drop_prob = drop_prob
        self.motion_scale = motion_scale
        self.T_bc = np.array(list(cfg.T_bc)).reshape(4, 4)
        self.R_bc, self.t_bc = self.T_bc[:3, :3], self.T_bc[:3, 3]
        # default pixel noise = sigma_im in pixels (about 1 px for EuRoC cam0)
        self.pixel_noise = float(max(cfg.sigma_px, cfg.sigma_py) * cfg.fx) if pixel_noise is None else pixel_noise
        rng = np.random.default_rng(7 + seed)
        # landmarks on the shell of a 10 x 10 x 6 m room centred on the trajectory
        n = n_landmarks
        face = rng.integers(0, 6, n)
        u = rng.uniform(-1, 1, (n, 2))
        half = np.array([5.0, 5.0, 3.0])
        L = np.zeros((n, 3))
        for f in range(6):
            ax, sgn = f // 2, 1.0 if f % 2 else -1.0
            m = face == f
            others = [a for a in range(3) if a != ax]
            L[m, ax] = sgn * half[ax]
            L[m, others[0]] = u[m, 0] * half[others[0]]
            L[m, others[1]] = u[m, 1] * half[others[1]]
        if scene == "sphere":
            v = rng.standard_normal((n, 3))
            L = 4.0 * v / np.linalg.norm(v, axis=1, keepdims=True)
        self.landmarks = L
        self.amps = np.random.default_rng(11 + seed).uniform(80, 200, n)
        self._imu_cache = None
        # background: 8 random low-frequency sinusoids, amplitude 30 (fixed per sequence, drifts with attitude)
        brng = np.random.default_rng(13 + seed)
        self._bg = (brng.uniform(0.004, 0.02, (8, 2)), brng.uniform(0, 2 * math.pi, 8))

    # ---------------------------------------------------------------- trajectory
    def pose(self, t):
        """(R_wb, p_w) at time t."""
        s = float(_ramp(np.asarray(t, dtype=float))) * self.motion_scale
        if self.motion == "stationary":
            return self._R0, np.zeros(3)
        if self.motion == "line":      # (t - 2) ramp(t) is C^2 at the start and exactly linear in t behind the ramp
            return self._R0, s * (t - 2.0) * np.array([0.3, 0.4, 0.05])
        p = s * np.array([2 * math.sin(0.5 * t), 1.5 * math.sin(0.7 * t + 1), 0.5 * math.sin(0.9 * t + 2)])
        th = s * np.array([0.2 * math.sin(0.6 * t), 0.15 * math.sin(0.8 * t), 0.3 * math.sin(0.4 * t)])
        # base attitude: camera looks roughly along world +x
        R = self._R0 @ _expm_so3(th)
        if self.motion == "rotation":  # the camera centre p + R t_bc stays where it was at rest
            return R, self._R0 @ self.t_bc - R @ self.t_bc
        return R, p

    @property
    def _R0(self):
        # body z (camera optical axis is ~ body z for EuRoC cam0 up to the extrinsic) -> world x
        return np.array([[0.0, 0.0, 1.0], [-1.0, 0.0, 0.0], [0.0, -1.0, 0.0]])

    def _kin(self, t, h=1e-4):
        """body angular rate and specific force (noise-free) at t by central differences."""
        Rm, pm = self.pose(t - h)
        R0, p0 = self.pose(t)
        Rp, pp = self.pose(t + h)
        acc = (pp - 2 * p0 + pm) / (h * h)
        dR = (Rp - Rm) / (2 * h)
        W = R0.T @ dR
        w = np.array([W[2, 1] - W[1, 2], W[0, 2] - W[2, 0], W[1, 0] - W[0, 1]]) * 0.5
        g_w = np.array([0.0, 0.0, self.cfg.gravity])
        a = R0.T @ (acc + g_w)
        return w, a

    def imu_all(self):
        """All IMU samples of the sequence (structured array, abi.IMU_DTYPE)."""
        if self._imu_cache is not None:
            return self._imu_cache
        n = int(round(self.duration * self.imu_hz)) + 1
        dt = 1.0 / self.imu_hz
        rng = np.random.default_rng(42 + self.seed)
        c = self.cfg
        bg = np.zeros(3)
        ba = np.zeros(3)
        out = np.zeros(n, dtype=abi.IMU_DTYPE)
        for i in range(n):
            t = i * dt
            w, a = self._kin(t)
            bg = bg + c.sigma_wg * math.sqrt(dt) * rng.standard_normal(3)
            ba = ba + c.sigma_wa * math.sqrt(dt) * rng.standard_normal(3)
            out["w"][i] = w + bg + c.sigma_g / math.sqrt(dt) * rng.standard_normal(3)
            out["a"][i] = a + ba + c.sigma_a / math.sqrt(dt) * rng.standard_normal(3)
            out["t"][i] = t
            out["dt"][i] = dt if i else 0.0
        self._imu_cache = out
        return out

    def frame_time(self, k):
        return k / self.cam_hz

    def n_frames(self):
        return int(self.duration * self.cam_hz)

    def imu_between(self, k):
        """IMU samples with t in (t_{k-1}, t_k] — what InputBuffer::GetMeasurements
        (InputBuffer.cc:53-81) hands to MonoVIO for frame k."""
        imu = self.imu_all()
        t1 = self.frame_time(k)
        t0 = self.frame_time(k - 1) if k > 0 else -1.0
        eps = 1e-9
        m = (imu["t"] > t0 + eps) & (imu["t"] <= t1 + eps)
        return np.ascontiguousarray(imu[m])

    # ---------------------------------------------------------------- camera
    def project(self, k, noise=True):
        """Pixel projections (float32 [N,2]) of all landmarks at frame k and the
        visibility mask (in front, inside the image with a 20 px margin)."""
        c = self.cfg
        R_wb, p_w = self.pose(self.frame_time(k))
        pb = (self.landmarks - p_w) @ R_wb            # R_wb^T (l - p)
        pc = (pb - self.t_bc) @ self.R_bc             # R_bc^T (pb - t_bc)
        z = pc[:, 2]
        ok = z > 0.5
        zs = np.where(ok, z, 1.0)
        x, y = pc[:, 0] / zs, pc[:, 1] / zs
        r2 = x * x + y * y
        k1, k2, p1, p2, k3 = float(c.k1), float(c.k2), float(c.p1), float(c.p2), float(c.k3)
        cd = 1 + ((k3 * r2 + k2) * r2 + k1) * r2
        xd = x * cd + 2 * p1 * x * y + p2 * (r2 + 2 * x * x)
        yd = y * cd + p1 * (r2 + 2 * y * y) + 2 * p2 * x * y
        u = float(c.fx) * xd + float(c.cx)
        v = float(c.fy) * yd + float(c.cy)
        ok &= r2 < 1.2  # keep the distortion model in its monotone range
        if noise and self.pixel_noise > 0:
            nr = np.random.default_rng(1000003 * (self.seed + 1) + k).standard_normal((len(u), 2))
            u = u + self.pixel_noise * nr[:, 0]
            v = v + self.pixel_noise * nr[:, 1]
        margin = 20.0
        ok &= (u > margin) & (u < c.width - margin) & (v > margin) & (v < c.height - margin)
        return np.stack([u, v], 1).astype(np.float32), ok

    def candidates(self, k, xy, vis):
        """The 'detector output' for frame k: visible projections in a deterministic
        shuffled order, at most n_features of them (maxCorners, FeatureDetector.cc:63)."""
        ids = np.flatnonzero(vis)
        np.random.default_rng(77 + 131 * k + self.seed).shuffle(ids)
        ids = ids[: self.cfg.n_features]
        return np.ascontiguousarray(xy[ids]), ids

    def drops(self, k, n):
        """Random track losses (simulated KLT failures) for frame k."""
        return np.random.default_rng(555 + 17 * k + self.seed).uniform(size=n) < self.drop_prob

    def render(self, k):
        """u8 H x W frame: Gaussian blobs (sigma 2.5 px) at the noise-free projections."""
        c = self.cfg
        W, H = c.width, c.height
        xy, vis = self.project(k, noise=False)
        fr, ph = self._bg
        yy, xx = np.mgrid[0:H, 0:W].astype(np.float32)
        R_wb, _ = self.pose(self.frame_time(k))
        shift = 40.0 * R_wb[:2, 2]
        img = np.full((H, W), 100.0, dtype=np.float32)
        for i in range(8):
            img += (30.0 / 8) * np.sin(fr[i, 0] * (xx + shift[0]) + fr[i, 1] * (yy + shift[1]) + ph[i]).astype(np.float32)
        sig, rad = 2.5, 8
        ax = np.arange(-rad, rad + 1, dtype=np.float32)
        for j in np.flatnonzero(vis):
            u, v = xy[j]
            iu, iv = int(round(float(u))), int(round(float(v)))
            x0, x1, y0, y1 = iu - rad, iu + rad + 1, iv - rad, iv + rad + 1
            if x0 < 0 or y0 < 0 or x1 > W or y1 > H:
                continue
            gx = np.exp(-((ax + iu - u) ** 2) / (2 * sig * sig))
            gy = np.exp(-((ax + iv - v) ** 2) / (2 * sig * sig))
            img[y0:y1, x0:x1] += (self.amps[j] * 0.5) * np.outer(gy, gx)
        return np.clip(np.rint(img), 0, 255).astype(np.uint8)

    # ---------------------------------------------------------------- init helper
    def init_from_static(self, k_first):
        """Average of the IMU samples before frame k_first: what MonoVIO's
        stationary gate (System.cc:183-249) feeds System::initialize."""
        imu = self.imu_all()
        m = imu["t"] <= self.frame_time(k_first) + 1e-9
        return imu["w"][m].mean(0), imu["a"][m].mean(0), int(m.sum())


class DirectTrackDriver:
    """Feeds a tracker object (oracle or HIP host wrapper; duck-typed:
    track_points / get_points) with simulated KLT results, keeping the
    landmark identity of every tracked feature by exact pixel-key lookup."""

    def __init__(self, seq):
        self.seq = seq
        self.ids = np.zeros(0, dtype=np.int64)

    @staticmethod
    def _keys(xy):
        return [xy[i].tobytes() for i in range(len(xy))]

    def inputs(self, k):
        seq = self.seq
        xy, vis = seq.project(k)
        cand, cand_ids = seq.candidates(k, xy, vis)
        tracked = np.ascontiguousarray(xy[self.ids]) if len(self.ids) else np.zeros((0, 2), np.float32)
        status = (vis[self.ids] & ~seq.drops(k, len(self.ids))).astype(np.uint8) if len(self.ids) else np.zeros(0, np.uint8)
        self._lookup = {key: int(i) for key, i in zip(self._keys(xy[vis]), np.flatnonzero(vis))}
        return dict(imu=seq.imu_between(k), tracked=tracked, status=status, cand=cand)

    def after(self, points_xy):
        """Call with the tracker's mvFeatsToTrack after the frame."""
        self.ids = np.array([self._lookup[key] for key in self._keys(np.asarray(points_xy, np.float32))], dtype=np.int64)


# ---------------------------------------------------------------- full update loads (no images, no tracker)
def _q2r_jpl(q):
    """QuatToRot of util/Numerics.h:111-120 (JPL): I - 2 w [q]x + 2 [q]x^2"""
    qx = _skew(q[:3])
    return np.eye(3) - 2 * q[3] * qx + 2 * qx @ qx


def _qmul_jpl(a, b):
    """QuatMul of util/Numerics.h:30-63 (JPL, normalised, w >= 0)"""
    q = np.array([a[3] * b[0] + a[2] * b[1] - a[1] * b[2] + a[0] * b[3],
                  -a[2] * b[0] + a[3] * b[1] + a[0] * b[2] + a[1] * b[3],
                  a[1] * b[0] - a[0] * b[1] + a[3] * b[2] + a[2] * b[3],
                  -a[0] * b[0] - a[1] * b[1] - a[2] * b[2] + a[3] * b[3]])
    q /= np.linalg.norm(q)
    return q if q[3] >= 0 else -q


def worst_case_tracks(cfg, x, n_feat=None, seed=0, mix="half"):
    """A full update load for the state vector x (window of n clones): ceil(F/2) features that are geometrically consistent with
    the clone poses of x — each a random 3-D point in front of its first camera, projected through the state's own relative-pose
    chain (Updater.cc:125-141) plus sigma_im noise, so that the chi-square gate accepts most of them.
    mix = "half": every second feature is type '2' at the maximum track length, the others type '1' with L ~ U[3, n+1] (the
    direct-track load of SURVEY.md 8d);  mix = "long": every feature is type '1' with L = n+1, i.e. 2L-3 stacked rows each —
    the worst case W_filter of SURVEY.md 8d is quoted on (cfg B: 100 x 19 = 1900 rows, 51 MFLOP)."""
    rng = np.random.default_rng(1234 + seed)
    n = (len(x) - 26) // 7
    Fu = abi.fu(cfg) if n_feat is None else n_feat
    ML = cfg.max_track_len
    T = np.array(list(cfg.T_bc)).reshape(4, 4)
    Ric, tic = T[:3, :3], T[:3, 3]
    Rci, tci = Ric.T, -Ric.T @ tic
    sig = float(max(cfg.sigma_px, cfg.sigma_py))
    types = np.zeros(Fu, np.uint8)
    lens = np.zeros(Fu, np.int32)
    meas = np.zeros((Fu, ML, 2), np.float32)
    for f in range(Fu):
        if mix == "long":
            ty, L = ord("1"), n + 1
        elif f % 2 == 0 and n + 1 == ML:
            ty, L = ord("2"), ML
        else:
            ty, L = ord("1"), int(rng.integers(3, n + 2))
        nph = L - 1
        rel = x[26 + 7 * n - 7 * nph:] if ty == ord("1") else x[26:26 + 7 * nph]
        qI = [rel[0:4]]
        tI = [-_q2r_jpl(rel[0:4]) @ rel[4:7]]
        for i in range(1, nph):
            qi = rel[7 * i:7 * i + 4]
            tI.append(_q2r_jpl(qi) @ (tI[-1] - rel[7 * i + 4:7 * i + 7]))
            qI.append(_qmul_jpl(qi, qI[-1]))
        depth = rng.uniform(2.0, 8.0)
        pc1 = np.array([rng.uniform(-0.5, 0.5) * depth, rng.uniform(-0.35, 0.35) * depth, depth])
        obs = [pc1[:2] / pc1[2]]
        for i in range(nph):
            RI = _q2r_jpl(qI[i])
            Rc = Rci @ RI @ Ric
            tc = Rci @ RI @ tic + Rci @ tI[i] + tci
            pc = Rc @ pc1 + tc
            obs.append(pc[:2] / pc[2])
        obs = np.array(obs) + sig * rng.standard_normal((L, 2))
        types[f], lens[f] = ty, L
        meas[f, :L] = obs.astype(np.float32)
    return types, lens, meas
