"""Round 6: NumPy models of three device-side reformulations, each against the serial form it replaces (SetPointPair, the squared-distance threshold,
the pose-chain scan).  First: RANSAC's SetPointPair (Ransac.cc:50-83) runs on the device as a WAVE-PARALLEL walk of glibc's rand() stream
(csrc/frontend_kernels.hip ransac_body): the additive-feedback generator r[i] = r[i-3] + r[i-31] advances 31 draws per "turn" of its ring as a
stride-3 prefix sum, a draw is accepted exactly when its value has not been drawn before (first occurrence), the 16 pairs are the first 32
accepted draws in order and the stream stops at the draw that delivered the 32nd.  This is the NumPy model of that algorithm, step for step
as the kernel does it (turn, prefix steps 3 / 6 / 12 / 24, atomicMin-style first-occurrence table, partial last turn), against the reference's
serial loops on the oracle's rand() (which tests/test_oracle_pins.py pins to glibc): same pairs, same generator state afterwards — for every
candidate count from the minimum 32 (where all 32 values must turn up: dozens of turns) upwards, on many seeds.  The GPU tests hold the kernel
itself to the oracle bit for bit; this test pins the ARGUMENT the kernel rests on."""
import numpy as np
import pytest

import oracle as O


def _seeded_state(seed):
    """glibc random_r TYPE_3 state as the oracle / device keep it: r[0..30], front index, rear index, seeded flag"""
    st = np.zeros(34, np.int64)
    word = seed if seed else 1
    st[0] = word
    for i in range(1, 31):
        hi, lo = word // 127773, word % 127773
        w = 16807 * lo - 2836 * hi
        if w < 0:
            w += 2147483647
        word = w
        st[i] = word
    st[31], st[32], st[33] = 3, 0, 1
    for _ in range(310):
        _next(st)
    return st


def _next(st):
    f, r = int(st[31]), int(st[32])
    v = (int(st[f]) + int(st[r])) & 0xffffffff
    st[f] = v
    st[31], st[32] = (f + 1) % 31, (r + 1) % 31
    return v >> 1


def serial_pairs(st, nc):
    """Ransac.cc:50-83 literally"""
    used = np.zeros(nc, bool)
    pairs = []
    for _ in range(16):
        while True:
            a = _next(st) % nc
            if not used[a]:
                break
        while True:
            b = _next(st) % nc
            if not (used[b] or a == b):
                break
        pairs.append((a, b))
        used[a] = used[b] = True
    return pairs


def wave_pairs(st, nc):
    """the kernel's formulation: 31 draws per turn by a stride-3 prefix sum over the ring, first occurrences, partial last turn"""
    first = np.full(nc, 0x7fffffff, np.int64)
    f = int(st[31])
    total, kbase, taken = 0, 0, []
    for _turn in range(4096):
        pos = [(f + t) % 31 for t in range(31)]
        x = np.array([int(st[pos[t]]) for t in range(31)], np.int64)
        for t in range(3):
            x[t] = (x[t] + int(st[(pos[t] + 28) % 31])) & 0xffffffff
        d = 3
        while d < 31:                                           # lane t adds lane t - d (all lanes at once: the values before the step)
            y = x.copy()
            for t in range(d, 31):
                x[t] = (y[t] + y[t - d]) & 0xffffffff
            d *= 2
        val = (x >> 1) % nc
        for t in range(31):                                     # atomicMin
            first[val[t]] = min(first[val[t]], kbase + t)
        acc = np.array([first[val[t]] == kbase + t for t in range(31)])
        rank = total + np.concatenate(([0], np.cumsum(acc)[:-1]))
        for t in range(31):
            if acc[t] and rank[t] < 32:
                taken.append(int(val[t]))
        na = int(acc.sum())
        done = total + na >= 32
        c = 31
        if done:
            c = int(np.flatnonzero(acc & (rank == 31))[0]) + 1
        for t in range(c):
            st[pos[t]] = x[t]
        f = (f + c) % 31
        if done:
            break
        total += na
        kbase += 31
    st[31], st[32] = f, (f + 28) % 31
    return [(taken[2 * i], taken[2 * i + 1]) for i in range(16)]


def test_the_model_generator_is_the_oracles_rand():
    st = _seeded_state(1)
    assert [_next(st) for _ in range(200)] == list(O.rand_stream(200, seed=1))


@pytest.mark.parametrize("nc", [32, 33, 34, 40, 64, 165, 200, 1600])
def test_wave_parallel_pairs_equal_the_serial_loops(nc):
    for seed in range(1, 13):
        a, b = _seeded_state(seed), _seeded_state(seed)
        for _frame in range(6):                                 # consecutive frames continue the same stream
            sp, wp = serial_pairs(a, nc), wave_pairs(b, nc)
            assert sp == wp, (nc, seed, _frame)
            assert np.array_equal(a[:33], b[:33]), (nc, seed, _frame)
        assert len({v for p in wp for v in p}) == 32


def _sqrt_gt_threshold(m):
    """csrc/frontend_kernels.hip sqrt_gt_threshold: the largest double T with not (sqrt(T) > m)"""
    if not (m >= 0):
        return -1.0
    T = np.float64(m) * np.float64(m)
    for _ in range(8):
        nx = np.nextafter(T, np.inf)
        if np.sqrt(nx) > m:
            break
        T = nx
    return T


def test_squared_distance_threshold_takes_the_same_decisions_as_the_sqrt_comparison():
    """book-keeping's ChessGrid walk (FeatureDetector.cc:78-150) compares sqrt(dx^2 + dy^2) with nMinDist; the device compares the squared
    distance with sqrt_gt_threshold(nMinDist) instead — the same decision for EVERY double, because the rounded root is monotonic and flips
    at one double, which the threshold finds with the same sqrt.  Checked on the doubles around the flip and on random ones."""
    rng = np.random.default_rng(0)
    for m in [np.float32(15.0), np.float32(7.5), np.float32(1e-3), np.float32(0.0), np.float32(123.456)] + list(rng.uniform(0.1, 300, 200).astype(np.float32)):
        m = np.float64(m)
        T = _sqrt_gt_threshold(m)
        s = m * m
        around = [s]
        for _ in range(6):
            around.append(np.nextafter(around[-1], np.inf))
        lo = s
        for _ in range(6):
            lo = np.nextafter(lo, -np.inf)
            around.append(lo)
        around += list(rng.uniform(0, 4 * s + 1, 50))
        for v in around:
            if v >= 0:
                assert (np.sqrt(v) > m) == (v > T), (m, v, T)


def test_pose_chain_as_a_prefix_scan_of_affine_maps():
    """U1 (Updater.cc:114-141): R_I(i) = R(q_i) R_I(i-1), t_I(i) = R(q_i) (t_I(i-1) - p_i).  The device evaluates it as an inclusive Hillis-Steele scan of
    the affine maps A_i(x) = R_i x - R_i p_i over a 16-lane row (csrc/filter_kernels.hip pose_chain_row16: steps 1, 2, 4, 8).  NumPy model of exactly
    those steps against the serial recursion: equal to rounding for every chain length the row holds."""
    rng = np.random.default_rng(3)
    for n in range(1, 16):
        q = rng.standard_normal((n, 4)); q /= np.linalg.norm(q, axis=1, keepdims=True)
        p = rng.standard_normal((n, 3))
        R = np.stack([O.quat_to_rot(qi) for qi in q])
        RI, tI, want = R[0], -R[0] @ p[0], []
        for i in range(n):
            if i > 0:
                tI = R[i] @ (tI - p[i]); RI = R[i] @ RI
            want.append((RI.copy(), tI.copy()))
        Rs, cs = R.copy(), np.stack([-R[i] @ p[i] for i in range(n)])
        d = 1
        while d < 16:
            Rp, cp = Rs.copy(), cs.copy()
            for l in range(d, n):
                cs[l] = Rp[l] @ cp[l - d] + cp[l]
                Rs[l] = Rp[l] @ Rp[l - d]
            d *= 2
        for i in range(n):
            assert np.max(np.abs(Rs[i] - want[i][0])) < 1e-14 and np.max(np.abs(cs[i] - want[i][1])) < 1e-13 * (1 + np.max(np.abs(want[i][1])))
