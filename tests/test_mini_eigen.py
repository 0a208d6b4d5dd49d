"""oracle/refshim/mini_eigen.hpp — the Eigen surface the reference's sources are compiled against in oracle/_ref — held against
NumPy / LAPACK and against what Eigen 3.3 documents.

oracle/_ref pins the oracle to the reference's OWN code, but that code computes through a home-made Eigen: these tests are what stands
behind that shim.  Numerical routines (PartialPivLU::inverse, ColPivHouseholderQR::solve, products) are compared with LAPACK-backed
NumPy / SciPy to a few ulps of the condition; semantics that no tolerance captures are checked exactly:
  * JacobiRotation::makeGivens' four cases and G.adjoint() applied on the left (Jacobi.h) — incl. the exact c / s of the zero cases the
    reference's sweeps depend on (tests/test_truncation.py);
  * `A = .5*(A + A.transpose())` WITHOUT a temporary, as Eigen's Release evaluator runs it (column-major, element by element: the
    documented aliasing of the transpose), against a NumPy walk of exactly that loop — bit for bit;
  * rank-deficient ColPivHouseholderQR::solve (zero in the free component);
  * the comma initialiser's row-major order, head / tail, blocks through .eval().
The probe (tests/hostemu/eigen_probe.cpp) spells the Eigen API exactly as the reference does (Updater.cc:239,388-400,420,501-510,543,
System.cc:297)."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest
import scipy.linalg as sla

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
SRC = os.path.join(HERE, "hostemu", "eigen_probe.cpp")
HDR = os.path.join(ROOT, "oracle", "refshim", "mini_eigen.hpp")
LIB = os.path.join(HERE, "hostemu", "libeigen_probe.so")
dp = C.POINTER(C.c_double)


def _p(a):
    return a.ctypes.data_as(dp)


@pytest.fixture(scope="module")
def L():
    if not os.path.exists(LIB) or os.path.getmtime(LIB) < max(os.path.getmtime(SRC), os.path.getmtime(HDR)):
        subprocess.check_call(["g++", "-O1", "-std=c++17", "-shared", "-fPIC", "-Wall", "-I" + os.path.join(ROOT, "oracle", "refshim"), SRC, "-o", LIB])
    return C.CDLL(LIB)


def F(a):
    return np.asfortranarray(a, dtype=float)


@pytest.mark.parametrize("n,kind", [(3, "general"), (19, "spd"), (60, "spd"), (60, "general"), (180, "spd")])
def test_inverse_against_lapack(L, n, kind):
    rng = np.random.default_rng(n)
    A = rng.standard_normal((n, n))
    if kind == "spd":   # an innovation covariance: H P H^T + s2 I
        A = A @ np.diag(rng.uniform(1e-4, 1.0, n)) @ A.T + 1e-6 * np.eye(n)
    A, X = F(A), F(np.zeros((n, n)))
    L.probe_inverse(_p(A), n, _p(X))
    ref = np.linalg.inv(A)
    assert np.max(np.abs(X - ref)) <= (10 * np.linalg.cond(A) * np.finfo(float).eps + 1e-13) * np.max(np.abs(ref))   # two backward-stable LU inverses
    assert np.max(np.abs(A @ X - np.eye(n))) <= 1e-13 * np.linalg.cond(A)


@pytest.mark.parametrize("r,c", [(3, 3), (19, 19), (27, 27), (40, 12)])
def test_col_piv_qr_solve_against_lapack(L, r, c):
    rng = np.random.default_rng(100 * r + c)
    A = rng.standard_normal((r, c))
    if r == c:
        A = A @ A.T + 0.01 * np.eye(r)          # the gate's S_f (Updater.cc:415-420) is SPD
    b = rng.standard_normal((r, 2))
    A, b, X = F(A), F(b), F(np.zeros((c, 2)))
    L.probe_qr_solve(_p(A), r, c, _p(b), 2, _p(X))
    ref = np.linalg.lstsq(A, b, rcond=None)[0]   # (square full rank: the solution; tall: the least-squares solution QR gives)
    assert np.max(np.abs(X - ref)) <= 1e-12 * np.linalg.cond(A) * np.max(np.abs(ref))


def test_col_piv_qr_solve_fixed_3x3_and_rank_deficient(L):
    rng = np.random.default_rng(5)
    A = rng.standard_normal((3, 3))
    A = A.T @ A + np.diag([1e-2, 1e-2, 1e-2])     # the LM normal matrix of Updater.cc:235-239
    b = rng.standard_normal(3)
    x = np.zeros(3)
    L.probe_qr_solve3(_p(F(A)), _p(b), _p(x))
    assert np.max(np.abs(x - np.linalg.solve(A, b))) <= 1e-13 * np.linalg.cond(A)
    # rank 2: a zero column.  Eigen's solve returns the basic solution: zero in the component of the dependent (last-pivoted) column
    A2 = A.copy()
    A2[:, 1] = 0.0
    b2 = A2 @ np.array([0.3, 123.0, -0.7])
    L.probe_qr_solve3(_p(F(A2)), _p(b2), _p(x))
    assert x[1] == 0.0 and np.max(np.abs(x[[0, 2]] - [0.3, -0.7])) <= 1e-12


def test_make_givens_cases_and_adjoint_on_the_left(L):
    rng = np.random.default_rng(9)
    cases = [(3.0, 4.0), (-3.0, 4.0), (4.0, -3.0), (-4.0, -3.0), (1e-300, 1.0), (1.0, 1e-300), (2.5, 0.0), (-2.5, 0.0), (0.0, 2.5), (0.0, -2.5)]
    cases += [tuple(rng.standard_normal(2) * 10.0 ** rng.integers(-8, 8)) for _ in range(200)]
    for p, q in cases:
        rows = F(np.vstack([[p] + list(rng.standard_normal(4)), [q] + list(rng.standard_normal(4))]))
        out, csr = F(np.zeros((2, 5))), np.zeros(3)
        L.probe_givens(C.c_double(p), C.c_double(q), _p(rows), 5, _p(out), _p(csr))
        c, s, r = csr
        # Jacobi.h: G = [c s; -s c], G^* (p, q)^T = (r, 0)^T
        if q == 0.0:
            assert (c, s, r) == (-1.0 if p < 0 else 1.0, 0.0, abs(p))
        elif p == 0.0:
            assert (c, s, r) == (0.0, 1.0 if q < 0 else -1.0, abs(q))
        else:
            assert abs(abs(r) - np.hypot(p, q)) <= 4e-16 * np.hypot(p, q)
            assert r > 0                                                     # u takes the sign of the larger entry, so r = |(p, q)|
        assert abs(c * c + s * s - 1.0) <= 1e-15
        assert abs(out[0, 0] - r) <= 1e-15 * max(abs(r), 1e-300) and abs(out[1, 0]) <= 1e-15 * max(abs(r), 1e-300)
        # the other columns: x' = c x - s y, y' = s x + c y (the adjoint of apply_rotation_in_the_plane's x' = c x + s y, y' = -s x + c y)
        exp0, exp1 = c * rows[0] - s * rows[1], s * rows[0] + c * rows[1]
        if not (c == 1.0 and s == 0.0):
            assert np.array_equal(out[0, 1:], exp0[1:]) and np.array_equal(out[1, 1:], exp1[1:])
        else:
            assert np.array_equal(out, rows)                                # the identity rotation leaves the rows untouched


def test_in_place_symmetrisation_aliases_like_the_release_evaluator(L):
    """A = .5*(A + A.transpose()) with no temporary: column by column, element (i, j) reads A(j, i) AS IT IS NOW — for i < j that entry was
    already overwritten while column i was written.  The result is not symmetric for a non-symmetric A (0.75 A_ij + 0.25 A_ji above the
    diagonal), and equals the symmetric mean to rounding when A is symmetric to rounding — the case the reference relies on."""
    rng = np.random.default_rng(3)
    n = 7
    A = rng.standard_normal((n, n))
    model = A.copy()
    for j in range(n):
        for i in range(n):
            model[i, j] = 0.5 * (model[i, j] + model[j, i])
    X = F(A.copy())
    L.probe_symmetrise(_p(X), n)
    assert np.array_equal(X, model)
    assert not np.allclose(X, X.T)                                           # the documented aliasing, reproduced
    iu = np.triu_indices(n, 1)
    assert np.allclose(X[iu], 0.75 * A[iu] + 0.25 * A.T[iu], rtol=0, atol=1e-15)
    Y = F(A.copy())
    L.probe_symmetrise_eval(_p(Y), n)                                        # through .eval(): the plain symmetric mean
    assert np.array_equal(Y, 0.5 * (A + A.T))
    # a covariance that is symmetric up to rounding: both forms agree to rounding, the in-place one is what the reference computes
    S0 = A @ A.T
    S1 = S0 + 1e-17 * rng.standard_normal((n, n))
    X = F(S1.copy())
    L.probe_symmetrise(_p(X), n)
    assert np.max(np.abs(X - 0.5 * (S1 + S1.T))) <= 1e-15 * np.max(np.abs(S0))


def test_products_and_block_assignment(L):
    rng = np.random.default_rng(4)
    m, k, n = 9, 13, 5
    A, B = F(rng.standard_normal((m, k))), F(rng.standard_normal((k, n)))
    AB, Cm, Dm = F(np.zeros((m, n))), F(rng.standard_normal((m, m))), F(rng.standard_normal((k, k)))
    C0 = Cm.copy()
    L.probe_products(_p(A), m, k, _p(B), n, _p(AB), _p(Cm), _p(Dm))
    assert np.max(np.abs(AB - A @ B)) <= 1e-14 * k
    assert np.max(np.abs(Cm - (A @ A.T + C0))) <= 1e-14 * k
    assert np.max(np.abs(Dm - 2.0 * (A.T @ A))) <= 1e-14 * m


def test_initialiser_order_blocks_norms(L):
    o = np.zeros(32)
    L.probe_misc(_p(o))
    assert list(o[:9]) == [1, 2, 3, 4, 5, 6, 7, 8, 10]                       # `<<` fills row by row
    assert o[9] == 169.0 and o[10] == 13.0
    assert np.allclose(o[11:14], [3 / 13, 4 / 13, 12 / 13], rtol=0, atol=1e-16)
    assert (o[14], o[15]) == (2.0, 5.0)                                      # head(2)(1), tail(2)(0)
    assert o[16] == 1.0
    assert (o[17], o[18]) == (10.0, 16.0)                                    # diagonal()(2), trace()
    assert o[19] == 7.0
    assert (o[20], o[21]) == (5.0, 10.0)                                     # overlapping blocks through .eval()
