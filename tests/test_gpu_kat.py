"""Hand-checkable single-edge known answers THROUGH THE DEVICE KERNELS.

tests/test_oracle_kat.py pins the CPU oracles with these triples; here the same hand values -- wrap cases at +-pi,
theta = -pi exactly, DCS at scale = 1 exactly -- go through k_chi2 (sgo_edge_chi2, sgo_chi2) and k_linearize
(sgo_linearize) on the GPU.  Every triple is an edge between a fixed and a free vertex of its own, so that the free
vertex's b and diagonal block are that one edge's -J^T W e and J^T W J with J = B (free second endpoint) or J = A
(free first endpoint); A and B are written out here from SURVEY.md section 8(a) a4, e from the hand table.
"""
import numpy as np
import pytest

from sparse_gslam_amd import capi

pytestmark = pytest.mark.gpu
PI = np.pi

# xi, xj, z, hand-derived e = Z^-1 * (Xi^-1 * Xj)   (the table of tests/test_oracle_kat.py + exact-boundary cases)
HAND = [
    ((0, 0, 0), (1, 0, 0), (1, 0, 0), (0, 0, 0)),
    ((0, 0, 0), (2, 1, 0.5), (1, 0, 0), (1, 1, 0.5)),
    ((1, 2, PI / 2), (1, 3, PI / 2), (1, 0, 0), (0, 0, 0)),
    ((1, 2, PI / 2), (0, 2, PI / 2), (0, 1, 0), (0, 0, 0)),
    ((0, 0, 3.0), (0, 0, -3.0), (0, 0, 0), (0, 0, 2 * PI - 6.0)),     # wrap through +-pi
    ((0, 0, -3.0), (0, 0, 3.0), (0, 0, 0), (0, 0, 6.0 - 2 * PI)),
    ((0, 0, 0), (0, 0, 0), (0, 0, PI / 2), (0, 0, -PI / 2)),
    ((0, 0, 0), (1, 1, 0), (1, 1, PI / 2), (0, 0, -PI / 2)),          # Z^-1 rotates the residual
    ((0, 0, 0), (0, 0, PI), (0, 0, 0), (0, 0, -PI)),                  # theta = pi is represented as -pi: range [-pi, pi)
    ((0, 0, 0), (0, 0, -PI), (0, 0, 0), (0, 0, -PI)),                 # theta = -pi exactly stays
    ((0, 0, 0), (0, 0, 3 * PI), (0, 0, 0), (0, 0, -PI)),
    ((0, 0, 0), (0, 0, 2 * PI + 0.5), (0, 0, 0), (0, 0, 0.5)),
    ((0, 0, 0), (0, 0, -7.0), (0, 0, 0), (0, 0, -7.0 + 2 * PI)),
    ((0, 0, PI - 1e-9), (0, 0, -PI + 1e-9), (0, 0, 0), (0, 0, 2e-9)),  # a hair across the wrap
]
OMEGA = np.array([[4.0, 0.5, 0.25], [0.5, 9.0, 0.75], [0.25, 0.75, 16.0]])
UT6 = OMEGA[np.triu_indices(3)]


def _jacobians(xi, xj, z):
    """A = d e / d Xi, B = d e / d Xj of EdgeSE2; Rz = blockdiag(R(theta of Z^-1), 1)."""
    ti = xi[2]
    s, c = np.sin(ti), np.cos(ti)
    dx, dy = xj[0] - xi[0], xj[1] - xi[1]
    tz = -z[2]
    Rz = np.array([[np.cos(tz), -np.sin(tz), 0], [np.sin(tz), np.cos(tz), 0], [0, 0, 1.0]])
    A = Rz @ np.array([[-c, -s, -s * dx + c * dy], [s, -c, -c * dx - s * dy], [0, 0, -1.0]])
    B = Rz @ np.array([[c, s, 0], [-s, c, 0], [0, 0, 1.0]])
    return A, B


def _graph(free_second: bool, phi: float):
    n = len(HAND)
    poses = np.zeros((2 * n, 3))
    fixed = np.zeros(2 * n, dtype=np.uint8)
    ei, ej = np.arange(0, 2 * n, 2, dtype=np.int32), np.arange(1, 2 * n, 2, dtype=np.int32)
    meas = np.zeros((n, 3))
    for k, (xi, xj, z, _) in enumerate(HAND):
        poses[2 * k], poses[2 * k + 1], meas[k] = xi, xj, z
        fixed[2 * k + (0 if free_second else 1)] = 1
    return poses, fixed, ei, ej, meas, np.tile(UT6, (n, 1)), np.full(n, phi)


@pytest.mark.parametrize("free_second", [True, False])
def test_hand_triples_through_the_device_kernels(free_second):
    args = _graph(free_second, -1.0)
    with capi.Optimizer(0) as o:
        o.set_graph(*args)
        e2 = o.edge_chi2()
        plain, robust = o.chi2()
        b, diag, c2, rc2 = o.linearize()
        ids = o.free_ids()
    want_e2 = np.array([np.array(e) @ OMEGA @ np.array(e) for *_, e in HAND])
    assert np.abs(e2 - want_e2).max() <= 1e-12 * max(1.0, want_e2.max())
    assert abs(plain - want_e2.sum()) <= 1e-12 * want_e2.sum() and abs(robust - plain) <= 1e-12 * plain
    assert abs(c2 - plain) <= 1e-12 * plain and abs(rc2 - robust) <= 1e-12 * plain
    assert list(ids) == [2 * k + (1 if free_second else 0) for k in range(len(HAND))]
    for k, (xi, xj, z, e) in enumerate(HAND):
        A, B = _jacobians(np.array(xi, float), np.array(xj, float), np.array(z, float))
        J = B if free_second else A
        assert np.abs(b[k] + J.T @ OMEGA @ np.array(e)).max() <= 1e-12 * 16.0 * max(1.0, np.abs(J).max()), (k, b[k])
        assert np.abs(diag[k] - J.T @ OMEGA @ J).max() <= 1e-12 * np.abs(J.T @ OMEGA @ J).max(), k


@pytest.mark.parametrize("phi,scale2", [(1.0, None), (17.0, 1.0), (100.0, 1.0), (-1.0, 1.0)])
def test_dcs_weight_on_the_device_including_scale_one(phi, scale2):
    """e = (1, 1, 0.5), Omega = diag(4, 9, 16): e2 = 17.  DCS: s = 2 phi / (phi + e2), clamped at 1 (phi = e2 is exactly the
    boundary); rho = (s^2 e2, s^2).  theta_i = 0 and Z has no rotation => B = I: the free vertex's diagonal block is rho1 * Omega."""
    info = np.array([[4.0, 0, 0, 9.0, 0, 16.0]])
    if scale2 is None:
        scale2 = (2 * phi / (phi + 17.0)) ** 2
    with capi.Optimizer(0) as o:
        o.set_graph(np.array([[0.0, 0, 0], [2, 1, 0.5]]), np.array([1, 0], np.uint8), np.array([0], np.int32), np.array([1], np.int32),
                    np.array([[1.0, 0, 0]]), info, np.array([phi]))
        e2 = o.edge_chi2()
        plain, robust = o.chi2()
        b, diag, _, _ = o.linearize()
    assert abs(e2[0] - 17.0) <= 1e-13 and abs(plain - 17.0) <= 1e-13
    assert abs(robust - scale2 * 17.0) <= 1e-13 * 17.0
    assert np.abs(diag[0] - scale2 * np.diag([4.0, 9.0, 16.0])).max() <= 1e-13 * 16.0
    assert np.abs(b[0] + scale2 * np.array([4.0, 9.0, 8.0])).max() <= 1e-13 * 9.0
    if phi == 17.0:
        assert robust == plain and diag[0][0, 0] == 4.0       # scale == 1 exactly: the un-robustified branch, bit for bit
