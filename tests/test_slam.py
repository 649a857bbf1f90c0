"""Host back end of SURVEY.md section 8(f) ranks 3-4 (include/pgslam_amd/slam.hpp): graph search, SE(3) maps,
candidate search (C++ checks), and the pose-graph least squares against an independent scipy solution of the
same cost -- the pin for the GTSAM replacement, since GTSAM itself is not available here."""
import os
import subprocess

import numpy as np
import pytest

from test_cpp_dropin import build, CPP, ROOT


def _so3_log(R):
    c = np.clip((np.trace(R) - 1) / 2, -1, 1)
    th = np.arccos(c)
    w = np.array([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]])
    return w * (0.5 + th * th / 12 if th < 1e-6 else th / (2 * np.sin(th)))


def _se3_log(Tm):
    w = _so3_log(Tm[:3, :3])
    th = np.linalg.norm(w)
    K = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])
    if th < 1e-6:
        Vinv = np.eye(3) - 0.5 * K + K @ K / 12
    else:
        Vinv = np.eye(3) - 0.5 * K + (1 - 0.5 * th * np.sin(th) / (1 - np.cos(th))) / th ** 2 * K @ K
    return np.concatenate([w, Vinv @ Tm[:3, 3]])


def _se3_exp(xi):
    w, v = xi[:3], xi[3:]
    th = np.linalg.norm(w)
    K = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])
    if th < 1e-9:
        R, V = np.eye(3) + K, np.eye(3) + 0.5 * K
    else:
        R = np.eye(3) + np.sin(th) / th * K + (1 - np.cos(th)) / th ** 2 * K @ K
        V = np.eye(3) + (1 - np.cos(th)) / th ** 2 * K + (th - np.sin(th)) / th ** 3 * K @ K
    Tm = np.eye(4)
    Tm[:3, :3], Tm[:3, 3] = R, V @ v
    return Tm


def _pose(vals):
    Tm = np.eye(4)
    Tm[:3, :3] = np.array(vals[:9]).reshape(3, 3)
    Tm[:3, 3] = vals[9:12]
    return Tm


def test_slam_host_cpu_and_least_squares_against_scipy():
    from scipy.optimize import least_squares
    out = subprocess.run([build("test_slam_cpu")], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert "slam cpu tests ok" in out.stdout
    edges, init, sol = [], {}, {}
    for line in out.stdout.splitlines():
        p = line.split()
        if p and p[0] == "EDGE":
            v = [float(x) for x in p[3:]]
            edges.append((int(p[1]), int(p[2]), _pose(v[:12]), np.array(v[12:]).reshape(6, 6)))
        elif p and p[0] == "INIT":
            init[int(p[1])] = _pose([float(x) for x in p[2:]])
        elif p and p[0] == "SOL":
            sol[int(p[1])] = _pose([float(x) for x in p[2:]])
        elif p and p[0] == "COST":
            c0, c1 = float(p[1]), float(p[2])
    N = len(init)
    Ws = [np.linalg.cholesky(np.linalg.inv(cov)).T for *_, cov in edges]      # r^T cov^-1 r = |W r|^2

    def residuals(x):
        X = [init[0]] + [init[i] @ _se3_exp(x[6 * (i - 1): 6 * i]) for i in range(1, N)]
        return np.concatenate([W @ _se3_log(np.linalg.inv(Z) @ np.linalg.inv(X[i]) @ X[j]) for (i, j, Z, _), W in zip(edges, Ws)])

    r0 = residuals(np.zeros(6 * (N - 1)))
    assert 0.5 * r0 @ r0 == pytest.approx(c0, rel=1e-9)                       # same cost function
    res = least_squares(residuals, np.zeros(6 * (N - 1)), method="lm", xtol=1e-14, ftol=1e-14, gtol=1e-14)
    assert c1 == pytest.approx(res.cost, rel=1e-4)                            # same optimum (the C++ LM stops at GTSAM's 1e-5 tolerances)
    Xs = [init[0]] + [init[i] @ _se3_exp(res.x[6 * (i - 1): 6 * i]) for i in range(1, N)]
    for i in range(N):
        d = np.linalg.inv(Xs[i]) @ sol[i]
        assert np.linalg.norm(d[:3, 3]) < 2e-3 and np.linalg.norm(_so3_log(d[:3, :3])) < 2e-3


@pytest.mark.gpu
def test_slam_facade_gpu():
    out = subprocess.run([build("test_slam_gpu")], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    assert "slam gpu tests ok" in out.stdout
