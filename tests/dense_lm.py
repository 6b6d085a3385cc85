"""The first Levenberg-Marquardt step of a local-BA window, computed densely with numpy from per-edge residuals and Jacobians:
(J^T W J + lambda I) dx = -J^T W r over ALL unknowns (free poses, then landmarks), lambda = 1e-5 max diag (g2o's initial lambda,
G/core/optimization_algorithm_levenberg.cpp:171-185), Huber weights rho' (G/core/robust_kernel_impl.cpp:78-91), applied with
exp(dx) * T / X + dx.  Independent of the oracle's and the product's solvers (Schur complement, LDL^T, back-substitution): both are
held against it.  The per-edge residuals / Jacobians come from the oracle's edge evaluation, which tests/test_oracle_lba.py and
tests/test_oracle_rig.py pin against central differences."""
import numpy as np

from multi_orbslam3_amd import _capi as capi
from multi_orbslam3_amd import views
from oracle import binding as ob


def quat_mul(a, b):
    ax, ay, az, aw = a
    bx, by, bz, bw = b
    return np.array([aw * bx + ax * bw + ay * bz - az * by, aw * by + ay * bw + az * bx - ax * bz, aw * bz + az * bw + ax * by - ay * bx,
                     aw * bw - ax * bx - ay * by - az * bz])


def quat_rot(q, v):
    u = np.array(q[:3])
    uv = 2 * np.cross(u, v)
    return v + q[3] * uv + np.cross(u, uv)


def quat_from_R(R):
    w = np.sqrt(max(0.0, 1 + R[0, 0] + R[1, 1] + R[2, 2])) / 2
    return np.array([(R[2, 1] - R[1, 2]) / (4 * w), (R[0, 2] - R[2, 0]) / (4 * w), (R[1, 0] - R[0, 1]) / (4 * w), w])


def oplus(q, t, upd):
    """VertexSE3Expmap::oplusImpl: exp(upd) * T."""
    eq, et = ob.se3_exp(upd)
    return quat_mul(eq, q), et + quat_rot(eq, t)


def dense_first_step(pr, rig=None):
    """pr: a synth.make_lba_*problem dict; rig: views.camera_rig(...) or None.  Returns (poses after the step as (P, 3, 4) float64 with
    the fixed ones unchanged, points after the step (L, 3), dx)."""
    E = pr["edges"]
    P = len(pr["poses"]); free = [i for i in range(P) if not pr["pose_fixed"][i]]
    pcol = {i: c for c, i in enumerate(free)}
    L = len(pr["points"])
    nu = 6 * len(free) + 3 * L
    H = np.zeros((nu, nu)); b = np.zeros(nu)
    quats, ts = [], []
    for i in range(P):
        T = np.asarray(pr["poses"][i], np.float32).reshape(4, 4).astype(np.float64)
        q = quat_from_R(T[:3, :3]); q /= np.linalg.norm(q)
        quats.append(q); ts.append(T[:3, 3].copy())
    X0 = np.asarray(pr["points"], np.float64)
    d_mono, d_stereo = float(np.float32(np.sqrt(5.991))), float(np.float32(np.sqrt(7.815)))
    for e in E:
        i, l = int(e["pose"]), int(e["point"])
        e1 = np.array([e], capi.EDGE_DTYPE)
        if rig is not None:
            err, A, B, _ = ob.lba_edge_eval_rig(quats[i], ts[i], X0[l], pr["cam"], rig, e1)
        else:
            err, A, B = ob.lba_edge_eval(quats[i], ts[i], X0[l], pr["cam"], e1)
        D = 2 if e["ur"] < 0 else 3
        delta = d_mono if D == 2 else d_stereo
        om = float(e["inv_sigma2"])
        c2 = om * float(err[:D] @ err[:D])
        w = 1.0 if c2 <= delta * delta else delta / np.sqrt(c2)
        J = np.zeros((D, nu))
        if i in pcol:
            J[:, 6 * pcol[i]:6 * pcol[i] + 6] = B[:D]
        J[:, 6 * len(free) + 3 * l:6 * len(free) + 3 * l + 3] = A[:D]
        H += J.T @ (w * om * J); b -= J.T @ (w * om * err[:D])
    lam = 1e-5 * np.abs(np.diag(H)).max()
    dx = np.linalg.solve(H + lam * np.eye(nu), b)
    poses = np.zeros((P, 3, 4))
    for i in range(P):
        q, t = (quats[i], ts[i]) if i not in pcol else oplus(quats[i], ts[i], dx[6 * pcol[i]:6 * pcol[i] + 6])
        q = q / np.linalg.norm(q)
        poses[i, :, :3] = np.array([quat_rot(q, ex) for ex in np.eye(3)]).T
        poses[i, :, 3] = t
    return poses, X0 + dx[6 * len(free):].reshape(L, 3), dx


def first_step_of(solve, pr, rig=None):
    """Run `solve(problem, stop_flag)` (the oracle's or the product's) with the flag that reads as raised once one LM trial has been
    evaluated; returns its output object."""
    p, keep = views.lba_problem(pr["poses"], pr["pose_fixed"], pr["points"], pr["edges"], pr["cam"], rig=rig)
    stop = np.array([-1], np.int32)
    out = solve(p, stop)
    assert out.iters == (1, 0), out.iters
    return out
