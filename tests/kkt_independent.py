"""Independent KKT evaluator for the NLPs of the hot path -- shares NO code with oracle/ or with the HIP kernels.

Why it exists: GPU == oracle only proves that two copies of one algorithm agree.  This module restates the
PROBLEM (not the solver) straight from the reference's sources and evaluates first-order optimality of a given
point (U, lambda) by reverse-mode differentiation:

* QuatMpc (legged_ctrl/src/mpc/QuatMpc.cpp:112-229; model src/utils/AltroUtils.cpp:363-392, midpoint :9-22,
  G(q) src/utils/QuaternionUtils.cpp:30-52): shooting problem in the stance inputs, torch autograd, with the
  tangent-space convention of SURVEY.md A.4/A.5 -- every perturbation of a knot's quaternion is restricted to
  q + G(q) phi, i.e. gradients w.r.t. q are multiplied by G(q) G(q)' at every knot (TangentProject below).
  Works for any number of contact points (4: Go1; 8: the synthetic biped of BASELINE config 5).
* ConvexMpc (src/mpc/ConvexMpc.cpp:81-198; model AltroUtils.cpp:224-293): the reference hands the solver an
  APPROXIMATE Jacobian (AltroUtils.cpp:295-359 omits d(I_world^-1)/d(yaw)), so the point its iLQR scheme seeks
  is a zero of the costate recursion built from THAT Jacobian pushed through the midpoint chain rule
  (AltroUtils.cpp:78-110) -- restated here in numpy; the exact gradient (autograd) is reported next to it.

Constraint rows (QuatMpc.cpp:47-52,194-215 / ConvexMpc.cpp:126-136): C_mat (R) f_l + b <= 0 per stance leg.
Swing legs carry zero force (their cone collapses to {0}); they are not variables here.
"""
from __future__ import annotations

import numpy as np
import torch

torch.set_default_dtype(torch.float64)
GRAV = 9.81


# ------------------------------------------------------------------------------------------------
# common: the pyramid rows
# ------------------------------------------------------------------------------------------------
def cone_matrix(mu: float) -> np.ndarray:
    """C_mat of QuatMpc.cpp:47-52: +-fx - mu fz <= 0, +-fy - mu fz <= 0, fz <= fz_max, -fz <= 0."""
    return np.array([[1, 0, -mu], [-1, 0, -mu], [0, 1, -mu], [0, -1, -mu], [0, 0, 1], [0, 0, -1.0]])


def cone_rows(U: np.ndarray, frame: np.ndarray, contacts: np.ndarray, mu: float, fz_max: float):
    """Row vectors a (6x3, acting on one leg's force as given in U) and values c [N][nl][6] (<= 0 when feasible)."""
    A = cone_matrix(mu) @ frame
    N = U.shape[0]
    nl = contacts.shape[0]
    F = U.reshape(N, nl, 3)
    c = np.einsum("ia,kla->kli", A, F)
    c[:, :, 4] -= fz_max * contacts[None, :]
    return A, c


def kkt_residuals(grad: np.ndarray, U: np.ndarray, lam: np.ndarray, frame, contacts, mu, fz_max) -> dict:
    """First-order optimality of (U, lam) for min J(U) s.t. cone rows, given grad = dJ/dU  [N][nu].

    stationarity   : max | grad_l + sum_i lam_i a_i | over stance legs
    complementarity: max_i min(s_i, lam_i) and max_i s_i lam_i  (s = -c)
    feasibility    : max(c, 0), min(lam)
    nnls           : stationarity with multipliers RE-DERIVED here (non-negative least squares over the rows with
                     slack <= 1e-6), i.e. without trusting the solver's multipliers at all
    """
    from scipy.optimize import nnls

    con = np.asarray(contacts) != 0
    N, nl = U.shape[0], con.shape[0]
    A, c = cone_rows(U, frame, con.astype(float), mu, fz_max)
    lam = lam.reshape(N, nl, 6)
    g = grad.reshape(N, nl, 3)
    r = g + np.einsum("kli,ia->kla", lam, A)
    s = -c
    st = float(np.abs(r[:, con]).max())
    comp_min = float(np.minimum(np.abs(s[:, con]), lam[:, con]).max())
    comp_prod = float(np.abs(s[:, con] * lam[:, con]).max())
    worst = 0.0
    for k in range(N):
        for l in np.where(con)[0]:
            act = np.where(s[k, l] <= 1e-6)[0]
            if act.size == 0:
                worst = max(worst, float(np.abs(g[k, l]).max()))
                continue
            _, rn = nnls(A[act].T, -g[k, l])
            worst = max(worst, float(rn))
    return {
        "stationarity": st,
        "complementarity_min": comp_min,
        "complementarity_prod": comp_prod,
        "violation": float(np.maximum(c[:, con], 0.0).max()),
        "lam_min": float(lam[:, con].min()),
        "swing_force": float(np.abs(U.reshape(N, nl, 3)[:, ~con]).max()) if (~con).any() else 0.0,
        "stationarity_nnls": worst,
    }


# ------------------------------------------------------------------------------------------------
# QuatMpc (4 or 8 contact points)
# ------------------------------------------------------------------------------------------------
def _G(q: torch.Tensor) -> torch.Tensor:
    """G(q) = L(q) [0; I3]  (QuaternionUtils.cpp:30-52), q = (w, x, y, z); 4x3."""
    s, x, y, z = q[0], q[1], q[2], q[3]
    return torch.stack([torch.stack([-x, -y, -z]), torch.stack([s, -z, y]), torch.stack([z, s, -x]),
                        torch.stack([-y, x, s])])


class TangentProject(torch.autograd.Function):
    """Identity on the state; the gradient w.r.t. its quaternion part is multiplied by G(q) G(q)' -- the
    perturbations the error-state formulation admits are q + G(q) phi (AltroUtils.cpp:153-168)."""

    @staticmethod
    def forward(ctx, x):
        ctx.save_for_backward(x)
        return x.clone()

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        Gm = _G(x[3:7].detach())
        out = g.clone()
        out[3:7] = Gm @ (Gm.T @ g[3:7])
        return out


class QuatProblem:
    """One QuatMpc instance.  `rec` is a numpy record with the qmpc_input / qmpc_input8 field names, `par` an object
    with the qmpc_params field names."""

    def __init__(self, par, rec):
        self.N = int(par.horizon)
        con = np.asarray(rec["contacts"], dtype=float)
        self.nl = con.shape[0]
        self.nu = 3 * self.nl
        self.con = (con != 0).astype(float)
        self.h = float(np.float32(par.h))                         # `float h` of the dynamics callbacks
        self.hh = float(np.float32(par.h) / np.float32(2))        # `h / 2` evaluated in float (AltroUtils.cpp:16)
        self.R = np.asarray(rec["rot"], dtype=float).reshape(3, 3)
        self.mass = float(par.mass)
        self.Iinv = torch.tensor(np.linalg.inv(np.asarray(par.inertia, dtype=float).reshape(3, 3)))
        self.feet = torch.tensor(np.asarray(rec["foot_pos_body"], dtype=float).reshape(self.nl, 3))
        self.gb = torch.tensor(self.R.T @ np.array([0.0, 0.0, -GRAV]))
        com = np.array([0.0223, 0.002, -0.0005])                  # AltroUtils.cpp:373
        self.mg = torch.tensor(np.cross(com, 5.204 * (self.R.T @ np.array([0.0, 0.0, -GRAV]))))
        self.Q = torch.tensor(np.asarray(par.q_weights, dtype=float))
        self.Rw = torch.tensor(np.array([par.r_weights[j % 12] for j in range(self.nu)], dtype=float))
        self.w = float(par.w)
        self.mu, self.fz_max = float(par.mu), float(par.fz_max)
        nc = int(self.con.sum())
        uref = np.zeros(self.nu)
        uref[2::3] = self.con * self.mass * GRAV / nc             # QuatMpc.cpp:121-125
        self.uref = torch.tensor(uref)
        x0 = np.zeros(13)                                         # QuatMpc.cpp:231-246
        x0[3:7] = rec["quat"]
        x0[7:10] = rec["lin_vel_body"]
        if not par.drop_ang_vel:
            x0[10:13] = rec["ang_vel_body"]
        self.x0 = torch.tensor(x0)
        h_ms = float(par.h_ref) * 1000.0
        xr = np.zeros((self.N + 1, 13))
        for i in range(self.N + 1):                               # QuatMpc.cpp:148-176 (+ acc term of the trot test)
            t = i * float(par.h_ref)
            xr[i, 0] = rec["pos_ref_body"][0] + rec["vel_ref_body"][0] * i * h_ms / 1000.0 + 0.5 * rec["acc_ref_body"][0] * t * t
            xr[i, 1] = rec["pos_ref_body"][1] + rec["vel_ref_body"][1] * i * h_ms / 1000.0 + 0.5 * rec["acc_ref_body"][1] * t * t
            xr[i, 2] = rec["pos_ref_body"][2] + 0.5 * rec["acc_ref_body"][2] * t * t
            xr[i, 3:7] = rec["quat_d"]
            xr[i, 7:10] = np.asarray(rec["vel_ref_body"]) + np.asarray(rec["acc_ref_body"]) * t
        self.xref = torch.tensor(xr)
        self.mask = torch.tensor(np.repeat(self.con, 3))

    # continuous dynamics, AltroUtils.cpp:363-392
    def f(self, x, u):
        F = u.reshape(self.nl, 3)
        tau = torch.cross(self.feet, F, dim=1).sum(0) + self.mg
        q, v, om = x[3:7], x[7:10], x[10:13]
        return torch.cat([v, 0.5 * (_G(q) @ om), F.sum(0) / self.mass + self.gb, self.Iinv @ tau])

    def step(self, x, u):                                         # AltroUtils.cpp:9-22
        xm = x + self.hh * self.f(x, u)
        return x + self.h * self.f(xm, u)

    def rollout(self, U, project=True):
        X = [self.x0]
        for k in range(self.N):
            xn = self.step(X[-1], U[k] * self.mask)
            X.append(TangentProject.apply(xn) if project else xn)
        return X

    def cost(self, X, U):                                         # SetQuaternionCost, SURVEY 0.3 / A.5
        J = 0.0
        for k in range(self.N + 1):
            e = X[k] - self.xref[k]
            J = J + 0.5 * (self.Q * e * e).sum() + self.w * (1.0 - torch.abs(torch.dot(self.xref[k, 3:7], X[k][3:7])))
            if k < self.N:
                d = U[k] * self.mask - self.uref
                J = J + 0.5 * (self.Rw * d * d).sum()
        return J

    def value_and_grad(self, U_np, project=True):
        U = torch.tensor(np.asarray(U_np, dtype=float).reshape(self.N, self.nu), requires_grad=True)
        J = self.cost(self.rollout(U, project), U)
        (g,) = torch.autograd.grad(J, U)
        return float(J.detach()), g.numpy() * np.repeat(self.con, 3)[None, :]

    def frame(self):
        return self.R                                             # cone rows act on R f (world frame), QuatMpc.cpp:203

    def kkt(self, U, lam):
        U = np.asarray(U, dtype=float).reshape(self.N, self.nu)
        _, g = self.value_and_grad(U)
        return kkt_residuals(g, U, np.asarray(lam, dtype=float), self.frame(), self.con, self.mu, self.fz_max)


# ------------------------------------------------------------------------------------------------
# ConvexMpc
# ------------------------------------------------------------------------------------------------
class ConvexProblem:
    """One ConvexMpc instance (numpy).  State [rpy, pos, ang_vel_world, lin_vel_world], world-frame forces."""

    def __init__(self, par, rec):
        self.N = int(par.horizon)
        self.nl, self.nu = 4, 12
        self.con = (np.asarray(rec["contacts"], dtype=float) != 0).astype(float)
        self.h = float(np.float32(par.h))
        self.hh = float(np.float32(par.h) / np.float32(2))
        self.mass = float(par.mass)
        self.Id = np.array([par.inertia[0], par.inertia[4], par.inertia[8]], dtype=float)
        self.feet = np.asarray(rec["foot_pos_abs_com"], dtype=float).reshape(4, 3)
        self.Q = np.asarray(par.q_weights, dtype=float)[:12]
        self.Rw = np.asarray(par.r_weights, dtype=float)
        self.mu, self.fz_max = float(par.mu), float(par.fz_max)
        nc = int(self.con.sum())
        self.uref = np.zeros(12)
        self.uref[2::3] = self.mass * GRAV / nc * self.con        # ConvexMpc.cpp:107-110
        self.x0 = np.concatenate([rec["euler"], rec["pos_world"], rec["ang_vel_world"], rec["lin_vel_world"]]).astype(float)
        h_ms = float(par.h_ref) * 1000.0
        self.xref = np.zeros((self.N + 1, 12))
        for k in range(self.N + 1):                               # ConvexMpc.cpp:95-106
            self.xref[k, 2] = rec["euler"][2] + rec["yaw_rate_d"] * h_ms / 1000.0 * k
            self.xref[k, 3:6] = rec["pos_d_world"]
            self.xref[k, 8] = rec["yaw_rate_d"]
            self.xref[k, 9:11] = np.asarray(rec["lin_vel_d_world"])[:2]
        self.mask = np.repeat(self.con, 3)

    @staticmethod
    def _skew(r):
        return np.array([[0, -r[2], r[1]], [r[2], 0, -r[0]], [-r[1], r[0], 0.0]])

    def _AcBc(self, x):
        cy, sy = np.cos(x[2]), np.sin(x[2])
        Ac = np.zeros((12, 12))
        Ac[0:3, 6:9] = [[cy, sy, 0], [-sy, cy, 0], [0, 0, 1]]     # AltroUtils.cpp:256-264
        Ac[3:6, 9:12] = np.eye(3)
        Rz = np.array([[cy, -sy, 0], [sy, cy, 0], [0, 0, 1.0]])
        Iw_inv = np.linalg.inv(Rz @ np.diag(self.Id) @ Rz.T)      # :282-286
        Bc = np.zeros((12, 12))
        for i in range(4):
            if self.con[i] == 0:
                continue
            Bc[6:9, 3 * i:3 * i + 3] = Iw_inv @ self._skew(self.feet[i])
            Bc[9:12, 3 * i:3 * i + 3] = np.eye(3) / self.mass
        return Ac, Bc

    def f(self, x, u):                                            # AltroUtils.cpp:224-293
        Ac, Bc = self._AcBc(x)
        g = np.zeros(12)
        g[11] = -GRAV
        return Ac @ x + Bc @ u + g

    def jac(self, x):                                             # AltroUtils.cpp:295-359 (approximate, as upstream)
        Ac, Bc = self._AcBc(x)
        A = np.zeros((12, 12))
        A[0, 2] = x[7] * np.cos(x[2]) - x[6] * np.sin(x[2])
        A[1, 2] = -x[6] * np.cos(x[2]) - x[7] * np.sin(x[2])
        A[0:6, 6:12] = Ac[0:6, 6:12]
        B = np.zeros((12, 12))
        B[6:12, :] = Bc[6:12, :]
        return A, B

    def step(self, x, u):
        xm = x + self.hh * self.f(x, u)
        return x + self.h * self.f(xm, u)

    def djac(self, x, u):                                         # AltroUtils.cpp:78-110
        xm = x + self.hh * self.f(x, u)
        A, B = self.jac(x)
        Am, Bm = self.jac(xm)
        I = np.eye(12)
        return I + self.h * Am @ (I + self.hh * A), self.h * (Am @ (self.hh * B) + Bm)

    def rollout(self, U):
        X = [self.x0]
        for k in range(self.N):
            X.append(self.step(X[-1], U[k] * self.mask))
        return np.array(X)

    def cost(self, U):
        X = self.rollout(U)
        e = X - self.xref
        d = U * self.mask - self.uref
        return 0.5 * (self.Q * e * e).sum() + 0.5 * (self.Rw * d * d).sum()

    def value_and_grad(self, U_np):
        """Costate recursion with the reference's Jacobian: the 'gradient' its solver drives to zero."""
        U = np.asarray(U_np, dtype=float).reshape(self.N, 12)
        X = self.rollout(U)
        y = self.Q * (X[self.N] - self.xref[self.N])
        g = np.zeros((self.N, 12))
        for k in range(self.N - 1, -1, -1):
            A, B = self.djac(X[k], U[k] * self.mask)
            g[k] = self.Rw * (U[k] * self.mask - self.uref) + B.T @ y
            y = self.Q * (X[k] - self.xref[k]) + A.T @ y
        return self.cost(U), g * self.mask[None, :]

    def exact_grad(self, U_np, eps=1e-6):
        """Central differences of the true objective (what an exact-Jacobian solver would see)."""
        U = np.asarray(U_np, dtype=float).reshape(self.N, 12).copy()
        g = np.zeros_like(U)
        for k in range(self.N):
            for j in np.where(self.mask != 0)[0]:
                U[k, j] += eps
                jp = self.cost(U)
                U[k, j] -= 2 * eps
                jm = self.cost(U)
                U[k, j] += eps
                g[k, j] = (jp - jm) / (2 * eps)
        return g

    def frame(self):
        return np.eye(3)                                          # the pyramid acts on world-frame forces directly

    def kkt(self, U, lam):
        U = np.asarray(U, dtype=float).reshape(self.N, 12)
        _, g = self.value_and_grad(U)
        return kkt_residuals(g, U, np.asarray(lam, dtype=float), self.frame(), self.con, self.mu, self.fz_max)


# ------------------------------------------------------------------------------------------------
# independent SOLVERS for a handful of instances
# ------------------------------------------------------------------------------------------------
def _stance_problem(prob):
    """Variables = stance inputs z [N * nv]; linear rows A z <= b; maps between z and U."""
    N, nu = prob.N, prob.nu
    con = prob.con != 0
    idx = np.where(np.repeat(con, 3))[0]
    nv = idx.size
    n = N * nv
    Ac = cone_matrix(prob.mu) @ prob.frame()
    rows, rhs = [], []
    for k in range(N):
        for li in range(nv // 3):
            for i in range(6):
                r = np.zeros(n)
                r[k * nv + 3 * li:k * nv + 3 * li + 3] = Ac[i]
                rows.append(r)
                rhs.append(prob.fz_max if i == 4 else 0.0)

    def unpack(z):
        U = np.zeros((N, nu))
        U[:, idx] = z.reshape(N, nv)
        return U

    def fun(z):
        J, g = prob.value_and_grad(unpack(z))
        return J, g[:, idx].ravel()

    z0 = np.tile(np.asarray(prob.uref)[idx], N)      # the reference's initial guess U = u_ref (QuatMpc.cpp:253)
    return np.array(rows), np.array(rhs), unpack, fun, z0


def active_set_newton(prob, iters=400, tol=1e-10):
    """Primal active-set Newton method -- a different algorithm CLASS from the interior-point / Riccati scheme of
    the oracle and the kernels: working set W of cone rows held with equality, Newton step of the equality-
    constrained model (dense KKT solve; Hessian = central differences of the gradient this module computes), ratio
    test adds the blocking row, a negative multiplier drops its row.  Returns (U, working set, multipliers, iters).
    Slow (a dense finite-difference Hessian per iteration): fixtures only."""
    A, b, unpack, fun, z = _stance_problem(prob)
    n = z.size

    def grad(v):
        return fun(v)[1]

    def hess(v, eps=1e-4):
        H = np.zeros((n, n))
        for j in range(n):
            e = np.zeros(n)
            e[j] = eps
            H[:, j] = (grad(v + e) - grad(v - e)) / (2 * eps)
        return 0.5 * (H + H.T)

    W: list[int] = []
    lam = np.zeros(0)
    H, fresh = None, False
    for it in range(iters):
        g = grad(z)
        # the model Hessian only shapes the steps (the gradient is exact, so a zero step certifies the point whatever
        # H is): refresh it every few working-set changes, and always once more before a zero step is accepted
        if H is None or it % 6 == 0:
            H, fresh = hess(z), True
        AW = A[W] if W else np.zeros((0, n))
        m = AW.shape[0]
        kkt = np.block([[H, AW.T], [AW, np.zeros((m, m))]])
        sol = np.linalg.lstsq(kkt, np.concatenate([-g, np.zeros(m)]), rcond=1e-14)[0]
        d, lam = sol[:n], sol[n:]
        if np.abs(d).max() < tol:
            if not fresh:
                H, fresh = hess(z), True
                continue
            if m == 0 or lam.min() >= -1e-9:
                return unpack(z), list(W), lam, it
            W.pop(int(np.argmin(lam)))
            continue
        Ad, sl = A @ d, b - A @ z
        alpha, blk = 1.0, -1
        for i in range(A.shape[0]):
            if i not in W and Ad[i] > 1e-14:
                a = sl[i] / Ad[i]
                if a < alpha:
                    alpha, blk = max(a, 0.0), i
        z = z + alpha * d
        fresh = False
        if blk >= 0:
            W.append(blk)
        elif np.abs(d).max() < 1e-3:
            H, fresh = hess(z), True       # full, small step: Newton's end game wants the current Hessian
    return unpack(z), list(W), lam, iters


def scipy_solve(prob, maxiter=400):
    """scipy.optimize SLSQP (tight ftol) from U = u_ref.  A quasi-Newton SQP cannot resolve the directions whose
    curvature is R = 1e-6 (cond ~ 1e6): it reaches the oracle's COST to ~1e-6 relative and never undercuts it, but
    its forces stay 1e-3 ... 1e-1 N away -- which is why active_set_newton (exact Newton steps) exists."""
    from scipy.optimize import minimize

    A, b, unpack, fun, z0 = _stance_problem(prob)
    cons = {"type": "ineq", "fun": lambda z: b - A @ z, "jac": lambda z: -A}
    res = minimize(fun, z0, jac=True, method="SLSQP", constraints=[cons],
                   options={"ftol": 1e-16, "maxiter": maxiter, "disp": False})
    return unpack(res.x), res
