"""An ADMM written independently of oracle/osqp_restate.cc, straight from the published OSQP 0.6.x algorithm as SURVEY.md
Appendix B records it: scipy sparse matrices, the quasi-definite KKT system solved by SuperLU (scipy.sparse.linalg.splu,
partial pivoting - not an LDL', not the oracle's ordering), numpy vectors.  It shares no code with the oracle; what the two
must share is the iterate PATH: iteration count at eps 1e-3, status, and the rho the adaptation picks at every check.
TEST INFRASTRUCTURE (tests/test_oracle_admm_path.py)."""
import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla

INFTY, RHO_MIN, RHO_MAX, RHO_TOL, RHO_EQ = 1e30, 1e-6, 1e6, 1e-4, 1e3
MIN_SCALING, MAX_SCALING = 1e-4, 1e4


def _limit(v):
    v = np.where(v < MIN_SCALING, 1.0, v)
    return np.where(v > MAX_SCALING, MAX_SCALING, v)


def _inf_norm_cols(M):
    M = sp.csc_matrix(M)
    out = np.zeros(M.shape[1])
    if M.nnz:
        a = abs(M)
        out = np.asarray(a.max(axis=0).todense()).ravel()
    return out


def _ninf(v):
    v = np.abs(v[~np.isnan(v)])
    return float(v.max()) if v.size else 0.0


def solve(P_triu, q, A, l, u, x_warm, max_iter=400, rho=0.1, sigma=1e-6, alpha=1.6, eps_abs=1e-3, eps_rel=1e-3,
          eps_pinf=1e-4, scaling=10, check=25, interval=25, tolerance=5.0):
    """Returns dict(x, y, status, iter, rho_hist, pri_hist, dua_hist): histories hold one entry per termination check."""
    Pu = sp.csc_matrix(P_triu, dtype=float)
    P = sp.csc_matrix(Pu + sp.triu(Pu, 1).T)
    A = sp.csc_matrix(A, dtype=float)
    q = np.array(q, float)
    l, u = np.array(l, float), np.array(u, float)
    n, m = P.shape[0], A.shape[0]
    # 1a. Ruiz equilibration + cost scaling
    D, E, c = np.ones(n), np.ones(m), 1.0
    for _ in range(scaling):
        dD = 1.0 / np.sqrt(_limit(np.maximum(_inf_norm_cols(P), _inf_norm_cols(A))))
        dE = 1.0 / np.sqrt(_limit(_inf_norm_cols(A.T)))
        P = sp.csc_matrix(sp.diags(dD) @ P @ sp.diags(dD))
        A = sp.csc_matrix(sp.diags(dE) @ A @ sp.diags(dD))
        q = dD * q
        D, E = dD * D, dE * E
        gamma = max(float(np.mean(_inf_norm_cols(P))), float(_limit(np.array([_ninf(q)]))[0]))
        gamma = 1.0 / float(_limit(np.array([gamma]))[0])
        P, q, c = P * gamma, q * gamma, c * gamma
    l, u = E * l, E * u
    # 1b. row classes
    loose = (l < -INFTY * MIN_SCALING) & (u > INFTY * MIN_SCALING)
    eq = ~loose & (u - l < RHO_TOL)

    def rho_vector(r):
        return np.where(loose, RHO_MIN, np.where(eq, RHO_EQ * r, r))

    def factor(rv):
        K = sp.bmat([[P + sigma * sp.identity(n), A.T], [A, -sp.diags(1.0 / rv)]], format="csc")
        return spla.splu(K)

    rv = rho_vector(rho)
    lu = factor(rv)
    # 2. warm start
    x = np.array(x_warm, float) / D
    z = A @ x
    y = np.zeros(m)
    status, it = -10, 0
    hist = dict(rho=[], pri=[], dua=[])
    At = sp.csc_matrix(A.T)

    def residuals():
        Ax, Px, Aty = A @ x, P @ x, At @ y
        r_p, r_d = Ax - z, Px + q + Aty
        unscaled = dict(pri=_ninf(r_p / E), ax=_ninf(Ax / E), z=_ninf(z / E), dua=_ninf(r_d / D) / c, px=_ninf(Px / D),
                        aty=_ninf(Aty / D), q=_ninf(q / D))
        scaled = dict(pri=_ninf(r_p), ax=_ninf(Ax), z=_ninf(z), dua=_ninf(r_d), px=_ninf(Px), aty=_ninf(Aty), q=_ninf(q))
        return unscaled, scaled

    def primal_infeasible(dy, eps):
        dy = dy.copy()
        up_inf, lo_inf = u > INFTY * MIN_SCALING, l < -INFTY * MIN_SCALING
        dy[up_inf & lo_inf] = 0.0
        only_up = up_inf & ~lo_inf
        dy[only_up] = np.minimum(dy[only_up], 0.0)
        only_lo = lo_inf & ~up_inf
        dy[only_lo] = np.maximum(dy[only_lo], 0.0)
        nrm = _ninf(E * dy)
        if not nrm > eps:
            return False
        with np.errstate(invalid="ignore"):
            s = float(np.sum(u * np.maximum(dy, 0.0) + l * np.minimum(dy, 0.0)))   # (-inf) * 0 = NaN as in C
        if not s < -eps * nrm:
            return False
        return _ninf((At @ dy) / D) < eps * nrm

    def terminated(un, dy, approx):
        ea, er, ep = (10 * eps_abs, 10 * eps_rel, 10 * eps_pinf) if approx else (eps_abs, eps_rel, eps_pinf)
        if un["pri"] > INFTY or un["dua"] > INFTY:
            return -7
        prim_ok = un["pri"] < ea + er * max(un["ax"], un["z"])
        prim_inf = False if prim_ok else primal_infeasible(dy, ep)
        dual_ok = un["dua"] < ea + er * max(un["px"], un["aty"], un["q"]) / c
        if prim_ok and dual_ok:
            return 2 if approx else 1
        if prim_inf:
            return 3 if approx else -3
        return None          # (dual infeasibility needs q'dx < 0; not reachable with q = 0, and tested after these)

    dy = np.zeros(m)
    checked = False
    while it < max_iter:
        it += 1
        xp, zp = x, z
        sol = lu.solve(np.concatenate([sigma * xp - q, zp - y / rv]))
        xt = sol[:n]
        zt = zp + (sol[n:] - y) / rv
        x = alpha * xt + (1 - alpha) * xp
        zr = alpha * zt + (1 - alpha) * zp
        z = np.minimum(np.maximum(zr + y / rv, l), u)
        dy = rv * (zr - z)
        y = y + dy
        checked = check and it % check == 0
        un = sc = None
        if checked:
            un, sc = residuals()
            hist["rho"].append(rho), hist["pri"].append(un["pri"]), hist["dua"].append(un["dua"])
            st = terminated(un, dy, False)
            if st is not None:
                status = st
                break
        if interval and it % interval == 0 and it < max_iter:
            if sc is None:
                un, sc = residuals()
            pri = sc["pri"] / (max(sc["ax"], sc["z"]) + 1e-10)
            dua = sc["dua"] / (max(sc["px"], sc["aty"], sc["q"]) + 1e-10)
            est = min(max(rho * np.sqrt(pri / (dua + 1e-10)), RHO_MIN), RHO_MAX)
            if est > rho * tolerance or est < rho / tolerance:
                rho = est
                rv = rho_vector(rho)
                lu = factor(rv)
    if status == -10:
        un, _ = residuals()
        if not checked:
            st = terminated(un, dy, False)
            status = st if st is not None else status
        if status == -10:
            st = terminated(un, dy, True)
            status = st if st is not None else -2
    return dict(x=D * x, y=(E * y) / c, status=int(status), iter=it, rho_hist=np.array(hist["rho"]),
                pri_hist=np.array(hist["pri"]), dua_hist=np.array(hist["dua"]))
