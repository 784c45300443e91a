"""Independent check of the oracle's ADMM ITERATE PATH (not only of its optimum): tests/admm_numpy.py - numpy / scipy, SuperLU on
the KKT matrix, written from SURVEY.md Appendix B without looking at oracle/osqp_restate.cc - must stop at the same iteration
with the same status and see the same rho and residuals at every termination check as the oracle on assembled agent QPs.
The reference's OSQP 0.6.3 itself is not available here: this pins the restatement against a second reading of the
published algorithm, not against the library."""
import numpy as np
import pytest

from tests import admm_numpy, helpers


def _first_qp(oracle, world, a):
    """The QP calcIndividualSQP (dsqp_solver.cc:99-223) hands to OSQP in its first iteration for agent a."""
    veh, parm, Nt = world.veh, world.parm, world.Nt
    g = world.x0_bar[a]
    sol0 = np.concatenate([g[:, 0], g[:, 1], g[:, 2], g[:, 3], g[:-1, 4], g[:-1, 5]])
    f32 = lambda v: v.astype(np.float32).astype(np.float64)          # State's disc centres are float members
    pts = np.concatenate([np.stack([f32(g[:, 0] + veh.f2x * np.cos(g[:, 2])), f32(g[:, 1] + veh.f2x * np.sin(g[:, 2]))], 1),
                          np.stack([f32(g[:, 0] + veh.r2x * np.cos(g[:, 2])), f32(g[:, 1] + veh.r2x * np.sin(g[:, 2]))], 1)])
    boxes, _ = oracle.generate_boxes(pts, world.obstacles, world.dimx, world.dimy, veh)
    bf, br = boxes[:Nt], boxes[Nt:]
    lb = np.concatenate([bf[:, 0], bf[:, 1], br[:, 0], br[:, 1]])
    ub = np.concatenate([bf[:, 2], bf[:, 3], br[:, 2], br[:, 3]])
    cfg = np.array([g[0, 0], g[-1, 0], g[0, 1], g[-1, 1], g[0, 2], g[-1, 2]])
    pl = world.planes[world.plane_off[a]:world.plane_off[a + 1]]
    P, A, l, u = oracle.assemble_qp(Nt, sol0, lb, ub, g[:, 0], g[:, 1], cfg, pl, veh, parm)
    return P, A, l, u, sol0


CASES = [("map50_agents0to5.npz", 1), ("map50_agents15to17.npz", 0), ("map100_agents0to3.npz", 2), ("map100_agents0to3.npz", 0)]


@pytest.mark.parametrize("name,agent", CASES)
def test_iterate_path_matches_an_independent_admm(oracle, veh_parm, name, agent):
    veh, parm = veh_parm
    world, _ = helpers.load_golden(name, veh, parm)
    P, A, l, u, sol0 = _first_qp(oracle, world, agent)
    q = np.zeros(P.shape[0])
    x_o, y_o, info, hist = oracle.osqp_hist(P, q, A, l, u, sol0, max_iter=int(parm.osqp_max_iter), adaptive_rho_interval=25)
    r = admm_numpy.solve(P, q, A, l, u, sol0, max_iter=int(parm.osqp_max_iter), interval=25)
    assert (r["iter"], r["status"]) == (info["iter"], info["status"]), (r["iter"], r["status"], info)
    assert len(r["rho_hist"]) == len(hist)
    # rho at every check: the adaptation's decisions (update or not) and values agree; residuals to solver accuracy
    np.testing.assert_allclose(r["rho_hist"], hist[:, 0], rtol=1e-6)
    np.testing.assert_allclose(r["pri_hist"], hist[:, 1], rtol=1e-4, atol=1e-9)
    np.testing.assert_allclose(r["dua_hist"], hist[:, 2], rtol=1e-4, atol=1e-9)
    assert np.abs(r["x"] - x_o).max() < 1e-6
    assert len(np.unique(np.round(np.log10(hist[:, 0]), 6))) >= 1


def test_the_cases_exercise_rho_updates_and_the_iteration_cap(oracle, veh_parm):
    """The comparison above is only worth something if the paths are not trivial: at least one case adapts rho (a
    refactorisation) and the cases stop at different checks."""
    veh, parm = veh_parm
    iters, updates = set(), 0
    for name, agent in CASES:
        world, _ = helpers.load_golden(name, veh, parm)
        P, A, l, u, sol0 = _first_qp(oracle, world, agent)
        _, _, info, hist = oracle.osqp_hist(P, np.zeros(P.shape[0]), A, l, u, sol0, max_iter=int(parm.osqp_max_iter))
        iters.add(info["iter"])
        updates += info["rho_updates"]
        assert info["iter"] % 25 == 0
    assert updates >= 2 and len(iters) >= 2


def test_restricted_iteration_cap_takes_the_inaccurate_branch(oracle, veh_parm):
    """max_iter = 50: both must leave through the approximate test at the cap (status 2 or -2) after the same 50 iterations."""
    veh, parm = veh_parm
    world, _ = helpers.load_golden("map100_agents0to3.npz", veh, parm)
    P, A, l, u, sol0 = _first_qp(oracle, world, 1)
    q = np.zeros(P.shape[0])
    x_o, _, info, hist = oracle.osqp_hist(P, q, A, l, u, sol0, max_iter=50)
    r = admm_numpy.solve(P, q, A, l, u, sol0, max_iter=50)
    assert (r["iter"], r["status"]) == (info["iter"], info["status"])
    np.testing.assert_allclose(r["rho_hist"], hist[:, 0], rtol=1e-6)
    if info["status"] in (1, 2, -2):
        assert np.abs(r["x"] - x_o).max() < 1e-6
