"""Shared parity rule between two DO-phase results (oracle vs HIP, oracle vs lane-serial build).

Bar: |dx| <= 1e-4 on x, y, yaw, steer, v, w of every agent and timestep, identical SQP iteration counts
and status codes.  One documented exception: the corridor boxes grow in 0.1 m steps behind a strict validity test
(sqp/corridor.cc:284-315), so a 1e-9 difference in a disc centre can flip one growth step; that changes the agent's
next QP and moves its trajectory by O(1e-3).  An agent whose final boxes differ by a growth step is therefore held to
the looser CORRIDOR_FLIP_TOL and must still have the same iteration counts/status.

A second, milder effect: the QP objective only prices (v_{k+1}-v_k)^2 and w^2 (P is singular on 4Nt of 6Nt-2 variables,
sqp/dsqp_solver.cc:163-197) and ADMM stops at eps = 1e-3, so the returned point of a QP is a sensitive function of its
data; the oracle factors the full KKT matrix (as OSQP does) while the device program factors the reduced block-tridiagonal
system, which differ by ~1e-9 per QP, and a chain of 5-10 such QPs can amplify that to a few 1e-4 for a handful of
agents.  Tests therefore allow a stated number of agents between TOL and LOOSE_TOL and none above it; the HIP path is
additionally compared with the lane-serial build of the same program, where the agreement is ~1e-9.
"""
import numpy as np

TOL = 1e-4
LOOSE_TOL = 1e-3          # see below: sensitive agents
CORRIDOR_FLIP_TOL = 2e-2


def compare(ref, got, tol=TOL):
    """Returns a dict with per-agent maxima and the list of agents that violated the rule."""
    d_sol = np.abs(ref.solutions - got.solutions).max(axis=(1, 2))
    d_cor = np.abs(ref.corridors - got.corridors).max(axis=(1, 2))
    flipped = d_cor > 0.05
    bad = []
    for a in range(len(d_sol)):
        lim = CORRIDOR_FLIP_TOL if flipped[a] else tol
        if not (d_sol[a] <= lim):
            bad.append((a, float(d_sol[a]), float(d_cor[a])))
    counts_equal = (np.array_equal(ref.sqp_iters, got.sqp_iters) and np.array_equal(ref.last_status, got.last_status)
                    and np.array_equal(ref.admm_iters, got.admm_iters))
    return dict(max_sol=float(d_sol.max()), max_cor=float(d_cor.max()), n_flipped=int(flipped.sum()), bad=bad,
                counts_equal=counts_equal, d_sol=d_sol, d_cor=d_cor)
