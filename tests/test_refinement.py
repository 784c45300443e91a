"""The exact-arithmetic arbiter and the refined solve (round 6), on the CPU.

ARBITER: oracle/libcsdo_oracle_q.so - the oracle with OSQP's linear algebra (Ruiz scaling, LDL', the ADMM updates, the residuals and every
test on them) in IEEE binary128; QP assembly, safe boxes and the SQP loop stay in double (the reference defines them so) and x* is rounded
to double once per QP.  It is the iterate path the reference algorithm would take in exact arithmetic on the same double-precision QP
data, and the yardstick for "who is closer to the reference OSQP path": the product, or a double-precision OSQP (the oracle)?

Finding (scripts/chain_parity.py, tests/golden/chain_outliers_*.json `arbiter`): the product as shipped is 1.3 - 1.5 x as often beyond 1e-4
of that path as the oracle - its reduced-system solve loses cond(H) eps, about fifty times what OSQP's LDL' of the KKT matrix loses
(scripts/solve_accuracy.py).  With csdo_qp_parm::solve_refinement = 1 (one refinement step on the KKT residual per solve) it is CLOSER
than the oracle on every workload.  This module holds both statements: recomputed on a few worlds, and on the committed totals."""
import json
import os

import numpy as np
import pytest

from tests import parity

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
THREADS = min(os.cpu_count() or 8, 16)


def _d(a, b):
    return np.concatenate([np.abs(g.solutions - r.solutions).max(axis=(1, 2)) for g, r in zip(a, b)])


def _same_counts(a, b):
    return all(np.array_equal(g.admm_iters, r.admm_iters) and np.array_equal(g.sqp_iters, r.sqp_iters) and
               np.array_equal(g.last_status, r.last_status) for g, r in zip(a, b))


@pytest.fixture(scope="module")
def worlds():
    from csdotrajectoryplanning_amd import workloads
    return [workloads.build_job(j)[0] for j in workloads.workload_jobs("map100", 2)]


def test_the_arbiter_is_resolved(oracle, worlds):
    """binary128 (113 bits) against x87 extended (64 bits) in the solve: two arbiters 49 bits apart agree to 1e-9 over two QPs - the
    exact path is resolved far below the 1e-9 .. 1e-8 at which the double-precision builds sit from it - and neither changes a count."""
    ws = [w.with_parm(max_iter=2) for w in worlds]
    q, ld, o = oracle.solve_batch_variant(ws, "q", THREADS), oracle.solve_batch_variant(ws, "ld", THREADS), oracle.solve_batch(ws, THREADS)
    assert _same_counts(q, ld) and _same_counts(q, o)
    assert _d(q, ld).max() <= 1e-9, _d(q, ld).max()
    assert 1e-12 < np.median(_d(o, q)) < 1e-7          # a double-precision OSQP is measurably off the exact path


@pytest.mark.parametrize("n_qp", [1, 3])
def test_refined_solve_is_closer_to_the_exact_path_than_a_double_precision_osqp(emu, oracle, worlds, n_qp):
    ws = [w.with_parm(max_iter=n_qp) for w in worlds]
    wr = [w.with_parm(max_iter=n_qp, solve_refinement=1) for w in worlds]
    q, o = oracle.solve_batch_variant(ws, "q", THREADS), oracle.solve_batch(ws, THREADS)
    plain, refined = emu.solve_batch(ws, 0, THREADS), emu.solve_batch(wr, 0, THREADS)
    lagged = emu.solve_batch([w.with_parm(max_iter=n_qp, solve_refinement=2) for w in worlds], 0, THREADS)
    assert _same_counts(plain, q) and _same_counts(refined, q) and _same_counts(o, q) and _same_counts(lagged, q)
    d_o, d_p, d_r, d_l = _d(o, q), _d(plain, q), _d(refined, q), _d(lagged, q)
    print("QPs %d: median / p90 distance to the arbiter: oracle %.1e / %.1e, product %.1e / %.1e, refined %.1e / %.1e, lagged %.1e / %.1e" % (
        n_qp, np.median(d_o), np.quantile(d_o, .9), np.median(d_p), np.quantile(d_p, .9), np.median(d_r), np.quantile(d_r, .9),
        np.median(d_l), np.quantile(d_l, .9)))
    assert np.median(d_r) <= np.median(d_o) and np.quantile(d_r, .9) <= np.quantile(d_o, .9)      # refined: closer than OSQP in double
    assert np.median(d_p) >= 2.0 * np.median(d_r)                                               # and that is the refinement's doing
    assert d_r.max() <= 1e-6 and d_p.max() <= 1e-5
    # the lagged form (one solve per iteration): about the distance of a double-precision OSQP, well inside the unrefined product's
    assert np.median(d_l) <= 1.5 * np.median(d_o) and np.median(d_l) <= 0.6 * np.median(d_p) and d_l.max() <= 1e-6
    # the other residency modes run the same refinement: the pair-split modes to the bit, the one-lane form to rounding
    for mode in (1, 2):
        other = emu.solve_batch(wr, mode, THREADS)
        assert all(np.array_equal(a.solutions, b.solutions) for a, b in zip(other, refined)), mode
    lane = emu.solve_batch(wr, 3, THREADS)
    assert _same_counts(lane, q) and np.median(_d(lane, q)) <= np.median(d_o)


@pytest.mark.parametrize("workload", ["map100", "map50", "synth1024", "room50", "agents100"])
def test_committed_totals_against_the_arbiter(workload):
    """Whole workloads (scripts/chain_parity.py; the arbiter alone takes 6 - 7 minutes per workload on 8 cores): agents beyond 1e-4 of the
    exact path or with other counts.  The bar VERDICT r5 set - the product no further from the exact OSQP path than a double-precision
    OSQP - holds with the refinement on, with room to spare; the shipped default pays for its cheaper solve with at most 1.5 x the
    oracle's count (+ 2), which is what the fitted `1.6 x + 3` of round 5 had been standing in for."""
    with open(os.path.join(GOLDEN, "chain_outliers_%s.json" % workload)) as f:
        fx = json.load(f)
    n = fx["arbiter"]["beyond_1e-4_or_other_counts"]
    qd = fx["arbiter"]["quantiles_of_d"]
    assert n["product_refined"] <= n["oracle"], n
    assert qd["product_refined"]["median"] <= qd["oracle"]["median"] and qd["product_refined"]["p90"] <= qd["oracle"]["p90"], qd
    # the lagged refinement (solve_refinement = 2, one solve per iteration): no further than the oracle either - by a smaller margin
    assert n["product_lagged"] <= n["oracle"] and qd["product_lagged"]["median"] <= qd["oracle"]["median"], (n, qd)
    assert n["product"] <= 1.5 * n["oracle"] + 2, n
    assert qd["product"]["median"] <= 4.0 * qd["oracle"]["median"], qd
    assert parity.TOL == 1e-4


@pytest.mark.parametrize("L,Na", [(1, 1), (2, 2), (3, 1), (5, 3), (21, 2), (64, 2)])
def test_refined_solve_on_edge_horizons(emu, oracle, veh_parm, L, Na):
    """Nt = 4 (no reduction level at all), 7, 10, 16, 64 (a wave boundary), 193 (the eight-node tail), one to three vehicles, planes
    between them: the refined program stops where the oracle stops and stays within 1e-6 of it, pair-split and one-lane form."""
    from tests import helpers
    veh, parm = veh_parm
    w = helpers.straight_line_world(veh, parm, Na=Na, L=L, dim=300.0, spacing=3.5)
    for k in (1, 3):
        ref = oracle.solve(w.with_parm(max_iter=k), 2)
        for mode in (0, 3):
            for refinement in (1, 2):
                got = emu.solve(w.with_parm(max_iter=k, solve_refinement=refinement), mode)
                assert np.isfinite(got.solutions).all() and np.isfinite(got.corridors).all()
                assert np.array_equal(got.admm_iters, ref.admm_iters) and np.array_equal(got.last_status, ref.last_status), (k, mode, refinement)
                assert np.abs(got.solutions - ref.solutions).max() <= 1e-6, (k, mode, refinement)
