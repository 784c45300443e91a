"""Parity at BASELINE.json's full sizes, every agent, through the C ABI (libcsdo_hip.so) against the oracle.

What is pinned per agent (not per cent):
  * ONE QP (QpParm.max_iter = 1: one SQP iteration from x0_bar, identical corridor boxes by construction): identical
    ADMM iteration count and OSQP status for EVERY agent of the set and |d| <= FIRST_QP_TOL on every state / control.
    This is the statement about the kernel (assembly, Ruiz, factor, ADMM, termination, adaptive rho).
  * TWO QPs (max_iter = 2: one corridor refresh in between): identical counts for every agent; agents whose refreshed boxes
    flipped a 0.1 m growth step (sqp/corridor.cc:284-315 is discontinuous) are LISTED and held to CORRIDOR_FLIP_TOL, every
    other agent to SECOND_QP_TOL.
  * the full chain (max_iter = 10): an ADMM stopped at eps = 1e-3 on a QP whose Hessian is singular in 4Nt of its 6Nt-2
    variables, re-linearised up to ten times, amplifies a last-bit difference by a factor 3-9 per SQP iteration (measured, DESIGN
    section 4): the oracle differs from ITSELF under a change of rounding by the same amounts.  So the chain is pinned (a) to
    the BIT against the lane-serial host build of the program's own source (round 5: one shared trigonometry) on every agent of
    all five workloads, (b) against the oracle by the EXACT committed list of agents beyond 1e-4 (tests/golden/chain_outliers_*.json,
    computed on the CPU; tests/test_chain_cpu.py recomputes it and runs the growth law on it) and (c) by an implementation-independent
    acceptance of the final trajectories (feasibility residuals, objective, obstacle validator: both solvers must agree).
"""
import os

import numpy as np
import pytest

from tests import parity

pytestmark = pytest.mark.gpu

FIRST_QP_TOL = 1e-5        # hard bar on every agent; the bulk is < 1e-8 (asserted below)
SECOND_QP_TOL = 1e-3
THREADS = os.cpu_count() or 8
_CACHE = {}


def _with_max_iter(world, k):
    from csdotrajectoryplanning_amd.abi import QpParm
    from csdotrajectoryplanning_amd.problem import World
    p = QpParm.from_buffer_copy(bytes(world.parm))
    p.max_iter = float(k)
    return World(world.x0_bar, world.plane_off, world.planes, world.dimx, world.dimy, world.obstacles, world.veh, p)


def _set(name, front="auto"):
    """All worlds of a BASELINE workload, built once per session by a spawn pool (the GPU is already initialised)."""
    key = (name, front)
    if key not in _CACHE:
        from csdotrajectoryplanning_amd import workloads
        _CACHE[key] = [w for w, _ in workloads.build_jobs_parallel(workloads.workload_jobs(name, front=front),
                                                                   min(THREADS, 32))]
    return _CACHE[key]


def _per_agent(got, ref):
    d = np.concatenate([np.abs(g.solutions - r.solutions).max(axis=(1, 2)) for g, r in zip(got, ref)])
    dc = np.concatenate([np.abs(g.corridors - r.corridors).max(axis=(1, 2)) for g, r in zip(got, ref)])
    same = np.concatenate([(g.sqp_iters == r.sqp_iters) & (g.admm_iters == r.admm_iters) &
                           (g.last_status == r.last_status) for g, r in zip(got, ref)])
    return d, dc, same


@pytest.mark.parametrize("front", ["auto", "stand-in"])
@pytest.mark.parametrize("workload", ["map100", "map50"])
def test_first_qp_every_agent(gpu_handle, oracle, workload, front):
    """front = auto: the measured workloads (front-end paths); stand-in: the round-1 worlds (longer horizons, up to 800 planes
    per agent, initial guesses that violate their planes): other residency decisions, harder QPs."""
    worlds = [_with_max_iter(w, 1) for w in _set(workload, front)]
    got = gpu_handle.solve_batch(worlds)
    ref = oracle.solve_batch(worlds, THREADS)
    d, dc, same = _per_agent(got, ref)
    assert len(d) == {"map100": 3000, "map50": 1500}[workload]
    assert same.all(), np.nonzero(~same)[0]
    assert d.max() <= FIRST_QP_TOL, (float(d.max()), int((d > FIRST_QP_TOL).sum()))
    assert np.median(d) < 1e-8 and np.mean(d <= 1e-6) >= 0.995, (float(np.median(d)), float(np.mean(d <= 1e-6)))
    for g, r in zip(got, ref):
        assert g.solver_status == r.solver_status and g.initial_static_legal == r.initial_static_legal


@pytest.mark.parametrize("workload", ["room50", "agents100"])
def test_first_qp_in_the_regimes_the_baseline_sets_do_not_reach(gpu_handle, oracle, workload):
    """room50: benchmark/room/agents50 (238 wall obstacles per world: the box phase at five times the obstacle count, most of
    the obstacle list culled per point); agents100: benchmark/map100by100/agents100/obstacle (100 vehicles: about twice the
    separating planes per agent, so agents whose plane state does not fit LDS beside the rest - the workspace path of the
    plane pass - are common).  One QP per agent: identical counts and status, |d| <= 1e-5, on every agent; the launch groups
    must show that the slow residency paths actually ran."""
    worlds = [_with_max_iter(w, 1) for w in _set(workload)]
    n_agents = {"room50": 600, "agents100": 1200}[workload]
    assert sum(w.Na for w in worlds) == n_agents
    if workload == "room50":
        n_obs = [w.obstacles.shape[0] for w in worlds]        # walls of 138 .. 238 discs (the BASELINE sets: 25 and 50)
        assert min(n_obs) >= 130 and max(n_obs) == 238, n_obs
    got = gpu_handle.solve_batch(worlds)
    ref = oracle.solve_batch(worlds, THREADS)
    d, dc, same = _per_agent(got, ref)
    assert same.all(), np.nonzero(~same)[0]
    assert d.max() <= FIRST_QP_TOL and np.median(d) < 1e-8, (float(d.max()), float(np.median(d)))
    # (the boxes returned are the ones refreshed at the QP's solution, corridor.cc via dsqp_solver.cc:251-253: they follow it)
    assert max(float(np.abs(g.corridors - r.corridors).max()) for g, r in zip(got, ref)) <= 1e-6
    for g, r in zip(got, ref):
        assert g.initial_static_legal == r.initial_static_legal
    if workload == "agents100":
        K = np.concatenate([w.plane_off[1:] - w.plane_off[:-1] for w in worlds])
        assert K.mean() > 150 and K.max() > 400, (K.mean(), K.max())


@pytest.mark.parametrize("workload", ["map100", "map50"])
def test_second_qp_with_flips_listed(gpu_handle, oracle, workload):
    worlds = [_with_max_iter(w, 2) for w in _set(workload)]
    got = gpu_handle.solve_batch(worlds)
    ref = oracle.solve_batch(worlds, THREADS)
    d, dc, same = _per_agent(got, ref)
    flipped = dc > 0.05                                   # a box edge moved by a 0.1 m growth step
    listed = [(int(a), float(d[a]), float(dc[a])) for a in np.nonzero(flipped | ~same)[0]]
    print("agents with a flipped growth step or different counts after one corridor refresh:", listed)
    assert len(listed) <= 0.005 * len(d), listed
    assert same[~flipped].all(), listed
    assert d[~flipped].max() <= SECOND_QP_TOL, float(d[~flipped].max())
    assert d[flipped].max(initial=0.0) <= 0.2, listed
    assert np.mean(d <= 1e-6) >= 0.98, float(np.mean(d <= 1e-6))


# The full chain (round 5).  The HIP build returns the BITS of the lane-serial host build of its own source (one shared sin / cos /
# tan / atan2, csrc/csdo_math.h): asserted below on every agent of all five workloads.  Everything about "the product against the
# oracle" is therefore computed on the CPU - scripts/chain_parity.py writes tests/golden/chain_outliers_<workload>.json (every agent
# beyond 1e-4 of the oracle or with other counts, with the oracle's own sensitivity on it: its FMA build and its build with the
# product's trigonometry) and profiles/r05_chain_*.json; tests/test_chain_cpu.py recomputes the lists and the growth law without a
# GPU - and the GPU tests only have to show (a) the bits and (b) that the oracle comparison of a GPU run reproduces the committed
# list EXACTLY: no fitted bars.
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CHAIN_WORKLOADS = ["map100", "map50", "room50", "agents100", "synth1024"]


def _fixture(workload):
    import json
    with open(os.path.join(GOLDEN, "chain_outliers_%s.json" % workload)) as f:
        return json.load(f)


def _world_agent(worlds, flat_index):
    first = np.cumsum([0] + [w.Na for w in worlds])
    wi = int(np.searchsorted(first, flat_index, side="right") - 1)
    return wi, int(flat_index - first[wi])


def _bits_equal(a, b):
    return (np.array_equal(a.solutions, b.solutions) and np.array_equal(a.corridors, b.corridors) and
            np.array_equal(a.sqp_iters, b.sqp_iters) and np.array_equal(a.admm_iters, b.admm_iters) and
            np.array_equal(a.last_status, b.last_status) and a.solver_status == b.solver_status and
            a.initial_static_legal == b.initial_static_legal)


@pytest.mark.parametrize("workload", CHAIN_WORKLOADS)
def test_full_chain_hip_build_is_bit_identical_to_its_lane_serial_build(gpu_handle, emu, workload):
    """Same program source as HIP device code and lane-serially on the host: solutions, corridors, SQP / ADMM counts and statuses of
    every agent of the set, np.array_equal."""
    worlds = _set(workload)
    got = gpu_handle.solve_batch(worlds)
    ref = emu.solve_batch(worlds, 0, THREADS)
    bad = [k for k, (g, r) in enumerate(zip(got, ref)) if not _bits_equal(g, r)]
    assert not bad, ("worlds whose results differ in some bit between HIP and the lane-serial build", bad)
    assert sum(w.Na for w in worlds) == {"map100": 3000, "map50": 1500, "room50": 600, "agents100": 1200, "synth1024": 1024}[workload]


@pytest.mark.parametrize("workload", ["map100", "room50", "agents100"])
def test_cut_chains_of_the_outliers_are_bit_identical_too(gpu_handle, emu, workload):
    """Every committed outlier alone with the chain cut after k = 1 .. 10 QPs (what the growth-law test of tests/test_chain_cpu.py
    runs on the lane-serial build): the HIP build returns the same bits at every cut."""
    fx = _fixture(workload)
    worlds = _set(workload)
    singles = [worlds[o["world"]].subset(o["agent"], o["agent"] + 1) for o in fx["outliers"][:40]]
    assert singles
    for k in range(1, 11):
        ws = [_with_max_iter(w, k) for w in singles]
        hip, em = gpu_handle.solve_batch(ws), emu.solve_batch(ws, 0, THREADS)
        bad = [j for j, (g, r) in enumerate(zip(hip, em)) if not _bits_equal(g, r)]
        assert not bad, (k, [(fx["outliers"][j]["world"], fx["outliers"][j]["agent"]) for j in bad])


@pytest.mark.parametrize("workload", ["map100", "map50", "room50", "agents100"])
def test_full_chain_against_oracle_with_acceptance(gpu_handle, oracle, workload):
    from csdotrajectoryplanning_amd import results
    worlds = _set(workload)
    got = gpu_handle.solve_batch(worlds)
    ref = oracle.solve_batch(worlds, THREADS)
    d, dc, same = _per_agent(got, ref)
    # the agents beyond north_star's 1e-4 (or with other counts) are EXACTLY the committed list, with the committed distances
    fx = _fixture(workload)
    listed = {(o["world"], o["agent"]): o for o in fx["outliers"]}
    found = {_world_agent(worlds, g): float(d[g]) for g in np.nonzero(~same | (d > parity.TOL))[0]}
    print("HIP vs oracle", workload, dict(agents=len(d), same_counts=int(same.sum()), beyond_1e4=int((d > parity.TOL).sum()),
                                          median=float(np.median(d)), max=float(d.max())), "oracle vs its own FMA build:",
          fx["outlier_counts"]["oracle_fma_vs_oracle"], "vs its build with the product's trigonometry:", fx["outlier_counts"]["oracle_xm_vs_oracle"])
    assert set(found) == set(listed), (sorted(set(found) - set(listed)), sorted(set(listed) - set(found)))
    for wa, dv in found.items():
        assert abs(dv - listed[wa]["d"]) <= 1e-9 + 1e-6 * listed[wa]["d"], (wa, dv, listed[wa]["d"])
    assert np.median(d) < 1e-7
    # ... and, independent of the committed list: an absolute envelope - no state further from the oracle's than the
    # diameter of the trust region both are confined to (sqp/dsqp_solver.cc:970-994: |x - x_guess| <= r_trust on either side) - and
    # the outliers sit on agents where the reference algorithm is itself rounding-sensitive: the oracle's FMA build, its build with
    # the product's trigonometry, or the oracle against the binary128 arbiter moves the agent by more than 1e-6 (all but a fifth).
    assert d.max() <= 2.0 * float(worlds[0].parm.r_trust) + 1e-3, float(d.max())
    calm = [wa for wa, o in listed.items() if not (o.get("d_oracle_fma", 0.0) > 1e-6 or o.get("d_oracle_xm", 0.0) > 1e-6 or
                                                   o.get("d_oracle_q", 0.0) > 1e-6)]
    assert len(calm) <= 0.2 * len(listed) + 1, calm
    # implementation-independent acceptance: the reference's own feasibility test (isFeasible, dsqp_solver.cc:292-420)
    # and the objective, evaluated in numpy on both results.  Agents within 1e-4 of the oracle: the same verdict unless a
    # residual sits within 1e-3 of its threshold, the same objective to 1e-3; the outliers: listed above.
    first = np.cumsum([0] + [w.Na for w in worlds])
    n_verdicts = n_whole = 0
    for k, (w, g, r) in enumerate(zip(worlds, got, ref)):
        close = d[first[k]:first[k + 1]] <= parity.TOL
        fg, fr = results.feasibility(w, g.solutions), results.feasibility(w, r.solutions)
        ok_g = (fg["kin"] < 1e-2) & (fg["planes"] < 1e-1)
        ok_r = (fr["kin"] < 1e-2) & (fr["planes"] < 1e-1)
        borderline = (np.abs(fr["kin"] - 1e-2) < 1e-3) | (np.abs(fr["planes"] - 1e-1) < 1e-3)
        assert np.all((ok_g == ok_r) | borderline | ~close), (k, np.nonzero(ok_g != ok_r)[0])
        both = (g.last_status == 1) & (r.last_status == 1) & close
        np.testing.assert_allclose(fg["objective"][both], fr["objective"][both], rtol=1e-3, atol=1e-4)
        # the authors' acceptance of a result (scripts/collision_detection.py), on the agents of the world that are within 1e-4 of the
        # oracle (all of them in a world without a listed outlier; the 100-vehicle set has an outlier in every world): the same
        # verdicts, vehicle against vehicle and vehicle against obstacle, to the collision count - unless the oracle's own result
        # is within 1e-3 of touching (then a count may differ by a frame or two)
        if close.sum() >= 2:
            vg = results.validate(g.solutions[close], w.veh, w.obstacles, w.dimx, w.dimy)
            vr = results.validate(r.solutions[close], w.veh, w.obstacles, w.dimx, w.dimy)
            near_touch = abs(vr.min_obstacle_clearance) < 1e-3
            assert (vg.obstacle_collisions == 0) == (vr.obstacle_collisions == 0) or near_touch, (k, vg, vr)
            assert (vg.vehicle_collisions == 0) == (vr.vehicle_collisions == 0), (k, vg.vehicle_collisions, vr.vehicle_collisions)
            assert abs(vg.vehicle_collisions - vr.vehicle_collisions) <= 2 and abs(vg.obstacle_collisions - vr.obstacle_collisions) <= 2, (k, vg, vr)
            n_verdicts += 1
            n_whole += int(close.all())
        assert g.initial_static_legal == r.initial_static_legal
    clean = len(worlds) - len({w for w, _ in listed})      # worlds without any listed outlier: compared whole
    print(workload, "worlds whose collision verdicts were compared:", n_verdicts, "of", len(worlds), "- whole worlds:", n_whole)
    assert n_verdicts == len(worlds) and n_whole == clean, (n_verdicts, n_whole, clean)


def test_synthetic_1024_batch(gpu_handle, oracle):
    """BASELINE configs[4]: 21 worlds / 1024 agents in one batch.  Agents are independent, so every world's result equals
    its solve alone, the truncated world equals the first 24 agents of the whole one, and a sample matches the oracle."""
    from csdotrajectoryplanning_amd import workloads
    worlds = _set("synth1024")
    assert sum(w.Na for w in worlds) == 1024 and len(worlds) == 21 and worlds[-1].Na == 24
    got = gpu_handle.solve_batch(worlds)
    assert sum(g["n_agents"] for g in gpu_handle.launch_groups()) == 1024
    full = _set("map100")
    for k in (0, 7, 20):
        alone = gpu_handle.solve(full[k])
        n = worlds[k].Na
        assert np.array_equal(alone.solutions[:n], got[k].solutions) and np.array_equal(alone.admm_iters[:n], got[k].admm_iters)
    # every world against the oracle: the first QP of all 1024 agents (identical counts, <= 1e-5), then the full chain with
    # the map100 bars (these are the first 21 map100 worlds; their chain outliers are in that set's list)
    first_qp = [_with_max_iter(w, 1) for w in worlds]
    d1, _, same1 = _per_agent(gpu_handle.solve_batch(first_qp), oracle.solve_batch(first_qp, THREADS))
    assert same1.all() and d1.max() <= FIRST_QP_TOL, (int((~same1).sum()), float(d1.max()))
    ref = oracle.solve_batch(worlds, THREADS)
    d, dc, same = _per_agent(got, ref)
    fx = _fixture("synth1024")
    found = {_world_agent(worlds, g) for g in np.nonzero(~same | (d > parity.TOL))[0]}
    assert found == {(o["world"], o["agent"]) for o in fx["outliers"]}, found
    for w, g in zip(worlds, got):
        ok = g.last_status == 1
        assert np.all(g.solutions[:, -1, 4:] == 0)
        assert np.all(np.abs(g.solutions[ok][:, :, :2] - w.x0_bar[ok][:, :, :2]) <= w.parm.r_trust + 0.5)
        assert np.all(g.sqp_iters >= 1) and np.all(g.sqp_iters <= 10) and np.all(g.admm_iters <= 4000)
