"""Parity at BASELINE.json's full sizes, every agent, through the C ABI (libcsdo_hip.so) against the oracle.

What is pinned per agent (not per cent):
  * ONE QP (QpParm.max_iter = 1: one SQP iteration from x0_bar, identical corridor boxes by construction): identical
    ADMM iteration count and OSQP status for EVERY agent of the set and |d| <= FIRST_QP_TOL on every state / control.
    This is the statement about the kernel (assembly, Ruiz, factor, ADMM, termination, adaptive rho).
  * TWO QPs (max_iter = 2: one corridor refresh in between): identical counts for every agent; agents whose refreshed boxes
    flipped a 0.1 m growth step (sqp/corridor.cc:284-315 is discontinuous) are LISTED and held to CORRIDOR_FLIP_TOL, every
    other agent to SECOND_QP_TOL.
  * the full chain (max_iter = 10): an ADMM stopped at eps = 1e-3 on a QP whose Hessian is singular in 4Nt of its 6Nt-2
    variables, re-linearised up to ten times, amplifies a last-bit difference by ~30x per SQP iteration (measured, DESIGN
    section 4): the oracle differs from ITSELF under a change of rounding by the same amounts, and so does the HIP build
    from the lane-serial host build of its own source (libm ulps).  So the chain is pinned (a) against that lane-serial
    build, (b) against the oracle by bars fitted to the measurements with every agent beyond 1e-4 held against a committed
    outlier list (tests/golden/chain_outliers_*.json) and against the oracle's own rounding sensitivity on that agent,
    (c) by a growth-law test on every listed outlier (max_iter = 1..10) and (d) by an implementation-independent
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


# Measured on MI355X (profiles/r03_chain_*.json, scripts/chain_parity.py): over a chain of up to ten QPs a last-bit difference
# grows by a factor 3-9 per cut of the chain for the sensitive agents, with single steps of 10^2-10^6 where a QP's termination
# check or a 0.1 m box growth step flips - and the ORACLE differs from ITSELF built with fused multiply-adds by the same
# factors on the same agents (growth 3.3-9.2 per cut against 3.3-8.7 for HIP vs oracle, the same worst single steps).  So:
#   * the bars below are fitted to the measured front-end workloads (map100: 2999/3000 identical counts, 2971 within 1e-4,
#     max 1.08 - one agent, the oracle's own two builds: 0.32; map50: 1500/1500 identical counts and 1500/1500 within 1e-4 - north_star's bar met on every agent - since the
#     instances the default search rules do not solve are planned with the reference's rules instead of the stand-in);
#   * every agent beyond 1e-4 must be LISTED in tests/golden/chain_outliers_<workload>.json (written by scripts/chain_parity.py
#     on the GPU) or be an agent on which the oracle differs from its own FMA build by more than 1e-6 in this very run: a new
#     outlier on an agent the reference algorithm is NOT sensitive on fails;
#   * test_outlier_growth_* runs every listed outlier alone with max_iter = 1..10 on HIP, the lane-serial build, the oracle and
#     the FMA oracle and asserts the growth law.
CHAIN_BARS = {   # min fraction with identical counts, min fraction <= 1e-6, min fraction <= 1e-4, max median, max
    "map100": dict(same=0.999, le_1e6=0.93, le_1e4=0.985, median=1e-7, max=1.4),   # (round 4: 2999 / 3000, 2971 within 1e-4, max 1.08 m)
    "map50": dict(same=1.0, le_1e6=0.98, le_1e4=1.0, median=1e-8, max=1.0e-4),
    # the two regimes the benchmark sets do not reach (profiles/r04_chain_room50.json, r04_chain_agents100.json): walls of obstacles,
    # where box growth steps flip (room50: 599 / 600 identical counts, 563 within 1e-4, max 1.13 m; the oracle against its own FMA
    # build: 599, 578, 1.51 m), and the seeded stand-in's colliding coarse paths of the 100-vehicle instances, QPs that run to the
    # iteration cap (agents100: 1181 / 1200, 1031, 1.93 m; oracle against itself: 1184, 1075, 1.62 m).  `max` is the measured
    # maximum + 25 %: what protects an agent is not that bar but the committed outlier list and the envelope test below.
    "room50": dict(same=0.995, le_1e6=0.77, le_1e4=0.925, median=1e-7, max=1.45),
    "agents100": dict(same=0.98, le_1e6=0.64, le_1e4=0.845, median=5e-7, max=2.4),
}
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _fixture(workload):
    import json
    with open(os.path.join(GOLDEN, "chain_outliers_%s.json" % workload)) as f:
        return json.load(f)


def _world_agent(worlds, flat_index):
    first = np.cumsum([0] + [w.Na for w in worlds])
    wi = int(np.searchsorted(first, flat_index, side="right") - 1)
    return wi, int(flat_index - first[wi])


def _chain_check(d, same, workload, what):
    bars = CHAIN_BARS[workload]
    stats = dict(same=float(np.mean(same)), le_1e6=float(np.mean(d <= 1e-6)), le_1e4=float(np.mean(d <= parity.TOL)),
                 median=float(np.median(d)), max=float(d.max()))
    print(what, workload, stats)
    assert stats["same"] >= bars["same"] and stats["le_1e6"] >= bars["le_1e6"] and stats["le_1e4"] >= bars["le_1e4"], stats
    assert stats["median"] <= bars["median"], stats
    assert stats["max"] <= bars["max"], stats      # (both stay inside the +-2 m trust region around x0_bar)


def _outliers_are_accounted_for(worlds, d, same, d_ref_sens, workload, what):
    """Every agent beyond 1e-4 (or with different counts) is in the committed list or is one the reference algorithm itself
    is rounding-sensitive on (d_ref_sens: oracle vs its FMA build, this run); returns the outliers found."""
    fx = _fixture(workload)
    listed = {(o["world"], o["agent"]) for o in fx["outliers"]}
    idx = np.nonzero(~same | (d > parity.TOL))[0]
    found = [_world_agent(worlds, g) for g in idx]
    # an outlier that is NOT in the committed list must be one the reference algorithm is rounding-sensitive on in this very run,
    # and by no more than the envelope of the growth test: 300 x what the oracle's two builds differ by on that agent
    new = [(wa, float(d[g]), float(d_ref_sens[g])) for wa, g in zip(found, idx)
           if wa not in listed and not (d_ref_sens[g] > 1e-6 and d[g] <= 300.0 * d_ref_sens[g])]
    print(what, workload, "outliers found: %d, of them not in the committed list: %s" % (len(found), [wa for wa in found if wa not in listed]))
    assert not new, ("outliers outside the list and outside 300 x the reference algorithm's own sensitivity on that agent", new)
    assert len(found) <= 1.25 * len(listed) + 2, (len(found), len(listed))
    return found


@pytest.mark.parametrize("workload", ["map100", "map50"])
def test_full_chain_hip_build_against_lane_serial_build(gpu_handle, emu, oracle, workload):
    """Same program source as HIP device code and lane-serially on the host, every agent of the set."""
    worlds = _set(workload)
    got = gpu_handle.solve_batch(worlds)
    ref = emu.solve_batch(worlds, 0, THREADS)
    d, dc, same = _per_agent(got, ref)
    _chain_check(d, same, workload, "HIP vs lane-serial build")
    assert same.all()                                  # measured: identical counts on every agent of both sets
    d_sens = _per_agent(oracle.solve_batch_fma(worlds, THREADS), oracle.solve_batch(worlds, THREADS))[0]
    # the two builds differ in libm ulps only: whatever exceeds 1e-4 must be an agent the oracle is itself sensitive on
    bad = [(_world_agent(worlds, g), float(d[g]), float(d_sens[g])) for g in np.nonzero(d > parity.TOL)[0] if not d_sens[g] > 1e-6]
    assert not bad, bad


@pytest.mark.parametrize("workload", ["map100", "map50", "room50", "agents100"])
def test_full_chain_against_oracle_with_acceptance(gpu_handle, oracle, workload):
    from csdotrajectoryplanning_amd import results
    worlds = _set(workload)
    got = gpu_handle.solve_batch(worlds)
    ref = oracle.solve_batch(worlds, THREADS)
    d, dc, same = _per_agent(got, ref)
    _chain_check(d, same, workload, "HIP vs oracle")
    d_sens = _per_agent(oracle.solve_batch_fma(worlds, THREADS), ref)[0]
    _outliers_are_accounted_for(worlds, d, same, d_sens, workload, "HIP vs oracle")
    # implementation-independent acceptance: the reference's own feasibility test (isFeasible, dsqp_solver.cc:292-420)
    # and the objective, evaluated in numpy on both results.  Agents within 1e-4 of the oracle: the same verdict unless a
    # residual sits within 1e-3 of its threshold, the same objective to 1e-3; the outliers: listed above.
    first = np.cumsum([0] + [w.Na for w in worlds])
    n_verdicts = 0
    for k, (w, g, r) in enumerate(zip(worlds, got, ref)):
        close = d[first[k]:first[k + 1]] <= parity.TOL
        fg, fr = results.feasibility(w, g.solutions), results.feasibility(w, r.solutions)
        ok_g = (fg["kin"] < 1e-2) & (fg["planes"] < 1e-1)
        ok_r = (fr["kin"] < 1e-2) & (fr["planes"] < 1e-1)
        borderline = (np.abs(fr["kin"] - 1e-2) < 1e-3) | (np.abs(fr["planes"] - 1e-1) < 1e-3)
        assert np.all((ok_g == ok_r) | borderline | ~close), (k, np.nonzero(ok_g != ok_r)[0])
        both = (g.last_status == 1) & (r.last_status == 1) & close
        np.testing.assert_allclose(fg["objective"][both], fr["objective"][both], rtol=1e-3, atol=1e-4)
        vg = results.validate(g.solutions, w.veh, w.obstacles, w.dimx, w.dimy)
        vr = results.validate(r.solutions, w.veh, w.obstacles, w.dimx, w.dimy)
        # the authors' acceptance of a result (scripts/collision_detection.py): a world all of whose agents are within 1e-4 of the
        # oracle gets the same verdicts, vehicle against vehicle and vehicle against obstacle, to the collision count - unless the
        # oracle's own result is within 1e-3 of touching (then a count may differ by a frame or two); a world with a chain outlier
        # (listed above) may differ
        if close.all():
            near_touch = abs(vr.min_obstacle_clearance) < 1e-3
            assert (vg.obstacle_collisions == 0) == (vr.obstacle_collisions == 0) or near_touch, (k, vg, vr)
            assert (vg.vehicle_collisions == 0) == (vr.vehicle_collisions == 0), (k, vg.vehicle_collisions, vr.vehicle_collisions)
            assert abs(vg.vehicle_collisions - vr.vehicle_collisions) <= 2 and abs(vg.obstacle_collisions - vr.obstacle_collisions) <= 2, (k, vg, vr)
            n_verdicts += 1
        assert g.initial_static_legal == r.initial_static_legal
    print(workload, "worlds whose collision verdicts were compared (all agents within 1e-4):", n_verdicts, "of", len(worlds))
    assert n_verdicts >= {"map100": 30, "map50": 60, "room50": 0, "agents100": 0}[workload]


@pytest.mark.parametrize("workload", ["map100", "map50", "room50", "agents100"])
def test_outlier_growth_is_the_reference_algorithms_own_amplification(gpu_handle, emu, oracle, workload):
    """Every committed outlier alone, the chain cut after k = 1..10 QPs (QpParm.max_iter = k), on HIP, the lane-serial build,
    the oracle and the oracle built with fused multiply-adds.  With d_k = max |difference| after the cut at k:
      * the seed is small: d_1(HIP, oracle) <= 1e-6 - one QP is where the kernel is compared, everything later is the chain;
      * HIP-vs-oracle grows like the oracle's own rounding sensitivity: geometric-mean growth per cut within a factor 2 of
        oracle-vs-FMA-oracle, worst single step within a factor 30 of its worst single step (no jump of HIP's own);
      * at every k HIP is no further from the oracle than 300x, and from the lane-serial build of its own source than 100x,
        what the oracle's two builds have differed by up to that k (floor 1e-9)."""
    fx = _fixture(workload)
    fx = dict(fx, outliers=fx["outliers"][:40])      # (sorted by size; agents100 has 169: its forty worst)
    worlds = _set(workload)
    singles = [worlds[o["world"]].subset(o["agent"], o["agent"] + 1) for o in fx["outliers"]]
    if not singles:
        pytest.skip("no agent of the %s set is beyond 1e-4 of the oracle (tests/golden/chain_outliers_%s.json)" % (workload, workload))
    D = {name: np.zeros((len(singles), 10)) for name in ("hip_oracle", "fma_oracle", "hip_emu")}
    for k in range(1, 11):
        ws = [_with_max_iter(w, k) for w in singles]
        hip, ref = gpu_handle.solve_batch(ws), oracle.solve_batch(ws, THREADS)
        fma, em = oracle.solve_batch_fma(ws, THREADS), emu.solve_batch(ws, 0, THREADS)
        for j in range(len(ws)):
            D["hip_oracle"][j, k - 1] = np.abs(hip[j].solutions - ref[j].solutions).max()
            D["fma_oracle"][j, k - 1] = np.abs(fma[j].solutions - ref[j].solutions).max()
            D["hip_emu"][j, k - 1] = np.abs(hip[j].solutions - em[j].solutions).max()
    ho, fo, he = D["hip_oracle"], D["fma_oracle"], D["hip_emu"]
    assert ho[:, 0].max() <= 1e-6, ho[:, 0]
    growth = lambda a: (np.maximum(a[:, -1], 1e-10) / np.maximum(a[:, 0], 1e-10)) ** (1.0 / 9.0)
    jump = lambda a: (np.maximum(a[:, 1:], 1e-10) / np.maximum(a[:, :-1], 1e-10)).max(axis=1)
    g_h, g_f, j_h, j_f = growth(ho), growth(fo), jump(ho), jump(fo)
    envelope = np.maximum.accumulate(np.maximum(fo, 1e-9), axis=1)
    for j, o in enumerate(fx["outliers"]):
        print("outlier world %d agent %d: growth per cut %.1f (oracle's own %.1f), worst step %.0f (%.0f), d_k = %s" %
              (o["world"], o["agent"], g_h[j], g_f[j], j_h[j], j_f[j], " ".join("%.0e" % v for v in ho[j])))
    assert np.all(g_h <= 2.0 * g_f), (g_h, g_f)
    # every listed outlier - except in the plane-dense set, whose inputs (the stand-in's colliding coarse paths) put one agent in
    # seven beyond 1e-4 for the oracle's own two builds as well: there the two builds of the oracle stay together on some agents
    # on which HIP and the oracle part, and the other way round (measured over its 169 outliers: 96 % inside the envelope, 98 %
    # inside the step bound; of the forty worst 93 % and 95 %), so the bound is on the fraction
    frac = 1.0 if workload != "agents100" else 0.85
    assert np.mean(j_h <= 30.0 * j_f) >= frac, (j_h, j_f)
    assert np.mean((ho <= 300.0 * envelope).all(axis=1)) >= frac, float((ho / envelope).max())
    assert np.mean((he <= 100.0 * envelope).all(axis=1)) >= frac, float((he / envelope).max())


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
    _chain_check(d, same, "map100", "synth1024: HIP vs oracle")
    d_sens = _per_agent(oracle.solve_batch_fma(worlds, THREADS), ref)[0]
    _outliers_are_accounted_for(worlds, d, same, d_sens, "map100", "synth1024: HIP vs oracle")
    for w, g in zip(worlds, got):
        ok = g.last_status == 1
        assert np.all(g.solutions[:, -1, 4:] == 0)
        assert np.all(np.abs(g.solutions[ok][:, :, :2] - w.x0_bar[ok][:, :, :2]) <= w.parm.r_trust + 0.5)
        assert np.all(g.sqp_iters >= 1) and np.all(g.sqp_iters <= 10) and np.all(g.admm_iters <= 4000)
