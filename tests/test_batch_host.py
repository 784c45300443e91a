"""Host logic around a batch: input validation of the packing (csrc/batch_pack.h, exercised through the lane-serial
build, which shares it with the HIP library), the oracle's pooled batch entry, the named workloads."""
import numpy as np

from csdotrajectoryplanning_amd import abi, config
from tests import helpers


def test_mixed_parameter_batch_is_rejected(emu, veh_parm):
    """A launch reads ONE parameter block: worlds with different QpParm / vehicle geometry in one batch -> CSDO_EINVAL
    (never a silent solve with worlds[0]'s values)."""
    veh, parm = veh_parm
    w1, _ = helpers.load_golden("map50_agents15to17.npz", veh, parm)
    rc, _ = emu.solve_batch_rc([w1, w1])
    assert rc == abi.CSDO_OK
    for change in ({"r_trust": 1.5}, {"max_iter": 3}, {"osqp_max_iter": 200}, {"fixed_corridor": True}):
        w2, _ = helpers.load_golden("map50_agents15to17.npz", veh, config.qp_parm_from_config(change))
        rc, _ = emu.solve_batch_rc([w1, w2])
        assert rc == abi.CSDO_EINVAL, change
    w3, _ = helpers.load_golden("map50_agents15to17.npz", config.vehicle_from_config({"LF": 2.5}), parm)
    assert emu.solve_batch_rc([w1, w3])[0] == abi.CSDO_EINVAL
    assert emu.solve_batch_rc([w3, w3])[0] == abi.CSDO_OK          # consistent batches of any parameter set are fine


def test_malformed_worlds_are_rejected(emu, veh_parm):
    from csdotrajectoryplanning_amd.problem import World
    veh, parm = veh_parm
    w, _ = helpers.load_golden("map50_agents15to17.npz", veh, parm)
    bad = World(w.x0_bar, w.plane_off.copy(), w.planes, w.dimx, w.dimy, w.obstacles, veh, parm)
    bad.plane_off[1] = bad.plane_off[2] + 1                         # non-monotone CSR offsets
    assert emu.solve_batch_rc([bad])[0] == abi.CSDO_EINVAL
    bad2 = World(w.x0_bar, w.plane_off, w.planes.copy(), w.dimx, w.dimy, w.obstacles, veh, parm)
    bad2.planes["t"][0] = w.Nt                                      # plane beyond the horizon
    assert emu.solve_batch_rc([bad2])[0] == abi.CSDO_EINVAL
    p = w.c_problem()
    p.n_obs = -1
    import ctypes as C
    from csdotrajectoryplanning_amd.problem import Solution
    s = Solution.allocate(w.Na, w.Nt)
    assert emu.lib().csdo_emu_solve_batch_mt(C.byref(p), 1, C.byref(s._c), 0, 1) == abi.CSDO_EINVAL


def test_oracle_pooled_batch_equals_separate_solves(oracle, veh_parm):
    veh, parm = veh_parm
    ws = [helpers.load_golden(n, veh, parm)[0] for n in ("map50_agents15to17.npz", "map100_agents0to3.npz")]
    pooled = oracle.solve_batch(ws, 4)
    for w, p in zip(ws, pooled):
        s = oracle.solve(w, 1)
        assert np.array_equal(s.solutions, p.solutions) and np.array_equal(s.corridors, p.corridors)
        assert np.array_equal(s.admm_iters, p.admm_iters) and s.solver_status == p.solver_status
        assert s.initial_static_legal == p.initial_static_legal


def test_named_workloads():
    from csdotrajectoryplanning_amd import sharding, workloads
    for name, n_worlds, n_agents in (("map100", 60, 3000), ("map50", 60, 1500), ("synth1024", 21, 1024)):
        jobs = workloads.workload_jobs(name)
        sizes = [workloads.job_agents(j) for j in jobs]
        assert len(jobs) == n_worlds and sum(sizes) == n_agents
        for N in (2, 4, 8):
            plans = [sharding.shard_batch_plan(sizes, r, N) for r in range(N)]
            per_rank = [sum(hi - lo for _, lo, hi in p) for p in plans]
            assert sum(per_rank) == n_agents and max(per_rank) - min(per_rank) <= 1
    w, info = workloads.build_job(workloads.workload_jobs("synth1024")[20])
    assert w.Na == 24 and info["truncated_to"] == 24 and int(w.plane_off[-1]) == len(w.planes)
    w50, _ = workloads.build_job(workloads.workload_jobs("map50")[7])
    assert w50.Na == 25 and w50.dimx == 50.0 and len(w50.obstacles) == 25


def test_feasibility_measures(oracle, veh_parm):
    """results.feasibility restates the kinematic / plane parts of isFeasible: a converged oracle result passes its
    thresholds, a perturbed one does not."""
    from csdotrajectoryplanning_amd import results
    veh, parm = veh_parm
    w, _ = helpers.load_golden("map50_agents15to17.npz", veh, parm)
    s = oracle.solve(w, 1)
    f = results.feasibility(w, s.solutions)
    ok = s.last_status == 1
    assert np.all(f["kin"][ok] < 1e-2) and np.all(f["planes"][ok] < 1e-1) and np.all(f["objective"] >= 0)
    moved = s.solutions.copy()
    moved[:, 5:, 0] += 1.0
    assert np.all(results.feasibility(w, moved)["kin"] > f["kin"])
