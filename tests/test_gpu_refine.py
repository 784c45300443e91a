"""csdo_qp_parm::solve_refinement (round 6): every ADMM iteration's linear solve followed by one step of iterative refinement on the
residual of OSQP's KKT system, formed through the constraint rows.  The backend solves the REDUCED system (P + sigma I + A' R A) x = b by
block cyclic reduction, whose error is cond(H) eps |x| - some fifty times that of OSQP's LDL' of the quasi-definite KKT matrix
(scripts/solve_accuracy.py) -, and over an SQP chain that leaves 1.3 - 1.5 times as many agents beyond 1e-4 of the exact-arithmetic
iterate path (the binary128 arbiter, oracle/libcsdo_oracle_q.so) as a double-precision OSQP is; with the refinement the product is
CLOSER to that path than the double-precision oracle (tests/test_refinement.py on the CPU, tests/golden/chain_outliers_*.json).
Here: the HIP kernels with the refinement compiled in (separate instantiations of every kernel class) return the bits of the
lane-serial build of the same source, in every kernel class, and meet the oracle's first-QP bar."""
import numpy as np
import pytest

from tests import helpers

pytestmark = pytest.mark.gpu


def _refined(world, k=None, mode=1):
    from csdotrajectoryplanning_amd.abi import QpParm
    from csdotrajectoryplanning_amd.problem import World
    p = QpParm.from_buffer_copy(bytes(world.parm))
    p.solve_refinement = mode
    if k is not None:
        p.max_iter = float(k)
    return World(world.x0_bar, world.plane_off, world.planes, world.dimx, world.dimy, world.obstacles, world.veh, p)


def _same(a, b):
    return (np.array_equal(a.solutions, b.solutions) and np.array_equal(a.corridors, b.corridors) and
            np.array_equal(a.sqp_iters, b.sqp_iters) and np.array_equal(a.admm_iters, b.admm_iters) and
            np.array_equal(a.last_status, b.last_status))


@pytest.mark.parametrize("refinement", [1, 2])
def test_refined_kernels_return_the_lane_serial_builds_bits_in_every_class(gpu_handle, emu, oracle, veh_parm, refinement):
    """solve_refinement = 1 (a second solve per iteration) and 2 (lagged: the residual joins the next rhs), all six kernel classes."""
    from csdotrajectoryplanning_amd import workloads
    veh, parm = veh_parm
    short, _ = helpers.load_golden("map50_agents15to17.npz", veh, parm)            # 256 threads
    mid, _ = helpers.load_golden("map100_agents0to3.npz", veh, parm)               # 512 threads, mode 0
    room = [workloads.build_job(j)[0] for j in workloads.workload_jobs("room50", 12)]
    room1 = [w for w in room if 232 <= w.Nt <= 256][:1]                            # 512 threads, mode 1 (obstacles leave no room)
    room2 = [w for w in room if w.Nt > 256][:1]                                    # 768 threads, mode 2
    line3 = helpers.straight_line_world(veh, parm, Na=2, L=126, dim=600.0, spacing=3.5)    # 768 threads, mode 3
    line4 = helpers.straight_line_world(veh, parm, Na=2, L=140, dim=700.0, spacing=3.5)    # 1024 threads
    batch = [_refined(w, mode=refinement) for w in [short, mid] + room1 + room2 + [line3, line4]]
    got = gpu_handle.solve_batch(batch)
    kinds = {(g["threads"], g["residency_mode"]) for g in gpu_handle.launch_groups()}
    assert {(256, 0), (512, 0), (768, 2), (768, 3), (1024, 3)} <= kinds, kinds
    # (the lane-serial build emulates ONE residency mode per call: the pair-split solve of modes 0, 1, 2 - the same bits in all three -
    #  for the first four worlds, the one-lane form of mode 3 for the two long lines)
    ref = emu.solve_batch(batch[:-2], 0, 16) + emu.solve_batch(batch[-2:], 3, 16)
    for k, (g, r) in enumerate(zip(got, ref)):
        assert _same(g, r), k
    # ... and differ from the unrefined kernels' (the flag reaches the launch), by rounding only on one QP
    plain = gpu_handle.solve(mid)
    assert not np.array_equal(plain.solutions, got[1].solutions)
    one = gpu_handle.solve(_refined(mid, 1, refinement))
    ref1 = oracle.solve(_refined(mid, 1), 4)
    assert np.array_equal(one.admm_iters, ref1.admm_iters) and np.array_equal(one.last_status, ref1.last_status)
    assert np.abs(one.solutions - ref1.solutions).max() <= 1e-6


def test_refined_map100_worlds_match_the_lane_serial_build_and_cost_about_twice(gpu_handle, emu):
    from csdotrajectoryplanning_amd import workloads
    worlds = [workloads.build_job(j)[0] for j in workloads.workload_jobs("map100", 6)]
    plain = gpu_handle.solve_batch(worlds)
    t_plain = sum(g["seconds"] for g in gpu_handle.launch_groups())
    got = gpu_handle.solve_batch([_refined(w) for w in worlds])
    t_ref = sum(g["seconds"] for g in gpu_handle.launch_groups())
    ref = emu.solve_batch([_refined(w) for w in worlds], 0, 16)
    assert all(_same(g, r) for g, r in zip(got, ref))
    assert 1.2 < t_ref / t_plain < 3.0, (t_plain, t_ref)
    print("refinement: %.2f ms against %.2f ms for %d agents (x %.2f)" % (t_ref * 1e3, t_plain * 1e3, sum(w.Na for w in worlds), t_ref / t_plain))
    assert sum(int(s.admm_iters.sum()) for s in plain) > 0


@pytest.mark.parametrize("workload,refinement", [("map100", 1), ("map50", 1), ("room50", 1), ("agents100", 1), ("synth1024", 1),
                                                 ("map100", 2), ("room50", 2), ("agents100", 2)])
def test_refined_full_chain_is_bit_identical_to_the_lane_serial_build(gpu_handle, emu, workload, refinement):
    """Every agent of the five workloads with the refinement on: the HIP kernels return the lane-serial build's bits - which is what
    lets scripts/chain_parity.py put `product_refined` against the arbiter on the CPU (tests/golden/chain_outliers_*.json `arbiter`)."""
    from tests.test_gpu_sets import _set          # (the session's cache of the built sets)
    worlds = [w.with_parm(solve_refinement=refinement) for w in _set(workload)]
    got = gpu_handle.solve_batch(worlds)
    ref = emu.solve_batch(worlds, 0, 16)
    bad = [k for k, (g, r) in enumerate(zip(got, ref)) if not _same(g, r)]
    assert not bad, bad
    assert sum(w.Na for w in worlds) == {"map100": 3000, "map50": 1500, "room50": 600, "agents100": 1200, "synth1024": 1024}[workload]
