"""The dense tail of the block cyclic reduction has six nodes, or eight for an agent of the 512-thread class whose reduction is one
level shorter with eight (horizons 193 .. 256, and 97 .. 128 where the 256-thread class has no room) - csrc/dsqp_class.h.  The tail's
size decides the elimination order of the last nodes, so it is the one item of an agent's kernel class that the results' last bits
depend on: the product library and the lane-serial test build must apply the same rule (here, on the CPU: the rule is host code),
and the arithmetic with eight nodes is held to the oracle like the rest (one QP: identical ADMM counts, 1e-6; two QPs: 1e-4)."""
import ctypes as C

import numpy as np
import pytest

from tests import parity


def _product_class(nt, n_obs, n_planes):
    from csdotrajectoryplanning_amd import _lib
    out = (C.c_int64 * 5)()
    assert _lib.lib().csdo_dsqp_agent_class(nt, n_obs, n_planes, out) == 0
    return tuple(int(v) for v in out)


def test_the_class_rule_is_the_same_in_the_product_and_in_the_lane_serial_build(emu):
    rng = np.random.default_rng(5)
    cases = [(214, 172, 496), (214, 172, 0), (166, 50, 100), (100, 25, 30), (115, 50, 40), (128, 50, 60), (129, 50, 60), (192, 50, 60),
             (193, 50, 60), (232, 211, 308), (253, 179, 176), (256, 10, 0), (257, 138, 100), (343, 138, 314), (400, 10, 10), (512, 0, 0)]
    cases += [(int(rng.integers(2, 513)), int(rng.integers(0, 400)), int(rng.integers(0, 600))) for _ in range(300)]
    for c in cases:
        assert _product_class(*c) == emu.agent_class(*c), c
    cap = 160 * 1024 - 64
    # the named cases: eight nodes where the horizon loses a level and the 48 x 50 inverse fits, six everywhere else
    assert _product_class(214, 172, 496)[:4] == (512, 0, 0, 8)       # the room set's capped agents
    assert _product_class(166, 50, 100)[:4] == (512, 0, 1, 6)        # map100: ceil(166 / 16) = 11 nodes would be needed
    assert _product_class(193, 50, 60)[3] == 8 and _product_class(192, 50, 60)[3] == 6
    assert _product_class(100, 25, 30)[:4] == (256, 0, 1, 6)         # two workgroups per CU: six
    assert _product_class(232, 211, 308)[3] == 6                     # 78 x 232 doubles + 211 obstacles: no room for 48 x 50
    assert _product_class(253, 179, 176)[:2] == (512, 1) and _product_class(253, 179, 176)[3] == 8   # the lean 512-thread layout
    assert _product_class(343, 138, 314)[0] == 768 and _product_class(343, 138, 314)[3] == 6
    for c in cases:   # whatever the rule picks fits, unless nothing does
        blk, mode, rows, tail, nbytes = _product_class(*c)
        if tail != 6:
            assert blk == 512 and nbytes <= cap, c


def _with_max_iter(world, k):
    from csdotrajectoryplanning_amd.abi import QpParm
    from csdotrajectoryplanning_amd.problem import World
    p = QpParm.from_buffer_copy(bytes(world.parm))
    p.max_iter = float(k)
    return World(world.x0_bar, world.plane_off, world.planes, world.dimx, world.dimy, world.obstacles, world.veh, p)


@pytest.mark.parametrize("workload,index", [("room50", 8), ("room50", 10), ("map50", 1)])
def test_eight_node_tail_against_the_oracle(emu, oracle, workload, index):
    """Worlds whose agents take the eight-node tail (214 and 202 timesteps; 115 timesteps in the 512-thread class)."""
    from csdotrajectoryplanning_amd import workloads
    w = workloads.build_job(workloads.workload_jobs(workload)[index])[0]
    per = np.diff(np.asarray(w.plane_off))
    tails = {emu.agent_class(w.Nt, len(w.obstacles), int(k))[3] for k in per}
    assert 8 in tails, (w.Nt, tails)
    for k in (1, 2):
        wk = _with_max_iter(w, k)
        got, ref = emu.solve_batch([wk], 0, 8)[0], oracle.solve_batch([wk], 8)[0]
        assert np.array_equal(got.admm_iters, ref.admm_iters) and np.array_equal(got.last_status, ref.last_status)
        d = np.abs(got.solutions - ref.solutions).max()
        # one QP: 1e-6 (measured 3.5e-8 over the whole room set); two QPs: north_star's 1e-4 - agent 16 of the 214-step world, whose
        # QPs run to OSQP's iteration cap, is at 2.1e-5 there and on the committed list of agents that part from the oracle later
        assert d <= (1e-6 if k == 1 else 1e-4), (k, d)
        for mode in (1, 2, 10):   # the other residency modes of the pair-split solve: the same bits
            other = emu.solve_batch([wk], mode, 8)[0]
            assert np.array_equal(other.solutions, got.solutions) and np.array_equal(other.admm_iters, got.admm_iters), mode
        if k == 1:   # the one-lane form of the solve (mode 3) absorbs the partials of a node at a multiple of 64 in another order
            other = emu.solve_batch([wk], 3, 8)[0]
            assert np.array_equal(other.admm_iters, got.admm_iters) and np.abs(other.solutions - got.solutions).max() < 1e-6


def test_run_time_tail_size_equals_the_compile_time_one_for_six_nodes(emu, veh_parm):
    """The 512-thread kernels read the tail's size per agent (BIGT), the other classes are compiled with six nodes as a constant, and
    the lane-serial build runs EVERY agent through the run-time form: for six-node agents the two must be the same program
    (only the GPU runs had checked that).  Lane-serial build in both forms, a map50 world (256-thread class, 97 timesteps)
    and a map100 world restricted to its six-node agents: the same bits."""
    from csdotrajectoryplanning_amd import workloads
    from tests import helpers
    veh, parm = veh_parm
    short, _ = helpers.load_golden("map50_agents15to17.npz", veh, parm)
    w100 = workloads.build_job(workloads.workload_jobs("map100", 1)[0])[0]
    six = [a for a, k in enumerate(np.diff(np.asarray(w100.plane_off))) if emu.agent_class(w100.Nt, len(w100.obstacles), int(k))[3] == 6]
    assert len(six) >= 10
    picks = [w100.subset(a, a + 1) for a in six[:12]]
    for w in [short] + picks:
        a, b = emu.solve_batch([w], 0, 4)[0], emu.solve_batch([w], 20, 4)[0]
        assert np.array_equal(a.solutions, b.solutions) and np.array_equal(a.corridors, b.corridors)
        assert np.array_equal(a.admm_iters, b.admm_iters) and np.array_equal(a.last_status, b.last_status)
