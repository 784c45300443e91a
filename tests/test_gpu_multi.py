"""Several GPUs behind ONE handle (csdo_dsqp_create_multi, include/csdo_dsqp.h; the loop that shards is
sqp/dsqp_solver.cc:1198-1220).  The metered box has one GPU: devices = [0, 0] runs two child handles on it - the plan, the host
threads, the scatter into the caller's arrays and the per-world scalars are what is tested; the results must be the bits of the
single-handle batch.  (tests/test_multi_host.py holds the sharding rule equal to the N-process path's on the CPU.)"""
import ctypes as C

import numpy as np
import pytest

from csdotrajectoryplanning_amd import abi
from csdotrajectoryplanning_amd._lib import CsdoError, lib
from tests import helpers

pytestmark = pytest.mark.gpu


def _same(a, b):
    return (np.array_equal(a.solutions, b.solutions) and np.array_equal(a.corridors, b.corridors) and
            np.array_equal(a.sqp_iters, b.sqp_iters) and np.array_equal(a.admm_iters, b.admm_iters) and
            np.array_equal(a.last_status, b.last_status) and a.solver_status == b.solver_status and
            a.initial_static_legal == b.initial_static_legal)


def _worlds(name, n):
    from csdotrajectoryplanning_amd import workloads
    return [workloads.build_job(j)[0] for j in workloads.workload_jobs(name, n)]


@pytest.mark.parametrize("n_dev", [2, 3])
def test_two_children_on_one_gpu_return_the_bits_of_the_single_handle(gpu_handle, n_dev):
    from csdotrajectoryplanning_amd.solver import DsqpHandle, estimate_work
    worlds = _worlds("map50", 5)           # 125 agents: the cuts fall inside worlds
    ref = gpu_handle.solve_batch(worlds)
    h = DsqpHandle(devices=[0] * n_dev)
    try:
        assert lib().csdo_dsqp_multi_count(h._h) == n_dev and lib().csdo_dsqp_multi_count(gpu_handle._h) == 0
        got = h.solve_batch(worlds)
        assert all(_same(g, r) for g, r in zip(got, ref))
        assert all(g.t_max_individual > 0 and g.t_device > 0 for g in got)
        # the children's launch groups one after the other cover every agent once; the blocks carry equal estimated work
        groups, of = h.launch_groups(), h.agent_groups()
        assert sum(g["n_agents"] for g in groups) == 125 == len(of) and of.max() == len(groups) - 1
        est = estimate_work(worlds)
        cuts = np.zeros(n_dev + 1, np.int32)
        assert lib().csdo_dsqp_shard_bounds(abi.as_double_p(est), len(est), n_dev, abi.as_int32_p(cuts)) == 0
        per_child = []
        for k in range(n_dev):
            kid = lib().csdo_dsqp_multi_child(h._h, k)
            assert kid
            buf = (abi.LaunchGroup * 8)()
            n = lib().csdo_dsqp_launch_groups(kid, buf, 8)
            per_child.append(sum(buf[i].n_agents for i in range(n)))
            nd = C.c_int64()
            assert lib().csdo_dsqp_device_solutions(kid, C.byref(nd)) and nd.value > 0     # one buffer per device, for a collective
        assert per_child == np.diff(cuts).tolist(), (per_child, cuts)
        assert lib().csdo_dsqp_device_solutions(h._h, None) is None
        assert not lib().csdo_dsqp_multi_child(h._h, n_dev)
        # split phase, twice: resident inputs, repeatable runs, results into the arrays of the first download
        h.upload(worlds)
        t1 = h.run()
        first = h.download()
        h.run_async()
        with pytest.raises(CsdoError):
            h.download()                      # a run is pending
        t2 = h.wait()
        again = h.download(out=first)
        assert t1 > 0 and t2 > 0 and all(a is f for a, f in zip(again, first)) and all(_same(g, r) for g, r in zip(again, ref))
        # the single-device entries run on the first device
        w = worlds[0]
        b, s = h.generate_boxes(w.x0_bar[0, :, :2], w.obstacles, w.dimx, w.dimy, w.veh)
        b0, s0 = gpu_handle.generate_boxes(w.x0_bar[0, :, :2], w.obstacles, w.dimx, w.dimy, w.veh)
        assert np.array_equal(b, b0) and np.array_equal(s, s0)
    finally:
        h.close()


def test_more_devices_than_agents_and_one_world(gpu_handle, veh_parm):
    from csdotrajectoryplanning_amd.solver import DsqpHandle
    veh, parm = veh_parm
    world, _ = helpers.load_golden("map50_agents15to17.npz", veh, parm)      # 3 agents
    ref = gpu_handle.solve(world)
    h = DsqpHandle(devices=[0, 0, 0, 0, 0])
    try:
        got = h.solve(world)
        assert _same(got, ref)
        one = h.solve(world.subset(1, 2))
        assert np.array_equal(one.solutions[0], ref.solutions[1])
    finally:
        h.close()


def test_a_world_that_does_not_fit_is_named_in_the_callers_indices(gpu_handle, veh_parm):
    """CSDO_ELIMIT from a child's upload: csdo_dsqp_last_limit of the multi handle speaks of the caller's world / agent."""
    from csdotrajectoryplanning_amd.problem import World
    from csdotrajectoryplanning_amd.solver import DsqpHandle
    veh, parm = veh_parm
    small, _ = helpers.load_golden("map50_agents15to17.npz", veh, parm)
    rng = np.random.default_rng(0)
    many = np.column_stack([rng.uniform(200, 300, 7000), rng.uniform(200, 300, 7000), np.full(7000, 0.5)])   # far away, too many for LDS
    big = World(small.x0_bar, small.plane_off, small.planes, 400.0, 400.0, many, veh, parm)
    h = DsqpHandle(devices=[0, 0])
    try:
        with pytest.raises(CsdoError) as e:
            h.solve_batch([small, small, big])
        assert "-4" in str(e.value)
        w, a, need = h.last_limit()
        assert w == 2 and 0 <= a < 3 and need > 160 * 1024 - 64
        assert _same(h.solve(small), gpu_handle.solve(small))      # the handle is usable afterwards
    finally:
        h.close()


def test_mixed_768_batch_is_batch_chunk_and_shard_independent(gpu_handle, veh_parm):
    """the 768-thread class has two residency modes whose solves differ in form - 2: pair-split, 3: one lane per
    node - and hence in last bits.  Round 5 put EVERY 768-thread agent of a batch into mode 3 as soon as one of them needed it, so a
    long-horizon agent's bits depended on its neighbours (and on which shard it landed in).  The mode is now the agent's own
    (dsqp_agent_class): a batch that mixes mode-2 agents (a room world's 257..343-step horizons, a 271-step line), a mode-3 agent by
    horizon (379 steps) and mode-3 agents by obstacle count (5000 obstacles beside 91 steps) returns, world for world, the bits of
    each world solved alone, and the same bits through create_multi with two and three children."""
    from csdotrajectoryplanning_amd import workloads
    from csdotrajectoryplanning_amd.problem import World
    from csdotrajectoryplanning_amd.solver import DsqpHandle
    veh, parm = veh_parm
    room = [workloads.build_job(j)[0] for j in workloads.workload_jobs("room50", 12)]
    room = [w for w in room if w.Nt > 256][:1]
    assert room, "the room set has worlds with horizons beyond 256"
    line2 = helpers.straight_line_world(veh, parm, Na=2, L=90, dim=600.0, spacing=3.5)      # Nt = 271: mode 2
    line3 = helpers.straight_line_world(veh, parm, Na=2, L=126, dim=600.0, spacing=3.5)     # Nt = 379: mode 3 by horizon
    short, _ = helpers.load_golden("map50_agents15to17.npz", veh, parm)
    rng = np.random.default_rng(2)
    far = np.column_stack([rng.uniform(300, 400, 5000), rng.uniform(300, 400, 5000), np.full(5000, 0.5)])
    heavy = World(short.x0_bar, short.plane_off, short.planes, 500.0, 500.0, np.vstack([short.obstacles, far]), veh, short.parm)
    batch = [line2, heavy] + room + [line3]
    got = gpu_handle.solve_batch(batch)
    kinds = {(g["threads"], g["residency_mode"]) for g in gpu_handle.launch_groups()}
    assert {(768, 2), (768, 3)} <= kinds, kinds
    for w, b in zip(batch, got):
        assert _same(gpu_handle.solve(w), b)
    # a batch WITHOUT the lean agents: the mode-2 worlds keep their bits
    sub = gpu_handle.solve_batch([line2] + room)
    assert {(g["threads"], g["residency_mode"]) for g in gpu_handle.launch_groups()} >= {(768, 2)}
    assert _same(sub[0], got[0]) and _same(sub[1], got[2])
    for n_dev in (2, 3):
        h = DsqpHandle(devices=[0] * n_dev)
        try:
            assert all(_same(g, r) for g, r in zip(h.solve_batch(batch), got))
        finally:
            h.close()
