"""The O(Nt Na^2) kernels either side of the solve (csrc/aux_kernels.hip): neighbour search + separating planes on the
device against the host bridge (bit for bit), the trajectory validator on the device against the numpy validator and the
reference's own verdicts (tests/golden/ref_collision_verdicts.npz)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _paths(k, which="map100"):
    from csdotrajectoryplanning_amd import workloads
    w, info = (workloads.map100_world(k) if which == "map100" else workloads.map50_world(k))
    return w, info["paths"]


def _same_bridge(a, b):
    (wa, pa, la), (wb, pb, lb) = a, b
    assert la == lb and np.array_equal(pa, pb)
    assert np.array_equal(wa.x0_bar, wb.x0_bar) and np.array_equal(wa.plane_off, wb.plane_off)
    assert np.array_equal(wa.planes["t"], wb.planes["t"]) and np.array_equal(wa.planes["c"], wb.planes["c"])


@pytest.mark.parametrize("which,k", [("map50", 0), ("map100", 0), ("map100", 7)])
def test_device_bridge_equals_host_bridge_and_the_oracle(gpu_handle, oracle, which, k):
    """The device bridge against the product's host bridge (bit for bit) AND, directly, against the oracle's restatement of
    sqp/inter_agent_cons.cc (the oracle used to be reached only through a CPU test of the host bridge): index
    work - horizon, pair list, plane order, legality flag - identical, poses and steer identical, floating point to 1e-12."""
    from csdotrajectoryplanning_amd.instance import Instance
    from csdotrajectoryplanning_amd.solver import interpolate_and_planes
    w, (st, ac, po, G) = _paths(k, which)
    host = interpolate_and_planes(st, ac, po, G, w.veh, w.parm, w.dimx, w.dimy, w.obstacles)
    dev = gpu_handle.interpolate_and_planes(st, ac, po, G, w.veh, w.parm, w.dimx, w.dimy, w.obstacles)
    assert len(host[1]) > 0
    _same_bridge(host, dev)
    starts = np.array([st[po[a]] for a in range(len(po) - 1)])
    wo, pairs_o, legal_o = oracle.preprocess(st, ac, po, G, w.veh, w.parm, Instance(w.dimx, w.dimy, w.obstacles, starts, G))
    wd, pairs_d, legal_d = dev
    assert wd.Nt == wo.Nt and wd.Na == wo.Na and legal_d == legal_o and np.array_equal(np.asarray(pairs_d), np.asarray(pairs_o))
    assert np.array_equal(wd.x0_bar[..., :4], wo.x0_bar[..., :4])
    np.testing.assert_allclose(wd.x0_bar, wo.x0_bar, atol=1e-12, rtol=0)
    assert np.array_equal(wd.plane_off, wo.plane_off) and np.array_equal(wd.planes["t"], wo.planes["t"])
    np.testing.assert_allclose(wd.planes["c"], wo.planes["c"], atol=1e-12, rtol=1e-14)


@pytest.mark.parametrize("seed", [1, 4, 5, 7, 8, 11])
def test_device_bridge_equals_host_bridge_on_random_worlds(gpu_handle, veh_parm, seed):
    """The random worlds of tests/test_bridge.py (a few pairs to 14655, single-state paths, overlapping rectangles, horizons off every
    block size): the device bridge, the host bridge on the library's threads and their batched forms return the same bytes."""
    from tests.test_bridge import _random_paths
    from csdotrajectoryplanning_amd.solver import interpolate_and_planes, interpolate_and_planes_batch_host
    veh, parm = veh_parm
    rng = np.random.default_rng(100 + seed)
    Na = [1, 2, 3, 7, 16, 25, 40, 64, 5, 33, 12, 50][seed]
    max_moves = [3, 1, 12, 30, 5, 21, 40, 11, 60, 17, 6, 26][seed]
    crowd = [0.1, 0.02, 0.05, 0.4, 0.1, 0.15, 0.3, 0.2, 0.05, 0.1, 0.01, 0.25][seed]
    st, ac, po, G = _random_paths(rng, Na, 100.0, max_moves, crowd)
    obs = np.zeros((0, 3))
    host = interpolate_and_planes(st, ac, po, G, veh, parm, 100.0, 100.0, obs)
    dev = gpu_handle.interpolate_and_planes(st, ac, po, G, veh, parm, 100.0, 100.0, obs)
    _same_bridge(host, dev)
    item = (st, ac, po, G, 100.0, 100.0, obs)
    for batched in (interpolate_and_planes_batch_host([item, item], veh, parm), gpu_handle.interpolate_and_planes_batch([item, item], veh, parm)):
        for b in batched:
            _same_bridge(host, b)


def test_device_bridge_1024_agents_in_one_world(gpu_handle):
    """The case K0 exists for: 1024 vehicles in ONE world, about 10^8 (t, i, j) candidates.  The coarse paths of
    21 instances are laid over each other (not a solvable planning instance: a stress input with very many neighbours)."""
    from csdotrajectoryplanning_amd import synth
    from csdotrajectoryplanning_amd.solver import interpolate_and_planes
    S, A, G = [], [], []
    w0 = None
    for k in range(21):
        w, (st, ac, po, g) = _paths(k)
        w0 = w0 or w
        for a in range(len(po) - 1):
            S.append(st[po[a]:po[a + 1]])
            A.append(ac[po[a] - a:po[a + 1] - a - 1])
        G.append(g)
    S, A, G = S[:1024], A[:1024], np.concatenate(G)[:1024]
    st, ac, po = synth.pack_paths(S, A)
    host = interpolate_and_planes(st, ac, po, G, w0.veh, w0.parm, w0.dimx, w0.dimy, w0.obstacles)
    dev = gpu_handle.interpolate_and_planes(st, ac, po, G, w0.veh, w0.parm, w0.dimx, w0.dimy, w0.obstacles)
    assert host[0].Na == 1024 and len(host[1]) > 100000
    _same_bridge(host, dev)


def test_device_bridge_without_neighbours(gpu_handle, veh_parm):
    from csdotrajectoryplanning_amd import synth
    veh, parm = veh_parm
    step = veh.r * veh.deltat
    S = [np.array([[8.0 + step * i, 10.0 + 40.0 * a, 0.0] for i in range(5)]) for a in range(2)]
    A = [np.zeros(4, np.int32) for _ in range(2)]
    st, ac, po = synth.pack_paths(S, A)
    G = np.array([s[-1] for s in S])
    w, pairs, legal = gpu_handle.interpolate_and_planes(st, ac, po, G, veh, parm, 100.0, 100.0, np.zeros((0, 3)))
    assert len(pairs) == 0 and legal == 1 and int(w.plane_off[-1]) == 0 and w.Nt == 13


def test_device_validator_equals_reference_verdicts(gpu_handle, veh_parm):
    """Every rectangle / rectangle and disc / rectangle verdict of the reference's collision_detection.py."""
    veh, _ = veh_parm
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "ref_collision_verdicts.npz"))
    n = len(z["rect_rect"])
    sol = np.zeros((2, n, 6))
    sol[0, :, :3], sol[1, :, :3] = z["rect_a"], z["rect_b"]
    rep = gpu_handle.validate(sol, veh)
    assert rep.vehicle_collisions == int(z["rect_rect"].sum())
    assert rep.first_vehicle_collision == (int(np.argmax(z["rect_rect"])), 0, 1)
    # disc / rectangle: one agent, one timestep and one obstacle per case
    hits = 0
    for k in range(0, 600):
        r = gpu_handle.validate(z["rect_c"][k][None, None, :], veh, z["circle"][k][None, :])
        assert (r.obstacle_collisions == 1) == bool(z["circle_rect"][k])
        hits += r.obstacle_collisions
    assert hits == int(z["circle_rect"][:600].sum())


@pytest.mark.parametrize("which,k", [("map50", 3), ("map100", 1)])
def test_device_validator_equals_numpy_validator_on_results(gpu_handle, which, k):
    from csdotrajectoryplanning_amd import results
    w, _ = _paths(k, which)
    sol = gpu_handle.solve(w).solutions
    for margin in (0.0, 0.3):
        ref = results.validate(sol, w.veh, w.obstacles, w.dimx, w.dimy, margin=margin)
        got = gpu_handle.validate(sol, w.veh, w.obstacles, w.dimx, w.dimy, margin=margin)
        assert (got.vehicle_collisions, got.obstacle_collisions, got.out_of_map) == \
               (ref.vehicle_collisions, ref.obstacle_collisions, ref.out_of_map)
        assert got.first_vehicle_collision == ref.first_vehicle_collision
        assert got.first_obstacle_collision == ref.first_obstacle_collision
        assert abs(got.min_obstacle_clearance - ref.min_obstacle_clearance) < 1e-9


def test_device_validator_between_states_equals_the_references_frames(gpu_handle, veh_parm):
    """csdo_validate_frames against the per-frame verdicts of scripts/visualize.py:219-247 with getState's interpolation
    (fixture generated by importing the reference's script: tests/golden/make_ref_fixtures.py)."""
    from csdotrajectoryplanning_amd import results
    veh, _ = veh_parm
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "ref_substep_frames.npz"))
    for S in (1, 3, 10):
        vh, oh = z["vehicle_hits_%d" % S], z["obstacle_hits_%d" % S]
        got = gpu_handle.validate(z["solutions"], veh, z["obstacles"], frames_per_move=S)
        assert (got.vehicle_collisions, got.obstacle_collisions) == (len(vh), len(oh))
        assert got.first_vehicle_collision == tuple(int(v) for v in min(map(tuple, vh)))
        assert got.first_obstacle_collision == tuple(int(v) for v in min(map(tuple, oh)))
        ref = results.validate(z["solutions"], veh, z["obstacles"], frames_per_move=S)
        assert abs(got.min_obstacle_clearance - ref.min_obstacle_clearance) < 1e-9


@pytest.mark.parametrize("which,k", [("map50", 3), ("map100", 1)])
def test_device_validator_between_states_on_results(gpu_handle, which, k):
    """A DO-phase result checked at 5 frames per move: device == numpy, and no more collisions appear between the states
    than the timestep check plus what a neighbouring state already shows (the trajectories are 0.7 m per step)."""
    from csdotrajectoryplanning_amd import results
    w, _ = _paths(k, which)
    sol = gpu_handle.solve(w).solutions
    ref = results.validate(sol, w.veh, w.obstacles, w.dimx, w.dimy, frames_per_move=5)
    got = gpu_handle.validate(sol, w.veh, w.obstacles, w.dimx, w.dimy, frames_per_move=5)
    assert (got.vehicle_collisions, got.obstacle_collisions, got.out_of_map) == \
           (ref.vehicle_collisions, ref.obstacle_collisions, ref.out_of_map)
    assert got.first_vehicle_collision == ref.first_vehicle_collision
    assert got.first_obstacle_collision == ref.first_obstacle_collision
    assert abs(got.min_obstacle_clearance - ref.min_obstacle_clearance) < 1e-9
    coarse = gpu_handle.validate(sol, w.veh, w.obstacles, w.dimx, w.dimy)
    assert got.min_obstacle_clearance <= coarse.min_obstacle_clearance + 1e-9


def test_batched_device_bridge_equals_host_bridge_world_by_world(gpu_handle):
    """csdo_preprocess_device_batch: a mixed batch (both maps, different horizons, a world without neighbours) in one call."""
    from csdotrajectoryplanning_amd import synth
    from csdotrajectoryplanning_amd.solver import interpolate_and_planes
    items, hosts = [], []
    veh = parm = None
    for which, k in (("map100", 0), ("map50", 1), ("map100", 7), ("map50", 4), ("map100", 2)):
        w, (st, ac, po, G) = _paths(k, which)
        veh, parm = w.veh, w.parm
        items.append((st, ac, po, G, w.dimx, w.dimy, w.obstacles))
        hosts.append(interpolate_and_planes(st, ac, po, G, w.veh, w.parm, w.dimx, w.dimy, w.obstacles))
    step = veh.r * veh.deltat
    S = [np.array([[8.0 + step * i, 10.0 + 40.0 * a, 0.0] for i in range(5)]) for a in range(2)]
    A = [np.zeros(4, np.int32) for _ in range(2)]
    st, ac, po = synth.pack_paths(S, A)
    G = np.array([s[-1] for s in S])
    items.append((st, ac, po, G, 100.0, 100.0, np.zeros((0, 3))))
    hosts.append(interpolate_and_planes(st, ac, po, G, veh, parm, 100.0, 100.0, np.zeros((0, 3))))
    for _ in range(2):                            # twice: the second call reuses every device / staging buffer
        got = gpu_handle.interpolate_and_planes_batch(items, veh, parm)
        assert len(got) == len(hosts)
        for a, b in zip(hosts, got):
            _same_bridge(a, b)
    assert len(got[-1][1]) == 0 and got[-1][2] == 1
    # a batch of one equals the single-world call
    _same_bridge(hosts[1], gpu_handle.interpolate_and_planes_batch(items[1:2], veh, parm)[0])
