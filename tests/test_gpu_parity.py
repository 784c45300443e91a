"""Parity tests proper: the HIP path, called through the C ABI (libcsdo_hip.so), against the oracle on the same inputs.
Bar (tests/parity.py): |dx| <= 1e-4 on every state/control of every agent and timestep, identical SQP / ADMM iteration
counts and OSQP status codes; agents whose corridor boxes flipped a 0.1 m growth step are held to 2e-2."""
import ctypes as C

import numpy as np
import pytest

from tests import helpers, parity

pytestmark = pytest.mark.gpu
from tests.test_program_serial import STAND_IN_MAP100_LOOSE   # noqa: E402


def _check(ref, got, loose=()):
    """loose: the agents (indices) known to end between TOL and LOOSE_TOL of the oracle - exactly those, no others (the HIP build
    returns the bits of the lane-serial host build, so the list is computed on the CPU: tests/test_program_serial.py)."""
    c = parity.compare(ref, got)
    assert c["counts_equal"], (ref.sqp_iters, got.sqp_iters, ref.admm_iters, got.admm_iters, ref.last_status,
                               got.last_status)
    # no agent above LOOSE_TOL (corridor-flipped agents: CORRIDOR_FLIP_TOL); between TOL and it exactly the agents named
    too_far = [b for b in c["bad"] if b[1] > parity.LOOSE_TOL]
    assert not too_far, too_far
    assert sorted(b[0] for b in c["bad"]) == sorted(loose), c["bad"]
    assert ref.solver_status == got.solver_status and ref.initial_static_legal == got.initial_static_legal
    return c


@pytest.mark.parametrize("name", ["map50_agents0to5.npz", "map50_agents15to17.npz", "map100_agents0to3.npz"])
def test_gpu_matches_golden_and_oracle(gpu_handle, oracle, veh_parm, name):
    veh, parm = veh_parm
    world, z = helpers.load_golden(name, veh, parm)
    got = gpu_handle.solve(world)
    assert np.array_equal(got.sqp_iters, z["sqp_iters"]) and np.array_equal(got.last_status, z["last_status"])
    assert np.array_equal(got.admm_iters, z["admm_iters"])
    c = _check(oracle.solve(world, 2), got)
    if c["n_flipped"] == 0:
        np.testing.assert_allclose(got.solutions, z["solutions"], atol=parity.TOL, rtol=0)


def test_gpu_full_map50_agents25(gpu_handle, oracle, world_map50):
    world, info = world_map50
    _check(oracle.solve(world, 8), gpu_handle.solve(world))


def test_gpu_full_map100_agents50(gpu_handle, oracle, emu, world_map100):
    world, info = world_map100
    got = gpu_handle.solve(world)
    ser = emu.solve(world)
    assert np.array_equal(got.solutions, ser.solutions) and np.array_equal(got.corridors, ser.corridors)      # the bits of the lane-serial build
    assert np.array_equal(got.admm_iters, ser.admm_iters) and np.array_equal(got.sqp_iters, ser.sqp_iters)
    # agents 13 and 16 of the stand-in world end 1.5e-4 and 1.4e-4 from the oracle, boxes unchanged, agent 6 1.6e-3 with a flipped
    # growth step (the oracle moves agent 6 by 6.1e-4 and agent 16 by 1.6e-4 when it is built with fused multiply-adds); every other
    # agent within 1e-4.
    # tests/test_program_serial.py::test_stand_in_world_chain_sensitive_agents holds the same list on the CPU.
    _check(oracle.solve(world, 8), got, loose=STAND_IN_MAP100_LOOSE)
    # size-independent properties at the full size
    x0 = world.x0_bar
    ok = got.last_status == 1
    # the trust-region / start-goal rows hold up to OSQP's primal tolerance (eps_abs + eps_rel * max(|Ax|, |z|)) of
    # the LAST QP only relative to that QP's own bounds; 0.5 m is a loose sanity bound on a 100 m map
    assert np.all(np.abs(got.solutions[ok][:, :, :2] - x0[ok][:, :, :2]) <= world.parm.r_trust + 0.5)
    np.testing.assert_allclose(got.solutions[ok][:, [0, -1], :3], x0[ok][:, [0, -1], :3], atol=0.5)
    assert np.all(got.solutions[:, -1, 4:] == 0)
    dt = world.parm.dt
    s = got.solutions[ok]
    kin = s[:, :-1, 0] + dt * s[:, :-1, 4] * np.cos(s[:, :-1, 2]) - s[:, 1:, 0]
    assert np.mean(kin ** 2) < 1e-2                                      # isFeasible's kinematic threshold


def test_gpu_matches_lane_serial_build_of_the_same_program(gpu_handle, emu, veh_parm):
    """Same source, once as HIP device code and once lane-serially on the host, the same trigonometry in both (csrc/csdo_math.h):
    the same bits, in every residency mode the 512-thread class can be put into."""
    veh, parm = veh_parm
    for name in ["map50_agents0to5.npz", "map100_agents0to3.npz", "map50_agents15to17.npz"]:
        world, _ = helpers.load_golden(name, veh, parm)
        e = emu.solve(world)
        for min_mode in (0, 1, 2):
            gpu_handle.set_min_residency_mode(min_mode)
            try:
                g = gpu_handle.solve(world)
            finally:
                gpu_handle.set_min_residency_mode(0)
            assert np.array_equal(g.sqp_iters, e.sqp_iters) and np.array_equal(g.admm_iters, e.admm_iters)
            assert np.array_equal(g.last_status, e.last_status)
            assert np.array_equal(g.solutions, e.solutions) and np.array_equal(g.corridors, e.corridors), (name, min_mode)


def test_gpu_is_deterministic_and_handle_is_reusable(gpu_handle, veh_parm):
    veh, parm = veh_parm
    w1, _ = helpers.load_golden("map50_agents15to17.npz", veh, parm)
    w2, _ = helpers.load_golden("map100_agents0to3.npz", veh, parm)
    a = gpu_handle.solve(w1)
    gpu_handle.solve(w2)
    b = gpu_handle.solve(w1)
    assert np.array_equal(a.solutions, b.solutions) and np.array_equal(a.corridors, b.corridors)


def test_gpu_lds_residency_modes_are_bit_identical(gpu_handle, veh_parm):
    """Where an ADMM block keeps its state only changes where the same doubles are read from: with the inter-vehicle rows'
    duals / slacks in the workspace instead of LDS (knob 1) and with the pivot inverse read from the workspace as well
    (knob 2, the 512-thread class's mode 1) the results are bit-identical.  (Modes 2, 3: the long-horizon tests.)"""
    veh, parm = veh_parm
    w1, _ = helpers.load_golden("map50_agents15to17.npz", veh, parm)
    w2, _ = helpers.load_golden("map100_agents0to3.npz", veh, parm)      # Nt = 169: 512-thread class
    ref = gpu_handle.solve_batch([w1, w2])
    assert all(g["residency_mode"] == 0 for g in gpu_handle.launch_groups())
    try:
        for knob in (1, 2):
            gpu_handle.set_min_residency_mode(knob)
            got = gpu_handle.solve_batch([w1, w2])
            groups = gpu_handle.launch_groups()
            assert sum(g["n_agents"] for g in groups) == w1.Na + w2.Na
            assert all(g["residency_mode"] == (1 if (knob == 2 and g["threads"] == 512) else 0) for g in groups)
            for r, g in zip(ref, got):
                assert np.array_equal(r.solutions, g.solutions) and np.array_equal(r.corridors, g.corridors)
                assert np.array_equal(r.admm_iters, g.admm_iters) and np.array_equal(r.last_status, g.last_status)
    finally:
        gpu_handle.set_min_residency_mode(0)


def test_gpu_wide_class_two_forms_of_the_solve_agree(gpu_handle, emu, oracle, veh_parm):
    """The 768-thread class runs the pair-split solve (mode 2: the lane's second block in LDS, its level-1 block fetched from the
    workspace in front of each use, the rows' coefficients streamed) while that fits the LDS, else the one-lane form (mode 3, knob 3),
    which absorbs the partials of the nodes at multiples of 64 in another order: on the first QP the two forms and the lane-serial
    build of the same program agree to rounding (same iteration count); both full chains meet the oracle bar.  Two lanes 3.5 m apart, so that
    the inter-vehicle rows take part."""
    from csdotrajectoryplanning_amd.abi import QpParm
    from csdotrajectoryplanning_amd.problem import World
    veh, parm = veh_parm
    w = helpers.straight_line_world(veh, parm, Na=2, L=90, dim=600.0, spacing=3.5)
    assert w.Nt == 271 and w.plane_off[-1] > 0
    p1 = QpParm.from_buffer_copy(bytes(w.parm))
    p1.max_iter = 1.0
    w1 = World(w.x0_bar, w.plane_off, w.planes, w.dimx, w.dimy, w.obstacles, w.veh, p1)
    ref, ref1 = gpu_handle.solve_batch([w])[0], gpu_handle.solve_batch([w1])[0]
    assert [(g["threads"], g["residency_mode"]) for g in gpu_handle.launch_groups()] == [(768, 2)]
    ser1 = emu.solve(w1, 2)                 # (device libm differs from glibc by ulps: to rounding, not to the bit)
    assert np.array_equal(ref1.admm_iters, ser1.admm_iters) and np.abs(ref1.solutions - ser1.solutions).max() < 1e-9
    oref = oracle.solve(w, 1)
    _check(oref, ref)
    try:
        gpu_handle.set_min_residency_mode(3)
        got1 = gpu_handle.solve_batch([w1])[0]
        assert [(g["threads"], g["residency_mode"]) for g in gpu_handle.launch_groups()] == [(768, 3)]
        assert np.array_equal(ref1.admm_iters, got1.admm_iters) and np.array_equal(ref1.last_status, got1.last_status)
        assert np.abs(ref1.solutions - got1.solutions).max() < 1e-9
        _check(oref, gpu_handle.solve_batch([w])[0])
    finally:
        gpu_handle.set_min_residency_mode(0)


def test_gpu_batch_equals_separate_solves(gpu_handle, veh_parm):
    veh, parm = veh_parm
    w1, _ = helpers.load_golden("map50_agents15to17.npz", veh, parm)
    w2, _ = helpers.load_golden("map100_agents0to3.npz", veh, parm)
    both = gpu_handle.solve_batch([w1, w2])
    for w, b in zip((w1, w2), both):
        s = gpu_handle.solve(w)
        assert np.array_equal(s.solutions, b.solutions) and np.array_equal(s.admm_iters, b.admm_iters)


def test_gpu_split_phase_equals_one_shot(gpu_handle, veh_parm):
    veh, parm = veh_parm
    w, _ = helpers.load_golden("map50_agents15to17.npz", veh, parm)
    one = gpu_handle.solve(w)
    gpu_handle.upload([w])
    t1 = gpu_handle.run()
    t2 = gpu_handle.run()                                                # re-running on resident inputs is idempotent
    got = gpu_handle.download()[0]
    assert t1 > 0 and t2 > 0
    assert np.array_equal(one.solutions, got.solutions) and np.array_equal(one.sqp_iters, got.sqp_iters)


def test_gpu_edge_cases(gpu_handle, oracle, veh_parm):
    veh, parm = veh_parm
    from csdotrajectoryplanning_amd.problem import World
    w = helpers.straight_line_world(veh, parm, Na=1, L=2)                # one agent, no obstacles, no planes
    _check(oracle.solve(w, 1), gpu_handle.solve(w))
    w = helpers.straight_line_world(veh, parm, Na=2, L=5, spacing=3.5, obstacles=[[20.0, 16.5, 0.8]])
    _check(oracle.solve(w, 1), gpu_handle.solve(w))
    w2 = World(w.x0_bar[:, :2].copy(), np.zeros(3, np.int32), w.planes[:0], w.dimx, w.dimy, w.obstacles, veh, parm)
    _check(oracle.solve(w2, 1), gpu_handle.solve(w2))                    # shortest horizon Nt = 2


@pytest.mark.parametrize("L,Nt,threads,mode", [(90, 271, 768, 2), (120, 361, 768, 2), (126, 379, 768, 3), (140, 421, 1024, 3)])
def test_gpu_long_horizons_use_the_wide_kernels(gpu_handle, oracle, veh_parm, L, Nt, threads, mode):
    """Horizons beyond 256 timesteps: 768 threads (168 registers per lane) up to 384 - the pair-split solve with the lane's second
    block in LDS (mode 2) while that fits, about 370 timesteps without obstacles -, 1024 threads (128 registers) up to 512."""
    veh, parm = veh_parm
    w = helpers.straight_line_world(veh, parm, Na=1, L=L, dim=600.0)
    assert w.Nt == Nt
    _check(oracle.solve(w, 1), gpu_handle.solve(w))
    gpu_handle.upload([w])
    assert [(g["threads"], g["residency_mode"]) for g in gpu_handle.launch_groups()] == [(threads, mode)]


@pytest.mark.parametrize("Nt", [385, 511, 512])
def test_gpu_longest_horizons_with_inter_vehicle_rows(gpu_handle, oracle, veh_parm, Nt):
    """The ends of the 1024-thread class with separating planes at work: its first horizon (385), the longest one the bridge can
    produce (511 = 3 * 170 + 1) and CSDO_MAX_NT itself (512: the 514-step world cut to 512 timesteps, planes beyond dropped)."""
    from csdotrajectoryplanning_amd import abi
    from csdotrajectoryplanning_amd.problem import World
    veh, parm = veh_parm
    L = {385: 128, 511: 170, 512: 171}[Nt]
    w = helpers.straight_line_world(veh, parm, Na=2, L=L, dim=700.0, spacing=3.5)
    if Nt == 512:
        assert w.Nt == 514 and abi.CSDO_MAX_NT == 512
        keep, off = [], [0]
        for a in range(w.Na):
            p = w.planes[w.plane_off[a]:w.plane_off[a + 1]]
            keep.append(p[p["t"] < 512])
            off.append(off[-1] + len(keep[-1]))
        w = World(np.ascontiguousarray(w.x0_bar[:, :512]), np.asarray(off, np.int32), np.concatenate(keep), w.dimx, w.dimy,
                  w.obstacles, veh, parm)
    assert w.Nt == Nt and w.plane_off[-1] > 2 * (Nt - 10)
    _check(oracle.solve(w, 2), gpu_handle.solve(w))
    assert [(g["threads"], g["residency_mode"]) for g in gpu_handle.launch_groups()] == [(1024, 3)]


def test_gpu_fixed_corridor_mode_is_tight(gpu_handle, oracle, veh_parm):
    from csdotrajectoryplanning_amd import config
    veh, _ = veh_parm
    parm = config.qp_parm_from_config({"fixed_corridor": True})
    world, _ = helpers.load_golden("map50_agents15to17.npz", veh, parm)
    ref, got = oracle.solve(world, 1), gpu_handle.solve(world)
    assert np.array_equal(ref.sqp_iters, got.sqp_iters) and np.array_equal(ref.admm_iters, got.admm_iters)
    np.testing.assert_allclose(got.solutions, ref.solutions, atol=1e-6, rtol=0)


def test_gpu_corridor_boxes_bit_exact(gpu_handle, oracle, veh_parm):
    veh, parm = veh_parm
    world, _ = helpers.load_golden("map100_agents0to3.npz", veh, parm)
    rng = np.random.default_rng(3)
    pts = np.concatenate([rng.uniform(-1, 101, (4000, 2)), world.x0_bar[0, :, :2],
                          world.obstacles[:, :2] + rng.uniform(-1.5, 1.5, (len(world.obstacles), 2))])
    bo, so = oracle.generate_boxes(pts, world.obstacles, world.dimx, world.dimy, veh)
    bg, sg = gpu_handle.generate_boxes(pts, world.obstacles, world.dimx, world.dimy, veh)
    assert np.array_equal(so, sg)
    legal = (so >> 1) != 2
    assert np.array_equal(bo[legal], bg[legal])                          # pure compare/add arithmetic: bit exact
    np.testing.assert_allclose(bo[~legal], bg[~legal], atol=1e-9, rtol=0)  # repair path calls atan2/cos/sin
    bx, sx = oracle.generate_boxes(pts, world.obstacles, world.dimx, world.dimy, veh, variant="xm")   # ... the program's own: bit exact everywhere
    assert np.array_equal(sx, sg) and np.array_equal(bx, bg)
    # empty obstacle list and empty point list
    b, s = gpu_handle.generate_boxes([[50.0, 50.0]], np.zeros((0, 3)), 100.0, 100.0, veh)
    np.testing.assert_allclose(b[0], [39.9, 39.9, 60.1, 60.1], atol=1e-9)
    b, s = gpu_handle.generate_boxes(np.zeros((0, 2)), np.zeros((0, 3)), 100.0, 100.0, veh)
    assert b.shape == (0, 4)


def test_gpu_error_codes(gpu_handle, veh_parm):
    from csdotrajectoryplanning_amd import _lib, abi
    from csdotrajectoryplanning_amd.problem import Solution, World
    veh, parm = veh_parm
    Nt = abi.CSDO_MAX_NT + 1
    w = World(np.zeros((1, Nt, 6)), np.zeros(2, np.int32), np.zeros(0, abi.PLANE_DTYPE), 50.0, 50.0, np.zeros((0, 3)),
              veh, parm)
    with pytest.raises(_lib.CsdoError, match="limit"):
        gpu_handle.solve(w)
    # the reference has no cap on the horizon (sqp/inter_agent_cons.cc:320-325); here Nt <= CSDO_MAX_NT, and the world beyond it is named
    ok = World(np.zeros((1, 8, 6)), np.zeros(2, np.int32), np.zeros(0, abi.PLANE_DTYPE), 50.0, 50.0, np.zeros((0, 3)), veh, parm)
    with pytest.raises(_lib.CsdoError, match="limit"):
        gpu_handle.upload([ok, ok, w, ok])
    assert gpu_handle.last_limit() == (2, 0, 0)
    sol = Solution.allocate(1, 4)
    assert _lib.lib().csdo_dsqp_solve(gpu_handle._h, None, C.byref(sol._c)) == abi.CSDO_EINVAL
    w1 = World(np.zeros((1, 1, 6)), np.zeros(2, np.int32), np.zeros(0, abi.PLANE_DTYPE), 50.0, 50.0, np.zeros((0, 3)),
               veh, parm)
    p = w1.c_problem()
    assert _lib.lib().csdo_dsqp_solve(gpu_handle._h, C.byref(p), C.byref(sol._c)) == abi.CSDO_EINVAL  # Nt < 2


@pytest.mark.parametrize("Nt,per_lane", [(100, 30), (200, 30), (300, 30)])
def test_obstacle_count_at_which_a_world_no_longer_fits(gpu_handle, veh_parm, Nt, per_lane):
    """CSDO_ELIMIT at upload: the obstacle list is staged in LDS beside the per-timestep arrays of the leanest residency mode the
    agent can run in - mode 3 of the wide classes, 30 doubles per timestep (round 5: an agent of the 512-thread class whose obstacles
    do not fit beside its 52-double layout moves to the 768-thread class instead of being turned away) -, the tail (1472 doubles) and
    nothing else.  The largest obstacle count that fits is pinned here for three horizons; one world beyond it rejects the whole
    batch, nothing is launched, and csdo_dsqp_last_limit says which world it was."""
    from csdotrajectoryplanning_amd import _lib
    from csdotrajectoryplanning_amd.problem import World
    veh, parm = veh_parm
    st = (Nt + 1) & ~1
    cap = 160 * 1024 - 64
    fixed = (per_lane * st + 32 + 72 + 36 * 38) * 8
    n_fit = (cap - fixed) // 8 // 3 + 2
    while ((3 * n_fit + 1) & ~1) * 8 + fixed > cap:        # (the staged list is padded to an even number of doubles)
        n_fit -= 1
    assert {100: 5333, 200: 4333, 300: 3333}[Nt] == n_fit      # the capability, in numbers (round 4: 4600, 2866, 3333)

    def world(n_obs):
        x0 = np.zeros((1, Nt, 6))
        x0[0, :, 0] = 5.0 + 0.3 * np.arange(Nt)
        x0[0, :, 1] = 5.0
        x0[0, :-1, 4] = 0.3 / parm.dt
        obs = np.column_stack([np.full(n_obs, 90.0), np.full(n_obs, 190.0), np.full(n_obs, 0.1)])
        return World(x0, np.zeros(2, np.int32), np.zeros(0, abi_plane()), 100.0 + 0.3 * Nt, 200.0, obs, veh, parm)

    gpu_handle.upload([world(10), world(n_fit)])            # fits: the upload succeeds
    assert gpu_handle.last_limit()[0] == -1
    with pytest.raises(_lib.CsdoError, match="limit"):
        gpu_handle.upload([world(10), world(n_fit + 2), world(5)])
    w_, a_, need = gpu_handle.last_limit()
    assert (w_, a_) == (1, 0) and need > cap


def abi_plane():
    from csdotrajectoryplanning_amd import abi
    return abi.PLANE_DTYPE


def test_solver_dsqp_mirror(oracle, veh_parm):
    """The host-side mirror keeps the reference's constructor-solves shape (sqp/dsqp_solver.h:24-47)."""
    from csdotrajectoryplanning_amd.solver import SolverDSQP
    veh, parm = veh_parm
    world, z = helpers.load_golden("map50_agents15to17.npz", veh, parm)
    s = SolverDSQP(world.x0_bar, (world.plane_off, world.planes), world.dimx, world.dimy, world.obstacles, parm, veh,
                   logger_level=0)
    assert s.getSolverStatus() == int(z["solver_status"])
    assert s.get_initial_static_legal() == bool(z["initial_static_legal"])
    assert np.array_equal(s.num_iterations, z["sqp_iters"]) and s.getMaxOfRuntimes() > 0
    np.testing.assert_allclose(s.solutions, z["solutions"], atol=parity.CORRIDOR_FLIP_TOL)


@pytest.mark.parametrize("n_obs", [40, 300])
def test_gpu_corridor_boxes_bit_exact_in_dense_obstacle_fields(gpu_handle, oracle, veh_parm, n_obs):
    """> 8 obstacles near a seed (register slots of grow_box) and > 256 obstacles (its cull mask): same boxes."""
    veh, _ = veh_parm
    rng = np.random.default_rng(11 + n_obs)
    side = 30.0
    obstacles = np.column_stack([rng.uniform(2, side - 2, n_obs), rng.uniform(2, side - 2, n_obs),
                                 rng.uniform(0.2, 0.6, n_obs)])
    pts = rng.uniform(0, side, (300, 2))
    bo, so = oracle.generate_boxes(pts, obstacles, side, side, veh)
    bg, sg = gpu_handle.generate_boxes(pts, obstacles, side, side, veh)
    assert np.array_equal(so, sg)
    legal = (so >> 1) != 2
    assert np.array_equal(bo[legal], bg[legal])
    np.testing.assert_allclose(bo[~legal], bg[~legal], atol=1e-9, rtol=0)
    bx, sx = oracle.generate_boxes(pts, obstacles, side, side, veh, variant="xm")
    assert np.array_equal(sx, sg) and np.array_equal(bx, bg)


def test_gpu_four_launch_groups_in_one_batch(gpu_handle, veh_parm):
    """256-thread, 512-thread (modes 0 and 1) and 768-thread (mode 2) agents in ONE batch: more launch groups than the handle has
    streams for second launches (capi.hip: `ng > 3` - every group then asks for all its workgroups at once and groups beyond the
    fourth share a stream).  Every world's result equals its solve alone, bit for bit."""
    from csdotrajectoryplanning_amd import workloads
    veh, parm = veh_parm
    room = [workloads.build_job(j)[0] for j in workloads.workload_jobs("room50")]
    short, _ = helpers.load_golden("map50_agents15to17.npz", veh, parm)
    mid, _ = helpers.load_golden("map100_agents0to3.npz", veh, parm)
    batch = [short, mid] + room
    got = gpu_handle.solve_batch(batch)
    groups = gpu_handle.launch_groups()
    kinds = {(g["threads"], g["residency_mode"]) for g in groups}
    assert len(groups) >= 4 and {(256, 0), (512, 0), (512, 1), (768, 2)} <= kinds, groups
    assert sum(g["n_agents"] for g in groups) == sum(w.Na for w in batch) and all(g["seconds"] > 0 for g in groups)
    for w, b in zip(batch, got):
        s = gpu_handle.solve(w)
        assert np.array_equal(s.solutions, b.solutions) and np.array_equal(s.corridors, b.corridors)
        assert np.array_equal(s.admm_iters, b.admm_iters) and np.array_equal(s.last_status, b.last_status)


def test_gpu_obstacle_heavy_short_horizon_runs_in_the_wide_class(gpu_handle, oracle, veh_parm):
    """5000 obstacles beside 91 timesteps: too many for the 512-thread class's leanest layout - the agents run in the 768-thread
    class (lean mode 3) and meet the oracle's first-QP bar; with the far-away obstacles removed the same world runs in its usual
    class and returns the same trajectories to 1e-9 (the one-lane form of the solve against the pair-split one)."""
    from csdotrajectoryplanning_amd.abi import QpParm
    from csdotrajectoryplanning_amd.problem import World
    veh, parm = veh_parm
    w, _ = helpers.load_golden("map50_agents15to17.npz", veh, parm)
    p1 = QpParm.from_buffer_copy(bytes(parm))
    p1.max_iter = 1.0
    rng = np.random.default_rng(2)
    far = np.column_stack([rng.uniform(300, 400, 5000), rng.uniform(300, 400, 5000), np.full(5000, 0.5)])
    obs = np.vstack([w.obstacles, far])
    heavy = World(w.x0_bar, w.plane_off, w.planes, 500.0, 500.0, obs, veh, p1)
    plain = World(w.x0_bar, w.plane_off, w.planes, 500.0, 500.0, w.obstacles, veh, p1)
    got = gpu_handle.solve(heavy)
    groups = gpu_handle.launch_groups()
    assert [(g["threads"], g["residency_mode"]) for g in groups] == [(768, 3)], groups
    ref = oracle.solve(heavy, 3)
    assert np.array_equal(got.admm_iters, ref.admm_iters) and np.array_equal(got.last_status, ref.last_status)
    assert np.abs(got.solutions - ref.solutions).max() <= 1e-5
    usual = gpu_handle.solve(plain)
    assert np.array_equal(got.admm_iters, usual.admm_iters) and np.abs(got.solutions - usual.solutions).max() <= 1e-9
    assert np.abs(got.corridors - usual.corridors).max() <= 1e-8      # (the boxes returned are grown at the QP's solution: they follow it)


def test_gpu_mixed_batch_launch_groups(gpu_handle, veh_parm):
    """A batch whose worlds fall into different kernel classes (Nt = 91: 256-thread workgroups, Nt = 169: 512) is split
    into concurrent launch groups; every agent is in exactly one and the results equal separate solves."""
    veh, parm = veh_parm
    w1, _ = helpers.load_golden("map50_agents15to17.npz", veh, parm)
    w2, _ = helpers.load_golden("map100_agents0to3.npz", veh, parm)
    gpu_handle.upload([w1, w2, w1])
    t = gpu_handle.run()
    groups, of = gpu_handle.launch_groups(), gpu_handle.agent_groups()
    got = gpu_handle.download()
    assert len(groups) == 2 and sorted(g["threads"] for g in groups) == [256, 512]
    assert sum(g["n_agents"] for g in groups) == 2 * w1.Na + w2.Na == len(of)
    assert all(0 < g["seconds"] <= t * 1.01 for g in groups)
    g256 = [i for i, g in enumerate(groups) if g["threads"] == 256][0]
    assert np.all(of[:w1.Na] == g256) and np.all(of[w1.Na:w1.Na + w2.Na] == 1 - g256) and np.all(of[-w1.Na:] == g256)
    for w, b in zip((w1, w2, w1), got):
        s = gpu_handle.solve(w)
        assert np.array_equal(s.solutions, b.solutions) and np.array_equal(s.admm_iters, b.admm_iters)
        assert b.agent_seconds.shape == (w.Na,) and np.all(b.agent_seconds > 0)
