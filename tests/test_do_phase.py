"""csdo_do_phase: the DO phase of a batch of worlds as ONE library call (coarse paths in, trajectories out).  CPU part: the chunking
rule of the C++ entry is the rule of solver.stream_cuts, the horizon is the bridge's.  GPU part: its results are the bits of
csdo_preprocess + csdo_dsqp_solve_batch and of the Python form (DsqpHandle.do_phase_stream), streamed and in one launch."""
import numpy as np
import pytest


def test_the_chunking_rule_is_the_one_of_stream_cuts():
    from csdotrajectoryplanning_amd import abi
    from csdotrajectoryplanning_amd._lib import lib
    from csdotrajectoryplanning_amd.solver import stream_cuts
    rng = np.random.default_rng(5)
    cases = [[50] * n for n in (1, 2, 3, 4, 5, 7, 12, 20, 59, 60, 61, 200)] + [[100] * 12, [25] * 60, [300], [10] * 3, [240, 10, 10, 10]]
    cases += [list(rng.integers(1, 120, size=int(n))) for n in rng.integers(1, 90, size=40)]
    for sizes in cases:
        for m in (230, 0, 1000):
            want = stream_cuts(sizes, min_first_agents=m)
            got = np.full(5, -7, np.int32)
            n_chunks = lib().csdo_do_phase_cuts(abi.as_int32_p(np.asarray(sizes, np.int32)), len(sizes), m, abi.as_int32_p(got))
            assert n_chunks == len(want) - 1 and list(got[:n_chunks + 1]) == list(want) and all(got[n_chunks + 1:] == -1), (sizes, m, want, got)


def test_the_horizon_is_the_bridges(oracle):
    import ctypes as C
    from csdotrajectoryplanning_amd import abi, workloads
    from csdotrajectoryplanning_amd._lib import lib
    for name in ("map50", "map100", "room50"):
        for j in workloads.workload_jobs(name, 3):
            w, info = workloads.build_job(j)
            po = np.ascontiguousarray(info["paths"][2], np.int32)
            if w.Na != len(po) - 1:
                continue
            assert lib().csdo_do_phase_horizon(abi.as_int32_p(po), len(po) - 1, C.byref(w.parm)) == w.Nt
    assert lib().csdo_do_phase_horizon(None, 3, None) < 0


def _same(a, b):
    return (np.array_equal(a.solutions, b.solutions) and np.array_equal(a.corridors, b.corridors) and
            np.array_equal(a.sqp_iters, b.sqp_iters) and np.array_equal(a.admm_iters, b.admm_iters) and
            np.array_equal(a.last_status, b.last_status) and a.solver_status == b.solver_status and
            a.initial_static_legal == b.initial_static_legal)


@pytest.mark.gpu
def test_do_phase_in_one_call_returns_the_bits_of_bridge_plus_batch(gpu_handle):
    from csdotrajectoryplanning_amd import workloads
    from csdotrajectoryplanning_amd._lib import CsdoError
    built = [workloads.build_job(j) for j in workloads.workload_jobs("map100", 8)]
    items = [(*info["paths"], w.dimx, w.dimy, w.obstacles) for w, info in built]
    worlds = [w for w, _ in built]
    ref = gpu_handle.solve_batch(worlds)
    got, tm, legal = gpu_handle.do_phase(items, worlds[0].veh, worlds[0].parm)
    # eight 50-vehicle worlds of one kernel class: the first chunk is enlarged to five worlds (230 agents), a fifth of the job: two chunks
    assert tm["streamed"] and [c["worlds"] for c in tm["chunks"]] == [5, 3], tm
    assert all(_same(g, r) for g, r in zip(got, ref))
    assert list(legal) == [info["initial_inter_legal"] for _, info in built]
    assert 0.0 < tm["first_launch"] < tm["kernels_done"] <= tm["total"] and all(g.t_total == tm["total"] for g in got)
    # a second call writes into the first call's arrays; the Python form returns the same
    got2, tm2, _ = gpu_handle.do_phase(items, worlds[0].veh, worlds[0].parm, out=got)
    assert all(g2 is g for g2, g in zip(got2, got)) and all(_same(g, r) for g, r in zip(got2, ref))
    got_py, tm_py = gpu_handle.do_phase_stream(items, worlds[0].veh, worlds[0].parm)
    assert [c["worlds"] for c in tm_py["chunks"]] == [5, 3] and all(_same(g, r) for g, r in zip(got_py, ref))
    # a job of several kernel classes (map50's horizons are below 129): one launch
    built50 = [workloads.build_job(j) for j in workloads.workload_jobs("map50", 5)]
    items50 = [(*info["paths"], w.dimx, w.dimy, w.obstacles) for w, info in built50]
    ref50 = gpu_handle.solve_batch([w for w, _ in built50])
    got50, tm50, _ = gpu_handle.do_phase(items50, worlds[0].veh, worlds[0].parm)
    assert not tm50["streamed"] and [c["worlds"] for c in tm50["chunks"]] == [5] and all(_same(g, r) for g, r in zip(got50, ref50))
    # one world; and the handle is as usable as before
    got1, tm1, _ = gpu_handle.do_phase(items[2:3], worlds[0].veh, worlds[0].parm)
    assert _same(got1[0], ref[2])
    assert all(_same(g, r) for g, r in zip(gpu_handle.solve_batch(worlds[:2]), ref[:2]))
    # invalid input: an action code out of range in one world - an error code, nothing left in flight
    bad = list(items)
    ac = np.array(bad[6][1], copy=True)
    ac[0] = 9
    bad[6] = (bad[6][0], ac, *bad[6][2:])
    with pytest.raises(CsdoError):
        gpu_handle.do_phase(bad, worlds[0].veh, worlds[0].parm)
    got3, _, _ = gpu_handle.do_phase(items, worlds[0].veh, worlds[0].parm)
    assert all(_same(g, r) for g, r in zip(got3, ref))


@pytest.mark.gpu
def test_do_phase_over_several_devices_deals_out_worlds(gpu_handle):
    """csdo_do_phase on a csdo_dsqp_create_multi handle: contiguous runs of worlds per device, one host thread each, no collective - the
    bits of the single-device call (the one metered GPU twice and three times: two and three independent handles side by side)."""
    from csdotrajectoryplanning_amd import workloads
    from csdotrajectoryplanning_amd.solver import DsqpHandle
    built = [workloads.build_job(j) for j in workloads.workload_jobs("map100", 3)] + [workloads.build_job(j) for j in workloads.workload_jobs("map50", 4)]
    items = [(*info["paths"], w.dimx, w.dimy, w.obstacles) for w, info in built]
    worlds = [w for w, _ in built]
    ref = gpu_handle.solve_batch(worlds)
    for devs in ([0, 0], [0, 0, 0]):
        hm = DsqpHandle(devices=devs)
        try:
            got, tm, legal = hm.do_phase(items, worlds[0].veh, worlds[0].parm)
            assert all(_same(g, r) for g, r in zip(got, ref)), devs
            assert list(legal) == [info["initial_inter_legal"] for _, info in built]
            assert tm["total"] >= tm["kernels_done"] > 0.0
            got1, _, _ = hm.do_phase(items[:1], worlds[0].veh, worlds[0].parm)      # fewer worlds than devices
            assert _same(got1[0], ref[0])
        finally:
            hm.close()


def _c_consumer():
    import os
    import subprocess
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "cpp")
    mk = subprocess.run(["make", "-C", here, "do_phase_main"], capture_output=True, text=True)
    import re
    assert mk.returncode == 0 and not re.search(r"\.[ch]:\d+:\d+: warning", mk.stdout + mk.stderr), mk.stdout + mk.stderr   # (the compiler's, not make's clock-skew notes)
    return subprocess.run([os.path.join(here, "do_phase_main")], capture_output=True, text=True, timeout=300)


def test_a_plain_c_program_binds_the_header_and_fails_loudly_without_a_device():
    """tests/cpp/do_phase_main.c (gcc -std=c99 -pedantic, no warning): the front end runs on the host; without a HIP device
    csdo_dsqp_create returns CSDO_ENODEV - there is no CPU fallback behind the ABI."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a device is present: the GPU test runs the same program to the end")
    run = _c_consumer()
    assert run.returncode == 3 and "front end ok" in run.stdout and "no device: error -2" in run.stdout, (run.stdout, run.stderr)


@pytest.mark.gpu
def test_a_plain_c_program_runs_front_end_and_do_phase():
    run = _c_consumer()
    assert run.returncode == 0 and "copies identical 1" in run.stdout, (run.stdout, run.stderr)


@pytest.mark.gpu
def test_two_handles_run_their_do_phases_at_once(gpu_handle):
    """Two handles, two host threads, one csdo_do_phase each at the same time (they share the library's host threads and the GPU):
    each gets the bits it gets alone."""
    import threading
    from csdotrajectoryplanning_amd import workloads
    from csdotrajectoryplanning_amd.solver import DsqpHandle
    jobs = {"a": ("map100", 6), "b": ("map50", 6)}
    items, refs, veh, parm = {}, {}, None, None
    for key, (name, n) in jobs.items():
        built = [workloads.build_job(j) for j in workloads.workload_jobs(name, n)]
        items[key] = [(*info["paths"], w.dimx, w.dimy, w.obstacles) for w, info in built]
        refs[key] = gpu_handle.solve_batch([w for w, _ in built])
        veh, parm = built[0][0].veh, built[0][0].parm
    other = DsqpHandle(0)
    try:
        for _ in range(3):
            got, err = {}, []

            def run(key, handle):
                try:
                    got[key] = handle.do_phase(items[key], veh, parm)[0]
                except Exception as e:   # noqa: BLE001
                    err.append(e)
            ta = threading.Thread(target=run, args=("a", gpu_handle))
            tb = threading.Thread(target=run, args=("b", other))
            ta.start(); tb.start(); ta.join(); tb.join()
            assert not err, err
            for key in jobs:
                assert all(_same(g, r) for g, r in zip(got[key], refs[key])), key
    finally:
        other.close()
