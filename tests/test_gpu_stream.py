"""Batches in flight (csdo_dsqp_run_async / csdo_dsqp_wait, csdo_dsqp_create_shared) and the streamed DO phase built on them:
whatever the overlap, every world's results are the bits the blocking calls return; handles are independent of each other, also
when two host threads drive them at once (include/csdo_dsqp.h: "different handles are independent")."""
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _items_and_worlds(name, n):
    from csdotrajectoryplanning_amd import solver, workloads
    built = [workloads.build_job(j) for j in workloads.workload_jobs(name, n)]
    items = [(*info["paths"], w.dimx, w.dimy, w.obstacles) for w, info in built]
    worlds = [w for w, _ in built]
    # the stored worlds are what the host bridge makes of the stored paths
    again = solver.interpolate_and_planes_batch_host(items, worlds[0].veh, worlds[0].parm)
    for (bw, _, _), w in zip(again, worlds):
        np.testing.assert_array_equal(np.asarray(bw.x0_bar), np.asarray(w.x0_bar))
        assert bytes(np.asarray(bw.planes)) == bytes(np.asarray(w.planes))
    return items, worlds


def _same(a, b):
    return (np.array_equal(a.solutions, b.solutions) and np.array_equal(a.corridors, b.corridors) and
            np.array_equal(a.sqp_iters, b.sqp_iters) and np.array_equal(a.admm_iters, b.admm_iters) and
            np.array_equal(a.last_status, b.last_status) and a.solver_status == b.solver_status and
            a.initial_static_legal == b.initial_static_legal)


def test_streamed_do_phase_returns_the_bits_of_the_batch(gpu_handle):
    """Seven worlds of the map50 set in three growing chunks, in the given order and in a permuted one; a second call that
    writes into the first call's arrays; a job smaller than the number of chunks; a job of mixed kernel classes."""
    items, worlds = _items_and_worlds("map50", 7)
    ref = gpu_handle.solve_batch(worlds)
    got, tm = gpu_handle.do_phase_stream(items, worlds[0].veh, worlds[0].parm, single_launch_if_mixed=False, min_first_agents=0)
    sizes = [c["worlds"] for c in tm["chunks"]]
    assert len(sizes) == 3 and sum(sizes) == 7 and sizes == sorted(sizes) and sizes[0] == 1, sizes
    assert tm["first_launch"] < tm["total"]
    assert all(_same(g, r) for g, r in zip(got, ref))
    order = [6, 2, 4, 0, 5, 1, 3]
    got2, _ = gpu_handle.do_phase_stream(items, worlds[0].veh, worlds[0].parm, order=order, out=got, single_launch_if_mixed=False, min_first_agents=0)
    assert all(g2 is g for g2, g in zip(got2, got)) and all(_same(g, r) for g, r in zip(got2, ref))
    got3, tm3 = gpu_handle.do_phase_stream(items[:2], worlds[0].veh, worlds[0].parm, single_launch_if_mixed=False, min_first_agents=0)
    assert [c["worlds"] for c in tm3["chunks"]] == [1, 1] and all(_same(g, r) for g, r in zip(got3, ref[:2]))
    # mixed kernel classes in one streamed job: the map100 worlds run in the 512-thread class, map50's mostly in the 256-thread one
    items100, worlds100 = _items_and_worlds("map100", 2)
    mixed_items, mixed_worlds = items[:3] + items100, worlds[:3] + worlds100
    # (one batch = one parameter block: both sets use the default vehicle and QP parameters)
    refm = gpu_handle.solve_batch(mixed_worlds)
    gotm, _ = gpu_handle.do_phase_stream(mixed_items, worlds[0].veh, worlds[0].parm, fractions=(0.2, 0.4, 0.4),
                                         single_launch_if_mixed=False, min_first_agents=0)
    assert all(_same(g, r) for g, r in zip(gotm, refm))
    # the default: only a job of ONE kernel class is streamed (horizons 129 .. 234, known from the coarse paths); map50's worlds
    # (horizons below 129, two classes) are bridged on the host pool and solved by one launch, map100's are streamed
    gotd, tmd = gpu_handle.do_phase_stream(items, worlds[0].veh, worlds[0].parm)
    assert all(_same(g, r) for g, r in zip(gotd, ref)) and tmd["streamed"] is False and len(tmd["chunks"]) == 1
    items100b, worlds100b = _items_and_worlds("map100", 4)
    ref100 = gpu_handle.solve_batch(worlds100b)
    got100, tm100 = gpu_handle.do_phase_stream(items100b, worlds100b[0].veh, worlds100b[0].parm, min_first_agents=0)
    assert all(_same(g, r) for g, r in zip(got100, ref100)) and tm100["streamed"] is True and len(tm100["chunks"]) == 3
    # the default: the first chunk is enlarged until it fills the GPU (about 230 agents) - these 200 agents are one chunk
    got100, tm100 = gpu_handle.do_phase_stream(items100b, worlds100b[0].veh, worlds100b[0].parm)
    assert all(_same(g, r) for g, r in zip(got100, ref100)) and [c["worlds"] for c in tm100["chunks"]] == [4]


def test_results_written_to_host_memory_are_the_results_copied_from_device_memory(gpu_handle):
    """csdo_dsqp_set_host_results: the kernels write trajectories, safe boxes and counters into page-locked host memory (nothing
    left to copy behind the last kernel - what do_phase_stream does by default) or into device memory: the same bits, on a plain
    handle, on the streamed chunks, switched on and off between uploads, and the per-agent device times arrive as well."""
    items, worlds = _items_and_worlds("map100", 4)
    ref = gpu_handle.solve_batch(worlds)
    try:
        gpu_handle.set_host_results(True)
        got = gpu_handle.solve_batch(worlds)
        assert all(_same(g, r) for g, r in zip(got, ref))
        assert all(g.t_max_individual > 0.0 for g in got)
        assert gpu_handle.transfer_seconds()["d2h"] < 2e-5            # nothing copied: the timer brackets an empty branch
        small = gpu_handle.solve_batch(worlds[1:2])           # a smaller batch in the same page-locked block
        assert _same(small[0], ref[1])
    finally:
        gpu_handle.set_host_results(False)
    again = gpu_handle.solve_batch(worlds)
    assert all(_same(g, r) for g, r in zip(again, ref)) and gpu_handle.transfer_seconds()["d2h"] > 0.0
    for hr in (True, False):
        got_s, tm = gpu_handle.do_phase_stream(items, worlds[0].veh, worlds[0].parm, min_first_agents=0, host_results=hr)
        assert tm["streamed"] is True and all(_same(g, r) for g, r in zip(got_s, ref))


def test_run_async_wait_and_their_guards(gpu_handle):
    from csdotrajectoryplanning_amd._lib import CsdoError
    _, worlds = _items_and_worlds("map50", 2)
    ref = gpu_handle.solve_batch(worlds)
    a, b = gpu_handle.shared(1), gpu_handle.shared(2)
    a.upload(worlds[:1])
    b.upload(worlds[1:])
    a.run_async()
    b.run_async()                       # two batches in flight on one GPU
    with pytest.raises(CsdoError):      # not again, not a new batch, not its results while it runs
        a.run_async()
    with pytest.raises(CsdoError):
        a.upload(worlds[:1])
    with pytest.raises(CsdoError):
        a.download()
    assert b.wait() > 0.0 and a.wait() > 0.0
    with pytest.raises(CsdoError):
        a.wait()                        # nothing pending any more
    assert _same(a.download()[0], ref[0]) and _same(b.download()[0], ref[1])
    a.run()                             # the blocking call is the two halves
    assert _same(a.download()[0], ref[0])


def test_two_handles_driven_by_two_host_threads(gpu_handle):
    """Two independent handles (own streams, own buffers), each driven by its own host thread at the same time, several solves
    each: every result equals the serial one bit for bit."""
    from csdotrajectoryplanning_amd.solver import DsqpHandle
    _, worlds = _items_and_worlds("map50", 6)
    ref = gpu_handle.solve_batch(worlds)
    jobs = {0: [[0, 1, 2], [3], [4, 5, 0]], 1: [[5, 4], [3, 2, 1, 0], [1]]}
    out, errs = {0: [], 1: []}, []
    start = threading.Barrier(2)

    def drive(k):
        try:
            h = DsqpHandle(0)
            try:
                start.wait()
                for sel in jobs[k]:
                    out[k].append((sel, h.solve_batch([worlds[i] for i in sel])))
            finally:
                h.close()
        except Exception as e:          # noqa: BLE001 - reported below
            errs.append(repr(e))

    threads = [threading.Thread(target=drive, args=(k,)) for k in (0, 1)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errs, errs
    for k in (0, 1):
        assert len(out[k]) == len(jobs[k])
        for sel, sols in out[k]:
            assert all(_same(s, ref[i]) for i, s in zip(sel, sols)), (k, sel)


def test_a_chunk_that_fails_leaves_the_handle_usable(gpu_handle):
    """The second chunk of a streamed job hits CSDO_ELIMIT at its upload (a world with more obstacles than fit LDS): the first
    chunk's kernels are already in flight on a cached child handle.  do_phase_stream collects them before it re-raises, so the
    same handle streams the next job (it used to fail with CSDO_EINVAL until the handle was closed)."""
    from csdotrajectoryplanning_amd._lib import CsdoError
    items, worlds = _items_and_worlds("map100", 4)
    ref = gpu_handle.solve_batch(worlds)
    rng = np.random.default_rng(1)
    walls = np.column_stack([rng.uniform(300, 400, 7000), rng.uniform(300, 400, 7000), np.full(7000, 0.5)])
    bad = list(items)
    bad[3] = (*items[3][:4], 500.0, 500.0, walls)        # the last world: it is in the last chunk
    with pytest.raises(CsdoError) as e:
        gpu_handle.do_phase_stream(bad, worlds[0].veh, worlds[0].parm, min_first_agents=0)
    assert "-4" in str(e.value)
    for c in gpu_handle._children.values():              # nothing is pending on any of the chunk handles
        with pytest.raises(CsdoError):
            c.wait()
    got, tm = gpu_handle.do_phase_stream(items, worlds[0].veh, worlds[0].parm, min_first_agents=0)
    assert tm["streamed"] is True and all(_same(g, r) for g, r in zip(got, ref))
