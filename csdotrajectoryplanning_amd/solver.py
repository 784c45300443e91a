"""Host-side mirror of the reference's `sqp/` interface on top of the C ABI.

`SolverDSQP` keeps the reference's shape (sqp/dsqp_solver.h:24-47): constructing it runs the whole DO phase; results
are read from `solutions`, `corridors`, `num_iterations`, `getSolverStatus()`, `getMaxOfRuntimes()`,
`get_initial_static_legal()`.  `interpolate_and_planes` mirrors the three bridge calls of csdo.cc:116-129.
"""
import ctypes as C

import numpy as np

from . import abi
from ._lib import check, lib
from .problem import Solution, World, bridge_to_world, bridge_views


class DsqpHandle:
    """Owns one csdo_handle (device buffers + stream) on one GPU."""

    def __init__(self, device=0, _parent=None, _lane=0, devices=None):
        """device: one GPU ordinal.  devices: a list of ordinals - ONE handle over several GPUs (csdo_dsqp_create_multi): an upload
        cuts the batch's agents into contiguous blocks of equal estimated work, one per device; everything else reads the same."""
        self._h = C.c_void_p()
        if devices is not None:
            devs = np.ascontiguousarray([int(d) for d in devices], dtype=np.int32)
            check(lib().csdo_dsqp_create_multi(C.byref(self._h), abi.as_int32_p(devs), len(devs)), "csdo_dsqp_create_multi")
            device = int(devs[0])
        elif _parent is None:
            check(lib().csdo_dsqp_create(C.byref(self._h), int(device)), "csdo_dsqp_create")
        else:
            check(lib().csdo_dsqp_create_shared(C.byref(self._h), _parent._h, int(_lane)), "csdo_dsqp_create_shared")
        self.devices = None if devices is None else [int(d) for d in devices]
        self.device = int(device)
        self._keep = None
        self._parent = _parent            # (keeps the owner of the streams alive)
        self._children = {}

    def shared(self, lane):
        """A handle for a further batch in flight on this GPU (csdo_dsqp_create_shared): own device buffers, this handle's
        streams starting at `lane`.  Cached per lane; closed with this handle."""
        if lane not in self._children:
            self._children[lane] = DsqpHandle(self.device, _parent=self, _lane=lane)
        return self._children[lane]

    def close(self):
        for c in list(getattr(self, "_children", {}).values()):
            c.close()
        self._children = {}
        if self._h:
            lib().csdo_dsqp_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- whole DO phase, host buffers in / out --------------------------------------------------------------
    def solve_batch(self, worlds):
        sols = [Solution.allocate(w.Na, w.Nt) for w in worlds]
        probs = (abi.Problem * len(worlds))(*[w.c_problem() for w in worlds])
        res = (abi.Result * len(worlds))(*[s._c for s in sols])
        check(lib().csdo_dsqp_solve_batch(self._h, probs, len(worlds), res), "csdo_dsqp_solve_batch")
        self._keep = list(worlds)          # (the batch stays uploaded: launch_groups / agent_groups / run describe it)
        for s, r in zip(sols, res):
            s._c = r
            s.finish()
        return sols

    def solve(self, world: World) -> Solution:
        return self.solve_batch([world])[0]

    # -- split phase (bench: inputs resident in HBM before the timed region) -----------------------------------
    def upload(self, worlds):
        probs = (abi.Problem * len(worlds))(*[w.c_problem() for w in worlds])
        check(lib().csdo_dsqp_upload(self._h, probs, len(worlds)), "csdo_dsqp_upload")
        self._keep = list(worlds)

    def run(self, stream=None):
        check(lib().csdo_dsqp_run(self._h, C.c_void_p(stream) if stream else None), "csdo_dsqp_run")
        return lib().csdo_dsqp_last_kernel_seconds(self._h)

    def set_lane(self, lane):
        check(lib().csdo_dsqp_set_lane(self._h, int(lane)), "csdo_dsqp_set_lane")

    def run_async(self, stream=None):
        """csdo_dsqp_run_async: enqueue the solve and return; wait() collects it."""
        check(lib().csdo_dsqp_run_async(self._h, C.c_void_p(stream) if stream else None), "csdo_dsqp_run_async")

    def wait(self):
        check(lib().csdo_dsqp_wait(self._h), "csdo_dsqp_wait")
        return lib().csdo_dsqp_last_kernel_seconds(self._h)

    def do_phase_stream(self, items, veh, parm, fractions=(0.08, 0.27, 0.65), out=None, order=None, single_launch_if_mixed=True,
                        min_first_agents=230, host_results=True, device_bridge=False):
        """The DO phase of csdo.cc:111-148 for a batch of worlds, streamed in chunks of worlds: the host bridge, the packing
        and the H2D copies of chunk k + 1 run under the solve of chunk k (csdo_dsqp_create_shared / csdo_dsqp_run_async), the
        results of a chunk come back under the solve of the later ones.  The chunks grow (a small first one starts the GPU
        within a millisecond or two; most agents are in the last one, where the launcher's longest-first order still has
        a large pool to work with - an agent that runs long and starts late is what a streamed batch can lose on).
        items: per world (states, actions, path_off, goals, dimx, dimy, obstacles); order: the worlds in the order they
        should be started (default: as given); out: what a previous call returned first (its arrays are written again).
        single_launch_if_mixed: a job whose first chunk needs more than one kernel class is solved by ONE launch of all its
        worlds on this handle instead (which replaces the batch this handle held).
        min_first_agents: the first chunk is enlarged until it holds that many agents (0: the fractions as given).
        host_results: the chunks' kernels write their results into page-locked host memory (csdo_dsqp_set_host_results).
        device_bridge: a job that is solved by one launch is bridged by csdo_preprocess_device_batch (pair search and planes on the
        device; round 4's choice) instead of the host threads (round 6: 60 worlds in 1 - 3 ms on the library's pool).
        Returns (solutions in the order of `items`, dict of host-side timings in seconds)."""
        import time
        n = len(items)
        idx = list(range(n)) if order is None else [int(i) for i in order]
        cuts = stream_cuts([len(items[i][2]) - 1 for i in idx], fractions, min_first_agents)
        fr = [None] * (len(cuts) - 1)
        t0 = time.perf_counter()
        if single_launch_if_mixed:
            # The horizon of a world is known from its coarse paths (sqp/inter_agent_cons.cc:320-325: the longest path, num_interpolation
            # points inserted per move); horizons 129 .. 234 run in ONE kernel class (512 threads, everything in LDS).  Anything else
            # makes a job of several classes: treated as the first chunk of one (the test below) and solved by one launch.
            ni = int(parm.num_interpolation) + 1
            nts = [ni * (int(np.max(np.diff(np.asarray(items[i][2])))) - 1) + 1 for i in idx]
            mixed_by_horizon = not all(128 < nt <= 234 for nt in nts)
        else:
            mixed_by_horizon = False
        if mixed_by_horizon:
            # known before anything is bridged: the device bridge of all worlds in one call (csdo_preprocess_device_batch), one launch
            tb = time.perf_counter()
            if device_bridge:
                bridged = self.interpolate_and_planes_batch([items[i] for i in idx], veh, parm)
            else:
                bridged = interpolate_and_planes_batch_host([items[i] for i in idx], veh, parm)
            tu = time.perf_counter()
            self.set_host_results(host_results)
            try:
                self.upload([b[0] for b in bridged])
            finally:
                self.set_host_results(False)          # (this handle's later uploads: the default again)
            tr = time.perf_counter()
            kern = self.run()
            kernel_end = time.perf_counter() - t0
            got = self.download(out=None if out is None else [out[i] for i in idx])
            sols = [None] * n
            for i, s_ in zip(idx, got):
                sols[i] = s_
            return sols, {"first_launch": tr - t0, "kernels_done": kernel_end, "total": time.perf_counter() - t0, "streamed": False,
                          "chunks": [{"worlds": n, "bridge": tu - tb, "upload": tr - tu, "kernel": kern}]}
        timing = {"first_launch": None, "chunks": []}
        inflight = []
        next_lane = 0
        try:
            for c in range(len(fr)):
                part = idx[cuts[c]:cuts[c + 1]]
                hc = self.shared(c)
                hc.set_host_results(host_results)   # the last chunk's D2H copy is the one that nothing hides
                tb = time.perf_counter()
                bridged = interpolate_and_planes_batch_host([items[i] for i in part], veh, parm)
                tu = time.perf_counter()
                hc.upload([b[0] for b in bridged])
                tr = time.perf_counter()
                # a batch's launch groups run side by side on streams of their own, and kernels that share a stream run one after the
                # other: deal the four streams out by the group counts (a chunk that comes around to a stream still in use queues
                # up behind an EARLIER, i.e. smaller, chunk's kernel)
                ng = len(hc.launch_groups())
                if c == 0 and (ng > 1 or mixed_by_horizon) and single_launch_if_mixed:
                    # Several kernel classes (workgroup sizes / residency modes) in the job: their CU shares are balanced per launch,
                    # and chunks of such launches in flight at once fragment the CUs (measured: map50 set 87 ms streamed against 59 ms,
                    # room set 115 against 68).  One launch of everything, on this handle, with the rest bridged at once.
                    rest = idx[cuts[1]:]
                    more = interpolate_and_planes_batch_host([items[i] for i in rest], veh, parm) if rest else []
                    tu2 = time.perf_counter()
                    self.set_host_results(host_results)
                    try:
                        self.upload([b[0] for b in bridged] + [b[0] for b in more])
                    finally:
                        self.set_host_results(False)
                    tr2 = time.perf_counter()
                    kern = self.run()
                    kernel_end = time.perf_counter() - t0
                    allidx = part + rest
                    got = self.download(out=None if out is None else [out[i] for i in allidx])
                    sols = [None] * n
                    for i, s_ in zip(allidx, got):
                        sols[i] = s_
                    return sols, {"first_launch": tr2 - t0, "kernels_done": kernel_end, "total": time.perf_counter() - t0,
                                  "streamed": False,
                                  "chunks": [{"worlds": n, "bridge": (tu - tb) + (tu2 - tr), "upload": tr2 - tu2, "kernel": kern}]}
                hc.set_lane(next_lane)
                next_lane = (next_lane + max(ng, 1)) % 4
                hc.run_async()
                if timing["first_launch"] is None:
                    timing["first_launch"] = time.perf_counter() - t0
                timing["chunks"].append({"worlds": len(part), "bridge": tu - tb, "upload": tr - tu, "upload_parts": hc.transfer_seconds()})
                inflight.append((hc, part))
        except BaseException:
            # a chunk failed (an obstacle-heavy world hits CSDO_ELIMIT at its upload, a device allocation fails): the chunks already
            # launched are still running and their handles are cached on this one - collect them, or every later call on this
            # handle would find them pending (CSDO_EINVAL at the next upload) with kernels in flight behind the caller's back
            for hc_, _ in inflight:
                try:
                    hc_.wait()
                except Exception:
                    pass
            raise
        sols = [None] * n
        kernel_end = 0.0
        for (hc, part), info in zip(inflight, timing["chunks"]):
            info["kernel"] = hc.wait()
            kernel_end = time.perf_counter() - t0
            got = hc.download(out=None if out is None else [out[i] for i in part])
            for i, s_ in zip(part, got):
                sols[i] = s_
        timing["kernels_done"] = kernel_end
        timing["total"] = time.perf_counter() - t0
        timing["streamed"] = True
        return sols, timing

    def do_phase(self, items, veh, parm, out=None):
        """csdo_do_phase: the DO phase of csdo.cc:111-148 for a batch of worlds in ONE library call - bridge on the library's host
        threads, streamed in chunks when the job is of one kernel class (else one launch), results written by the kernels into
        page-locked host memory and scattered into the arrays returned here.  The C++ form of do_phase_stream (same chunks, same
        bits), without an interpreter between its stages.
        items: per world (states, actions, path_off, goals, dimx, dimy, obstacles); out: what a previous call returned first.
        Returns (solutions in the order of `items`, dict of host-side timings in seconds, initial_inter_legal per world);
        timing["total"] is the library call's own clock, timing["total_with_binding"] includes this method's marshalling."""
        import time
        t_py0 = time.perf_counter()
        n = len(items)
        keep = []
        cw = (abi.CoarseWorld * n)()
        for k, (st, ac, po, G, dimx, dimy, obs) in enumerate(items):
            st = np.ascontiguousarray(st, dtype=np.float64)
            ac = np.ascontiguousarray(ac, dtype=np.int32)
            po = np.ascontiguousarray(po, dtype=np.int32)
            G = np.ascontiguousarray(G, dtype=np.float64)
            obs = np.ascontiguousarray(obs, dtype=np.float64).reshape(-1, 3)
            keep.append((st, ac, po, G, obs))
            c = cw[k]
            c.states, c.actions, c.path_off, c.goals = abi.as_double_p(st), abi.as_int32_p(ac), abi.as_int32_p(po), abi.as_double_p(G)
            c.obstacles, c.Na, c.n_obs, c.dimx, c.dimy = abi.as_double_p(obs), len(po) - 1, obs.shape[0], float(dimx), float(dimy)
        if out is None:
            ni = int(parm.num_interpolation) + 1
            out = [Solution.allocate(len(it[2]) - 1, ni * (int(np.max(np.diff(np.asarray(it[2])))) - 1) + 1) for it in items]
        res = (abi.Result * n)(*[s_._c for s_ in out])
        legal = np.zeros(n, np.int32)
        tm = abi.DoPhaseTiming()
        check(lib().csdo_do_phase(self._h, cw, n, C.byref(veh), C.byref(parm), res, abi.as_int32_p(legal), C.byref(tm)), "csdo_do_phase")
        del keep
        for s_, r in zip(out, res):
            s_._c = r
            s_.finish()
        nc = int(tm.n_chunks)
        timing = {"first_launch": tm.first_launch, "kernels_done": tm.kernels_done, "total": tm.total, "streamed": bool(tm.streamed),
                  "chunks": [{"worlds": int(tm.chunk_worlds[c]), "bridge": tm.chunk_bridge[c], "upload": tm.chunk_upload[c],
                              "kernel": tm.chunk_kernel[c]} for c in range(nc)]}
        timing["total_with_binding"] = time.perf_counter() - t_py0      # ... with this method's marshalling around the library call
        return out, timing, legal

    def transfer_seconds(self):
        """Host seconds of the last upload / download: dict(pack, stage, h2d, d2h, unpack)."""
        out = (C.c_double * 5)()
        check(lib().csdo_dsqp_last_transfer_seconds(self._h, C.byref(out)), "csdo_dsqp_last_transfer_seconds")
        return dict(zip(("pack", "stage", "h2d", "d2h", "unpack"), [float(v) for v in out]))

    def set_host_results(self, on=True):
        """csdo_dsqp_set_host_results: from the next upload on the kernels write their results into page-locked host memory."""
        check(lib().csdo_dsqp_set_host_results(self._h, int(bool(on))), "csdo_dsqp_set_host_results")

    def set_min_residency_mode(self, mode):
        check(lib().csdo_dsqp_set_min_residency_mode(self._h, int(mode)), "csdo_dsqp_set_min_residency_mode")

    def last_limit(self):
        """(world, agent, lds_bytes_needed) of the last CSDO_ELIMIT of an upload (csdo_dsqp_last_limit); world = -1 if none."""
        w, a, b = C.c_int32(), C.c_int32(), C.c_int64()
        check(lib().csdo_dsqp_last_limit(self._h, C.byref(w), C.byref(a), C.byref(b)), "csdo_dsqp_last_limit")
        return int(w.value), int(a.value), int(b.value)

    def launch_groups(self):
        """How the uploaded batch is launched (csdo_dsqp_launch_groups): a list of dicts, one per concurrent kernel."""
        buf = (abi.LaunchGroup * 64)()
        n = lib().csdo_dsqp_launch_groups(self._h, buf, 64)
        if n < 0:
            check(n, "csdo_dsqp_launch_groups")
        return [{f: getattr(buf[i], f) for f, _ in abi.LaunchGroup._fields_} for i in range(min(n, 64))]

    def agent_groups(self):
        """Launch group of every agent of the uploaded batch, in upload order (csdo_dsqp_agent_groups)."""
        n = sum(w.Na for w in self._keep)
        out = np.zeros(n, np.int32)
        check(lib().csdo_dsqp_agent_groups(self._h, abi.as_int32_p(out), n), "csdo_dsqp_agent_groups")
        return out

    def download(self, out=None):
        """Results of the last run.  out: the list a previous download of the same batch returned - its arrays are written again
        instead of allocating (and first touching) new ones."""
        worlds = self._keep
        sols = out if out is not None else [Solution.allocate(w.Na, w.Nt) for w in worlds]
        res = (abi.Result * len(worlds))(*[s._c for s in sols])
        check(lib().csdo_dsqp_download(self._h, res, len(worlds)), "csdo_dsqp_download")
        for s, r in zip(sols, res):
            s._c = r
            s.finish()
        return sols

    def device_solutions(self):
        n = C.c_int64()
        p = lib().csdo_dsqp_device_solutions(self._h, C.byref(n))
        return p, int(n.value)

    def interpolate_and_planes(self, states, actions, path_off, goals, veh, parm, dimx, dimy, obstacles):
        """The bridge with the neighbour search and plane generation on this GPU (csdo_preprocess_device); same return
        value as the module-level interpolate_and_planes, bit for bit."""
        bo = abi.BridgeOut()
        states = np.ascontiguousarray(states, dtype=np.float64)
        actions = np.ascontiguousarray(actions, dtype=np.int32)
        path_off = np.ascontiguousarray(path_off, dtype=np.int32)
        goals = np.ascontiguousarray(goals, dtype=np.float64)
        check(lib().csdo_preprocess_device(self._h, abi.as_double_p(states), abi.as_int32_p(actions),
                                           abi.as_int32_p(path_off), len(path_off) - 1, abi.as_double_p(goals),
                                           C.byref(veh), C.byref(parm), C.byref(bo)), "csdo_preprocess_device")
        out = bridge_to_world(bo, dimx, dimy, obstacles, veh, parm)
        lib().csdo_bridge_free(C.byref(bo))
        return out

    def interpolate_and_planes_batch(self, items, veh, parm):
        """csdo_preprocess_device_batch: items = [(states, actions, path_off, goals, dimx, dimy, obstacles), ...], one per world;
        returns [(World, pairs, initial_inter_legal), ...] - per world what interpolate_and_planes returns."""
        n = len(items)
        keep, st_p, ac_p, po_p, na, g_p = _marshal_paths(items)
        outs = (abi.BridgeOut * n)()
        check(lib().csdo_preprocess_device_batch(self._h, n, st_p, ac_p, po_p, abi.as_int32_p(na), g_p, C.byref(veh),
                                                 C.byref(parm), outs), "csdo_preprocess_device_batch")
        # views of the library-owned results (released with them): no second copy of 29 MB of planes on the way to the upload
        return [bridge_views(outs[k], lib().csdo_bridge_free, it[4], it[5], it[6], veh, parm) for k, it in enumerate(items)]

    def validate(self, solutions, veh, obstacles=None, dimx=0.0, dimy=0.0, margin=0.0, frames_per_move=None):
        """Trajectory validator on this GPU (csdo_validate; with frames_per_move = S >= 1 csdo_validate_frames: the frames
        of the authors' animation, indices in the report are frames); returns a results.ValidationReport."""
        from .results import ValidationReport
        sol = np.ascontiguousarray(np.asarray(solutions, dtype=np.float64)[..., :6])
        if sol.shape[-1] < 6:
            sol = np.ascontiguousarray(np.concatenate([sol, np.zeros(sol.shape[:-1] + (6 - sol.shape[-1],))], -1))
        obs = np.zeros((0, 3)) if obstacles is None else np.ascontiguousarray(obstacles, dtype=np.float64).reshape(-1, 3)
        v = abi.Validation()
        if frames_per_move is None:
            check(lib().csdo_validate(self._h, abi.as_double_p(sol), sol.shape[0], sol.shape[1], abi.as_double_p(obs),
                                      obs.shape[0], float(dimx or 0.0), float(dimy or 0.0), C.byref(veh), float(margin),
                                      C.byref(v)), "csdo_validate")
        else:
            check(lib().csdo_validate_frames(self._h, abi.as_double_p(sol), sol.shape[0], sol.shape[1], int(frames_per_move),
                                             abi.as_double_p(obs), obs.shape[0], float(dimx or 0.0), float(dimy or 0.0),
                                             C.byref(veh), float(margin), C.byref(v)), "csdo_validate_frames")
        fv = tuple(v.first_vehicle) if v.vehicle_collisions else None
        fo = tuple(v.first_obstacle) if v.obstacle_collisions else None
        return ValidationReport(int(v.vehicle_collisions), int(v.obstacle_collisions), int(v.out_of_map), fv, fo,
                                float(v.min_obstacle_clearance))

    def generate_boxes(self, points, obstacles, dimx, dimy, veh):
        points = np.ascontiguousarray(points, dtype=np.float64).reshape(-1, 2)
        obstacles = np.ascontiguousarray(obstacles, dtype=np.float64).reshape(-1, 3)
        n = points.shape[0]
        boxes = np.zeros((n, 4))
        status = np.zeros(n, np.int32)
        check(lib().csdo_generate_boxes(self._h, abi.as_double_p(points), n, abi.as_double_p(obstacles),
                                        obstacles.shape[0], float(dimx), float(dimy), C.byref(veh),
                                        abi.as_double_p(boxes), abi.as_int32_p(status)), "csdo_generate_boxes")
        return boxes, status

    def math_eval(self, fn, a, b=None):
        """Diagnostic: the device program's sin (fn 0), cos (1), tan (2) of a, atan2 (3) of (a, b), evaluated on the device."""
        a = np.ascontiguousarray(a, dtype=np.float64).ravel()
        b = a if b is None else np.ascontiguousarray(b, dtype=np.float64).ravel()
        out = np.zeros_like(a)
        check(lib().csdo_math_eval(self._h, int(fn), abi.as_double_p(a), abi.as_double_p(b), abi.as_double_p(out), a.size),
              "csdo_math_eval")
        return out


class SolverDSQP:
    """Drop-in shape of the reference class: the constructor solves (sqp/dsqp_solver.cc:1133-1249)."""

    def __init__(self, x0_bar, inter_planes, dimx, dimy, obstacles, param, veh=None, logger_level=2, handle=None):
        from .config import vehicle_from_config
        veh = veh or vehicle_from_config()
        plane_off, planes = inter_planes  # (CSR offsets, abi.PLANE_DTYPE array)
        world = World(x0_bar, plane_off, planes, dimx, dimy, obstacles, veh, param, logger_level)
        own = handle is None
        handle = handle or DsqpHandle(0)
        try:
            sol = handle.solve(world)
        finally:
            if own:
                handle.close()
        self._sol = sol
        self.solutions = sol.solutions
        self.corridors = sol.corridors
        self.num_iterations = sol.sqp_iters
        self.admm_iterations = sol.admm_iters
        self.last_status = sol.last_status

    def getSolverStatus(self):
        return self._sol.solver_status

    def getMaxOfRuntimes(self):
        return self._sol.t_max_individual

    def get_initial_static_legal(self):
        return bool(self._sol.initial_static_legal)


def _marshal_paths(items):
    n = len(items)
    keep = []
    st_p, ac_p, po_p, g_p = (abi.c_double_p * n)(), (abi.c_int32_p * n)(), (abi.c_int32_p * n)(), (abi.c_double_p * n)()
    na = np.zeros(n, np.int32)
    for k, (st, ac, po, G, _, _, _) in enumerate(items):
        st = np.ascontiguousarray(st, dtype=np.float64)
        ac = np.ascontiguousarray(ac, dtype=np.int32)
        po = np.ascontiguousarray(po, dtype=np.int32)
        G = np.ascontiguousarray(G, dtype=np.float64)
        keep.append((st, ac, po, G))
        st_p[k], ac_p[k], po_p[k], g_p[k] = abi.as_double_p(st), abi.as_int32_p(ac), abi.as_int32_p(po), abi.as_double_p(G)
        na[k] = len(po) - 1
    return keep, st_p, ac_p, po_p, na, g_p


def stream_cuts(agents_per_world, fractions=(0.08, 0.27, 0.65), min_first_agents=230):
    """Chunk boundaries of a streamed DO phase over len(agents_per_world) worlds (DsqpHandle.do_phase_stream): growing chunks by
    `fractions` (at most four, never an empty one), the first one enlarged until it holds `min_first_agents` agents - it has to fill
    the GPU by itself (one workgroup per CU), else the CUs it leaves idle are lost until the second chunk arrives -, and a job that
    such a first chunk takes a fifth of is split in two, not three (100-vehicle worlds, 12 of them: 1 + 3 + 8 worlds 70.7 ms,
    3 + 3 + 6 72.1, 3 + 9 66.6, one launch 67.1).  Returns [0, c1, ..., n]."""
    n = len(agents_per_world)
    fr = [f for f in fractions if f > 0]
    fr = fr[-min(len(fr), n, 4):]                      # at most four batches in flight (four streams), never an empty chunk
    cuts = [0]
    tot = float(sum(fr))
    acc = 0.0
    for c, f in enumerate(fr):
        acc += f
        hi = n if c == len(fr) - 1 else max(cuts[-1] + 1, min(n - (len(fr) - 1 - c), int(round(n * acc / tot))))
        cuts.append(hi)
    na = np.cumsum(agents_per_world)
    fill = int(np.searchsorted(na, min_first_agents)) + 1
    if len(cuts) > 2 and cuts[1] < fill:
        if fill >= n:
            cuts = [0, n]
        elif fill >= 0.2 * n:
            cuts = [0, fill, n]
        else:
            cuts = [0, fill] + [max(c_, fill + k_ + 1) for k_, c_ in enumerate(cuts[2:-1])] + [n]
    return cuts


def interpolate_and_planes_batch_host(items, veh, parm):
    """csdo_preprocess_batch: the host bridge of every world of `items` (as DsqpHandle.interpolate_and_planes_batch takes them)
    on a pool of host threads, no device work.  Returns [(World, pairs, initial_inter_legal), ...], views of library memory."""
    n = len(items)
    keep, st_p, ac_p, po_p, na, g_p = _marshal_paths(items)
    outs = (abi.BridgeOut * n)()
    check(lib().csdo_preprocess_batch(n, st_p, ac_p, po_p, abi.as_int32_p(na), g_p, C.byref(veh), C.byref(parm), outs),
          "csdo_preprocess_batch")
    del keep
    return [bridge_views(outs[k], lib().csdo_bridge_free, it[4], it[5], it[6], veh, parm) for k, it in enumerate(items)]


def estimate_work(worlds):
    """csdo_dsqp_estimate_work: the launcher's relative work estimate of every agent of `worlds`, in order (host code)."""
    probs = (abi.Problem * len(worlds))(*[w.c_problem() for w in worlds])
    est = np.zeros(int(sum(w.Na for w in worlds)))
    check(lib().csdo_dsqp_estimate_work(probs, len(worlds), abi.as_double_p(est)), "csdo_dsqp_estimate_work")
    return est


def interpolate_and_planes(states, actions, path_off, goals, veh, parm, dimx, dimy, obstacles):
    """InterpolateInitalGuess + findNeighborPairsByTrustRegion + calcEqualInterPlanes (csdo.cc:116-129).
    Returns (World, pairs[n,3] = (t,i,j), initial_inter_legal)."""
    bo = abi.BridgeOut()
    states = np.ascontiguousarray(states, dtype=np.float64)
    actions = np.ascontiguousarray(actions, dtype=np.int32)
    path_off = np.ascontiguousarray(path_off, dtype=np.int32)
    goals = np.ascontiguousarray(goals, dtype=np.float64)
    check(lib().csdo_preprocess(abi.as_double_p(states), abi.as_int32_p(actions), abi.as_int32_p(path_off),
                                len(path_off) - 1, abi.as_double_p(goals), C.byref(veh), C.byref(parm),
                                C.byref(bo)), "csdo_preprocess")
    out = bridge_to_world(bo, dimx, dimy, obstacles, veh, parm)
    lib().csdo_bridge_free(C.byref(bo))
    return out
