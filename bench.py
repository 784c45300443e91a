#!/usr/bin/env python
"""bench.py — DO-phase throughput of the MI355X backend on BASELINE.json's workloads.

One step = one pass of the hot path (the whole SolverDSQP-equivalent solve: initial corridors + every agent's SQP with
its ADMM QPs) over one batch of worlds whose inputs are already resident in HBM.

--workload map100     (default; BASELINE.json configs[2]/[3]) the 60 instances of map100by100/agents50/obstacle,
                      3000 agents in ONE batch
           map50      (configs[1]) the 60 instances of map50by50/agents25/obstacle, 1500 agents
           synth1024  (configs[4]) 21 worlds (ex0..ex20 of the map100 set) truncated to 1024 agents
           room50     benchmark/room/agents50 ex0..ex11: 238 obstacles per world (the obstacle-dense regime)
           agents100  benchmark/map100by100/agents100/obstacle ex0..ex11: 100 vehicles per world (the plane-dense regime)
--gpus N > 1 outside torch.distributed.run: bench.py starts the N ranks ITSELF - a child `python -m torch.distributed.run --standalone
--local-addr 127.0.0.1 --nnodes=1 --nproc-per-node N bench.py <the same arguments>`, started before this process has touched the
GPU, rank 0's JSON line comes through on the shared stdout, the exit code is the children's.  Launched under torch.distributed.run
(RANK and WORLD_SIZE set, as the driver does for N > 1) it is one of the ranks: --gpus left out adopts WORLD_SIZE, --gpus given must
equal it.  For N > 1 the line carries BOTH scaling curves: `value` is --scaling's job, `value_weak` / `value_strong` the other's
(a second timed loop of the same K steps).
N > 1 ranks (one process per GPU, RCCL):
Either way the agents of the job are SHARDED by sharding.shard_batch_plan: rank r owns one contiguous block of the job's
concatenated agents (the reference's loop over agents is what shards, sqp/dsqp_solver.cc:1198-1220), builds only the worlds
its block overlaps, solves its block, and the step ends with the path's only collective, the all-gather of the final
trajectories on the device pointer.
--scaling strong      (default) the job is ONE copy of the workload whatever N - BASELINE's metric read literally: one workload
                      at 1 / 2 / 4 / 8 GPUs.  Its floor is the workload's longest agent (map100: ~35 ms of a ~65 ms step), so
                      the expected curve flattens at that floor, not 1/N; the line also carries `strong_scaling_floor_ms`
--scaling weak        the job is N copies of the workload (the stand-in worlds of copy c seeded with 60 c): the per-GPU
                      work is fixed as N grows
Blocks are balanced by the launcher's own per-agent work estimate (csdo_dsqp_estimate_work), not by agent count.
--force-dist          N = 1 with a one-rank "nccl" group: RCCL initialisation, the all-gather on the solver's device buffer
                      and the stream ordering run on the one GPU there is
--dry                 CPU check of the N-rank plumbing (tests/test_bench_launch.py): backend gloo, no GPU, the lane-serial host build
                      of the device program (tests/emu) as the solver: plan, shard, solve, stage / collect, the JSON line with
                      n_gpus = the group's size and `value` null - a dry line is not a measurement
Rank 0 prints ONE JSON line.

metric  = agent-QP-iterations/sec: ADMM iterations executed by all agents of all ranks / wall time of the K steps (max
          over ranks): kernels only, inputs resident in HBM.  `value_e2e` is the same count over the PCIe-inclusive DO phase
          of csdo.cc:111-148 (`do_phase_e2e`: ONE csdo_do_phase call - bridge on the library's host threads, pack + H2D,
          kernels writing their results into page-locked host memory, scatter; streamed in chunks of worlds where the job is
          of one kernel class - host wall clock, nothing resident); `single_instance` (ex0 alone) is the metric's "DO-phase
          ms, 50-agent instance".
roofline: `fp64` and `lds` price the ADMM iterations against the two resources the kernel actually uses (DESIGN section 5 writes
          the counts out: F_iter = 718 Nt + 88 K + 2592 flop and L_iter = 8 (139 Nt + 42 K + 1440) LDS bytes per agent-iteration,
          against 78.6 TFLOP/s fp64 vector and 157 TB/s LDS); the contract's figure is the nominal HBM roofline of SURVEY 8(d): algorithmic bytes W_iter = 2280*Nt + 416*K_a per agent-iteration summed
          over the iterations the dominant kernel's launch executes / that kernel's average duration (HIP events on the
          launch stream inside csdo_dsqp_run) against 8 TB/s.  The working set is on-chip, so the kernel is latency
          bound, not HBM bound: `traffic` (HBM bytes per launch from rocprofv3 FETCH_SIZE/WRITE_SIZE passes),
          `hbm_counter_frac` and `valu_fp64_issue_frac` come from the newest profiles/rNN_pmc_summary.json and are null
          when that profile was taken on a different workload, from other kernel sources (its `kernel_source_hash` against the
          running library's csdo_source_hash()) or its step time is more than 20 % off this run's.
cpu_baseline: the oracle (CPU restatement of the reference + OSQP 0.6.3, kind "port") on the same batch with ONE thread
          pool over all its agents on all host cores (rank 0, N = 1 only).
"""
import argparse
import glob
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec


class _DevArray:
    """Exposes a raw device pointer through __cuda_array_interface__ so torch can wrap it without a copy."""

    def __init__(self, ptr, n):
        self.__cuda_array_interface__ = {"data": (int(ptr), False), "shape": (int(n),), "typestr": "<f8",
                                         "version": 2, "strides": None}


def _build(job):
    from csdotrajectoryplanning_amd import workloads
    world, info = workloads.build_job(job)
    return world, info


def algorithmic_bytes(world, admm_iters):
    """Per agent: iterations * W_iter, W_iter = 2280*Nt + 416*K_a bytes."""
    K = (world.plane_off[1:] - world.plane_off[:-1]).astype("float64")
    return admm_iters.astype("float64") * (2280.0 * world.Nt + 416.0 * K)


def effective_cpus():
    """CPUs this process may actually use: the affinity mask, capped by a cgroup CPU quota when there is one."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(float(quota) / float(period) + 0.5)))
    except Exception:
        try:
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f:
                q = int(f.read())
            with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
                per = int(f.read())
            if q > 0:
                n = min(n, max(1, int(q / per + 0.5)))
        except Exception:
            pass
    return n


def _newest_pmc(workload, step_ms, lib_hash):
    """HBM traffic / VALU figures of the dominant kernel from the newest committed PMC summary of this workload - but only if that
    summary was taken from THE KERNEL THAT IS RUNNING: its `kernel_source_hash` (scripts/summarize_profiles.py) must equal the
    library's csdo_source_hash(), and its step time must be within 20 % of this run's.  Otherwise Nones and the reason: counters of
    another build are not evidence for this one."""
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_summary.json")))
    for path in reversed(files):
        try:
            with open(path) as f:
                d = json.load(f)
        except Exception:
            continue
        if d.get("workload", "map100") != workload:
            continue
        name = os.path.basename(path)
        if d.get("kernel_source_hash") != lib_hash:
            return None, None, None, name + " (not quoted: taken from kernel sources %s, this library is %s)" % (
                d.get("kernel_source_hash", "without a hash"), lib_hash)
        ref_ms = d.get("ms_per_step")      # span of all agent-kernel dispatches of a step in the profiled run
        if ref_ms and abs(ref_ms - step_ms) > 0.2 * step_ms:
            return None, None, None, name + " (stale: %.1f ms per step there)" % ref_ms
        return (d.get("hbm_bytes_per_launch_dominant_kernel"), d.get("hbm_counter_frac_of_peak"),
                d.get("valu_fp64_issue_frac"), name)
    return None, None, None, None


def _dry_run(args, rank, world_size, worlds, jobs, sizes, shard_balance, strong, other=None):
    """--dry: the N-rank step on the CPU - gloo group, this rank's block solved by the lane-serial host build of the device program
    (tests/emu, test infrastructure), FlatGather.stage / collect as in the GPU step - and rank 0's line.  Nothing here is timed as a
    result: `value` is null.  N > 1: the other scaling's job runs the same way behind the first (`other_scaling`)."""
    import hashlib
    import numpy as np
    import torch
    import torch.distributed as dist
    from csdotrajectoryplanning_amd import sharding
    from tests import emu_lib
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", str(29400 + os.getpid() % 500))
    dist.init_process_group(backend="gloo", rank=rank, world_size=world_size)
    dev = torch.device("cpu")

    def job(job_worlds):
        n_dbl = int(sum(w.Na * w.Nt * 6 for w in job_worlds))
        fg = sharding.FlatGather(n_dbl, dist, dev)
        iters_step, local = 0, None
        for _ in range(args.warmup + args.steps):
            sols = emu_lib.solve_batch(job_worlds, 0, max(1, (os.cpu_count() or 2) // world_size)) if job_worlds else []
            local = torch.from_numpy(np.concatenate([s.solutions.reshape(-1) for s in sols]) if sols else np.zeros(0))
            fg.stage(local)
            fg.collect()
            iters_step = int(sum(int(s.admm_iters.sum()) for s in sols))
        mine = fg.parts()[rank]
        ok = bool(torch.equal(mine, local)) and fg.lengths[rank] == n_dbl
        t = torch.tensor([float(iters_step), float(sum(w.Na for w in job_worlds)), float(ok)], dtype=torch.float64)
        allr = [torch.zeros_like(t) for _ in range(world_size)]
        dist.all_gather(allr, t)
        sha = hashlib.sha256(np.concatenate([p.numpy() for p in fg.parts()]).tobytes()).hexdigest() if rank == 0 else None
        return allr, sha, int(sum(fg.lengths))

    t0 = time.perf_counter()
    allr, gathered_sha, gathered = job(worlds)
    other_line = None
    if other is not None:
        o_jobs, o_sizes, _, _, o_worlds, _, _ = other
        o_allr, o_sha, o_gathered = job(o_worlds)
        other_line = {"scaling": "weak" if strong else "strong", "value": None, "worlds_total": len(o_jobs),
                      "agents_total": int(sum(o_sizes)), "gathered_doubles": o_gathered, "gathered_sha256": o_sha,
                      "per_rank": [{"rank": r, "agents": int(v[1]), "admm_iterations_per_step": int(v[0]),
                                    "gathered_block_equals_local": bool(v[2])} for r, v in enumerate(o_allr)]}
    elapsed = time.perf_counter() - t0
    n_ranks = dist.get_world_size()
    dist.barrier()
    dist.destroy_process_group()
    if rank == 0:
        print(json.dumps({
            "metric": "agent_qp_iterations_per_sec", "value": None, "unit": "agent-QP-iterations/s", "dry": True,
            "n_gpus": n_ranks, "rccl_ranks": None, "gloo_ranks": n_ranks, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": None, "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None,
            ("value_weak" if strong else "value_strong"): None, "other_scaling": other_line,
            "dtype": "f64", "data": "dry run on the CPU (lane-serial host build of the device program): plumbing only, no measurement",
            "config": {"workload_key": args.workload, "worlds_total": len(jobs), "agents_total": int(sum(sizes)),
                       "per_rank": [{"rank": r, "agents": int(v[1]), "admm_iterations_per_step": int(v[0]),
                                     "gathered_block_equals_local": bool(v[2])} for r, v in enumerate(allr)],
                       "shard_balance": shard_balance, "gathered_doubles": gathered,
                       "gathered_sha256": gathered_sha, "wall_s": elapsed}}), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=None,
                    help="ranks = GPUs of the job (default: WORLD_SIZE under torch.distributed.run, else 1)")
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", choices=("map100", "map50", "synth1024", "room50", "agents100"), default="map100")
    ap.add_argument("--front", choices=("auto", "stand-in"), default="auto",
                    help="initial guesses: the front end's stored paths where it solves the instance (auto, default) or the "
                         "seeded stand-in for every instance (the round-1 workload, for like-for-like comparisons)")
    ap.add_argument("--scaling", choices=("weak", "strong"), default="strong",
                    help="N > 1: the sharded job is one copy of the workload whatever N (strong, default: BASELINE's metric is "
                         "ONE workload at 1/2/4/8 GPUs) or N copies of it (weak)")
    ap.add_argument("--instances", type=int, default=None, help="instances of the set (default: the whole set)")
    ap.add_argument("--setup-procs", type=int, default=32, help="processes building the worlds (1: in-process, no fork; "
                                                                  "use that under rocprofv3)")
    ap.add_argument("--skip-single-instance", action="store_true",
                    help="do not measure ex0 alone before the batch (profiling runs: keeps the kernel statistics of the "
                         "batch launches free of the small launches)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-e2e", action="store_true", help="skip the PCIe-inclusive DO-phase measurement")
    ap.add_argument("--force-dist", action="store_true",
                    help="N = 1: run the sharded path with a one-rank nccl group (RCCL init, all-gather on the device buffer)")
    ap.add_argument("--solve-refinement", type=int, nargs="?", const=1, default=0, choices=(0, 1, 2),
                    help="csdo_qp_parm::solve_refinement: 1 = every linear solve refined on the KKT residual (a second solve per iteration, "
                         "about 1.8 x the kernel time), 2 = lagged (the residual joins the next rhs: one solve per iteration, 1.3 - 1.45 x) - "
                         "what the accurate modes cost, not the headline configuration")
    ap.add_argument("--dry", action="store_true",
                    help="CPU check of the multi-rank plumbing: gloo, no GPU, the lane-serial host build as the solver; not a measurement")
    args = ap.parse_args()

    # Under torch.distributed.run (RANK and WORLD_SIZE set: the driver's launch for N > 1) this process is one of the ranks: --gpus left
    # out adopts WORLD_SIZE, --gpus given must equal it.  Otherwise (a stray WORLD_SIZE without RANK counts for nothing) --gpus N > 1
    # starts the N ranks here.
    under_launcher = "RANK" in os.environ and "WORLD_SIZE" in os.environ
    if args.gpus is None:
        args.gpus = int(os.environ["WORLD_SIZE"]) if under_launcher else 1
    if args.gpus < 1:
        ap.error("--gpus must be >= 1")
    if args.gpus > 1 and not under_launcher:
        # One process per GPU, started HERE: fresh children (no exec of this process, which must not have touched the GPU - it has
        # not: nothing above imports torch or the library), rank 0's line on the inherited stdout, the children's exit code.
        # --standalone: the launcher binds its own rendezvous port (port 0) - no port probed here and taken by someone else before
        # torchrun binds it; --local-addr: the container's hostname may not resolve.
        import subprocess
        cmd = [sys.executable, "-m", "torch.distributed.run", "--standalone", "--local-addr", "127.0.0.1", "--nnodes=1",
               "--nproc-per-node", str(args.gpus), os.path.abspath(__file__)] + sys.argv[1:]
        env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
        env["OMP_NUM_THREADS"] = os.environ.get("OMP_NUM_THREADS", "1")
        sys.stdout.flush()
        sys.exit(subprocess.run(cmd, env=env).returncode)

    rank = int(os.environ.get("RANK", "0")) if under_launcher else 0
    local_rank = int(os.environ.get("LOCAL_RANK", "0")) if under_launcher else 0
    world_size = int(os.environ["WORLD_SIZE"]) if under_launcher else 1
    if world_size != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d: launch with --nproc-per-node equal to --gpus (or without "
                         "torch.distributed.run: bench.py starts its ranks itself)" % (args.gpus, world_size))
    strong = args.scaling == "strong"
    copies = 1 if (strong or world_size == 1) else world_size
    sharded = world_size > 1 or args.force_dist

    # torch first: it brings its own HIP runtime, which must be the one libcsdo_hip.so binds to (loading the library
    # before torch puts two runtimes into the process and aborts under rocprofv3); importing does not touch the GPU
    import torch

    # ---- workload (host side, before anything touches the GPU: the pool forks) ----
    from multiprocessing import get_context
    from csdotrajectoryplanning_amd import sharding, workloads
    t_pre0 = time.perf_counter()

    def _build_all(js):
        procs = min(len(js), os.cpu_count() or 1, max(args.setup_procs, 1))
        if procs > 1:
            with get_context("fork").Pool(procs) as pool:
                return pool.map(_build, js)
        return [_build(j) for j in js]

    built0_cache = [None]

    def build_rank_job(n_copies):
        """This rank's share of a job of `n_copies` copies of the workload: (jobs, sizes, plan, my_jobs, worlds, infos, shard_balance)."""
        jobs_ = []
        for c in range(n_copies):
            jobs_ += workloads.workload_jobs(args.workload, args.instances, seed_offset=60 * c, front=args.front)
        sizes_ = [workloads.job_agents(j) for j in jobs_]
        balance = None
        if sharded:
            # blocks of equal estimated WORK (the launcher's own estimate), not of equal agent count: every rank builds copy 0 of the
            # workload for the estimates (the copies of a weak-scaling job differ only in the seeds of the few stand-in worlds)
            import numpy as _np
            from csdotrajectoryplanning_amd.solver import estimate_work
            base = jobs_[:len(jobs_) // n_copies]
            if built0_cache[0] is None:
                built0_cache[0] = _build_all(base)
            built0 = built0_cache[0]
            est = _np.tile(estimate_work([w for w, _ in built0]), n_copies)
            plan_ = sharding.shard_batch_plan(sizes_, rank, world_size, weights=est)
            loads = [float(est[lo:hi].sum()) for lo, hi in sharding.shard_bounds_weighted(est, world_size)]
            by_count = [float(est[lo:hi].sum()) for lo, hi in sharding.shard_bounds(len(est), world_size)]
            balance = {"estimated_work_max_over_mean": max(loads) / (sum(loads) / len(loads)),
                       "if_balanced_by_agent_count": max(by_count) / (sum(by_count) / len(by_count))}
            my_jobs_ = [jobs_[w] for w, _, _ in plan_]
            built = []
            todo = []
            for w, _, _ in plan_:
                c, k = divmod(w, len(base))
                if c == 0 or str(built0[k][1]["generator"]).startswith("front_end"):
                    built.append(built0[k])
                else:
                    built.append(None)
                    todo.append((len(built) - 1, jobs_[w]))
            for (slot, _), b in zip(todo, _build_all([j for _, j in todo]) if todo else []):
                built[slot] = b
        else:
            plan_ = [(w, 0, n) for w, n in enumerate(sizes_)]
            my_jobs_ = [jobs_[w] for w, _, _ in plan_]
            built = _build_all(my_jobs_)
        worlds_, infos_ = [], []
        for (w, lo, hi), (world, info) in zip(plan_, built):
            if args.solve_refinement:
                world = world.with_parm(solve_refinement=int(args.solve_refinement))
            worlds_.append(world if (lo == 0 and hi == world.Na) else world.subset(lo, hi))
            infos_.append(info)
        return jobs_, sizes_, plan_, my_jobs_, worlds_, infos_, balance

    jobs, sizes, plan, my_jobs, worlds, infos, shard_balance = build_rank_job(copies)
    # N > 1: the OTHER scaling's job too, so that one line carries both curves (`value` is --scaling's, `value_weak` / `value_strong`
    # the other's; two timed loops)
    other = build_rank_job(world_size if strong else 1) if world_size > 1 else None
    t_pre = time.perf_counter() - t_pre0

    import numpy as np
    if args.dry:
        return _dry_run(args, rank, world_size, worlds, jobs, sizes, shard_balance, strong, other)
    from csdotrajectoryplanning_amd.solver import DsqpHandle, interpolate_and_planes
    n_front = sum(1 for i in infos if str(i["generator"]).startswith("front_end"))
    guesses = ("initial guesses: %d worlds from this repository's front end (PBS over hybrid A*, paths stored by "
               "tests/golden/make_front_end_paths.py), %d worlds the search does not solve from the seeded stand-in %s"
               % (n_front, len(infos) - n_front, next((i["generator"] for i in infos
                                                       if not str(i["generator"]).startswith("front_end")), "-")))

    dist = None
    torch.cuda.set_device(local_rank)
    if sharded:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(29400 + os.getpid() % 500))
        dist.init_process_group(backend="nccl", rank=rank, world_size=world_size)
    dev = torch.device("cuda", local_rank)
    # an explicit (non-null) stream: the solver's kernels, the copy into the send buffer and the collective are all
    # ordered on it (torch's default stream has handle 0, which csdo_dsqp_run would replace by the handle's own stream)
    tstream = torch.cuda.Stream(device=dev)
    stream = tstream.cuda_stream

    w0 = worlds[0]
    h = DsqpHandle(local_rank)

    # ---- the metric's "DO-phase ms, 50-agent instance": the first instance alone, outside the timed region ----
    single = None
    if not args.skip_single_instance and rank == 0:
        st, ac, po, G = infos[0]["paths"]
        full0 = _build(my_jobs[0])[0] if worlds[0].Na != sizes[plan[0][0]] else w0
        t_bridge1 = None
        for _ in range(3):                     # best of 3, each after a pause: the library's host threads are asleep when it starts
            time.sleep(0.05)
            t_b0 = time.perf_counter()
            interpolate_and_planes(st, ac, po, G, full0.veh, full0.parm, full0.dimx, full0.dimy, full0.obstacles)
            t_b1 = time.perf_counter() - t_b0
            t_bridge1 = t_b1 if t_bridge1 is None else min(t_bridge1, t_b1)
        h.upload([full0])
        t_u0 = time.perf_counter()
        h.upload([full0])                      # device buffers exist now: this is the steady-state upload
        t_upload1 = time.perf_counter() - t_u0
        h.run(stream)
        single_kernel, t_download1, sol0 = None, None, None
        for _ in range(3):                     # best of 3; everything reported about the run comes from that one run
            ks = h.run(stream)
            t_d0 = time.perf_counter()
            s0 = h.download()[0]
            t_dl = time.perf_counter() - t_d0
            if single_kernel is None or ks < single_kernel:
                single_kernel, t_download1, sol0 = ks, t_dl, s0
        single = {
            "workload": "%s alone: Na=%d, Nt=%d, %d planes" % (infos[0]["instance"], full0.Na, full0.Nt,
                                                               int(full0.plane_off[-1])),
            "admm_iterations": int(sol0.admm_iters.sum()), "solver_status": int(sol0.solver_status),
            "do_phase_ms": {"bridge_host": t_bridge1 * 1e3, "upload_h2d": t_upload1 * 1e3,
                            "solve_kernel": single_kernel * 1e3, "download_d2h": t_download1 * 1e3,
                            "total": (t_bridge1 + t_upload1 + single_kernel + t_download1) * 1e3,
                            "max_individual_agent": sol0.t_max_individual * 1e3},
            "agent_qp_iterations_per_sec": float(sol0.admm_iters.sum()) / single_kernel,
        }

    # ---- the whole pipeline of csdo.cc:93-159 for that instance: real front end (host PBS over hybrid A*), bridge with its
    # pair search and planes on the device, DO phase, validator, with the search run now (the batch below reads the stored
    # paths of the same search).
    pipeline = None
    if not args.skip_single_instance and rank == 0 and single is not None:
        from csdotrajectoryplanning_amd import front_end as fe, instance as inst_mod
        from csdotrajectoryplanning_amd.workloads import INSTANCE_DIR
        inst0 = inst_mod.load_instance(os.path.join(INSTANCE_DIR, infos[0]["instance"]), obs_radius=w0.veh.obs_radius)
        cp = fe.plan(inst0.starts, inst0.goals, inst0.dimx, inst0.dimy, inst0.obstacles, w0.veh)
        if cp is None:
            pipeline = {"instance": infos[0]["instance"], "search_status": 0}
        else:
            t_b0 = time.perf_counter()
            pw = h.interpolate_and_planes(cp.states, cp.actions, cp.path_off, inst0.goals, w0.veh, w0.parm, inst0.dimx,
                                          inst0.dimy, inst0.obstacles)[0]
            t_pb = time.perf_counter() - t_b0
            t_s0 = time.perf_counter()
            psol = h.solve(pw)
            t_ps = time.perf_counter() - t_s0
            t_v0 = time.perf_counter()
            prep = h.validate(psol.solutions, pw.veh, pw.obstacles, pw.dimx, pw.dimy)
            t_pv = time.perf_counter() - t_v0
            pipeline = {"instance": infos[0]["instance"], "search_status": 1, "Na": pw.Na, "Nt": pw.Nt,
                        "planes": int(pw.plane_off[-1]), "high_level_nodes": cp.hl_expanded,
                        "low_level_expansions": cp.ll_expanded,
                        "ms": {"front_end_host": cp.seconds * 1e3, "bridge_device": t_pb * 1e3,
                               "do_phase_upload_solve_download": t_ps * 1e3, "validator_device": t_pv * 1e3},
                        "admm_iterations": int(psol.admm_iters.sum()), "solver_status": int(psol.solver_status),
                        "vehicle_collision_triples": prep.vehicle_collisions,
                        "obstacle_collision_triples": prep.obstacle_collisions, "out_of_map": prep.out_of_map}

    # ---- the batch ----
    # Sharded step: solve enqueued (csdo_dsqp_run_async), the copy out of the solver's buffer ordered behind it on the same stream,
    # the all-gather on a SECOND stream behind that copy - so the collective of step k runs beside the solve of step k + 1 (the
    # next copy into the send buffer waits for it).  The barrier / synchronize pair around the timed region covers both streams.
    gstream = torch.cuda.Stream(device=dev) if sharded else None

    def timed_job(job_worlds):
        """Upload, W warm-up steps, K timed steps between barrier + synchronize pairs, download: one job of this rank."""
        t_u0 = time.perf_counter()
        h.upload(job_worlds)
        t_up = time.perf_counter() - t_u0
        ptr, n_dbl = h.device_solutions()
        sol_dev = torch.as_tensor(_DevArray(ptr, n_dbl), device=dev)
        fg = None
        if sharded:
            with torch.cuda.stream(tstream):
                fg = sharding.FlatGather(n_dbl, dist, dev)     # ranks differ in sum(Nt): padded to the largest
            tstream.synchronize()
        collected = [None]

        def step():
            if fg is None:
                return h.run(stream)
            h.run_async(stream)
            with torch.cuda.stream(tstream):
                if collected[0] is not None:
                    tstream.wait_event(collected[0])
                fg.stage(sol_dev)
                staged = tstream.record_event()
            with torch.cuda.stream(gstream):
                gstream.wait_event(staged)
                fg.collect()
                collected[0] = gstream.record_event()
            return h.wait()

        for _ in range(args.warmup):
            step()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        k_s, g_s = 0.0, None
        for _ in range(args.steps):
            k_s += step()
            gs = [g["seconds"] for g in h.launch_groups()]
            g_s = gs if g_s is None else [a + b for a, b in zip(g_s, gs)]
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        t_d0 = time.perf_counter()
        job_sols = h.download()
        return {"elapsed": el, "kernel_s": k_s, "group_s": g_s, "sols": job_sols, "t_upload": t_up,
                "t_download": time.perf_counter() - t_d0}

    def over_ranks(el, iters):
        """(max over ranks of the elapsed time, sum over ranks of the iterations per step)"""
        if dist is None:
            return el, float(iters)
        t = torch.tensor([el, float(iters)], dtype=torch.float64, device=dev)
        tmax, tsum = t.clone(), t.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
        return float(tmax[0]), float(tsum[1])

    main_job = timed_job(worlds)
    elapsed, kernel_s, group_s, sols = main_job["elapsed"], main_job["kernel_s"], main_job["group_s"], main_job["sols"]
    t_upload_first, t_download = main_job["t_upload"], main_job["t_download"]
    iters_step = int(sum(int(s.admm_iters.sum()) for s in sols))
    groups = h.launch_groups()
    group_of = h.agent_groups()
    bytes_agent = np.concatenate([algorithmic_bytes(w, s.admm_iters) for w, s in zip(worlds, sols)])
    iters_agent = np.concatenate([s.admm_iters for s in sols])

    # ---- N > 1: the other scaling's job, same K steps (`value_weak` beside a strong `value`, or the other way round)
    other_line = None
    if other is not None:
        o_jobs, o_sizes, _, _, o_worlds, _, _ = other
        oj = timed_job(o_worlds)
        o_iters = int(sum(int(s_.admm_iters.sum()) for s_ in oj["sols"]))
        o_el, o_it = over_ranks(oj["elapsed"], o_iters)
        other_line = {"scaling": "weak" if strong else "strong", "value": o_it * args.steps / o_el,
                      "ms_per_step": o_el / max(args.steps, 1) * 1e3, "worlds_total": len(o_jobs), "agents_total": int(sum(o_sizes)),
                      "admm_iterations_per_step_all_ranks": int(o_it)}

    # ---- the PCIe-inclusive DO phase of the batch (csdo.cc:111-148: preprocess + SolverDSQP), outside the timed region, host
    # wall clock around everything, nothing resident: bridge of every world (ONE csdo_preprocess_device_batch call: host
    # interpolation on a thread pool, pair search + planes on the device), then csdo_dsqp_solve_batch's three stages
    e2e = None
    if not args.no_e2e and rank == 0:
        items = [(*infos[i]["paths"], worlds[i].dimx, worlds[i].dimy, worlds[i].obstacles) for i in range(len(worlds))
                 if worlds[i].Na == len(infos[i]["paths"][2]) - 1]
        whole = [i for i in range(len(worlds)) if worlds[i].Na == len(infos[i]["paths"][2]) - 1]
        from csdotrajectoryplanning_amd.solver import interpolate_and_planes_batch_host
        h.interpolate_and_planes_batch(items[:2], w0.veh, w0.parm)      # (first call: device buffers, page-locked staging)
        t_bridge_dev = None
        for _ in range(3):                   # round 4's device bridge of all worlds in one call, for the record (the pipeline below
            t_b0 = time.perf_counter()       # uses the host threads' bridge, which round 6 made the faster one)
            h.interpolate_and_planes_batch(items, w0.veh, w0.parm)
            t_b1 = time.perf_counter() - t_b0
            t_bridge_dev = t_b1 if t_bridge_dev is None else min(t_bridge_dev, t_b1)
        best = None
        out_e2e = None                       # steady state: the caller's output arrays are reused from call to call
        for _ in range(3):
            time.sleep(0.05)
            t_b0 = time.perf_counter()
            bridged = interpolate_and_planes_batch_host(items, w0.veh, w0.parm)
            t_bridge_pool = time.perf_counter() - t_b0
            bw = [worlds[i] for i in range(len(worlds))]
            for i, (bworld, _, _) in zip(whole, bridged):
                bw[i] = bworld
            t_u0 = time.perf_counter()
            h.set_host_results(True)         # the kernels write their results into page-locked host memory: no D2H copy behind them
            try:
                h.upload(bw)
            finally:
                h.set_host_results(False)
            t_upload = time.perf_counter() - t_u0
            t_k0 = time.perf_counter()
            t_k = h.run(stream)
            t_kw = time.perf_counter() - t_k0
            t_d0 = time.perf_counter()
            out_e2e = h.download(out=out_e2e)
            t_dl = time.perf_counter() - t_d0
            tot = time.perf_counter() - t_b0
            if best is None or tot < best[0]:
                best = (tot, t_bridge_pool, t_upload, t_k, t_kw, t_dl, h.transfer_seconds())
        tot, t_bridge_pool, t_upload, t_k, t_kw, t_dl, xfer = best

        # ---- the same DO phase as ONE library call, STREAMED in chunks of worlds where the job allows it (csdo_do_phase: host bridge +
        # packing + H2D of chunk k + 1 under the solve of chunk k on csdo_dsqp_create_shared handles, results of a chunk back under the
        # later solves; DsqpHandle.do_phase_stream is the same pipeline driven from Python)
        streamed = None
        if len(whole) == len(worlds) and len(worlds) >= 3:
            out_s, best_s = None, None
            for _ in range(4):                 # (first pass: device buffers and page-locked staging of the chunk handles)
                time.sleep(0.05)               # a planner's DO phase follows a search: host threads asleep, nothing in flight
                sols_s, tm, _ = h.do_phase(items, w0.veh, w0.parm, out=out_s)      # csdo_do_phase: ONE library call
                out_s = sols_s
                if best_s is None or tm["total"] < best_s["total"]:
                    best_s = tm
            same = all(np.array_equal(a_.solutions, b_.solutions) and np.array_equal(a_.admm_iters, b_.admm_iters)
                       for a_, b_ in zip(sols_s, sols))
            streamed = {"total_ms": best_s["total"] * 1e3, "total_with_python_binding_ms": best_s.get("total_with_binding", best_s["total"]) * 1e3,
                        "first_launch_ms": best_s["first_launch"] * 1e3,
                        "kernels_done_ms": best_s["kernels_done"] * 1e3,
                        "chunks": [{"worlds": c_["worlds"], "bridge_host_ms": c_["bridge"] * 1e3,
                                    "upload_pack_h2d_ms": c_["upload"] * 1e3, "kernels_ms": c_["kernel"] * 1e3}
                                   for c_ in best_s["chunks"]],
                        "agent_qp_iterations_per_sec": iters_step / best_s["total"],
                        "in_chunks": bool(best_s["streamed"]),
                        "results_equal_the_resident_batch": bool(same),
                        "results_written_to_host_memory_by_the_kernels": True,
                        "entry": "csdo_do_phase",
                        "note": "best of 3 after a first pass, 50 ms of sleep in front of each; host wall clock around ONE library call "
                                "(csdo_do_phase), from the coarse paths to the results in the "
                                "caller's arrays; chunk k + 1 is bridged (host threads), packed and copied under chunk k's solve; "
                                "every workgroup writes its agent's results into page-locked host memory when the agent is done "
                                "(csdo_dsqp_set_host_results): nothing is copied behind the last kernel, only scattered; "
                                "a job of several kernel classes (in_chunks false) is bridged at once (host threads) and solved by "
                                "one launch instead"}
        e2e = {"total_ms": (streamed["total_ms"] if streamed else tot * 1e3),
               "streamed": streamed,
               "single_launch_total_ms": tot * 1e3,
               "bridge_host_pool_ms": t_bridge_pool * 1e3, "bridge_device_batch_ms": t_bridge_dev * 1e3, "upload_pack_h2d_ms": t_upload * 1e3,
               "solve_kernels_ms": t_k * 1e3, "solve_host_wall_ms": t_kw * 1e3, "download_d2h_scatter_ms": t_dl * 1e3,
               "upload_first_call_ms": t_upload_first * 1e3,
               "library_breakdown_ms": {k: v * 1e3 for k, v in xfer.items()},
               "agent_qp_iterations_per_sec": iters_step / (streamed["total_ms"] * 1e-3 if streamed else tot),
               "agent_qp_iterations_per_sec_single_launch": iters_step / tot,
               "note": "PCIe-inclusive DO phase of rank 0's batch, best of 3, host wall clock from the coarse paths to the "
                       "results in the caller's arrays; never `value`.  total_ms = the streamed form when the batch has at "
                       "least three whole worlds (`streamed`), else the single launch; the stage times below are the single "
                       "launch's, driven from Python (host-thread bridge of all whole worlds, pack + H2D, kernels writing their "
                       "results into page-locked host memory, scatter; one after the other; bridge_device_batch_ms: round 4's "
                       "device bridge of the same worlds, not part of the total)"}

    # ---- the authors' own acceptance of a result: the trajectory validator (device kernel), per world
    validation = None
    if rank == 0:
        ok_w = ok_obs = 0
        veh_hits = obs_hits = 0
        t_v0 = time.perf_counter()
        colliding = []      # WHO collides, and how their solves ended: per world with a hit, from the numpy validator
        from csdotrajectoryplanning_amd import results as results_mod
        for wi_, (w, s_) in enumerate(zip(worlds, sols)):
            rep = h.validate(s_.solutions, w.veh, w.obstacles, w.dimx, w.dimy)
            if rep.vehicle_collisions or rep.obstacle_collisions:
                who = {}
                results_mod.validate(s_.solutions, w.veh, w.obstacles, w.dimx, w.dimy, participants=who)
                agents_ = sorted({a_ for i_, j_, _ in who.get("vehicle_pairs", []) for a_ in (i_, j_)} |
                                 {a_ for a_, _ in who.get("obstacle_agents", [])})
                colliding.append({"world": wi_, "instance": infos[wi_].get("instance"),
                                  "vehicle_pairs": who.get("vehicle_pairs", []), "obstacle_agents": who.get("obstacle_agents", []),
                                  "agents": [{"agent": a_, "last_status": int(s_.last_status[a_]), "sqp_iterations": int(s_.sqp_iters[a_]),
                                              "admm_iterations": int(s_.admm_iters[a_])} for a_ in agents_]})
            ok_w += int(rep.ok)
            ok_obs += int(rep.obstacle_collisions == 0 and rep.out_of_map == 0)
            veh_hits += rep.vehicle_collisions
            obs_hits += rep.obstacle_collisions
        validation = {"worlds": len(worlds), "worlds_without_any_collision": ok_w,
                      "worlds_without_static_collision": ok_obs, "vehicle_collision_triples": int(veh_hits),
                      "obstacle_collision_triples": int(obs_hits), "solver_status_ok_worlds":
                          int(sum(1 for s_ in sols if abs(int(s_.solver_status)) <= 2)),
                      "validator_ms": (time.perf_counter() - t_v0) * 1e3,
                      "colliding": colliding,
                      "colliding_agents_by_last_status": {str(k_): v_ for k_, v_ in sorted(
                          __import__("collections").Counter(a_["last_status"] for c_ in colliding for a_ in c_["agents"]).items())},
                      "colliding_agents_that_ran_all_ten_qps": int(sum(1 for c_ in colliding for a_ in c_["agents"] if a_["sqp_iterations"] >= 10)),
                      "note": "rectangle/rectangle and disc/rectangle checks of the final trajectories (csdo_validate); "
                              "the stand-in's coarse paths (worlds the search does not solve) are not collision-free"}

    per_rank = None
    elapsed_max, iters_all = over_ranks(elapsed, iters_step)
    if dist is not None:
        mine = torch.tensor([float(sum(w.Na for w in worlds)), float(iters_step), kernel_s / max(args.steps, 1)],
                            dtype=torch.float64, device=dev)
        allr = [torch.zeros_like(mine) for _ in range(world_size)]
        dist.all_gather(allr, mine)
        per_rank = [{"rank": r, "agents": int(v[0]), "admm_iterations_per_step": int(v[1]),
                     "solve_kernels_ms": float(v[2]) * 1e3} for r, v in enumerate(allr)]

    if rank == 0:
        steps = max(args.steps, 1)
        value = iters_all * args.steps / elapsed_max
        kernel_avg = kernel_s / steps
        # dominant kernel = the launch group that executes the most algorithmic bytes
        gbytes = [float(bytes_agent[group_of == g].sum()) for g in range(len(groups))]
        gd = int(np.argmax(gbytes))
        dom = groups[gd]
        dom_avg = group_s[gd] / steps
        achieved = gbytes[gd] / dom_avg / 1e9
        traffic, hbm_frac, valu_frac, pmc_src = (None, None, None, None)
        from csdotrajectoryplanning_amd import _lib as _csdo_lib
        lib_hash = _csdo_lib.lib().csdo_source_hash().decode()
        if world_size == 1:
            # (the refined kernels are other kernels: their counters come from a profile of their own, `<workload>_refine<m>`)
            pmc_label = args.workload + ("_refine%d" % int(args.solve_refinement) if args.solve_refinement else "")
            traffic, hbm_frac, valu_frac, pmc_src = _newest_pmc(pmc_label, elapsed_max / steps * 1e3, lib_hash)
        # what the ADMM iterations cost in the two resources the kernel does use (counts per agent-iteration)
        K_agent = np.concatenate([(w.plane_off[1:] - w.plane_off[:-1]).astype("float64") for w in worlds])
        Nt_agent = np.concatenate([np.full(w.Na, float(w.Nt)) for w in worlds])
        it_f = iters_agent.astype("float64")
        flops_step = float((it_f * (718.0 * Nt_agent + 88.0 * K_agent + 2592.0)).sum())
        lds_bytes_step = float((it_f * 8.0 * (139.0 * Nt_agent + 42.0 * K_agent + 1440.0)).sum())
        FP64_PEAK_TF, LDS_PEAK_TBS = 78.6, 157.0
        # ---- the roofline object.  The contract's figure (algorithmic bytes per agent-iteration over the kernel's
        # duration against 8 TB/s) prices every iteration as if factor, rows and iterate streamed from HBM; the design keeps them
        # in registers and LDS, so that figure is NOMINAL and passes 1.0 as the kernel gets faster: it is reported under
        # `nominal_hbm`, flagged when it does.  `bound` / `achieved` / `peak` / `frac` describe the resource that is closest to its
        # own ceiling among those the kernel really uses - every one of them a fraction that cannot pass 1:
        #   valu_issue  SQ_INSTS_VALU x 4 cycles per SIMD-seconds available (PMC; every vector instruction priced as one 4-cycle
        #               issue slot of a lone wave; 1024 SIMDs x 2.4 GHz)            [from the newest matching profiles/ summary]
        #   hbm         counter bytes (2 x FETCH_SIZE + WRITE_SIZE) / time against 8 TB/s                             [PMC]
        #   fp64        F_iter flops of the executed iterations / time against 78.6 TFLOP/s                          [model]
        #   lds         L_iter bytes / time against 157 TB/s                                                         [model]
        fp64_ach = flops_step / kernel_avg / 1e12
        lds_ach = lds_bytes_step / kernel_avg / 1e12
        SIMD_CYCLES_PEAK = 1024 * 2.4e9
        resources = {
            "fp64": {"achieved": fp64_ach, "peak": FP64_PEAK_TF, "unit": "TFLOP/s", "frac": fp64_ach / FP64_PEAK_TF,
                     "source": "model: 718 Nt + 88 K + 2592 flop per agent-iteration (ADMM iterations only)"},
            "lds": {"achieved": lds_ach, "peak": LDS_PEAK_TBS, "unit": "TB/s", "frac": lds_ach / LDS_PEAK_TBS,
                    "source": "model: 8 (139 Nt + 42 K + 1440) bytes per agent-iteration (ADMM iterations only)"},
        }
        if hbm_frac is not None and traffic is not None:
            resources["hbm"] = {"achieved": hbm_frac * HBM_PEAK_GBS, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": hbm_frac,
                                "source": "rocprofv3 FETCH_SIZE x 2 + WRITE_SIZE, %s" % pmc_src}
        if valu_frac is not None:
            resources["valu_issue"] = {"achieved": valu_frac * SIMD_CYCLES_PEAK / 1e9, "peak": SIMD_CYCLES_PEAK / 1e9,
                                       "unit": "G SIMD-cycles/s", "frac": valu_frac,
                                       "source": "rocprofv3 SQ_INSTS_VALU x 4 cycles, %s" % pmc_src}
        bound = max(resources, key=lambda k: resources[k]["frac"])
        nominal = achieved / HBM_PEAK_GBS
        roofline = {"bound": bound, "achieved": resources[bound]["achieved"], "peak": resources[bound]["peak"],
                    "unit": resources[bound]["unit"], "frac": resources[bound]["frac"], "traffic": traffic,
                    "binding": "dependency latency of one workgroup per agent (cross-lane exchange + barriers + dependent fp64 "
                               "chains): no throughput resource is near its ceiling; `bound` names the one that is closest "
                               "(highest measured fraction), `resources` lists all of them",
                    "resources": resources,
                    "nominal_hbm": {"achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": nominal,
                                    "exceeds_peak": bool(nominal > 1.0),
                                    "note": "SURVEY 8(d)'s contract figure: W_iter = 2280 Nt + 416 K_a algorithmic bytes per "
                                            "agent-iteration of the dominant kernel's launch / its average duration against 8 TB/s; "
                                            "nominal - the working set is on-chip, so it is not a ceiling and can pass 1"},
                    "hbm_counter_frac": hbm_frac, "valu_fp64_issue_frac": valu_frac, "pmc_source": pmc_src,
                    "kernel": "dsqp_agent_kernel<%d, %d, true>" % (dom["threads"], dom["residency_mode"]),
                    "kernel_avg_ms": dom_avg * 1e3, "algorithmic_bytes_per_launch": gbytes[gd],
                    "all_kernels": {"algorithmic_bytes_per_step": float(bytes_agent.sum()), "avg_ms": kernel_avg * 1e3,
                                    "achieved": float(bytes_agent.sum()) / kernel_avg / 1e9}}
        # floor of the strong-scaling curve: the job's longest agent (device time of the last step)
        floor_ms = max(float(s_.t_max_individual) for s_ in sols) * 1e3
        Nts = sorted(w.Nt for w in worlds)
        n_agents = int(sum(w.Na for w in worlds))
        wl_names = {"map100": "map100by100/agents50/obstacle set", "map50": "map50by50/agents25/obstacle set",
                    "room50": "room/agents50 set, ex0..ex11 (238 wall obstacles per world)",
                    "agents100": "map100by100/agents100/obstacle set, ex0..ex11 (100 vehicles per world)",
                    "synth1024": "synthetic 1024-agent stress batch (21 worlds = ex0..ex20 of the map100by100/agents50/"
                                 "obstacle set, truncated to 1024 agents)"}
        out = {
            "metric": "agent_qp_iterations_per_sec",
            "value": value,
            "unit": "agent-QP-iterations/s",
            "n_gpus": world_size,
            "rccl_ranks": (dist.get_world_size() if dist is not None else None),
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed_max / steps * 1e3,
            "higher_is_better": True,
            "scaling": "strong" if strong else "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "solve_refinement": int(args.solve_refinement),
            "kernel_source_hash": lib_hash,
            "kernel_source_hash_of_tree": _csdo_lib.source_hash_of_tree(),
            "strong_scaling_floor_ms": floor_ms,
            ("value_weak" if strong else "value_strong"): (other_line["value"] if other_line else None),
            "other_scaling": other_line,
            "data": "the reference's benchmark instance files (tests/golden/instances = benchmark/map100by100, map50by50, room of "
                    "the reference), no synthetic maps; coarse paths: %d of %d worlds from this repository's own front end "
                    "(stored), %d from the seeded stand-in generator (instances the search does not solve)"
                    % (n_front, len(infos), len(infos) - n_front),
            "value_e2e": (e2e["agent_qp_iterations_per_sec"] if e2e else None),
            "value_is": "kernels only, inputs resident in HBM (the contract's `value`); value_e2e = the same iterations over the "
                        "PCIe-inclusive DO phase (do_phase_e2e.total_ms)",
            "config": {
                "workload": "%s: %d worlds, %d agents on rank 0 in one batch, Nt %d..%d, %d inter-vehicle planes; "
                            "benchmark instance files; %s"
                            % (wl_names[args.workload], len(worlds), n_agents, Nts[0], Nts[-1],
                               int(sum(int(w.plane_off[-1]) for w in worlds)), guesses),
                "workload_key": args.workload,
                "worlds_total": len(jobs), "agents_total": int(sum(sizes)),
                "agents_rank0": n_agents,
                "admm_iterations_per_step_rank0": iters_step,
                "sqp_iterations_rank0": int(sum(int(s.sqp_iters.sum()) for s in sols)),
                "launch_groups_rank0": [{"agents": g["n_agents"], "threads": g["threads"],
                                         "lds_residency_mode": g["residency_mode"], "lds_bytes": g["lds_bytes"],
                                         "avg_ms": group_s[i] / steps * 1e3,
                                         "admm_iterations": int(iters_agent[group_of == i].sum())}
                                        for i, g in enumerate(groups)],
                "parallelism": ("1 GPU" if world_size == 1 else
                                ("one copy of the workload, agents sharded in contiguous blocks over %d ranks" % world_size
                                 if strong else
                                 "%d copies of the workload, agents sharded in contiguous blocks over %d ranks"
                                 % (copies, world_size))),
                "collective": ("all_gather(final trajectories, device pointers) per step, on a second stream: the gather of step k "
                               "runs beside the solve of step k + 1" +
                               (" on a ONE-rank nccl group (--force-dist)" if world_size == 1 else "")) if sharded else "none",
                "shard_balance": shard_balance,
                "per_rank": per_rank,
            },
            "single_instance": single,
            "pipeline_single_instance": pipeline,
            "do_phase_e2e": e2e,
            "validation": validation,
            "batch_ms": {"load_paths_and_bridge_host": t_pre * 1e3, "solve_kernels": kernel_avg * 1e3,
                         "download_d2h": t_download * 1e3},
            "roofline": roofline,
        }
        if not args.no_cpu_baseline and world_size == 1:
            from tests import oracle_lib
            cores = effective_cpus()
            # bounded sample: the whole batch when the host has the cores for it, else a prefix sized for ~20 s
            est_rate = 1.9e4 * cores                           # measured: 19 k agent-iterations/s per core
            n_s = len(worlds)
            while n_s > 1 and sum(int(s.admm_iters.sum()) for s in sols[:n_s]) / est_rate > 30.0:
                n_s -= 1
            sample = worlds[:n_s]
            tc0 = time.perf_counter()
            so = oracle_lib.solve_batch(sample, cores)
            t_all = time.perf_counter() - tc0
            it_cpu = float(sum(int(s.admm_iters.sum()) for s in so))
            tc0 = time.perf_counter()
            so1 = oracle_lib.solve(w0, 1)
            t_one = time.perf_counter() - tc0
            tc0 = time.perf_counter()
            oracle_lib.solve(w0, min(cores, w0.Na))
            t_inst = time.perf_counter() - tc0
            out["cpu_baseline"] = {"value": it_cpu / t_all, "unit": "agent-QP-iterations/s", "cores": cores,
                                   "os_cpu_count": os.cpu_count(),
                                   "kind": "port",
                                   "sample": "the oracle (OSQP-0.6.3-equivalent restatement) over the first %d of the "
                                             "batch's %d worlds = %d agents, ONE thread pool of %d threads over all "
                                             "those agents (%.1f s of wall time)"
                                             % (len(sample), len(worlds), int(sum(w.Na for w in sample)), cores, t_all),
                                   "single_core_value": float(so1.admm_iters.sum()) / t_one,
                                   "do_phase_ms_single_instance_one_thread_per_agent": t_inst * 1e3,
                                   "do_phase_ms_single_instance_single_core": t_one * 1e3}
        line = json.dumps(out)
    h.close()
    if dist is not None:
        dist.destroy_process_group()
    if rank == 0:       # the ONE line, LAST: RCCL leaves a version banner in C stdio's buffer, which would otherwise come out at exit
        sys.stdout.flush()
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        print(line, flush=True)


if __name__ == "__main__":
    main()
