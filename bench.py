#!/usr/bin/env python
"""bench.py — DO-phase throughput of the MI355X backend on BASELINE.json's map100by100/agents50/obstacle set.

One step = one pass of the hot path (the whole SolverDSQP-equivalent solve: initial corridors + every agent's SQP with
its ADMM QPs) over the 60 instances of the set, 3000 agents, in ONE batch whose inputs are already resident in HBM
(BASELINE.json configs[2]: "map100by100 agents50 obstacle set, 1xMI355X (large per-GPU batch)").  Agents are independent
once their separating planes are fixed, so every agent is one workgroup and the batch fills the 256 CUs.
With N > 1 ranks (torch.distributed.run, one process per GPU) every rank owns its own copy of the set with differently
seeded initial guesses (weak scaling; no data-path exchange) and the step ends with the only real collective of the
path, the RCCL all-gather of the final trajectories.  Rank 0 prints ONE JSON line.

metric  = agent-QP-iterations/sec: ADMM iterations executed by all agents of all ranks / wall time of the K steps.
          The DO-phase time of a single 50-agent instance (the metric's "DO-phase ms") is measured outside the timed
          region and reported as single_instance.
roofline: SURVEY 8(d) algorithmic bytes W_iter = 2280*Nt + 416*K_a per agent-iteration, summed over the iterations the
          dominant kernel's launch executes, divided by that kernel's average duration (HIP events recorded on the stream
          the kernel is launched on, inside csdo_dsqp_run) against 8 TB/s.  `traffic` is the HBM traffic per launch from
          rocprofv3 PMC passes (profiles/r01_pmc_summary.json), or null when that file is absent.
cpu_baseline: the oracle (CPU restatement of the reference + OSQP 0.6.3, kind "port") on a bounded sample of the same
          worlds, rank 0 only.
"""
import argparse
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


def _make_world(args):
    k, seed_offset = args
    from csdotrajectoryplanning_amd import workloads
    world, info = workloads.map100_world(k, seed_offset=seed_offset)
    return world, info["paths"], info["n_planes"]


def algorithmic_bytes(world, admm_iters):
    """Per agent: iterations * W_iter, W_iter = 2280*Nt + 416*K_a bytes."""
    K = (world.plane_off[1:] - world.plane_off[:-1]).astype("float64")
    return admm_iters.astype("float64") * (2280.0 * world.Nt + 416.0 * K)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--instances", type=int, default=60, help="instances of the set per GPU (60 = the whole set)")
    ap.add_argument("--setup-procs", type=int, default=32, help="processes building the worlds (1: in-process, no fork; "
                                                                  "use that under rocprofv3)")
    ap.add_argument("--skip-single-instance", action="store_true",
                    help="do not measure ex0 alone before the batch (profiling runs: keeps the kernel statistics of the "
                         "batch launches free of the small launches)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world_size = int(os.environ.get("WORLD_SIZE", "1"))

    # torch first: it brings its own HIP runtime, which must be the one libcsdo_hip.so binds to (loading the library
    # before torch puts two runtimes into the process and aborts under rocprofv3); importing does not touch the GPU
    import torch

    # ---- workload (host side, before anything touches the GPU: the pool forks) ----
    from multiprocessing import get_context
    n_inst = max(1, min(args.instances, 60))
    t_pre0 = time.perf_counter()
    jobs = [(k, 60 * rank) for k in range(n_inst)]
    procs = min(n_inst, os.cpu_count() or 1, max(args.setup_procs, 1))
    if procs > 1:
        with get_context("fork").Pool(procs) as pool:
            built = pool.map(_make_world, jobs)
    else:
        built = [_make_world(j) for j in jobs]
    worlds = [b[0] for b in built]
    t_pre = time.perf_counter() - t_pre0

    import numpy as np
    from csdotrajectoryplanning_amd.solver import DsqpHandle, interpolate_and_planes
    from csdotrajectoryplanning_amd.synth import GENERATOR_NAME

    dist = None
    torch.cuda.set_device(local_rank)
    if world_size > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", rank=rank, world_size=world_size)
    dev = torch.device("cuda", local_rank)

    # bridge (host preprocess of the reference, csdo.cc:116-129) timed on the first instance
    st, ac, po, G = built[0][1]
    w0 = worlds[0]
    t_b0 = time.perf_counter()
    interpolate_and_planes(st, ac, po, G, w0.veh, w0.parm, w0.dimx, w0.dimy, w0.obstacles)
    t_bridge = time.perf_counter() - t_b0

    h = DsqpHandle(local_rank)
    stream = torch.cuda.current_stream().cuda_stream

    # ---- the metric's "DO-phase ms, 50-agent instance": ex0 alone, outside the timed region ----
    single = None
    if not args.skip_single_instance:
        t_u0 = time.perf_counter()
        h.upload([w0])
        t_upload1 = time.perf_counter() - t_u0
        h.run(stream)
        single_kernel = min(h.run(stream) for _ in range(3))
        t_d0 = time.perf_counter()
        sol0 = h.download()[0]
        t_download1 = time.perf_counter() - t_d0
        single = {
            "workload": "map100by100/agents50/obstacle ex0 alone: Na=50, Nt=%d, %d planes" % (w0.Nt, int(w0.plane_off[-1])),
            "admm_iterations": int(sol0.admm_iters.sum()), "solver_status": int(sol0.solver_status),
            "do_phase_ms": {"bridge_host": t_bridge * 1e3, "upload_h2d": t_upload1 * 1e3,
                            "solve_kernel": single_kernel * 1e3, "download_d2h": t_download1 * 1e3,
                            "total": (t_bridge + t_upload1 + single_kernel + t_download1) * 1e3,
                            "max_individual_agent": sol0.t_max_individual * 1e3},
            "agent_qp_iterations_per_sec": float(sol0.admm_iters.sum()) / single_kernel,
        }

    # ---- the batch ----
    t_u0 = time.perf_counter()
    h.upload(worlds)
    t_upload = time.perf_counter() - t_u0
    ptr, n_dbl = h.device_solutions()
    sol_dev = torch.as_tensor(_DevArray(ptr, n_dbl), device=dev)
    gathered = None
    if world_size > 1:
        sizes = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world_size)]
        dist.all_gather(sizes, torch.tensor([n_dbl], dtype=torch.int64, device=dev))
        n_max = int(max(int(s) for s in sizes))          # ranks differ slightly in sum(Nt): pad to the largest
        send = torch.zeros(n_max, dtype=torch.float64, device=dev)
        gathered = torch.empty(world_size * n_max, dtype=torch.float64, device=dev)

    def step():
        ks = h.run(stream)
        if world_size > 1:
            send[:n_dbl].copy_(sol_dev)
            dist.all_gather_into_tensor(gathered, send)
        return ks

    for _ in range(args.warmup):
        step()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    kernel_s = 0.0
    group_s = None
    for _ in range(args.steps):
        kernel_s += step()
        gs = [g["seconds"] for g in h.launch_groups()]
        group_s = gs if group_s is None else [a + b for a, b in zip(group_s, gs)]
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    elapsed = time.perf_counter() - t0

    t_d0 = time.perf_counter()
    sols = h.download()
    t_download = time.perf_counter() - t_d0
    iters_step = int(sum(int(s.admm_iters.sum()) for s in sols))
    groups = h.launch_groups()
    group_of = h.agent_groups()
    bytes_agent = np.concatenate([algorithmic_bytes(w, s.admm_iters) for w, s in zip(worlds, sols)])
    iters_agent = np.concatenate([s.admm_iters for s in sols])

    if dist is not None:
        t = torch.tensor([elapsed, float(iters_step)], dtype=torch.float64, device=dev)
        tmax = t.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        tsum = t.clone()
        dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
        elapsed_max = float(tmax[0])
        iters_all = float(tsum[1])
    else:
        elapsed_max, iters_all = elapsed, float(iters_step)

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
        traffic = None
        pmc = os.path.join(ROOT, "profiles", "r01_pmc_summary.json")
        if os.path.exists(pmc):
            try:
                with open(pmc) as f:
                    traffic = json.load(f).get("hbm_bytes_per_launch_dominant_kernel")
            except Exception:
                traffic = None
        Nts = sorted(w.Nt for w in worlds)
        out = {
            "metric": "agent_qp_iterations_per_sec",
            "value": value,
            "unit": "agent-QP-iterations/s",
            "n_gpus": world_size,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed_max / steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": "map100by100/agents50/obstacle set per GPU: %d instances (ex0..ex%d) x 50 agents = %d agents "
                            "in one batch, Nt %d..%d, %d inter-vehicle planes (rank 0); benchmark instance files + %s "
                            "initial guesses" % (n_inst, n_inst - 1, sum(w.Na for w in worlds), Nts[0], Nts[-1],
                                                 int(sum(b[2] for b in built)), GENERATOR_NAME),
                "instances_per_gpu": n_inst, "agents_per_gpu": int(sum(w.Na for w in worlds)),
                "admm_iterations_per_step_rank0": iters_step,
                "sqp_iterations_rank0": int(sum(int(s.sqp_iters.sum()) for s in sols)),
                "launch_groups_rank0": [{"agents": g["n_agents"], "threads": g["threads"],
                                         "lds_residency_mode": g["residency_mode"], "lds_bytes": g["lds_bytes"],
                                         "avg_ms": group_s[i] / steps * 1e3,
                                         "admm_iterations": int(iters_agent[group_of == i].sum())}
                                        for i, g in enumerate(groups)],
                "collective": "all_gather(final trajectories) per step" if world_size > 1 else "none",
            },
            "single_instance": single,
            "batch_ms": {"front_end_stand_in_and_bridge_host": t_pre * 1e3, "upload_h2d": t_upload * 1e3,
                         "solve_kernels": kernel_avg * 1e3, "download_d2h": t_download * 1e3},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel": "dsqp_agent_kernel<%d, %d, true>" % (dom["threads"], dom["residency_mode"]),
                         "kernel_avg_ms": dom_avg * 1e3, "algorithmic_bytes_per_launch": gbytes[gd],
                         "all_kernels": {"algorithmic_bytes_per_step": float(bytes_agent.sum()),
                                         "avg_ms": kernel_avg * 1e3,
                                         "achieved": float(bytes_agent.sum()) / kernel_avg / 1e9}},
        }
        if not args.no_cpu_baseline:
            from tests import oracle_lib
            cores = os.cpu_count() or 1
            sample = worlds[:min(16, len(worlds))]
            tc0 = time.perf_counter()
            it_cpu = 0.0
            for w in sample:
                it_cpu += float(oracle_lib.solve(w, cores).admm_iters.sum())
            t_all = time.perf_counter() - tc0
            tc0 = time.perf_counter()
            so1 = oracle_lib.solve(w0, 1)
            t_one = time.perf_counter() - tc0
            out["cpu_baseline"] = {"value": it_cpu / t_all, "unit": "agent-QP-iterations/s", "cores": cores,
                                   "kind": "port",
                                   "sample": "the oracle (OSQP-0.6.3-equivalent restatement, one agent per thread) over "
                                             "the first %d instances of the same batch, one instance after the other "
                                             "(%.1f s of CPU wall time)" % (len(sample), t_all),
                                   "single_core_value": float(so1.admm_iters.sum()) / t_one,
                                   "do_phase_ms_all_cores_per_instance": t_all / len(sample) * 1e3,
                                   "do_phase_ms_single_core_ex0": t_one * 1e3}
        print(json.dumps(out))
    h.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
