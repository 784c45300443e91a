#!/usr/bin/env python
"""bench.py — DO-phase throughput of the MI355X backend on BASELINE.json's 50-agent configuration.

One step = one pass of the hot path (the whole SolverDSQP-equivalent solve: initial corridors + every agent's SQP with
its ADMM QPs) over one 50-agent map100by100 instance whose inputs are already resident in HBM.  With N > 1 ranks
(torch.distributed.run, one process per GPU) every rank owns its own 50-agent world (ex{rank}: weak scaling, agents
are independent so there is no data-path exchange) and the step ends with the only real collective of the path, the
RCCL all-gather of the final trajectories.  Rank 0 prints ONE JSON line.

metric  = agent-QP-iterations/sec: ADMM iterations executed by all agents of all ranks / wall time of the K steps.
roofline: SURVEY 8(d) algorithmic bytes W_iter = 2280*Nt + 416*K_a per agent-iteration, summed over the iterations one
          launch executes, divided by the kernel's average duration (HIP events on the launch stream, taken inside
          csdo_dsqp_run) against 8 TB/s.
cpu_baseline: the oracle (CPU restatement of the reference + OSQP 0.6.3, kind "port") on the same world, rank 0 only.
"""
import argparse
import ctypes
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


def algorithmic_bytes(world, admm_iters):
    """Sum over agents of iterations * W_iter, W_iter = 2280*Nt + 416*K_a bytes."""
    K = (world.plane_off[1:] - world.plane_off[:-1]).astype("float64")
    return float((admm_iters.astype("float64") * (2280.0 * world.Nt + 416.0 * K)).sum())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import numpy as np
    import torch
    from csdotrajectoryplanning_amd import workloads
    from csdotrajectoryplanning_amd.solver import DsqpHandle
    from csdotrajectoryplanning_amd.synth import GENERATOR_NAME

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world_size = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    if world_size > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group(backend="nccl", rank=rank, world_size=world_size)
    else:
        torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    # ---- workload: map100by100/agents50/obstacle, instance ex{rank} (8 instance files are shipped) ----
    t_pre0 = time.perf_counter()
    world, info = workloads.map100_world(rank % 8)
    t_pre = time.perf_counter() - t_pre0  # includes the front-end stand-in; the bridge alone is timed below
    st, ac, po, G = info["paths"]
    from csdotrajectoryplanning_amd.solver import interpolate_and_planes
    t_b0 = time.perf_counter()
    interpolate_and_planes(st, ac, po, G, world.veh, world.parm, world.dimx, world.dimy, world.obstacles)
    t_bridge = time.perf_counter() - t_b0

    h = DsqpHandle(local_rank)
    t_u0 = time.perf_counter()
    h.upload([world])
    t_upload = time.perf_counter() - t_u0
    stream = torch.cuda.current_stream().cuda_stream
    ptr, n_dbl = h.device_solutions()
    sol_dev = torch.as_tensor(_DevArray(ptr, n_dbl), device=dev)
    gathered = torch.empty(world_size * n_dbl, dtype=torch.float64, device=dev) if world_size > 1 else None

    def step():
        ks = h.run(stream)
        if world_size > 1:
            dist.all_gather_into_tensor(gathered, sol_dev)
        return ks

    for _ in range(args.warmup):
        step()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    kernel_s = 0.0
    for _ in range(args.steps):
        kernel_s += step()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    elapsed = time.perf_counter() - t0

    t_d0 = time.perf_counter()
    sol = h.download()[0]
    t_download = time.perf_counter() - t_d0
    iters_step = int(sol.admm_iters.sum())
    alg_bytes = algorithmic_bytes(world, sol.admm_iters)

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
        value = iters_all * args.steps / elapsed_max
        kernel_avg = kernel_s / max(args.steps, 1)
        achieved = alg_bytes / kernel_avg / 1e9
        out = {
            "metric": "agent_qp_iterations_per_sec",
            "value": value,
            "unit": "agent-QP-iterations/s",
            "n_gpus": world_size,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed_max / max(args.steps, 1) * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": "map100by100/agents50/obstacle ex{rank} per GPU: Na=50, Nt=%d, %d inter-vehicle planes "
                            "(rank 0); benchmark instance file + %s initial guesses" % (world.Nt, int(world.plane_off[-1]),
                                                                                      GENERATOR_NAME),
                "agents_per_gpu": world.Na, "horizon_Nt": world.Nt, "admm_iterations_per_step_rank0": iters_step,
                "sqp_iterations_rank0": int(sol.sqp_iters.sum()), "solver_status_rank0": int(sol.solver_status),
                "collective": "all_gather(final trajectories) per step" if world_size > 1 else "none",
            },
            "do_phase_ms": {"bridge_host": t_bridge * 1e3, "upload_h2d": t_upload * 1e3,
                            "solve_kernel": kernel_avg * 1e3, "download_d2h": t_download * 1e3,
                            "total": (t_bridge + t_upload + kernel_avg + t_download) * 1e3,
                            "max_individual_agent": sol.t_max_individual * 1e3},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                         "kernel": "dsqp_agent_kernel<256>", "kernel_avg_ms": kernel_avg * 1e3,
                         "algorithmic_bytes_per_launch": alg_bytes},
        }
        if not args.no_cpu_baseline:
            from tests import oracle_lib
            cores = os.cpu_count() or 1
            tc0 = time.perf_counter()
            so1 = oracle_lib.solve(world, cores)
            t_all = time.perf_counter() - tc0
            tc0 = time.perf_counter()
            oracle_lib.solve(world, 1)
            t_one = time.perf_counter() - tc0
            it_cpu = float(so1.admm_iters.sum())
            out["cpu_baseline"] = {"value": it_cpu / t_all, "unit": "agent-QP-iterations/s", "cores": cores,
                                   "kind": "port",
                                   "sample": "one pass of the oracle (OSQP-0.6.3-equivalent restatement, one agent per "
                                             "thread) over the same 50-agent world",
                                   "single_core_value": it_cpu / t_one, "do_phase_ms_all_cores": t_all * 1e3,
                                   "do_phase_ms_single_core": t_one * 1e3}
        print(json.dumps(out))
    h.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
