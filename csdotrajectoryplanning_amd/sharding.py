"""Multi-GPU execution: the DO phase shards by agent (or by world) with no data-path exchange — after the planes are
fixed every agent's SQP is independent (sqp/dsqp_solver.cc:36-43,1198-1205).  The only collective is the gather of
the final trajectories.  One process per GPU; torch.distributed carries the gather (backend "nccl" = RCCL
over xGMI on the GPU box, "gloo" in the CPU tests)."""
import numpy as np


def shard_bounds(n_items, world_size):
    """Contiguous blocks [lo, hi) per rank, sizes differ by at most one (SURVEY 8e partitioning)."""
    base, rem = divmod(n_items, world_size)
    bounds, lo = [], 0
    for r in range(world_size):
        hi = lo + base + (1 if r < rem else 0)
        bounds.append((lo, hi))
        lo = hi
    return bounds


def shard_world(world, rank, world_size):
    lo, hi = shard_bounds(world.Na, world_size)[rank]
    return (world.subset(lo, hi) if hi > lo else None), (lo, hi)


def gather_solutions(local_solutions, n_total, Nt, rank, world_size, dist, device=None):
    """all_gather of per-rank [Na_r, Nt, 6] blocks into [n_total, Nt, 6] on every rank.  Blocks are padded to the
    largest shard so one all_gather_into_tensor moves everything (payload: 50 agents x 169 steps x 6 x 8 B = 0.4 MB)."""
    import torch
    bounds = shard_bounds(n_total, world_size)
    cap = max(hi - lo for lo, hi in bounds)
    if isinstance(local_solutions, np.ndarray):
        local = torch.from_numpy(np.ascontiguousarray(local_solutions))
    else:
        local = local_solutions
    if device is not None:
        local = local.to(device)
    pad = torch.zeros((cap, Nt, 6), dtype=torch.float64, device=local.device)
    n_loc = bounds[rank][1] - bounds[rank][0]
    if n_loc:
        pad[:n_loc] = local.reshape(n_loc, Nt, 6)
    out = torch.empty((world_size * cap, Nt, 6), dtype=torch.float64, device=local.device)
    dist.all_gather_into_tensor(out, pad)
    parts = [out[r * cap: r * cap + (hi - lo)] for r, (lo, hi) in enumerate(bounds)]
    return torch.cat(parts, dim=0)
