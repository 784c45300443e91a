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


def shard_bounds_weighted(weights, world_size):
    """Contiguous blocks [lo, hi) per rank of near-equal total WEIGHT: block r ends where the running sum first reaches
    (r + 1) / world_size of the total ("balance by Nt (13 + 4 K_a / Nt) if K is skewed"; here the weights are the
    launcher's own work estimate, csdo_dsqp_estimate_work).  Agents take 3-6x the median time when their initial guess sits in
    tight spots, so equal COUNTS leave the ranks unequal; every rank keeps at least one item while there are enough."""
    w = np.asarray(weights, dtype=np.float64)
    n = len(w)
    if n == 0 or world_size <= 0:
        return [(0, 0)] * max(world_size, 0)
    cum = np.cumsum(np.maximum(w, 0.0))
    total = float(cum[-1])
    if not total > 0.0:
        return shard_bounds(n, world_size)
    cuts = [0]
    for r in range(1, world_size):
        target = total * r / world_size
        i = int(np.searchsorted(cum, target, side="left")) + 1        # first prefix whose sum reaches the target
        # take the closer of the two candidate cuts, stay monotone, leave enough items for the ranks behind
        if i - 1 > cuts[-1] and abs(cum[i - 2] - target) <= abs(cum[i - 1] - target):
            i -= 1
        i = max(i, cuts[-1] + (1 if n - cuts[-1] > world_size - r else 0))
        i = min(i, n - min(world_size - r, n - cuts[-1]) if n - cuts[-1] > world_size - r else n)
        cuts.append(min(max(i, cuts[-1]), n))
    cuts.append(n)
    return [(cuts[r], cuts[r + 1]) for r in range(world_size)]


def shard_world(world, rank, world_size):
    lo, hi = shard_bounds(world.Na, world_size)[rank]
    return (world.subset(lo, hi) if hi > lo else None), (lo, hi)


def shard_batch_plan(world_sizes, rank, world_size, weights=None):
    """Contiguous block of the batch's agents (worlds concatenated in order) owned by `rank`: a list of
    (world index, lo, hi) with agents [lo, hi) of that world; only worlds that overlap the block appear.
    weights: per-agent work estimates (len = sum(world_sizes)): blocks of equal work instead of equal agent count."""
    total = int(sum(world_sizes))
    if weights is not None:
        assert len(weights) == total
        lo, hi = shard_bounds_weighted(weights, world_size)[rank]
    else:
        lo, hi = shard_bounds(total, world_size)[rank]
    plan, first = [], 0
    for w, n in enumerate(world_sizes):
        a, b = max(lo, first), min(hi, first + n)
        if b > a:
            plan.append((w, a - first, b - first))
        first += n
    return plan


def shard_batch(worlds, rank, world_size):
    """This rank's share of a batch of worlds: whole worlds where the block covers them, World.subset otherwise."""
    out = []
    for w, lo, hi in shard_batch_plan([x.Na for x in worlds], rank, world_size):
        out.append(worlds[w] if (lo == 0 and hi == worlds[w].Na) else worlds[w].subset(lo, hi))
    return out


class FlatGather:
    """The step-end collective of a sharded batch: all_gather of per-rank flat fp64 buffers of different lengths (a rank's
    packed trajectories, [sum Nt][6]) with preallocated buffers - padded to the longest, one all_gather_into_tensor per
    call, on whatever stream is current.  `local` tensors live where `device` says (device memory on the GPU path: the
    collective reads the solver's own output buffer through a device-to-device copy, no host hop)."""

    def __init__(self, n_local, dist, device):
        import torch
        self.dist, self.n_local = dist, int(n_local)
        ws = dist.get_world_size()
        lens = [torch.zeros(1, dtype=torch.int64, device=device) for _ in range(ws)]
        dist.all_gather(lens, torch.tensor([self.n_local], dtype=torch.int64, device=device))
        self.lengths = [int(x) for x in lens]
        self.n_max = max(self.lengths)
        self.send = torch.zeros(self.n_max, dtype=torch.float64, device=device)
        self.out = torch.empty(ws * self.n_max, dtype=torch.float64, device=device)

    def gather(self, local):
        self.stage(local)
        return self.collect()

    # The two halves of gather(), for a step that overlaps the collective with the NEXT solve: stage() on the stream the solve
    # was launched on (the copy out of the solver's buffer is then ordered behind the solve, and the buffer is free for the
    # next one), collect() on a second stream that waits for the copy; the next stage() must wait for that collect().
    def stage(self, local):
        self.send[:self.n_local].copy_(local[:self.n_local])

    def collect(self):
        self.dist.all_gather_into_tensor(self.out, self.send)
        return self.out.view(len(self.lengths), self.n_max)

    def parts(self):
        """The ranks' buffers without padding, in rank order (views into the gathered tensor)."""
        v = self.out.view(len(self.lengths), self.n_max)
        return [v[r, :n] for r, n in enumerate(self.lengths)]


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
