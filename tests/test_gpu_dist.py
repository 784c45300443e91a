"""The N > 1 step of bench.py on the hardware there is: a ONE-rank "nccl" (= RCCL) process group on the metered GPU, the solver's
kernels, the copy out of its device buffer and all_gather_into_tensor ordered on one torch stream, the gathered bytes compared
with what csdo_dsqp_download returns.  (Two ranks need two GPUs; the sharding logic itself runs with gloo in test_sharding.py.)"""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_one_rank_rccl_all_gather_on_the_solvers_device_buffer(veh_parm):
    import torch
    import torch.distributed as dist
    from csdotrajectoryplanning_amd import sharding, workloads
    from csdotrajectoryplanning_amd.solver import DsqpHandle
    sys.path.insert(0, ROOT)
    import bench
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(29600 + os.getpid() % 300)
    torch.cuda.set_device(0)
    dist.init_process_group(backend="nccl", rank=0, world_size=1)
    try:
        dev = torch.device("cuda", 0)
        worlds = [workloads.map100_world(1)[0], workloads.map50_world(2)[0].subset(3, 20)]
        h = DsqpHandle(0)
        h.upload(worlds)
        ptr, n_dbl = h.device_solutions()
        assert n_dbl == sum(w.Na * w.Nt * 6 for w in worlds)
        sol_dev = torch.as_tensor(bench._DevArray(ptr, n_dbl), device=dev)
        tstream = torch.cuda.Stream(device=dev)
        with torch.cuda.stream(tstream):
            fg = sharding.FlatGather(n_dbl, dist, dev)
        tstream.synchronize()
        for _ in range(2):                       # twice: the second gather must see the second run's bytes, not stale ones
            h.run(tstream.cuda_stream)
            with torch.cuda.stream(tstream):
                out = fg.gather(sol_dev)
        tstream.synchronize()
        sols = h.download()
        ref = np.concatenate([s.solutions.reshape(-1) for s in sols])
        got = fg.parts()[0].cpu().numpy()
        assert got.shape == ref.shape and np.array_equal(got, ref)
        assert out.shape == (1, n_dbl)
        h.close()
    finally:
        dist.destroy_process_group()


def test_bench_force_dist_line():
    """bench.py --force-dist end to end: one JSON line that says the collective ran, with the work-balanced plan in it."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--force-dist", "--instances", "6", "--steps", "2",
                          "--warmup", "1", "--no-cpu-baseline", "--no-e2e", "--skip-single-instance"],
                         capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and "ONE-rank nccl" in d["config"]["collective"]
    assert d["config"]["shard_balance"]["estimated_work_max_over_mean"] == 1.0
    assert d["value"] > 0 and d["config"]["agents_rank0"] == 300
