"""bench.py --gpus N starts its N ranks itself (run as `python bench.py --gpus 8` it measured ONE GPU).  CPU check with
--dry: gloo, the lane-serial host build as the solver; what is checked is the launch, the plan, the gather and the line."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra, env=None, timeout=600):
    e = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "map50", "--instances", "2", "--steps", "1",
                           "--warmup", "0", "--setup-procs", "1"] + extra, env=e, capture_output=True, text=True, timeout=timeout)


def test_gpus_2_starts_two_ranks_by_itself():
    out = _run(["--gpus", "2", "--dry"])
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout                       # ONE line, rank 0's
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["gloo_ranks"] == 2 and d["dry"] is True and d["value"] is None
    ranks = d["config"]["per_rank"]
    assert [r["rank"] for r in ranks] == [0, 1] and all(r["gathered_block_equals_local"] for r in ranks)
    assert sum(r["agents"] for r in ranks) == d["config"]["agents_total"] == 50 and min(r["agents"] for r in ranks) >= 15
    # one rank alone gathers the same bytes: the job is the same whatever N (strong scaling)
    one = _run(["--gpus", "1", "--dry"])
    assert one.returncode == 0, one.stderr[-2000:]
    d1 = json.loads([l for l in one.stdout.splitlines() if l.startswith("{")][0])
    assert d1["n_gpus"] == 1 and d1["config"]["gathered_sha256"] == d["config"]["gathered_sha256"]
    assert d1["config"]["gathered_doubles"] == d["config"]["gathered_doubles"]


def test_world_size_must_equal_gpus():
    out = _run(["--gpus", "2", "--dry"], env={"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert out.returncode != 0 and "WORLD_SIZE" in (out.stderr + out.stdout)


def test_gpus_8_dry_carries_both_scaling_curves():
    """the first 8-GPU run anyone gets must record both curves.  `--gpus 8 --dry`: eight gloo ranks, the strong job
    (one copy of the workload) and the weak job (eight copies) one after the other, every rank's block gathered in both."""
    out = _run(["--gpus", "8", "--dry"], timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 8 and d["gloo_ranks"] == 8 and d["scaling"] == "strong" and "value_weak" in d
    ranks = d["config"]["per_rank"]
    assert [r["rank"] for r in ranks] == list(range(8)) and all(r["gathered_block_equals_local"] for r in ranks)
    assert sum(r["agents"] for r in ranks) == d["config"]["agents_total"] == 50 and min(r["agents"] for r in ranks) >= 3
    o = d["other_scaling"]
    assert o["scaling"] == "weak" and o["agents_total"] == 8 * 50 and o["worlds_total"] == 8 * d["config"]["worlds_total"]
    assert all(r["gathered_block_equals_local"] for r in o["per_rank"]) and sum(r["agents"] for r in o["per_rank"]) == 400
    assert o["gathered_doubles"] == 8 * d["config"]["gathered_doubles"]      # (front-end paths: the copies are the same worlds)


def test_gpus_defaults_to_world_size_and_a_stray_world_size_counts_for_nothing():
    """under the launcher `--gpus` left out adopts WORLD_SIZE; WORLD_SIZE exported without RANK is not a launcher."""
    e = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--standalone", "--local-addr", "127.0.0.1", "--nnodes=1",
           "--nproc-per-node", "2", os.path.join(ROOT, "bench.py"), "--workload", "map50", "--instances", "2", "--steps", "1",
           "--warmup", "0", "--setup-procs", "1", "--dry"]
    out = subprocess.run(cmd, env=e, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][0])
    assert d["n_gpus"] == 2 and d["other_scaling"]["scaling"] == "weak"
    stray = _run(["--dry"], env={"WORLD_SIZE": "4"})
    assert stray.returncode == 0, stray.stderr[-2000:]
    assert json.loads([l for l in stray.stdout.splitlines() if l.startswith("{")][0])["n_gpus"] == 1
