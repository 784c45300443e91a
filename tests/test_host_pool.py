"""The library's host threads (csrc/host_pool.h): the bridge, the packing and the work estimate are cut into blocks for a persistent pool.
Their outputs do not depend on the pool - one thread (CSDO_HOST_THREADS=1), three, the default -, a process forked after the pool
was started (its threads do not exist in the child) still gets its answers, and many threads may call into the library at once."""
import hashlib
import os
import subprocess
import sys
import threading

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_SCRIPT = r"""
import hashlib, os, sys
sys.path.insert(0, %r)
import numpy as np
from csdotrajectoryplanning_amd import workloads
from csdotrajectoryplanning_amd.solver import estimate_work, interpolate_and_planes_batch_host

def digest():
    h = hashlib.sha256()
    for name, n in (("map100", 3), ("map50", 4), ("room50", 1)):
        built = [workloads.build_job(j) for j in workloads.workload_jobs(name, n)]
        items = [(*info["paths"], w.dimx, w.dimy, w.obstacles) for w, info in built if w.Na == len(info["paths"][2]) - 1]
        w0 = built[0][0]
        for world, pairs, legal in interpolate_and_planes_batch_host(items, w0.veh, w0.parm):
            for a in (world.x0_bar, world.plane_off, world.planes, pairs):
                h.update(np.ascontiguousarray(a).tobytes())
            h.update(bytes([legal]))
            h.update(estimate_work([world]).tobytes())
    return h.hexdigest()

if __name__ == "__main__":
    first = digest()
    if len(sys.argv) > 1 and sys.argv[1] in ("fork", "fork_busy"):
        if sys.argv[1] == "fork_busy":              # another thread keeps the pool busy while this one forks (ctypes releases the GIL)
            import threading
            stop = []
            def churn():
                while not stop:
                    digest()
            bg = threading.Thread(target=churn, daemon=True)
            bg.start()
            import time
            time.sleep(0.05)
        r, w = os.pipe()
        pid = os.fork()
        if pid == 0:
            os.write(w, digest().encode())        # the pool's threads are not in this process: every loop runs on its caller
            os._exit(0)
        os.waitpid(pid, 0)
        child = os.read(r, 100).decode()
        assert child == first, (child, first)
        if sys.argv[1] == "fork_busy":
            stop.append(1)
            bg.join()
    print(first)
""" % ROOT


def _run(env_threads, *args):
    env = dict(os.environ)
    env.pop("CSDO_HOST_THREADS", None)
    if env_threads is not None:
        env["CSDO_HOST_THREADS"] = str(env_threads)
    out = subprocess.run([sys.executable, "-c", _SCRIPT, *args], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    return out.stdout.strip().splitlines()[-1]


def test_outputs_do_not_depend_on_the_number_of_host_threads():
    one = _run(1)
    assert len(one) == 64
    assert _run(3) == one
    assert _run(None) == one


def test_a_forked_child_runs_its_loops_alone():
    assert len(_run(None, "fork")) == 64


def test_a_fork_while_the_pool_is_busy_does_not_leave_the_child_with_a_locked_pool():
    """pthread_atfork: the pool's lock is held around fork(), the child gets it unlocked and an empty pool (ten forks in a row)."""
    for _ in range(10):
        assert len(_run(None, "fork_busy")) == 64


def test_many_caller_threads_at_once():
    """Eight Python threads (ctypes releases the GIL) bridge and estimate different worlds at once, each call cut into blocks for the one
    pool: the answers are the ones of the calls made one after the other."""
    from csdotrajectoryplanning_amd import workloads
    from csdotrajectoryplanning_amd.solver import estimate_work, interpolate_and_planes_batch_host
    built = [workloads.build_job(j) for j in workloads.workload_jobs("map50", 8)]
    items = [(*info["paths"], w.dimx, w.dimy, w.obstacles) for w, info in built]
    w0 = built[0][0]

    def one(k):
        world, pairs, legal = interpolate_and_planes_batch_host(items[k:k + 1], w0.veh, w0.parm)[0]
        h = hashlib.sha256()
        for a in (world.x0_bar, world.plane_off, world.planes, pairs, estimate_work([world])):
            h.update(np.ascontiguousarray(a).tobytes())
        return h.hexdigest()
    want = [one(k) for k in range(len(items))]
    for _ in range(3):
        got = [None] * len(items)
        thr = [threading.Thread(target=lambda k=k: got.__setitem__(k, one(k))) for k in range(len(items))]
        for t in thr:
            t.start()
        for t in thr:
            t.join()
        assert got == want


def test_pool_under_thread_sanitizer(tmp_path):
    """tests/cpp/pool_stress.cc built with -fsanitize=thread: nested loops from four caller threads, no report and the right sums."""
    exe = str(tmp_path / "pool_stress_tsan")
    src = os.path.join(ROOT, "tests", "cpp", "pool_stress.cc")
    cc = subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-pthread", "-fsanitize=thread", src, "-o", exe], capture_output=True, text=True)
    if cc.returncode != 0 and "tsan" in (cc.stderr or "").lower():
        import pytest
        pytest.skip("no ThreadSanitizer runtime here")
    assert cc.returncode == 0, cc.stderr[-2000:]
    run = subprocess.run([exe], env=dict(os.environ, CSDO_HOST_THREADS="8"), capture_output=True, text=True, timeout=600)
    assert run.returncode == 0 and "ThreadSanitizer" not in run.stderr and run.stdout.strip().endswith("ok"), (run.stdout, run.stderr[-3000:])


def test_bridge_and_packing_under_address_and_ub_sanitizers(tmp_path):
    """tests/cpp/sanitize_host.cc + csrc/bridge_host.cc built with -fsanitize=address,undefined: no report on thirty random worlds."""
    exe = str(tmp_path / "sanitize_host")
    cc = subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-pthread", "-fsanitize=address,undefined", "-fno-omit-frame-pointer",
                         os.path.join(ROOT, "tests", "cpp", "sanitize_host.cc"),
                         os.path.join(ROOT, "csdotrajectoryplanning_amd", "csrc", "bridge_host.cc"), "-o", exe], capture_output=True, text=True)
    if cc.returncode != 0 and "asan" in (cc.stderr or "").lower():
        import pytest
        pytest.skip("no sanitizer runtime here")
    assert cc.returncode == 0, cc.stderr[-2000:]
    run = subprocess.run([exe], env=dict(os.environ, CSDO_HOST_THREADS="8"), capture_output=True, text=True, timeout=600)
    assert run.returncode == 0 and run.stdout.strip().endswith("ok") and "runtime error" not in run.stderr and "AddressSanitizer" not in run.stderr, \
        (run.stdout[-500:], run.stderr[-3000:])
