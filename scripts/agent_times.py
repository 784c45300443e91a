"""Per-agent device seconds of one batch solve, with the features a launch-order predictor could use.
usage (GPU box): python scripts/agent_times.py [map100|map50] out.npz"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from csdotrajectoryplanning_amd import workloads  # noqa: E402
from csdotrajectoryplanning_amd.solver import DsqpHandle  # noqa: E402


def plane_residual(w):
    """Largest separating-plane residual of the initial guess per agent (> 0: a plane is violated)."""
    x0 = np.asarray(w.x0_bar).reshape(w.Na, w.Nt, 6)
    pl = np.asarray(w.planes)
    off = np.asarray(w.plane_off)
    out = np.full(w.Na, -np.inf)
    for a in range(w.Na):
        p = pl[off[a]:off[a + 1]]
        if not len(p):
            continue
        xs = x0[a, p["t"]]
        cy, sy = np.cos(xs[:, 2]), np.sin(xs[:, 2])
        res = []
        for r in range(4):
            ox = w.veh.f2x if r < 2 else w.veh.r2x
            res.append(p["c"][:, 3 * r] * (xs[:, 0] + ox * cy) + p["c"][:, 3 * r + 1] * (xs[:, 1] + ox * sy) + p["c"][:, 3 * r + 2])
        out[a] = np.max(res)
    return out


def main():
    wl = sys.argv[1] if len(sys.argv) > 1 else "map100"
    out = sys.argv[2] if len(sys.argv) > 2 else "agent_times.npz"
    built = workloads.build_jobs_parallel(workloads.workload_jobs(wl), 16)
    worlds = [w for w, _ in built]
    h = DsqpHandle(0)
    h.upload(worlds)
    h.run()
    h.run()
    sols = h.download()
    secs = np.concatenate([s.agent_seconds for s in sols])
    np.savez(out, seconds=secs, Nt=np.concatenate([np.full(w.Na, w.Nt) for w in worlds]),
             K=np.concatenate([np.diff(np.asarray(w.plane_off)) for w in worlds]),
             sqp=np.concatenate([s.sqp_iters for s in sols]), admm=np.concatenate([s.admm_iters for s in sols]),
             status=np.concatenate([s.last_status for s in sols]),
             residual=np.concatenate([plane_residual(w) for w in worlds]),
             kernel_seconds=h.launch_groups()[0]["seconds"] if h.launch_groups() else 0.0,
             groups=np.asarray(h.agent_groups()))
    print(wl, "agents", len(secs), "sum/256 = %.1f ms" % (secs.sum() / 256 * 1e3), "longest %.1f ms" % (secs.max() * 1e3),
          "kernel", [round(g["seconds"] * 1e3, 1) for g in h.launch_groups()])


if __name__ == "__main__":
    main()
