"""Diagnostic: per-phase shader-clock share of the slowest agents (needs `make -C csdotrajectoryplanning_amd/csrc prof`)."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, '.')
from csdotrajectoryplanning_amd import _lib, workloads  # noqa: E402

_lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), "libcsdo_hip_prof.so")
from csdotrajectoryplanning_amd.solver import DsqpHandle  # noqa: E402

NAMES = ["other", "corridor", "assemble", "ruiz", "warmstart", "factor", "rhs", "solve_fwd", "solve_bwd", "update",
         "info/check", "bookkeeping", "hot load/save", "fwd barrier", "bwd barrier"]
which = sys.argv[1] if len(sys.argv) > 1 else "map100"
if which in ("map100", "map50"):
    world, info = (workloads.map100_world(0) if which == "map100" else workloads.build_world(workloads.MAP50_AGENTS25, 0))
else:   # any workload of workloads.workload_jobs, world index in argv[2]
    world, info = workloads.build_job(workloads.workload_jobs(which)[int(sys.argv[2]) if len(sys.argv) > 2 else 0])
h = DsqpHandle(0)
h.upload([world])
h.run()
ks = h.run()
sol = h.download()[0]
L = _lib.lib()
ph = np.zeros((world.Na, 48), np.int64)
tk = np.zeros(world.Na, np.int64)
L.csdo_debug_phase_ticks.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
assert L.csdo_debug_phase_ticks(h._h, ph.ctypes.data, tk.ctypes.data) == 0
K = world.plane_off[1:] - world.plane_off[:-1]
print("kernel %.1f ms, Nt=%d" % (ks * 1e3, world.Nt))
order = np.argsort(-tk)[:4].tolist() + np.argsort(tk)[:1].tolist()
for a in order:
    tot = ph[a][:16].sum()
    print("agent %d: wall %.1f ms, sqp %d, admm %d, planes %d, clock %.2f GHz, us/iter %.1f" % (
        a, tk[a] * 1e-5, sol.sqp_iters[a], sol.admm_iters[a], K[a], tot / (tk[a] * 10.0) , tk[a] * 1e-2 / max(sol.admm_iters[a], 1)))
    print("   " + "  ".join("%s %.1f%%" % (n, 100.0 * ph[a][i] / tot) for i, n in enumerate(NAMES)))
    it = max(sol.admm_iters[a], 1)
    print("   level compute (lane t=2^l) cycles/iter: fwd " + " ".join("%d" % (ph[a][16 + l] / it) for l in range(8)) + " | bwd " + " ".join("%d" % (ph[a][32 + l] / it) for l in range(8)))
    print("   per-level cycles/iter in the eliminated solver lane: fwd " + " ".join("%d" % (ph[a][16 + l] / it if True else 0) for l in range(8)) + " | bwd " + " ".join("%d" % (ph[a][32 + l] / it) for l in range(8))) if False else None
    print("   cycles/iter: " + "  ".join("%s %d" % (NAMES[i], ph[a][i] / it) for i in (6, 7, 13, 8, 14, 9, 10)))
