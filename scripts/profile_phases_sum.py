"""Diagnostic: where the CU time of a batch goes, summed over all agents (needs `make -C csdotrajectoryplanning_amd/csrc prof`)."""
import ctypes as C
import os
import sys
import numpy as np
sys.path.insert(0, '.')
from csdotrajectoryplanning_amd import _lib, workloads  # noqa: E402
_lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), "libcsdo_hip_prof.so")
from csdotrajectoryplanning_amd.solver import DsqpHandle  # noqa: E402
NAMES = ["other", "corridor", "assemble", "ruiz", "warmstart", "factor", "rhs", "solve_fwd", "solve_bwd", "update",
         "info/check", "bookkeeping", "hot load/save", "fwd barrier", "bwd barrier", "-"]
ids = [int(x) for x in sys.argv[1].split(',')] if len(sys.argv) > 1 else [0, 1, 2, 4]
worlds = [workloads.map100_world(k)[0] for k in ids]
h = DsqpHandle(0)
h.upload(worlds); h.run(); ks = h.run(); sols = h.download()
Na = sum(w.Na for w in worlds)
L = _lib.lib()
ph = np.zeros((Na, 48), np.int64); tk = np.zeros(Na, np.int64)
L.csdo_debug_phase_ticks.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
assert L.csdo_debug_phase_ticks(h._h, ph.ctypes.data, tk.ctypes.data) == 0
it = np.concatenate([s.admm_iters for s in sols]); sq = np.concatenate([s.sqp_iters for s in sols])
for label, sel in (("all agents", it >= 0), ("short agents (< 1000 iterations)", it < 1000), ("long agents", it >= 1000)):
    p = ph[sel][:, :16].sum(0).astype(float); tot = p.sum()
    print("%s: %d agents, %d ADMM iterations, %d SQP iterations, %.1f ms of CU time (%.1f us per iteration, %.2f ms per SQP iteration)" % (
        label, sel.sum(), it[sel].sum(), sq[sel].sum(), tk[sel].sum() * 1e-5, tk[sel].sum() * 1e-2 / max(it[sel].sum(), 1), tk[sel].sum() * 1e-5 / max(sq[sel].sum(), 1)))
    print("   " + "  ".join("%s %.1f%%" % (n, 100.0 * p[i] / tot) for i, n in enumerate(NAMES[:15])))

    print("   cycles per SQP iteration: " + "  ".join("%s %.0fk" % (NAMES[i], p[i] / max(sq[sel].sum(), 1) / 1e3) for i in (1, 2, 3, 4, 5, 10, 11, 12)))
