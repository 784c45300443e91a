"""Diagnostic: where the CU time of a batch goes, summed over all agents (needs `make -C csdotrajectoryplanning_amd/csrc prof`).
usage: python scripts/profile_phases_sum.py [instance ids, comma separated] [map100|map50|room50|agents100]"""
import ctypes as C
import os
import sys
import numpy as np
sys.path.insert(0, '.')
from csdotrajectoryplanning_amd import _lib, workloads  # noqa: E402
if not os.environ.get("CSDO_DIAG_LIB"):      # (CSDO_DIAG_LIB: a phase-timer build given by path, scripts/gpu_run.sh phases:<workload>:<lib>)
    _lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), os.environ.get("CSDO_PROF_LIB", "libcsdo_hip_prof.so"))
from csdotrajectoryplanning_amd.solver import DsqpHandle  # noqa: E402
NAMES = ["other", "corridor", "assemble", "ruiz", "warmstart", "factor", "rhs", "solve_fwd", "solve_bwd", "update",
         "info/check", "bookkeeping", "hot load/save", "fwd barrier", "bwd barrier", "-"]
ids = [int(x) for x in sys.argv[1].split(',')] if len(sys.argv) > 1 else [0, 1, 2, 4]
wl = sys.argv[2] if len(sys.argv) > 2 else "map100"
if wl in ("map100", "map50"):
    worlds = [(workloads.map100_world(k) if wl == "map100" else workloads.map50_world(k))[0] for k in ids]
else:   # any other workload of workloads.WORKLOADS (room50, agents100, ...): `ids` index its job list
    jobs = workloads.workload_jobs(wl)
    worlds = [workloads.build_job(jobs[k])[0] for k in ids]
h = DsqpHandle(0)
h.upload(worlds); h.run(); ks = h.run(); sols = h.download()
Na = sum(w.Na for w in worlds)
L = _lib.lib()
ph = np.zeros((Na, 48), np.int64); tk = np.zeros(Na, np.int64)
L.csdo_debug_phase_ticks.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
assert L.csdo_debug_phase_ticks(h._h, ph.ctypes.data, tk.ctypes.data) == 0
it = np.concatenate([s.admm_iters for s in sols]); sq = np.concatenate([s.sqp_iters for s in sols])
print("workload %s instances %s: kernel %.1f ms, Nt %s" % (wl, ids, ks * 1e3, sorted(set(w.Nt for w in worlds))))
for label, sel in (("all agents", it >= 0), ("short agents (< 1000 iterations)", it < 1000), ("long agents", it >= 1000)):
    if sel.sum() == 0:
        continue
    p = ph[sel][:, :16].sum(0).astype(float); tot = p.sum()
    print("%s: %d agents, %d ADMM iterations, %d SQP iterations, %.1f ms of CU time (%.1f us per iteration, %.2f ms per SQP iteration)" % (
        label, sel.sum(), it[sel].sum(), sq[sel].sum(), tk[sel].sum() * 1e-5, tk[sel].sum() * 1e-2 / max(it[sel].sum(), 1), tk[sel].sum() * 1e-5 / max(sq[sel].sum(), 1)))
    print("   " + "  ".join("%s %.1f%%" % (n, 100.0 * p[i] / tot) for i, n in enumerate(NAMES[:15])))
    print("   cycles per SQP iteration: " + "  ".join("%s %.0fk" % (NAMES[i], p[i] / max(sq[sel].sum(), 1) / 1e3) for i in (1, 2, 3, 4, 5, 10, 11, 12)))
    fsub = ph[sel][:, 24:28].sum(0).astype(float)    # factor sub-phases (slots 16..19 of the kernel's timers)
    print("   factor, cycles per SQP iteration: assembly %.0fk  levels: elimination %.0fk + absorption %.0fk  tail inversion %.0fk  (the factor column above counts only the rest)" % tuple(fsub[[0, 3, 1, 2]] / max(sq[sel].sum(), 1) / 1e3))
    subs = ph[sel][:, 40:46].sum(0).astype(float) / max(sq[sel].sum(), 1) / 1e3
    print("   sub-timers, k cycles per SQP iteration: ruiz phase 1 %.0f  phase 2 %.0f  phase 3 %.0f | info lane loop %.0f  info fold %.0f | "
          "block load %.0f" % tuple(subs))
    n_it = max(it[sel].sum(), 1)
    print("   cycles per ADMM iteration: " + "  ".join("%s %.0f" % (NAMES[i], p[i] / n_it) for i in (6, 7, 13, 8, 14, 9)) +
          "  | iteration total %.0f" % (sum(p[i] for i in (6, 7, 13, 8, 14, 9)) / n_it))
    sub = ph[sel][:, 28:31].sum(0).astype(float) / n_it    # slots 20..22
    print("   of the forward sweep, per ADMM iteration: rhs assembly + first level %.0f  |  w pass + tail gather %.0f  |  tail product %.0f" % tuple(sub))
    # per-level cycles inside the elimination block of solver lane t = 2^lv (forward / backward), per ADMM iteration of the
    # agents that have that level
    fw = ph[sel][:, 16:24].astype(float); bw = ph[sel][:, 32:40].astype(float)
    itv = it[sel][:, None].astype(float)
    has = fw > 0
    print("   cycles inside a level's elimination block (lane 2^lv), fwd: " +
          " ".join("%.0f" % ((fw[:, lv][has[:, lv]] / itv[has[:, lv], 0]).mean() if has[:, lv].any() else 0) for lv in range(6)) +
          "  bwd: " + " ".join("%.0f" % ((bw[:, lv][has[:, lv]] / itv[has[:, lv], 0]).mean() if has[:, lv].any() else 0) for lv in range(6)))
    xs = ph[sel][:, 16:24].astype(float); xs2 = ph[sel][:, 32:40].astype(float)
    lab = ["rhs assembly", "fwd level 1", "fwd levels >= 2 + gather + hand-over", "wait at barrier", "-", "w pass", "wait for tail product", "backward sweep"]
    print("   pair-split solve, solver wave 0 (cycles per ADMM iteration): " + "  ".join("%s %.0f" % (lab[k], xs[:, k].sum() / n_it) for k in range(8)) + "  | sum %.0f" % (xs.sum() / n_it))
    print("   pair-split solve, solver wave 2 (cycles per ADMM iteration): " + "  ".join("%s %.0f" % (lab[k], xs2[:, k].sum() / n_it) for k in range(8)) + "  | sum %.0f" % (xs2.sum() / n_it))
