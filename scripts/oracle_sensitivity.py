"""How sensitive is the reference algorithm itself to rounding?  The oracle built without and with fused multiply-adds
(-ffp-contract=off vs -ffp-contract=fast -mfma: same source, last-bit differences in some products) on the same inputs.
usage: python scripts/oracle_sensitivity.py /path/to/alternative_liboracle.so [n_instances]"""
import ctypes as C
import json
import os
import sys
import numpy as np
sys.path.insert(0, '.')
from csdotrajectoryplanning_amd import abi, workloads
from csdotrajectoryplanning_amd.problem import Solution
from tests import oracle_lib, parity

alt = C.CDLL(sys.argv[1])
alt.csdo_oracle_solve.argtypes = [C.POINTER(abi.Problem), C.POINTER(abi.Result), C.c_int]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 8
agents = mism = 0
dall, dcor = [], []
for k in range(n):
    w = workloads.map100_world(k)[0]
    r = oracle_lib.solve(w, os.cpu_count())
    g = Solution.allocate(w.Na, w.Nt)
    p = w.c_problem()
    assert alt.csdo_oracle_solve(C.byref(p), C.byref(g._c), os.cpu_count()) == 0
    g.finish()
    same = (r.sqp_iters == g.sqp_iters) & (r.admm_iters == g.admm_iters) & (r.last_status == g.last_status)
    agents += w.Na
    mism += int((~same).sum())
    c = parity.compare(r, g)
    dall.extend(c["d_sol"][same].tolist())
    dcor.extend(c["d_cor"][same].tolist())
dall, dcor = np.array(dall), np.array(dcor)
nf = dcor < 0.05
print(json.dumps({"instances": n, "agents": agents, "agents_with_different_iteration_counts_or_status": mism,
                  "equal_counts": {"median": float(np.median(dall)), "p99": float(np.percentile(dall, 99)), "max": float(dall.max()),
                                   "above_1e-4": int((dall > 1e-4).sum()), "above_1e-3": int((dall > 1e-3).sum()),
                                   "above_2e-2": int((dall > 2e-2).sum())},
                  "equal_counts_and_identical_box_growth": {"agents": int(nf.sum()), "above_1e-4": int((dall[nf] > 1e-4).sum()),
                                                            "above_1e-3": int((dall[nf] > 1e-3).sum()), "max": float(dall[nf].max())}}))
