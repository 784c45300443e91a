"""GPU vs oracle over the whole map100by100/agents50/obstacle set (3000 agents): iteration counts, statuses, |dx|."""
import json
import os
import sys
import time
from multiprocessing import Pool
import numpy as np
sys.path.insert(0, '.')
from csdotrajectoryplanning_amd import workloads


def make(k):
    return workloads.map100_world(k)[0]


if __name__ == '__main__':
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    with Pool(32) as pool:
        worlds = pool.map(make, range(n))
    from csdotrajectoryplanning_amd.solver import DsqpHandle
    from tests import oracle_lib, parity
    h = DsqpHandle(0)
    got = h.solve_batch(worlds)
    t0 = time.time()
    ref = [oracle_lib.solve(w, os.cpu_count()) for w in worlds]
    print('oracle: %.1f s for %d instances' % (time.time() - t0, n))
    agents = mism = 0
    dmax = []
    dcor = []
    worst = []
    for k, (w, r, g) in enumerate(zip(worlds, ref, got)):
        same = (r.sqp_iters == g.sqp_iters) & (r.admm_iters == g.admm_iters) & (r.last_status == g.last_status)
        agents += w.Na
        mism += int((~same).sum())
        c = parity.compare(r, g)
        d = c["d_sol"][same]
        dmax.extend(d.tolist())
        dcor.extend(c["d_cor"][same].tolist())
        for a in np.where(~same)[0]:
            worst.append((k, int(a), int(r.admm_iters[a]), int(g.admm_iters[a]), int(r.last_status[a]), int(g.last_status[a])))
        assert r.solver_status == g.solver_status or (~same).any(), (k, r.solver_status, g.solver_status)
    dmax = np.array(dmax)
    dcor = np.array(dcor)
    noflip = dcor < 0.05          # no safe box differs by a 0.1 m growth step anywhere along the agent's SQP
    dn = dmax[noflip]
    out = {"instances": n, "agents": agents, "agents_with_different_iteration_counts_or_status": mism,
           "max_abs_diff_on_agents_with_equal_counts": {"median": float(np.median(dmax)), "p90": float(np.percentile(dmax, 90)),
                                                         "p99": float(np.percentile(dmax, 99)), "max": float(dmax.max()),
                                                         "above_1e-4": int((dmax > 1e-4).sum()), "above_1e-3": int((dmax > 1e-3).sum()),
                                                         "above_2e-2": int((dmax > 2e-2).sum())},
           "agents_with_equal_counts_and_identical_box_growth": int(noflip.sum()),
           "max_abs_diff_on_those": {"median": float(np.median(dn)), "p99": float(np.percentile(dn, 99)), "max": float(dn.max()),
                                     "above_1e-4": int((dn > 1e-4).sum()), "above_1e-3": int((dn > 1e-3).sum())},
           "different": worst[:20]}
    print(json.dumps(out))
