"""Diagnostic: scheduling efficiency of the 60-instance batch: sum of per-agent device seconds / 256 CUs vs kernel time."""
import sys
from multiprocessing import Pool
import numpy as np
sys.path.insert(0, '.')
from csdotrajectoryplanning_amd import workloads


def make(k):
    return workloads.map100_world(k)[0]


if __name__ == '__main__':
    with Pool(32) as pool:
        worlds = pool.map(make, range(60))
    from csdotrajectoryplanning_amd.solver import DsqpHandle
    h = DsqpHandle(0)
    h.upload(worlds); h.run(); ks = h.run(); sols = h.download()
    sec = np.concatenate([s.agent_seconds for s in sols]); it = np.concatenate([s.admm_iters for s in sols])
    of = h.agent_groups(); groups = h.launch_groups()
    print('kernel %.1f ms; sum of agent seconds / 256 = %.1f ms; slowest agent %.1f ms' % (ks * 1e3, sec.sum() / 256 * 1e3, sec.max() * 1e3))
    for g, G in enumerate(groups):
        m = of == g
        print('  group %d (mode %d): %d agents, %.1f ms, CU-seconds share %.1f ms, us/it long agents %.1f, short %.1f' % (
            g, G['residency_mode'], m.sum(), G['seconds'] * 1e3, sec[m].sum() / 256 * 1e3,
            np.median(sec[m & (it >= 2000)] / it[m & (it >= 2000)]) * 1e6, np.median(sec[m & (it < 1000)] / np.maximum(it[m & (it < 1000)], 1)) * 1e6))
    est = np.concatenate([np.zeros(w.Na) for w in worlds])
    order = np.argsort(-sec)[:10]
    print('  slowest: ' + ', '.join('%.0fms/%dit/g%d' % (sec[i] * 1e3, it[i], of[i]) for i in order))
