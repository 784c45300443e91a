"""Diagnostic: per-agent microseconds per ADMM iteration alone vs inside a full batch."""
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
    small = [w for w in worlds if w.Nt <= 181]
    alone = []
    for w in small[:8]:
        h.upload([w]); h.run(); h.run(); s = h.download()[0]
        alone.append(s.agent_seconds / np.maximum(s.admm_iters, 1) * 1e6)
    h.upload(small); h.run(); ks = h.run(); sols = h.download()
    print('batch kernel %.1f ms' % (ks * 1e3))
    for i, (w, s) in enumerate(zip(small[:8], sols[:8])):
        inb = s.agent_seconds / np.maximum(s.admm_iters, 1) * 1e6
        K = w.plane_off[1:] - w.plane_off[:-1]
        long_ = s.admm_iters >= 1000
        print('instance %d Nt %d: alone us/it median %.1f (long agents %.1f), in batch median %.1f (long %.1f); ratio long %.2f' % (
            i, w.Nt, np.median(alone[i]), np.median(alone[i][long_]) if long_.any() else 0, np.median(inb),
            np.median(inb[long_]) if long_.any() else 0, np.median(inb[long_] / alone[i][long_]) if long_.any() else 0))
    allsec = np.concatenate([s.agent_seconds for s in sols]); allit = np.concatenate([s.admm_iters for s in sols])
    allK = np.concatenate([w.plane_off[1:] - w.plane_off[:-1] for w in small])
    order = np.argsort(-allsec)[:12]
    print('slowest agents in batch: ' + ', '.join('%.0fms/%dit/K%d' % (allsec[i] * 1e3, allit[i], allK[i]) for i in order))
    print('sum of agent seconds / 256 = %.1f ms; corr(K, iterations) = %.2f' % (allsec.sum() / 256 * 1e3, np.corrcoef(allK, allit)[0, 1]))
