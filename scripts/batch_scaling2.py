"""Diagnostic: where does the batch throughput go?  copies of ex0 vs mixed instances, one vs two launch groups."""
import sys
import time
from multiprocessing import Pool
import numpy as np
sys.path.insert(0, '.')
from csdotrajectoryplanning_amd import workloads


def make(k):
    return workloads.map100_world(k)[0]


def run(h, ws, label):
    h.upload(ws); h.run(); ks = min(h.run() for _ in range(2)); sols = h.download()
    it = sum(int(s.admm_iters.sum()) for s in sols)
    tm = max(s.t_max_individual for s in sols)
    print('%-34s %5d agents: kernel %6.1f ms, slowest agent %6.1f ms, %8d it -> %.2f M it/s, CU-us/it %.1f  groups %s' % (
        label, sum(w.Na for w in ws), ks * 1e3, tm * 1e3, it, it / ks / 1e6, ks * 256 / it * 1e6,
        [(g['n_agents'], g['residency_mode'], round(g['seconds'] * 1e3, 1)) for g in h.launch_groups()]))


if __name__ == '__main__':
    with Pool(32) as pool:
        worlds = pool.map(make, range(60))
    from csdotrajectoryplanning_amd.solver import DsqpHandle
    h = DsqpHandle(0)
    w0 = worlds[0]
    for n in (5, 6, 10, 20):
        run(h, [w0] * n, '%d copies of ex0' % n)
    small = [w for w in worlds if w.Nt <= 181]
    run(h, small, 'instances with Nt <= 181')
    run(h, small[:5], '5 instances with Nt <= 181')
    run(h, small[:10], '10 instances with Nt <= 181')
    h.set_min_residency_mode(1)
    run(h, worlds, 'all 60, everything mode 1')
    run(h, [w0] * 5, '5 copies of ex0, mode 1')
    h.set_min_residency_mode(0)
    run(h, worlds, 'all 60, automatic')
