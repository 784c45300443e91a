"""Diagnostic: kernel time of one launch over a batch of 50-agent instances ex0..ex{n-1} (all groups concurrent)."""
import sys
import time
from multiprocessing import Pool
sys.path.insert(0, '.')
from csdotrajectoryplanning_amd import workloads


def make(k):
    return workloads.map100_world(k)[0]


if __name__ == '__main__':
    counts = [int(x) for x in sys.argv[1].split(',')] if len(sys.argv) > 1 else [1, 8, 16, 32, 60]
    t0 = time.time()
    with Pool(min(32, max(counts))) as pool:
        worlds = pool.map(make, range(max(counts)))
    print('built %d worlds in %.1f s; Nt: %s' % (len(worlds), time.time() - t0, sorted(w.Nt for w in worlds)))
    from csdotrajectoryplanning_amd.solver import DsqpHandle
    h = DsqpHandle(0)
    for n in counts:
        ws = worlds[:n]
        h.upload(ws); h.run(); ks = min(h.run() for _ in range(2)); sols = h.download()
        it = sum(int(s.admm_iters.sum()) for s in sols)
        tm = max(s.t_max_individual for s in sols)
        print('%d instances (%d agents): kernel %.1f ms, slowest agent %.1f ms, %d iterations -> %.2f M it/s' % (
            n, sum(w.Na for w in ws), ks * 1e3, tm * 1e3, it, it / ks / 1e6))
        for g in h.launch_groups():
            print('    group:', g)
