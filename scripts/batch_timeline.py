"""Diagnostic (prof build): when do the workgroups of the 60-instance batch start and end?"""
import ctypes as C
import os
import sys
from multiprocessing import Pool
import numpy as np
sys.path.insert(0, '.')
from csdotrajectoryplanning_amd import _lib, workloads


def make(k):
    return workloads.map100_world(k)[0]


if __name__ == '__main__':
    with Pool(32) as pool:
        worlds = pool.map(make, range(60))
    _lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), "libcsdo_hip_prof.so")
    from csdotrajectoryplanning_amd.solver import DsqpHandle
    h = DsqpHandle(0)
    h.upload(worlds); h.run(); ks = h.run(); sols = h.download()
    Na = sum(w.Na for w in worlds)
    ph = np.zeros((Na, 48), np.int64); tk = np.zeros(Na, np.int64)
    L = _lib.lib()
    L.csdo_debug_phase_ticks.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    assert L.csdo_debug_phase_ticks(h._h, ph.ctypes.data, tk.ctypes.data) == 0
    start = ph[:, 47].astype(float); start -= start.min(); start *= 1e-5   # ms
    dur = tk * 1e-5
    end = start + dur
    of = h.agent_groups()
    print('kernel %.1f ms (prof build); last end %.1f ms; sum dur / 256 = %.1f ms' % (ks * 1e3, end.max(), dur.sum() / 256))
    for t in range(0, int(end.max()) + 10, 10):
        running = ((start <= t) & (end > t))
        print('t=%3d ms: running %3d (g0 %3d, g1 %3d)  started so far %4d' % (t, running.sum(), (running & (of == 0)).sum(), (running & (of == 1)).sum(), (start <= t).sum()))
    ev = np.concatenate([np.stack([start, np.ones_like(start)], 1), np.stack([end, -np.ones_like(end)], 1)])
    ev = ev[np.lexsort((ev[:, 1], ev[:, 0]))]
    run = np.cumsum(ev[:, 1])
    dt = np.diff(ev[:, 0])
    mid = (ev[:-1, 0] > 15) & (ev[:-1, 0] < 0.8 * end.max())
    print('peak concurrency %d; time-averaged concurrency over the middle of the run %.1f' % (run.max(), (run[:-1][mid] * dt[mid]).sum() / dt[mid].sum()))
    gaps = []
    order = np.argsort(start)
    print('start times of the first 300 workgroups (ms): ' + ' '.join('%.2f' % v for v in np.sort(start)[[0, 17, 18, 19, 20, 50, 100, 150, 200, 220, 230, 240, 260, 299]]))
    hw = ph[:, 46]
    cu = ((hw >> 32) & 0xf) * 4096 + (((hw >> 13) & 0x7) * 64 + ((hw >> 12) & 1) * 32 + ((hw >> 8) & 0xf))   # xcc, se, sh, cu
    ids = np.unique(cu)
    gaps = []
    for c in ids:
        m = np.where(cu == c)[0]
        o = m[np.argsort(start[m])]
        gaps.extend((start[o[1:]] - end[o[:-1]]).tolist())
    gaps = np.array(gaps)
    print('%d distinct CU slots; %d hand-overs; gap between a workgroup ending and the next one starting on the same CU: median %.3f ms, mean %.3f ms, p90 %.3f ms, max %.2f ms; total gap time / 256 = %.1f ms' % (
        len(ids), len(gaps), np.median(gaps), gaps.mean(), np.percentile(gaps, 90), gaps.max(), gaps.sum() / 256))
    per_x = [(int(x), int((((hw >> 32) & 0xf) == x).sum()), float(end[((hw >> 32) & 0xf) == x].max())) for x in np.unique((hw >> 32) & 0xf)]
    print('per XCC: (id, workgroups, last end ms): ' + ' '.join('(%d,%d,%.0f)' % t for t in per_x))
    K = np.concatenate([w.plane_off[1:] - w.plane_off[:-1] for w in worlds])
    NT = np.concatenate([np.full(w.Na, w.Nt) for w in worlds])
    IT = np.concatenate([s_.admm_iters for s_ in sols]); SQ = np.concatenate([s_.sqp_iters for s_ in sols])
    m = np.where((start > 0.75 * end.max()) & (dur > 12))[0]
    print('late and long (start, dur, K, Nt, admm, sqp): ' + ' '.join('(%.0f,%.0f,%d,%d,%d,%d)' % (start[i], dur[i], K[i], NT[i], IT[i], SQ[i]) for i in m[:16]))
    late = np.argsort(-end)[:8]
    print('last to finish: ' + ', '.join('g%d start %.0f dur %.0f' % (of[i], start[i], dur[i]) for i in late))
