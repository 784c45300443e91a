"""Diagnostic: cost of one safe box on the device (csdo_generate_boxes: one lane per point, the same make_box as the solve kernel)."""
import sys, time
sys.path.insert(0, '.')
import torch  # noqa
import numpy as np
from csdotrajectoryplanning_amd import workloads
from csdotrajectoryplanning_amd.solver import DsqpHandle
w = workloads.map100_world(0)[0]
h = DsqpHandle(0)
rng = np.random.default_rng(0)
for n in (64 * 256 * 4, 64 * 256 * 16):
    # points along the world's own initial guess (what the kernel sees), jittered
    base = w.x0_bar[:, :, :2].reshape(-1, 2)
    pts = base[rng.integers(0, len(base), n)] + rng.normal(0, 0.3, (n, 2))
    h.generate_boxes(pts[:1024], w.obstacles, w.dimx, w.dimy, w.veh)
    t = time.perf_counter()
    for _ in range(3):
        h.generate_boxes(pts, w.obstacles, w.dimx, w.dimy, w.veh)
    dt = (time.perf_counter() - t) / 3
    waves = n / 64
    print("n %d: %.2f ms per call (incl. PCIe); upper bound %.0f cycles per wave-box at 1024 waves in flight" % (n, dt * 1e3, dt * 2.4e9 / (waves / 1024.0)))
