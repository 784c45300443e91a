"""Diagnostic: time the fixed-work ablation builds (make -C csdotrajectoryplanning_amd/csrc ablate)."""
import os
import subprocess
import sys

CODE = r"""
import os, sys
sys.path.insert(0, '.')
from csdotrajectoryplanning_amd import _lib, workloads
V = sys.argv[1]
_lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), 'libcsdo_hip_abl_' + V + '.so')
from csdotrajectoryplanning_amd.solver import DsqpHandle
w, _ = workloads.map100_world(0)
h = DsqpHandle(0); h.upload([w]); h.run(); ks = min(h.run() for _ in range(3)); s = h.download()[0]
it = max(int(s.admm_iters.max()), 1)
print('%-16s kernel %.2f ms  iters/agent %d  -> %.2f us per iteration' % (V, ks * 1e3, it, ks * 1e6 / it))
"""
VARIANTS = sys.argv[1].split(",") if len(sys.argv) > 1 else ["FIXED", "FIXED_NOSOLVE", "FIXED_NOLEVELWORK", "FIXED_NOBWDWORK"]
for v in VARIANTS:
    subprocess.run([sys.executable, "-c", CODE, v])
