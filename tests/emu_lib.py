"""Loader for the lane-serial host build of the device program (tests/emu/libcsdo_emu.so).  Test infrastructure."""
import ctypes as C
import os
import subprocess

import numpy as np

from csdotrajectoryplanning_amd import abi
from csdotrajectoryplanning_amd.problem import Solution, World

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_LIB = None


def build():
    subprocess.run(["make", "-s", "-C", os.path.join(_ROOT, "tests", "emu")], check=True)


def lib():
    global _LIB
    if _LIB is None:
        build()
        # CSDO_EMU_LIB: another build of the same source (experiments: scripts/emu_regress.py, scripts/chain_parity.py)
        _LIB = C.CDLL(os.environ.get("CSDO_EMU_LIB") or os.path.join(_ROOT, "tests", "emu", "libcsdo_emu.so"))
        _LIB.csdo_emu_solve_batch.argtypes = [C.POINTER(abi.Problem), C.c_int32, C.POINTER(abi.Result)]
        _LIB.csdo_emu_solve_batch_mode.argtypes = [C.POINTER(abi.Problem), C.c_int32, C.POINTER(abi.Result), C.c_int]
        _LIB.csdo_emu_solve_batch_mt.argtypes = [C.POINTER(abi.Problem), C.c_int32, C.POINTER(abi.Result), C.c_int,
                                                 C.c_int]
        _LIB.csdo_emu_generate_boxes.argtypes = [abi.c_double_p, C.c_int32, abi.c_double_p, C.c_int32, C.c_double,
                                                 C.c_double, C.POINTER(abi.Vehicle), abi.c_double_p, abi.c_int32_p]
        _LIB.csdo_emu_math_eval.argtypes = [C.c_int32, abi.c_double_p, abi.c_double_p, abi.c_double_p, C.c_int32]
        _LIB.csdo_emu_agent_class.argtypes = [C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_int64)]
    return _LIB


def math_eval(fn, a, b=None):
    """csrc/csdo_math.h as the host build compiles it (fn 0 sin, 1 cos, 2 tan, 3 atan2(a, b); 10..13 the C library's)."""
    a = np.ascontiguousarray(a, dtype=np.float64)
    b = a if b is None else np.ascontiguousarray(b, dtype=np.float64)
    out = np.zeros_like(a)
    assert lib().csdo_emu_math_eval(fn, abi.as_double_p(a), abi.as_double_p(b), abi.as_double_p(out), a.size) == 0
    return out


def solve_batch_rc(worlds, mode=0, n_threads=1):
    """Return code of the packing + solve (error-path tests) and the solutions."""
    sols = [Solution.allocate(w.Na, w.Nt) for w in worlds]
    probs = (abi.Problem * len(worlds))(*[w.c_problem() for w in worlds])
    res = (abi.Result * len(worlds))(*[s._c for s in sols])
    rc = lib().csdo_emu_solve_batch_mt(probs, len(worlds), res, mode, n_threads)
    for s, r in zip(sols, res):
        s._c = r
        s.finish()
    return rc, sols


def solve_batch(worlds, mode=0, n_threads=1):
    rc, sols = solve_batch_rc(worlds, mode, n_threads)
    assert rc == 0, rc
    return sols


def solve(world: World, mode=0) -> Solution:
    return solve_batch([world], mode)[0]


def generate_boxes(points, obstacles, dimx, dimy, veh):
    points = np.ascontiguousarray(points, dtype=np.float64).reshape(-1, 2)
    obstacles = np.ascontiguousarray(obstacles, dtype=np.float64).reshape(-1, 3)
    n = points.shape[0]
    boxes = np.zeros((n, 4))
    status = np.zeros(n, np.int32)
    lib().csdo_emu_generate_boxes(abi.as_double_p(points), n, abi.as_double_p(obstacles), obstacles.shape[0], dimx,
                                  dimy, C.byref(veh), abi.as_double_p(boxes), abi.as_int32_p(status))
    return boxes, status


def agent_class(nt, n_obs, n_planes):
    """(threads, mode, rows_lds, tail_nodes, lds_bytes) as this build's class rule gives them (csrc/dsqp_class.h)."""
    out = (C.c_int64 * 5)()
    assert lib().csdo_emu_agent_class(nt, n_obs, n_planes, out) == 0
    return tuple(int(v) for v in out)
