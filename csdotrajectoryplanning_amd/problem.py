"""Host-side containers for what crosses the C ABI: one `World` per SolverDSQP construction."""
import ctypes as C
from dataclasses import dataclass, field

import numpy as np

from . import abi


@dataclass
class World:
    """Inputs of the reference's SolverDSQP constructor (sqp/dsqp_solver.h:26-34) as flat arrays."""
    x0_bar: np.ndarray      # [Na, Nt, 6] x,y,yaw,steer,v,d_steer
    plane_off: np.ndarray   # [Na+1] int32
    planes: np.ndarray      # abi.PLANE_DTYPE
    dimx: float
    dimy: float
    obstacles: np.ndarray   # [n_obs, 3]
    veh: abi.Vehicle
    parm: abi.QpParm
    logger_level: int = 0

    def __post_init__(self):
        self.x0_bar = np.ascontiguousarray(self.x0_bar, dtype=np.float64)
        self.plane_off = np.ascontiguousarray(self.plane_off, dtype=np.int32)
        self.planes = np.ascontiguousarray(self.planes, dtype=abi.PLANE_DTYPE)
        self.obstacles = np.ascontiguousarray(self.obstacles, dtype=np.float64).reshape(-1, 3)
        assert self.x0_bar.ndim == 3 and self.x0_bar.shape[2] == 6
        assert self.plane_off.shape[0] == self.Na + 1

    @property
    def Na(self):
        return int(self.x0_bar.shape[0])

    @property
    def Nt(self):
        return int(self.x0_bar.shape[1])

    def with_parm(self, **fields) -> "World":
        """The same world with fields of its csdo_qp_parm changed (max_iter, solve_refinement, adaptive_rho_interval, ...)."""
        p = abi.QpParm.from_buffer_copy(bytes(self.parm))
        for k, v in fields.items():
            setattr(p, k, type(getattr(p, k))(v))
        return World(self.x0_bar, self.plane_off, self.planes, self.dimx, self.dimy, self.obstacles, self.veh, p, self.logger_level)

    def c_problem(self) -> abi.Problem:
        p = abi.Problem()
        p.Na, p.Nt = self.Na, self.Nt
        p.x0_bar = abi.as_double_p(self.x0_bar)
        p.plane_off = abi.as_int32_p(self.plane_off)
        p.planes = abi.as_plane_p(self.planes)
        p.dimx, p.dimy = float(self.dimx), float(self.dimy)
        p.n_obs = int(self.obstacles.shape[0])
        p.obstacles = abi.as_double_p(self.obstacles)
        p.veh, p.parm = self.veh, self.parm
        p.logger_level = int(self.logger_level)
        return p

    def subset(self, lo, hi) -> "World":
        """Agents [lo, hi) of this world with their planes (agents are independent once planes are fixed)."""
        po = self.plane_off
        return World(self.x0_bar[lo:hi], po[lo:hi + 1] - po[lo], self.planes[po[lo]:po[hi]], self.dimx, self.dimy,
                     self.obstacles, self.veh, self.parm, self.logger_level)


@dataclass
class Solution:
    """Outputs of the constructor + getters (sqp/dsqp_solver.h:41-47)."""
    solutions: np.ndarray   # [Na, Nt, 6]
    corridors: np.ndarray   # [Na, Nt, 8]
    sqp_iters: np.ndarray
    admm_iters: np.ndarray
    last_status: np.ndarray
    solver_status: int = 0
    initial_static_legal: int = 0
    t_total: float = 0.0
    t_device: float = 0.0
    t_max_individual: float = 0.0
    agent_seconds: np.ndarray = None   # [Na] per-agent solve time
    _c: abi.Result = field(default=None, repr=False)

    @staticmethod
    def allocate(Na, Nt) -> "Solution":
        s = Solution(np.zeros((Na, Nt, 6)), np.zeros((Na, Nt, 8)), np.zeros(Na, np.int32), np.zeros(Na, np.int32),
                     np.zeros(Na, np.int32))
        r = abi.Result()
        r.solutions = abi.as_double_p(s.solutions)
        r.corridors = abi.as_double_p(s.corridors)
        r.sqp_iters = abi.as_int32_p(s.sqp_iters)
        r.admm_iters = abi.as_int32_p(s.admm_iters)
        r.last_status = abi.as_int32_p(s.last_status)
        s.agent_seconds = np.zeros(Na)
        r.agent_seconds = abi.as_double_p(s.agent_seconds)
        s._c = r
        return s

    def finish(self):
        r = self._c
        self.solver_status, self.initial_static_legal = int(r.solver_status), int(r.initial_static_legal)
        self.t_total, self.t_device, self.t_max_individual = r.t_total, r.t_device, r.t_max_individual
        return self


class _BridgeMemory:
    """Keeps a csdo_bridge_out (library-owned memory) alive for the numpy views built over it and releases it with them."""

    def __init__(self, bo, free):
        self.bo, self._free = bo, free

    def __del__(self):
        try:
            self._free(C.byref(self.bo))
        except Exception:
            pass


def bridge_views(bo: abi.BridgeOut, free, inst_dimx, inst_dimy, obstacles, veh, parm):
    """(World, pairs, initial_inter_legal) as VIEWS of the library-owned csdo_bridge_out: no copy of x0_bar, planes or pairs (29 MB
    for the 60 worlds of the map100 set).  Every array (and every slice of it, World.subset included) keeps the owner of the memory
    alive; it is released (`free` = csdo_bridge_free) when the last one goes."""
    Na, Nt = bo.Na, bo.Nt
    keep = _BridgeMemory(abi.BridgeOut.from_buffer_copy(bytes(bo)), free)

    def view(ptr, n_items, dtype):
        if n_items == 0:
            return np.zeros(0, dtype=dtype)
        raw = (C.c_char * (n_items * np.dtype(dtype).itemsize)).from_address(C.cast(ptr, C.c_void_p).value)
        raw._keep = keep                      # the buffer object is the base of every numpy view taken from it
        return np.frombuffer(raw, dtype=dtype)

    x0 = view(bo.x0_bar, Na * Nt * 6, np.float64).reshape(Na, Nt, 6)
    po = view(bo.plane_off, Na + 1, np.int32)
    planes = view(bo.planes, int(po[-1]), abi.PLANE_DTYPE)
    pairs = view(bo.pairs, 3 * bo.n_pairs, np.int32).reshape(-1, 3)
    return World(x0, po, planes, inst_dimx, inst_dimy, obstacles, veh, parm), pairs, int(bo.initial_inter_legal)


def bridge_to_world(bo: abi.BridgeOut, inst_dimx, inst_dimy, obstacles, veh, parm):
    """Copy a csdo_bridge_out (library-owned memory) into numpy arrays; returns (World, pairs, initial_inter_legal)."""
    Na, Nt = bo.Na, bo.Nt
    x0 = np.ctypeslib.as_array(bo.x0_bar, shape=(Na, Nt, 6)).copy()
    po = np.ctypeslib.as_array(bo.plane_off, shape=(Na + 1,)).copy()
    npl = int(po[-1])
    planes = np.zeros(npl, dtype=abi.PLANE_DTYPE)
    if npl:
        C.memmove(planes.ctypes.data, bo.planes, npl * abi.PLANE_DTYPE.itemsize)
    pairs = np.ctypeslib.as_array(bo.pairs, shape=(max(bo.n_pairs, 1), 3))[:bo.n_pairs].copy()
    return World(x0, po, planes, inst_dimx, inst_dimy, obstacles, veh, parm), pairs, int(bo.initial_inter_legal)
