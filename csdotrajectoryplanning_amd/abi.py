"""ctypes mirror of include/csdo_dsqp.h (struct layouts only; no library is loaded here).

Field order and types follow the header one for one; the header cites the reference interface each struct replaces.
"""
import ctypes as C

import numpy as np

c_double_p = C.POINTER(C.c_double)
c_int32_p = C.POINTER(C.c_int32)

CSDO_OK = 0
CSDO_EINVAL = -1
CSDO_ENODEV = -2
CSDO_ENOMEM = -3
CSDO_ELIMIT = -4
CSDO_EDEVICE = -5
CSDO_MAX_NT = 512


class Vehicle(C.Structure):
    _fields_ = [(n, C.c_double) for n in
                ("r", "deltat", "LF", "LB", "car_width", "WB", "f2x", "r2x", "rv", "obs_radius")]


class QpParm(C.Structure):
    _fields_ = [("r_trust", C.c_double), ("max_omega", C.c_double), ("max_v", C.c_double),
                ("max_iter", C.c_double), ("delta_solution_threshold", C.c_double),
                ("max_violation", C.c_double), ("osqp_max_iter", C.c_int32),
                ("num_interpolation", C.c_int32), ("dt", C.c_double), ("fixed_corridor", C.c_int32),
                ("adaptive_rho_interval", C.c_int32), ("solve_refinement", C.c_int32), ("_reserved", C.c_int32)]


class Plane(C.Structure):
    _fields_ = [("t", C.c_int32), ("_pad", C.c_int32), ("c", C.c_double * 12)]


PLANE_DTYPE = np.dtype([("t", np.int32), ("_pad", np.int32), ("c", np.float64, (12,))])
assert PLANE_DTYPE.itemsize == C.sizeof(Plane) == 104


class Problem(C.Structure):
    _fields_ = [("Na", C.c_int32), ("Nt", C.c_int32), ("x0_bar", c_double_p), ("plane_off", c_int32_p),
                ("planes", C.POINTER(Plane)), ("dimx", C.c_double), ("dimy", C.c_double),
                ("n_obs", C.c_int32), ("_pad", C.c_int32), ("obstacles", c_double_p), ("veh", Vehicle),
                ("parm", QpParm), ("logger_level", C.c_int32), ("_pad2", C.c_int32)]


class Result(C.Structure):
    _fields_ = [("solutions", c_double_p), ("corridors", c_double_p), ("sqp_iters", c_int32_p),
                ("admm_iters", c_int32_p), ("last_status", c_int32_p), ("solver_status", C.c_int32),
                ("initial_static_legal", C.c_int32), ("t_total", C.c_double), ("t_device", C.c_double),
                ("t_max_individual", C.c_double), ("agent_seconds", c_double_p)]


class Validation(C.Structure):
    """csdo_validation (include/csdo_dsqp.h)."""
    _fields_ = [("vehicle_collisions", C.c_int64), ("obstacle_collisions", C.c_int64), ("out_of_map", C.c_int64),
                ("first_vehicle", C.c_int32 * 3), ("first_obstacle", C.c_int32 * 3),
                ("min_obstacle_clearance", C.c_double)]


class BridgeOut(C.Structure):
    _fields_ = [("Na", C.c_int32), ("Nt", C.c_int32), ("x0_bar", c_double_p), ("plane_off", c_int32_p),
                ("planes", C.POINTER(Plane)), ("n_pairs", C.c_int32), ("initial_inter_legal", C.c_int32),
                ("pairs", c_int32_p)]


class CoarseWorld(C.Structure):
    """csdo_coarse_world (include/csdo_dsqp.h): one world's front-end paths and map, as csdo_do_phase takes them."""
    _fields_ = [("states", c_double_p), ("actions", c_int32_p), ("path_off", c_int32_p), ("goals", c_double_p),
                ("obstacles", c_double_p), ("Na", C.c_int32), ("n_obs", C.c_int32), ("dimx", C.c_double), ("dimy", C.c_double)]


class DoPhaseTiming(C.Structure):
    """csdo_do_phase_timing."""
    _fields_ = [("first_launch", C.c_double), ("kernels_done", C.c_double), ("total", C.c_double),
                ("n_chunks", C.c_int32), ("streamed", C.c_int32), ("chunk_worlds", C.c_int32 * 4),
                ("chunk_bridge", C.c_double * 4), ("chunk_upload", C.c_double * 4), ("chunk_kernel", C.c_double * 4)]


class FrontEndParm(C.Structure):
    """csdo_front_end_parm (include/csdo_dsqp.h)."""
    _fields_ = [("penalty_turning", C.c_double), ("penalty_reversing", C.c_double), ("penalty_cod", C.c_double),
                ("map_resolution", C.c_double), ("max_closed_set_size", C.c_double), ("time_limit_s", C.c_double),
                ("node_limit", C.c_int32), ("rand_seed", C.c_uint32), ("keep_off_lower_goals", C.c_int32),
                ("rand_glibc", C.c_int32)]


class Paths(C.Structure):
    """csdo_paths (include/csdo_dsqp.h)."""
    _fields_ = [("Na", C.c_int32), ("status", C.c_int32), ("path_off", c_int32_p), ("states", c_double_p),
                ("actions", c_int32_p), ("seconds", C.c_double), ("hl_expanded", C.c_int32),
                ("hl_generated", C.c_int32), ("ll_expanded", C.c_int64)]


def as_double_p(a):
    return a.ctypes.data_as(c_double_p)


def as_int32_p(a):
    return a.ctypes.data_as(c_int32_p)


def as_plane_p(a):
    return a.ctypes.data_as(C.POINTER(Plane))


# Every symbol include/csdo_dsqp.h declares (checked by tests/test_abi.py against the built library).
class LaunchGroup(C.Structure):
    """csdo_launch_group (include/csdo_dsqp.h)."""
    _fields_ = [("n_agents", C.c_int32), ("threads", C.c_int32), ("residency_mode", C.c_int32),
                ("max_nt", C.c_int32), ("lds_bytes", C.c_int64), ("seconds", C.c_double)]


EXPORTED_SYMBOLS = (
    "csdo_dsqp_create", "csdo_dsqp_create_multi", "csdo_dsqp_multi_count", "csdo_dsqp_multi_child", "csdo_dsqp_shard_bounds", "csdo_dsqp_destroy", "csdo_dsqp_solve", "csdo_dsqp_solve_batch", "csdo_dsqp_upload",
    "csdo_dsqp_run", "csdo_dsqp_run_async", "csdo_dsqp_wait", "csdo_dsqp_create_shared", "csdo_dsqp_set_lane", "csdo_dsqp_download", "csdo_dsqp_last_kernel_seconds", "csdo_dsqp_last_transfer_seconds", "csdo_dsqp_launch_groups", "csdo_dsqp_agent_groups", "csdo_dsqp_set_min_residency_mode", "csdo_dsqp_set_host_results", "csdo_do_phase", "csdo_do_phase_horizon", "csdo_do_phase_cuts",
    "csdo_dsqp_device_solutions",
    "csdo_preprocess", "csdo_preprocess_device", "csdo_preprocess_device_batch", "csdo_preprocess_batch", "csdo_dsqp_estimate_work", "csdo_dsqp_agent_class", "csdo_dsqp_last_limit", "csdo_front_end_gate_draws", "csdo_bridge_free", "csdo_validate", "csdo_validate_frames", "csdo_generate_boxes", "csdo_math_eval", "csdo_vehicle_default",
    "csdo_qp_parm_default", "csdo_backend_name", "csdo_source_hash",
    "csdo_front_end_parm_default", "csdo_front_end_plan", "csdo_paths_free", "csdo_reeds_shepp",
)
