"""ctypes binding of the shipped HIP library (libcsdo_hip.so).  There is no CPU fallback: if the library is missing
or no HIP device is usable, the calls raise."""
import ctypes as C
import os

from . import abi

_HERE = os.path.dirname(os.path.abspath(__file__))
# (CSDO_DIAG_LIB: diagnostic builds of the same library - phase timers, A/B kernel experiments - for scripts/; never set in tests)
LIB_PATH = os.environ.get("CSDO_DIAG_LIB") or os.path.join(_HERE, "libcsdo_hip.so")
_LIB = None


class CsdoError(RuntimeError):
    pass


_ERR = {abi.CSDO_EINVAL: "invalid argument", abi.CSDO_ENODEV: "no usable HIP device (MI355X required)",
        abi.CSDO_ENOMEM: "allocation failed", abi.CSDO_ELIMIT: "problem exceeds a compiled limit (Nt > 512, or an obstacle list that does not fit the LDS beside the horizon)",
        abi.CSDO_EDEVICE: "HIP kernel launch or execution failed"}


def check(rc, what):
    if rc != abi.CSDO_OK:
        raise CsdoError(f"{what} failed: {_ERR.get(rc, rc)} (code {rc})")


def lib():
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise CsdoError(f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; "
                            f"g.build()'` (hipcc --offload-arch=gfx950); there is no CPU fallback")
        L = C.CDLL(LIB_PATH)
        H = C.c_void_p
        L.csdo_backend_name.restype = C.c_char_p
        L.csdo_source_hash.restype = C.c_char_p
        L.csdo_dsqp_create.argtypes = [C.POINTER(H), C.c_int]
        L.csdo_dsqp_create_multi.argtypes = [C.POINTER(H), abi.c_int32_p, C.c_int32]
        L.csdo_dsqp_multi_count.argtypes = [H]
        L.csdo_dsqp_multi_count.restype = C.c_int32
        L.csdo_dsqp_multi_child.argtypes = [H, C.c_int32]
        L.csdo_dsqp_multi_child.restype = H
        L.csdo_dsqp_shard_bounds.argtypes = [abi.c_double_p, C.c_int32, C.c_int32, abi.c_int32_p]
        L.csdo_dsqp_destroy.argtypes = [H]
        L.csdo_dsqp_destroy.restype = None
        L.csdo_dsqp_solve.argtypes = [H, C.POINTER(abi.Problem), C.POINTER(abi.Result)]
        L.csdo_dsqp_solve_batch.argtypes = [H, C.POINTER(abi.Problem), C.c_int32, C.POINTER(abi.Result)]
        L.csdo_dsqp_upload.argtypes = [H, C.POINTER(abi.Problem), C.c_int32]
        L.csdo_dsqp_run.argtypes = [H, C.c_void_p]
        L.csdo_dsqp_download.argtypes = [H, C.POINTER(abi.Result), C.c_int32]
        L.csdo_dsqp_last_kernel_seconds.argtypes = [H]
        L.csdo_dsqp_last_kernel_seconds.restype = C.c_double
        L.csdo_dsqp_last_transfer_seconds.argtypes = [H, C.POINTER(C.c_double * 5)]
        L.csdo_dsqp_launch_groups.argtypes = [H, C.POINTER(abi.LaunchGroup), C.c_int32]
        L.csdo_dsqp_launch_groups.restype = C.c_int32
        L.csdo_dsqp_set_min_residency_mode.argtypes = [H, C.c_int32]
        L.csdo_dsqp_set_host_results.argtypes = [H, C.c_int32]
        L.csdo_do_phase.argtypes = [H, C.POINTER(abi.CoarseWorld), C.c_int32, C.POINTER(abi.Vehicle), C.POINTER(abi.QpParm),
                                    C.POINTER(abi.Result), abi.c_int32_p, C.POINTER(abi.DoPhaseTiming)]
        L.csdo_do_phase_horizon.argtypes = [abi.c_int32_p, C.c_int32, C.POINTER(abi.QpParm)]
        L.csdo_do_phase_horizon.restype = C.c_int32
        L.csdo_do_phase_cuts.argtypes = [abi.c_int32_p, C.c_int32, C.c_int32, abi.c_int32_p]
        L.csdo_dsqp_agent_groups.argtypes = [H, abi.c_int32_p, C.c_int32]
        L.csdo_dsqp_device_solutions.argtypes = [H, C.POINTER(C.c_int64)]
        L.csdo_dsqp_device_solutions.restype = C.c_void_p
        L.csdo_preprocess.argtypes = [abi.c_double_p, abi.c_int32_p, abi.c_int32_p, C.c_int32, abi.c_double_p,
                                      C.POINTER(abi.Vehicle), C.POINTER(abi.QpParm), C.POINTER(abi.BridgeOut)]
        L.csdo_preprocess_device.argtypes = [H] + L.csdo_preprocess.argtypes
        PP = C.POINTER
        L.csdo_preprocess_device_batch.argtypes = [H, C.c_int32, PP(abi.c_double_p), PP(abi.c_int32_p), PP(abi.c_int32_p),
                                                   abi.c_int32_p, PP(abi.c_double_p), PP(abi.Vehicle), PP(abi.QpParm),
                                                   PP(abi.BridgeOut)]
        L.csdo_dsqp_last_limit.argtypes = [H, PP(C.c_int32), PP(C.c_int32), PP(C.c_int64)]
        L.csdo_dsqp_estimate_work.argtypes = [PP(abi.Problem), C.c_int32, abi.c_double_p]
        L.csdo_dsqp_agent_class.argtypes = [C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_int64)]
        L.csdo_validate.argtypes = [H, abi.c_double_p, C.c_int32, C.c_int32, abi.c_double_p, C.c_int32, C.c_double,
                                    C.c_double, C.POINTER(abi.Vehicle), C.c_double, C.POINTER(abi.Validation)]
        L.csdo_validate_frames.argtypes = [H, abi.c_double_p, C.c_int32, C.c_int32, C.c_int32, abi.c_double_p, C.c_int32,
                                           C.c_double, C.c_double, C.POINTER(abi.Vehicle), C.c_double, C.POINTER(abi.Validation)]
        L.csdo_bridge_free.argtypes = [C.POINTER(abi.BridgeOut)]
        L.csdo_bridge_free.restype = None
        L.csdo_generate_boxes.argtypes = [H, abi.c_double_p, C.c_int32, abi.c_double_p, C.c_int32, C.c_double,
                                          C.c_double, C.POINTER(abi.Vehicle), abi.c_double_p, abi.c_int32_p]
        L.csdo_math_eval.argtypes = [H, C.c_int32, abi.c_double_p, abi.c_double_p, abi.c_double_p, C.c_int32]
        L.csdo_vehicle_default.argtypes = [C.POINTER(abi.Vehicle)]
        L.csdo_vehicle_default.restype = None
        L.csdo_qp_parm_default.argtypes = [C.POINTER(abi.Vehicle), C.POINTER(abi.QpParm)]
        L.csdo_qp_parm_default.restype = None
        L.csdo_front_end_parm_default.argtypes = [C.POINTER(abi.FrontEndParm)]
        L.csdo_front_end_parm_default.restype = None
        L.csdo_front_end_plan.argtypes = [abi.c_double_p, abi.c_double_p, C.c_int32, C.c_double, C.c_double,
                                          abi.c_double_p, C.c_int32, C.POINTER(abi.Vehicle),
                                          C.POINTER(abi.FrontEndParm), C.POINTER(abi.Paths)]
        L.csdo_paths_free.argtypes = [C.POINTER(abi.Paths)]
        L.csdo_paths_free.restype = None
        L.csdo_reeds_shepp.argtypes = [C.c_double * 3, C.c_double * 3, C.c_double, C.c_int32 * 5, C.c_double * 5]
        L.csdo_front_end_gate_draws.argtypes = [C.c_uint32, C.c_int32, C.c_int32, C.POINTER(C.c_uint32)]
        L.csdo_reeds_shepp.restype = C.c_double
        _LIB = L
    return _LIB


def source_hash_of_tree():
    """What csrc/Makefile bakes into the library (csdo_source_hash), recomputed from the tree: the first 16 hex digits of the SHA-256
    over the device sources in the Makefile's order.  Equal to lib().csdo_source_hash() when the library is a build of this tree."""
    import hashlib
    import re
    csrc = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
    with open(os.path.join(csrc, "Makefile")) as f:
        mk = f.read()
    hdrs = re.search(r"^HDRS = (.*)$", mk, re.M).group(1).split()
    hips = re.search(r"^HIPSRC = (.*)$", mk, re.M).group(1).split()
    h = hashlib.sha256()
    for name in sorted(hdrs + hips):          # make's $(sort ...): lexical, duplicates removed
        with open(os.path.join(csrc, name), "rb") as f:
            h.update(f.read())
    h.update(b"\n")                           # (the Makefile appends its XDEFS - empty for the shipped library - and a newline)
    return h.hexdigest()[:16]
