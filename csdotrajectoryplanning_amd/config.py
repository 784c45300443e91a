"""Vehicle / optimizer parameters evaluated the way the reference does.

readAgentConfig (common/motion_planning.cc:54-93) stores YAML doubles into `float` statics and derives f2x, r2x,
rv from them; readQpSolverConfig (sqp/utils.cc:34-59) forms dt from the float product r*deltat.  The defaults
below are the reference's shipped config.yaml:4-57.
"""
import numpy as np
import yaml

from .abi import QpParm, Vehicle

_f32 = np.float32

DEFAULT_CONFIG = dict(r=3, deltat=0.706, carWidth=2.0, LF=2.0, LB=1.0, WB=1.0, obsRadius=0.8, max_omega=0.07,
                      max_v=1, decelerate_factor=0.8, fixed_corridor=False, max_iter=10,
                      delta_solution_threshold=1, max_violation=0.001, osqp_max_iter=400, r_trust=2.0,
                      num_interpolation=2)


def vehicle_from_config(cfg=None) -> Vehicle:
    cfg = {**DEFAULT_CONFIG, **(cfg or {})}
    r, deltat = _f32(cfg["r"]), _f32(cfg["deltat"])
    LF, LB, W, WB = _f32(cfg["LF"]), _f32(cfg["LB"]), _f32(cfg["carWidth"]), _f32(cfg["WB"])
    # motion_planning.cc:82-85: double expressions over float operands, stored back into floats
    f2x = _f32(0.25 * (3.0 * float(LF) - float(LB)))
    r2x = _f32(0.25 * (float(LF) - 3.0 * float(LB)))
    lsum = float(_f32(LF + LB))
    rv = _f32(0.5 * (lsum ** 2 / 4 + float(_f32(W * W))) ** 0.5)
    v = Vehicle()
    v.r, v.deltat, v.LF, v.LB, v.car_width, v.WB = map(float, (r, deltat, LF, LB, W, WB))
    v.f2x, v.r2x, v.rv = float(f2x), float(r2x), float(rv)
    v.obs_radius = float(_f32(cfg["obsRadius"]))
    return v


def qp_parm_from_config(cfg=None, adaptive_rho_interval=25, solve_refinement=0) -> QpParm:
    cfg = {**DEFAULT_CONFIG, **(cfg or {})}
    p = QpParm()
    p.r_trust = float(cfg["r_trust"])
    p.max_omega = float(cfg["max_omega"])
    p.max_v = float(cfg["max_v"])
    p.max_iter = float(cfg["max_iter"])
    p.delta_solution_threshold = float(cfg["delta_solution_threshold"])
    p.max_violation = float(cfg["max_violation"])
    p.osqp_max_iter = int(cfg["osqp_max_iter"])
    p.num_interpolation = int(cfg["num_interpolation"])
    # utils.cc:55-56: float product, then double divisions
    step = float(_f32(_f32(cfg["r"]) * _f32(cfg["deltat"])))
    p.dt = step / p.max_v / (p.num_interpolation + 1) / float(cfg["decelerate_factor"])
    p.fixed_corridor = int(bool(cfg["fixed_corridor"]))
    p.adaptive_rho_interval = int(adaptive_rho_interval)
    # csdo_qp_parm::solve_refinement (include/csdo_dsqp.h): 1 = a second solve on the KKT residual per iteration (~1.8 x the time),
    # 2 = the lagged form (~1.3 x); True counts as 1
    p.solve_refinement = int(solve_refinement) if int(solve_refinement) in (1, 2) else 0
    return p


def load_config_yaml(path):
    with open(path) as f:
        return yaml.safe_load(f)
