"""Host-side tools on the output side of the DO path.

write_solutions   the reference's result YAML (dumpSolutions, sqp/inter_agent_cons.cc:411-453): same keys in the same
                  order (scripts/analysis_result.py reads the header positionally), %.3f, steer / omega in "degrees" with
                  the reference's 180/3.14 factor, no v / omega on the last timestep.
read_solutions    the inverse, for round-trip tests and for feeding the validator from a file.
write_corridors   dumpCorridors (sqp/utils.cc:62-89): per agent and timestep the front and the rear disc centre (float-rounded, as
                  State stores them) with its box, default ostream formatting; read_corridors is the inverse.
write_guesses     the initial-guess dump of csdo.cc:138-140 (dumpSolutions on x0_bar); output_paths derives the three file
                  names the way csdo.cc:76,139,164 does.
interpolate_states  getState of scripts/visualize.py:256-281: what the authors' per-frame collision print looks at between states.
validate          independent geometric check of final trajectories in the spirit of scripts/collision_detection.py:
                  vehicle rectangles against each other (separating axes) and against the circular obstacles, per
                  timestep.  Own formulation, vectorised numpy; not on the hot path.
"""
from dataclasses import dataclass

import numpy as np

_HEADER = ("cost", "makespan", "flowtime", "runtime", "runtime_search", "runtime_preprocess", "runtime_optimization",
           "runtime_decentralized_optimization", "search_status", "solver_status")
_DEG = 180 / 3.14   # the reference's constant, not pi


def write_solutions(path, solutions, stats=None):
    """solutions [Na][Nt][6] = x, y, yaw, steer, v, d_steer; stats: dict with any of the header keys.  Missing keys take the
    defaults of the reference's SolutionStatistics (sqp/common.h:25-36): -1 for the floats (cost, makespan, flowtime stay
    -1 for csdo runs), search_status 2, solver_status 0."""
    stats = stats or {}
    sol = np.asarray(solutions, dtype=np.float64)
    Na, Nt = sol.shape[:2]
    out = ["statistics:"]
    for k in _HEADER:
        v = stats.get(k, {"search_status": 2, "solver_status": 0}.get(k, -1))
        out.append("  %s: %s" % (k, ("%d" % v) if k.endswith("_status") else ("%.3f" % float(v))))
    out.append("schedule:")
    for a in range(Na):
        out.append("  agent%d:" % a)
        for t in range(Nt):
            x, y, yaw, steer, v, w = sol[a, t]
            out.append("    - x: %.3f" % x)
            out.append("      y: %.3f" % y)
            out.append("      yaw: %.3f" % yaw)
            out.append("      steer: %.3f" % (steer * _DEG))
            out.append("      t: %d" % t)
            if t == Nt - 1:
                continue
            out.append("      v: %.3f" % v)
            out.append("      omega: %.3f" % (w * _DEG))
    with open(path, "w") as f:
        f.write("\n".join(out) + "\n")


def solution_statistics(sol, rt_search=-1.0, rt_preprocess=-1.0, search_status=2):
    """The reference's SolutionStatistics (sqp/common.h:25-36) for a DO-phase result, with the reference's semantics
    (csdo.cc:98-161, sqp/dsqp_solver.cc:1194-1248) mapped onto this backend's clocks:
      runtime_optimization                  wall clock of the SolverDSQP construction = `sol.t_total` (upload + kernels + download)
      runtime_decentralized_optimization    "ideal parallel" time: the slowest agent + everything outside the per-agent loop.  The
                                            reference adds the slowest agent's SQP time, the shared overhead and the slowest agent's
                                            initial-corridor time; here an agent's device time already contains its corridors, so it
                                            is `t_max_individual + (t_total - t_device)`
      runtime                               runtime_search + runtime_preprocess + runtime_decentralized_optimization (csdo.cc:160-161)
      search_status                         lowered to 1 ("minor collision") when the initial guess was not statically legal (csdo.cc:152-154)
    cost / makespan / flowtime stay -1 as in csdo runs.  Returns the dict write_solutions takes."""
    other = max(float(sol.t_total) - float(sol.t_device), 0.0)
    rt_max = float(sol.t_max_individual) + other
    if not sol.initial_static_legal:
        search_status = 1
    known = [t for t in (rt_search, rt_preprocess) if t >= 0]
    return {"runtime": (sum(known) + rt_max) if len(known) == 2 else -1.0, "runtime_search": rt_search,
            "runtime_preprocess": rt_preprocess, "runtime_optimization": float(sol.t_total),
            "runtime_decentralized_optimization": rt_max, "search_status": int(search_status),
            "solver_status": int(sol.solver_status)}


def output_paths(output_file):
    """(result, guesses, corridors) file names as csdo.cc:76,139,164 derives them: the output file must end in `.yaml`;
    the prefix is everything but those five characters."""
    output_file = str(output_file)
    if not output_file.endswith(".yaml"):
        raise ValueError("the output file must end with .yaml (csdo.cc:76 cuts five characters)")
    prefix = output_file[:-5]
    return output_file, prefix + "_guesses.yaml", prefix + "_corridors.yaml"


def write_guesses(path, x0_bar, stats=None):
    """csdo.cc:138-140 (--initial_guess): the interpolated initial guess in the result format, with the statistics gathered so
    far (the reference passes the same SolutionStatistics object: search and preprocess times set, the rest at its defaults)."""
    write_solutions(path, x0_bar, stats)


def _g(v):
    """A double through a default-constructed std::ofstream: %g with six significant digits."""
    return "%g" % float(v)


def write_corridors(path, corridors, x0_bar, veh):
    """dumpCorridors, sqp/utils.cc:62-89.  corridors [Na][Nt][8] = xf_min, xf_max, yf_min, yf_max, xr_min, xr_max, yr_min, yr_max
    (csdo_result.corridors); x0_bar [Na][Nt][>=3].  Two list items per timestep: [xf, yf, xf_min, xf_max, yf_min, yf_max] then
    the same for the rear disc; the disc centres are State's float members (common/motion_planning.h:111-118)."""
    cor = np.asarray(corridors, dtype=np.float64)
    g = np.asarray(x0_bar, dtype=np.float64)
    Na, Nt = cor.shape[:2]
    f32 = lambda v: np.asarray(v, dtype=np.float64).astype(np.float32).astype(np.float64)
    xf, yf = f32(g[..., 0] + veh.f2x * np.cos(g[..., 2])), f32(g[..., 1] + veh.f2x * np.sin(g[..., 2]))
    xr, yr = f32(g[..., 0] + veh.r2x * np.cos(g[..., 2])), f32(g[..., 1] + veh.r2x * np.sin(g[..., 2]))
    out = []
    for a in range(Na):
        out.append("agent%d:" % a)
        for t in range(Nt):
            c = cor[a, t]
            out.append("  - [" + ", ".join(_g(v) for v in (xf[a, t], yf[a, t], c[0], c[1], c[2], c[3])) + "]")
            out.append("  - [" + ", ".join(_g(v) for v in (xr[a, t], yr[a, t], c[4], c[5], c[6], c[7])) + "]")
    with open(path, "w") as f:
        f.write("\n".join(out) + "\n")


def read_corridors(path):
    """Inverse of write_corridors: (centres [Na][Nt][4] = xf, yf, xr, yr, corridors [Na][Nt][8])."""
    agents, cur = [], None
    with open(path) as f:
        for line in f:
            body = line.strip()
            if not body:
                continue
            if body.startswith("agent") and body.endswith(":"):
                cur = []
                agents.append(cur)
            else:
                cur.append([float(v) for v in body[body.index("[") + 1:body.rindex("]")].split(",")])
    arr = np.array(agents, dtype=np.float64)                      # [Na][2 Nt][6]
    front, rear = arr[:, 0::2], arr[:, 1::2]
    centres = np.concatenate([front[..., :2], rear[..., :2]], -1)
    return centres, np.concatenate([front[..., 2:], rear[..., 2:]], -1)


def read_solutions(path):
    """Returns (solutions [Na][Nt][6], stats dict) from a file in the reference's result format."""
    stats, agents, cur, rec = {}, [], None, None
    section = None
    with open(path) as f:
        for raw in f:
            line = raw.rstrip("\n")
            if not line.strip():
                continue
            if line.startswith("statistics:"):
                section = "stat"
                continue
            if line.startswith("schedule:"):
                section = "sched"
                continue
            body = line.strip()
            if section == "stat":
                k, v = body.split(":", 1)
                stats[k.strip()] = float(v)
            elif section == "sched":
                if body.startswith("agent") and body.endswith(":"):
                    cur = []
                    agents.append(cur)
                    continue
                if body.startswith("- "):
                    rec = {}
                    cur.append(rec)
                    body = body[2:]
                k, v = body.split(":", 1)
                rec[k.strip()] = float(v)
    Na, Nt = len(agents), max(len(a) for a in agents)
    sol = np.zeros((Na, Nt, 6))
    for a, recs in enumerate(agents):
        for t, r in enumerate(recs):
            sol[a, t] = [r["x"], r["y"], r["yaw"], r["steer"] / _DEG, r.get("v", 0.0), r.get("omega", 0.0) / _DEG]
    return sol, stats


@dataclass
class ValidationReport:
    vehicle_collisions: int          # (timestep, i, j) triples with overlapping rectangles
    obstacle_collisions: int         # (timestep, agent, obstacle) triples
    out_of_map: int                  # (timestep, agent) pairs with a rectangle corner outside the map
    first_vehicle_collision: tuple   # (t, i, j) or None
    first_obstacle_collision: tuple  # (t, agent, obstacle) or None
    min_obstacle_clearance: float    # smallest distance rectangle - obstacle disc over the trajectory (negative: overlap)

    @property
    def ok(self):
        return self.vehicle_collisions == 0 and self.obstacle_collisions == 0 and self.out_of_map == 0


def _rect_frames(sol, veh):
    """Centres [Na,Nt,2], unit heading [Na,Nt,2], half length, half width of the vehicle rectangles.  The reference point
    (x, y) sits LB in front of the rear bumper: the body spans [-LB, +LF] along the heading (common/motion_planning.h)."""
    x, y, yaw = sol[..., 0], sol[..., 1], sol[..., 2]
    u = np.stack([np.cos(yaw), np.sin(yaw)], -1)
    c = np.stack([x, y], -1) + 0.5 * (veh.LF - veh.LB) * u
    return c, u, 0.5 * (veh.LF + veh.LB), 0.5 * veh.car_width


def _pose_frames(p, veh, margin=0.0):
    p = np.asarray(p, dtype=np.float64)
    u = np.stack([np.cos(p[..., 2]), np.sin(p[..., 2])], -1)
    c = p[..., :2] + 0.5 * (veh.LF - veh.LB) * u
    n = np.stack([-u[..., 1], u[..., 0]], -1)
    return c, u, n, 0.5 * (veh.LF + veh.LB) + margin, 0.5 * veh.car_width + margin


def rect_rect_collision(p, q, veh, margin=0.0):
    """Vehicle rectangles at rear-axle poses p, q [..., 3] overlap (separating axes; touching counts, as in the
    reference's collision_rect_and_rect, scripts/collision_detection.py:20-56).  Pinned to the reference's verdicts by
    tests/golden/ref_collision_verdicts.npz."""
    ca, ua, na, hl, hw = _pose_frames(p, veh, margin)
    cb, ub, nb, _, _ = _pose_frames(q, veh, margin)
    d = cb - ca
    def sep(ax_u, ax_n, ou, on):
        eu = hl * np.abs((ou * ax_u).sum(-1)) + hw * np.abs((on * ax_u).sum(-1)) + hl
        en = hl * np.abs((ou * ax_n).sum(-1)) + hw * np.abs((on * ax_n).sum(-1)) + hw
        return (np.abs((d * ax_u).sum(-1)) <= eu) & (np.abs((d * ax_n).sum(-1)) <= en)
    return sep(ua, na, ub, nb) & sep(ub, nb, ua, na)


def circle_rect_distance(p, obs, veh, margin=0.0):
    """Signed distance between the vehicle rectangle at pose p [..., 3] and the disc obs [..., 3] = x, y, r (negative:
    overlap; the reference's collision_circle_and_rect, scripts/collision_detection.py:59-96, returns True exactly then)."""
    c, u, n, hl, hw = _pose_frames(p, veh, margin)
    ob = np.asarray(obs, dtype=np.float64)
    rel = ob[..., :2] - c
    lx = np.abs((rel * u).sum(-1)) - hl
    ly = np.abs((rel * n).sum(-1)) - hw
    return np.hypot(np.maximum(lx, 0), np.maximum(ly, 0)) + np.minimum(np.maximum(lx, ly), 0) - ob[..., 2]


def interpolate_states(solutions, frames_per_move):
    """The poses the authors' animation checks, frame by frame (getState, scripts/visualize.py:256-281): frame f looks at
    time f / frames_per_move; between two states x, y and yaw are interpolated linearly - `(next - last) * tau + last`, also
    at whole times, where tau = 1 - after the earlier yaw has been moved by 2 pi towards the later one when they are more
    than pi apart.  Returns [Na][(Nt - 1) * frames_per_move + 1][3]."""
    sol = np.asarray(solutions, dtype=np.float64)[..., :3]
    S = int(frames_per_move)
    if S < 1:
        raise ValueError("frames_per_move must be >= 1")
    Na, Nt = sol.shape[:2]
    f = np.arange((Nt - 1) * S + 1)
    t = f / float(S)
    idx = np.ceil(t).astype(np.int64)                               # first state whose time is >= t
    last, nxt = sol[:, np.maximum(idx - 1, 0)].copy(), sol[:, idx]
    dyaw = last[..., 2] - nxt[..., 2]
    last[..., 2] = np.where(dyaw > np.pi, last[..., 2] - 2 * np.pi, np.where(-dyaw > np.pi, last[..., 2] + 2 * np.pi, last[..., 2]))
    tau = (t - (idx - 1)) / 1
    out = (nxt - last) * tau[None, :, None] + last
    out[:, idx == 0] = sol[:, :1]
    return out


def validate(solutions, veh, obstacles=None, dimx=None, dimy=None, margin=0.0, frames_per_move=None, participants=None) -> ValidationReport:
    """solutions [Na][Nt][>=3]; obstacles [n][3] = x, y, r.  `margin` inflates every vehicle rectangle on all sides.
    frames_per_move = None: the Nt states as they are.  An integer S >= 1: the frames of the authors' animation
    (interpolate_states: S frames per move, scripts/visualize.py:219-247); every index in the report is then a frame.
    participants: a dict that receives WHO collides - "vehicle_pairs": [(i, j, frames in collision)], "obstacle_agents":
    [(agent, frames in collision)] - beside the counts of the report."""
    sol = np.asarray(solutions, dtype=np.float64)
    if frames_per_move is not None:
        sol = interpolate_states(sol, frames_per_move)
    Na, Nt = sol.shape[:2]
    c, u, hl, hw = _rect_frames(sol, veh)
    hl, hw = hl + margin, hw + margin
    n = np.stack([-u[..., 1], u[..., 0]], -1)                       # unit normal
    # ---- rectangle / rectangle: separating axes = the two axes of either rectangle
    veh_hits, first_v = 0, None
    if Na > 1:
        d = c[None, :, :, :] - c[:, None, :, :]                     # [i, j, t, 2]  centre j - centre i
        def gap(axis_owner):                                        # axes of rectangle i (0) or j (1), both axes at once
            ua = u[:, None] if axis_owner == 0 else u[None, :]
            na = n[:, None] if axis_owner == 0 else n[None, :]
            uo = u[None, :] if axis_owner == 0 else u[:, None]
            no = n[None, :] if axis_owner == 0 else n[:, None]
            ext_u = hl * np.abs((uo * ua).sum(-1)) + hw * np.abs((no * ua).sum(-1)) + hl
            ext_n = hl * np.abs((uo * na).sum(-1)) + hw * np.abs((no * na).sum(-1)) + hw
            return np.abs((d * ua).sum(-1)) <= ext_u, np.abs((d * na).sum(-1)) <= ext_n
        a1, a2 = gap(0)
        b1, b2 = gap(1)
        overlap = a1 & a2 & b1 & b2                                 # no separating axis
        iu = np.triu_indices(Na, 1)
        ov = overlap[iu]                                            # [pairs, t]
        veh_hits = int(ov.sum())
        if veh_hits:
            p, t = np.argwhere(ov)[np.argmin(np.argwhere(ov)[:, 1])]
            first_v = (int(t), int(iu[0][p]), int(iu[1][p]))
        if participants is not None:
            participants["vehicle_pairs"] = [(int(iu[0][p_]), int(iu[1][p_]), int(ov[p_].sum())) for p_ in np.nonzero(ov.any(axis=1))[0]]
    # ---- circle / rectangle: distance from the disc centre to the rectangle in the rectangle's frame
    obs_hits, first_o, clearance = 0, None, np.inf
    if obstacles is not None and len(obstacles):
        ob = np.asarray(obstacles, dtype=np.float64).reshape(-1, 3)
        rel = ob[None, None, :, :2] - c[:, :, None, :]               # [a, t, o, 2]
        lx = np.abs((rel * u[:, :, None, :]).sum(-1)) - hl
        ly = np.abs((rel * n[:, :, None, :]).sum(-1)) - hw
        outside = np.hypot(np.maximum(lx, 0), np.maximum(ly, 0))
        inside = np.minimum(np.maximum(lx, ly), 0)
        dist = outside + inside - ob[None, None, :, 2]
        clearance = float(dist.min())
        hit = dist < 0
        obs_hits = int(hit.sum())
        if obs_hits:
            idx = np.argwhere(hit)
            a, t, o = idx[np.argmin(idx[:, 1])]
            first_o = (int(t), int(a), int(o))
        if participants is not None:
            participants["obstacle_agents"] = [(int(a_), int(hit[a_].any(axis=1).sum())) for a_ in np.nonzero(hit.any(axis=(1, 2)))[0]]
    out = 0
    if dimx is not None and dimy is not None:
        corners = [c + sx * hl * u + sy * hw * n for sx in (-1, 1) for sy in (-1, 1)]
        bad = np.zeros((Na, Nt), bool)
        for k in corners:
            bad |= (k[..., 0] < 0) | (k[..., 0] > dimx) | (k[..., 1] < 0) | (k[..., 1] > dimy)
        out = int(bad.sum())
    return ValidationReport(veh_hits, obs_hits, out, first_v, first_o, clearance)


def feasibility(world, solutions):
    """Per-agent acceptance measures of a DO-phase result, independent of any solver: the kinematic and separating-plane
    parts of the reference's own isFeasible (sqp/dsqp_solver.cc:292-420: mean-square bicycle-model residual, largest
    plane violation with the exact disc centres) and the QP objective 1/2 sum (v_{k+1}-v_k)^2 + 1/2 sum w^2 (:163-197).
    Returns dict(kin [Na], planes [Na], objective [Na]); isFeasible's thresholds are 1e-2 and 1e-1."""
    sol = np.asarray(solutions, dtype=np.float64)
    Na, Nt = sol.shape[:2]
    P, V = world.parm, world.veh
    x, y, yaw, st, v, w = (sol[..., k] for k in range(6))
    r1 = x[:, :-1] + v[:, :-1] * np.cos(yaw[:, :-1]) * P.dt - x[:, 1:]
    r2 = y[:, :-1] + v[:, :-1] * np.sin(yaw[:, :-1]) * P.dt - y[:, 1:]
    r3 = yaw[:, :-1] + v[:, :-1] * np.tan(st[:, :-1]) / V.WB * P.dt - yaw[:, 1:]
    r4 = st[:, :-1] + w[:, :-1] * P.dt - st[:, 1:]
    kin = ((r1 ** 2).sum(1) + (r2 ** 2).sum(1) + (r3 ** 2).sum(1) + (r4 ** 2).sum(1)) / Nt
    xf, yf = x + V.f2x * np.cos(yaw), y + V.f2x * np.sin(yaw)
    xr, yr = x + V.r2x * np.cos(yaw), y + V.r2x * np.sin(yaw)
    planes = np.zeros(Na)
    po = world.plane_off
    for a in range(Na):
        pl = world.planes[po[a]:po[a + 1]]
        if len(pl) == 0:
            continue
        t, c = pl["t"], pl["c"]
        res = np.stack([c[:, 0] * xf[a, t] + c[:, 1] * yf[a, t] + c[:, 2], c[:, 3] * xf[a, t] + c[:, 4] * yf[a, t] + c[:, 5],
                        c[:, 6] * xr[a, t] + c[:, 7] * yr[a, t] + c[:, 8], c[:, 9] * xr[a, t] + c[:, 10] * yr[a, t] + c[:, 11]])
        planes[a] = max(0.0, float(res.max()))
    vv, ww = v[:, :-1], w[:, :-1]
    objective = 0.5 * ((vv[:, 1:] - vv[:, :-1]) ** 2).sum(1) + 0.5 * (ww ** 2).sum(1)
    return dict(kin=kin, planes=planes, objective=objective)
