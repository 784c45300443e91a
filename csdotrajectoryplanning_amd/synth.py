"""Deterministic stand-in for the reference front end (PBS + hybrid A*), which is out of scope for this backend.

The DO phase needs coarse per-agent paths (states + motion-primitive actions, `PlanResult<State,Action,double>` in
the reference).  The reference's own front end cannot be built here, so inputs come from a *prioritised primitive
roll-out*: agents are planned one after another from the instance's start poses with the reference's motion
primitives (common/motion_planning.cc:95-108: forward straight / right / left, plus wait), rejecting steps that
leave the map, enter an inflated obstacle, or come too close to an earlier agent at the same timestep — the same
sequential scheme PBS uses for its root node (pbs/PBS.cc:665-719).  The goal of every agent is defined as the final
pose of its roll-out.  Everything is driven by a SplitMix64 stream seeded with 1000*instance_seed + agent, so the
paths are reproducible bit for bit on any machine.  Reports name this generator wherever its inputs are used.
"""
import math

import numpy as np

from .abi import Vehicle

GENERATOR_NAME = "prioritised-primitive-rollout-v3"
_MASK = (1 << 64) - 1


class SplitMix64:
    def __init__(self, seed):
        self.s = seed & _MASK

    def next_u64(self):
        self.s = (self.s + 0x9E3779B97F4A7C15) & _MASK
        z = self.s
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & _MASK
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & _MASK
        return z ^ (z >> 31)

    def uniform(self):
        return (self.next_u64() >> 11) * (1.0 / (1 << 53))

    def randint(self, lo, hi):  # inclusive
        return lo + int(self.uniform() * (hi - lo + 1))


def _discs(x, y, yaw, veh):
    c, s = math.cos(yaw), math.sin(yaw)
    return (x + veh.f2x * c, y + veh.f2x * s, x + veh.r2x * c, y + veh.r2x * s)


def _static_ok(x, y, yaw, inst, veh, margin):
    xf, yf, xr, yr = _discs(x, y, yaw, veh)
    lo = veh.rv + 1e-2
    for px, py in ((xf, yf), (xr, yr)):
        if px < lo or px > inst.dimx - lo or py < lo or py > inst.dimy - lo:
            return False
        for ox, oy, orad in inst.obstacles:
            if abs(px - ox) < orad + veh.rv + margin and abs(py - oy) < orad + veh.rv + margin:
                return False
    return True


def _sub_discs(p0, p1, veh, nsub=3):
    """Disc centres [nsub,4] of poses interpolated from p0 (exclusive) to p1 (inclusive)."""
    out = np.empty((nsub, 4))
    dyaw = (p1[2] - p0[2] + math.pi) % (2.0 * math.pi) - math.pi
    for k in range(nsub):
        s = (k + 1) / nsub
        out[k] = _discs(p0[0] + s * (p1[0] - p0[0]), p0[1] + s * (p1[1] - p0[1]), p0[2] + s * dyaw, veh)
    return out


def _dynamic_ok(cur, cand, t, fine, veh, sep, nsub=3):
    """fine: [n_other, T_fine, 4] disc centres of the other agents at sub-step resolution (index nsub*t)."""
    if fine.shape[0] == 0:
        return True
    mine = _sub_discs(cur, cand, veh, nsub)                     # [nsub,4]
    idx = np.minimum(nsub * (t - 1) + 1 + np.arange(nsub), fine.shape[1] - 1)
    other = fine[:, idx, :]                                     # [n, nsub, 4]
    s2 = sep * sep
    for (ax, ay) in ((0, 1), (2, 3)):
        for (bx, by) in ((0, 1), (2, 3)):
            d2 = (mine[None, :, ax] - other[:, :, bx]) ** 2 + (mine[None, :, ay] - other[:, :, by]) ** 2
            if (d2 < s2).any():
                return False
    return True


def _rest_ok(pose, t_last, fine, veh, sep, nsub=3):
    """True if resting at `pose` from coarse step t_last onwards never comes within `sep` of another agent."""
    if fine.shape[0] == 0:
        return True
    mine = np.array(_discs(pose[0], pose[1], pose[2], veh))
    other = fine[:, nsub * t_last:, :]
    s2 = sep * sep
    for (ax, ay) in ((0, 1), (2, 3)):
        for (bx, by) in ((0, 1), (2, 3)):
            d2 = (mine[ax] - other[:, :, bx]) ** 2 + (mine[ay] - other[:, :, by]) ** 2
            if (d2 < s2).any():
                return False
    return True


def rollout_paths(inst, veh: Vehicle, instance_seed=0, len_scale=1.0, nsub=3):
    """Returns (states_list, actions_list, goals): per agent [L,3] float64 poses, [L-1] int32 actions, [Na,3] goals."""
    step = veh.r * veh.deltat
    arc = veh.r * math.sin(veh.deltat)
    lat = veh.r * (1.0 - math.cos(veh.deltat))
    prim = {0: (step, 0.0, 0.0), 1: (arc, -lat, -veh.deltat), 2: (arc, lat, veh.deltat),
            3: (-step, 0.0, 0.0), 4: (-arc, -lat, veh.deltat), 5: (-arc, lat, -veh.deltat)}
    sep = 2.0 * veh.rv + 0.3
    Na = inst.num_agents
    # horizon bound for the sub-step table
    lens = []
    for a in range(Na):
        d = math.hypot(inst.goals[a][0] - inst.starts[a][0], inst.goals[a][1] - inst.starts[a][1])
        lens.append(max(2, int(math.ceil(d / step) * len_scale)))
    T_fine = nsub * int(math.ceil(1.3 * max(lens))) + 2
    # every agent occupies its start pose until it has been planned (later agents are static blockers)
    fine = np.empty((Na, T_fine, 4))
    for a in range(Na):
        fine[a, :, :] = _discs(*(float(v) for v in inst.starts[a]), veh)
    states_list, actions_list = [], []

    def successor(pose, act):
        x, y, yaw = pose
        if act == 6:
            return pose
        dx, dy, dyaw = prim[act]
        return (x + dx * math.cos(yaw) - dy * math.sin(yaw), y + dx * math.sin(yaw) + dy * math.cos(yaw),
                (yaw + dyaw) % (2.0 * math.pi))

    for a in range(Na):
        rng = SplitMix64(1000 * instance_seed + a)
        others = np.delete(fine, a, axis=0)
        cur = tuple(float(v) for v in inst.starts[a])
        L = rng.randint(lens[a], int(math.ceil(1.3 * lens[a])))
        L_cap = (T_fine - 2) // nsub
        poses, acts = [cur], []
        backtracks = 0
        while True:
            t = len(acts) + 1
            if t > L:
                # the agent rests at its last pose for the rest of the horizon: that spot has to stay free,
                # otherwise keep rolling (bounded by the table horizon)
                if t > L_cap or _rest_ok(cur, t - 1, others, veh, sep, nsub):
                    break
            chosen = None
            tries = []
            for _ in range(8):
                u = rng.uniform()
                tries.append(0 if u < 0.6 else (1 if u < 0.75 else (2 if u < 0.9 else 6)))
            for act in tries + [1, 2, 0, 6, 3, 4, 5]:  # escape hatch: the reversing primitives
                cand = successor(cur, act)
                if (act == 6 or _static_ok(cand[0], cand[1], cand[2], inst, veh, 0.15)) and \
                        _dynamic_ok(cur, cand, t, others, veh, sep, nsub):
                    chosen = (act, cand)
                    break
            if chosen is None:
                if backtracks < 60 and len(acts) > 0:
                    # boxed in by an earlier agent: undo up to 3 steps and roll again with fresh draws
                    backtracks += 1
                    for _ in range(min(3, len(acts))):
                        acts.pop()
                        poses.pop()
                    cur = poses[-1]
                    continue
                chosen = (6, cur)
            act, cur = chosen
            poses.append(cur)
            acts.append(act)
        L = len(acts)
        # publish this agent's sub-step disc table
        fine[a, 0, :] = _discs(*poses[0], veh)
        for t in range(1, L + 1):
            sd = _sub_discs(poses[t - 1], poses[t], veh, nsub)
            fine[a, nsub * (t - 1) + 1: nsub * t + 1, :] = sd
        fine[a, nsub * L + 1:, :] = fine[a, nsub * L, :]
        states_list.append(np.array(poses, dtype=np.float64))
        actions_list.append(np.array(acts, dtype=np.int32))
    goals = np.array([s[-1] for s in states_list], dtype=np.float64)
    return states_list, actions_list, goals


def pack_paths(states_list, actions_list):
    """Concatenate per-agent paths into the flat arrays csdo_preprocess takes."""
    path_off = np.zeros(len(states_list) + 1, dtype=np.int32)
    for a, s in enumerate(states_list):
        path_off[a + 1] = path_off[a] + s.shape[0]
    states = np.ascontiguousarray(np.concatenate(states_list, axis=0), dtype=np.float64)
    actions = np.ascontiguousarray(np.concatenate(actions_list, axis=0), dtype=np.int32)
    return states, actions, path_off
