"""Named workloads: benchmark instance (data fixture) -> front-end stand-in -> bridge -> World.

Instance files under tests/golden/instances/ are data files of the reference's benchmark set (benchmark/map50by50,
benchmark/map100by100, benchmark/room); BASELINE.json's configs name them.  Coarse paths (SURVEY 8d "Initial guesses"):
  front="auto"      the named workloads' default: the paths this repository's own front end (front_end.plan, PBS over hybrid
                    A*) produced for the instance, stored under tests/golden/front_end_paths/ by make_front_end_paths.py;
                    the instances that search does not solve (unsolved.json: 1 of map100's 60, 3 of map50's 60, 1 of room50's 12, all 12 of agents100) fall back
                    to the stand-in so that every instance of a set takes part
  front="pbs"       run the search now; raises FrontEndFailed where it finds nothing
  front="stand-in"  synth.rollout_paths (GENERATOR_NAME): seeded primitive roll-outs, never fails, not collision-free;
                    the committed golden fixtures and the small test worlds are built from it
"""
import os

from . import config, instance, synth
from .solver import interpolate_and_planes

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
INSTANCE_DIR = os.path.join(_ROOT, "tests", "golden", "instances")

MAP50_AGENTS25 = "map_50by50_obst25_agents25_ex0.yaml"
MAP50_AGENTS25_SET = "map_50by50_obst25_agents25_ex{}.yaml"
MAP100_AGENTS50 = "map_100by100_obst50_agents50_ex{}.yaml"


class FrontEndFailed(RuntimeError):
    """The priority-based search found no set of paths within its limits (search_status 0 of the reference)."""


PATHS_DIR = os.path.join(_ROOT, "tests", "golden", "front_end_paths")


def stored_paths(instance_name):
    """(states, actions, path_off) the front end produced for a benchmark instance, or None (not solved / not stored)."""
    f = os.path.join(PATHS_DIR, os.path.basename(instance_name).replace(".yaml", ".npz"))
    if not os.path.exists(f):
        return None
    import numpy as np
    with np.load(f) as z:
        return z["states"], z["actions"], z["path_off"]


def build_world(instance_file, seed=0, veh=None, parm=None, preprocess=None, front="stand-in", front_parm=None):
    """Returns (World, info).  `preprocess` defaults to the shipped bridge (csdo_preprocess); tests pass the oracle's.
    front: see the module docstring; `seed` only matters where the stand-in is used."""
    veh = veh or config.vehicle_from_config()
    parm = parm or config.qp_parm_from_config()
    path = instance_file if os.path.isabs(instance_file) else os.path.join(INSTANCE_DIR, instance_file)
    inst = instance.load_instance(path, obs_radius=veh.obs_radius)
    search = None
    if front == "auto":
        stored = stored_paths(path)
        front = "stored" if stored is not None else "stand-in"
    if front == "stored":
        st, ac, po = stored
        G = inst.goals
        generator = "front_end.plan (stored)"
    elif front == "pbs":
        from . import front_end
        cp = front_end.plan(inst.starts, inst.goals, inst.dimx, inst.dimy, inst.obstacles, veh, front_parm)
        if cp is None:
            raise FrontEndFailed(os.path.basename(path))
        st, ac, po, G = cp.states, cp.actions, cp.path_off, inst.goals
        generator = "front_end.plan"
        search = dict(seconds=cp.seconds, hl_expanded=cp.hl_expanded, hl_generated=cp.hl_generated,
                      ll_expanded=cp.ll_expanded)
    elif front == "stand-in":
        S, A, G = synth.rollout_paths(inst, veh, seed)
        st, ac, po = synth.pack_paths(S, A)
        generator = synth.GENERATOR_NAME
    else:
        raise ValueError("front must be 'auto', 'stand-in' or 'pbs'")
    if preprocess is None:
        world, pairs, legal = interpolate_and_planes(st, ac, po, G, veh, parm, inst.dimx, inst.dimy, inst.obstacles)
    else:
        world, pairs, legal = preprocess(st, ac, po, G, veh, parm, inst)
    info = dict(instance=os.path.basename(path), generator=generator, search=search,
                seed=seed, Na=world.Na, Nt=world.Nt,
                n_pairs=int(len(pairs)), n_planes=int(world.plane_off[-1]), initial_inter_legal=int(legal),
                paths=(st, ac, po, G))
    return world, info


MAP100_SET_SIZE = 60   # benchmark/map100by100/agents50/obstacle holds ex0 .. ex59


def map100_world(k, veh=None, parm=None, seed_offset=0, front="auto"):
    """Instance ex{k} of the map100by100/agents50/obstacle set; where the stand-in is used it is seeded with k + seed_offset."""
    return build_world(MAP100_AGENTS50.format(k), seed=k + seed_offset, veh=veh, parm=parm, front=front)


MAP50_SET_SIZE = 60    # benchmark/map50by50/agents25/obstacle holds ex0 .. ex59


def map50_world(k, veh=None, parm=None, seed_offset=0, front="auto"):
    """Instance ex{k} of the map50by50/agents25/obstacle set (BASELINE.json configs[1]); seed k + seed_offset.
    (ex0 with seed 0 is the world the single-instance fixtures and tests use.)"""
    return build_world(MAP50_AGENTS25_SET.format(k), seed=k + seed_offset, veh=veh, parm=parm, front=front)


SYNTH1024_AGENTS = 1024


def synthetic_1024(veh=None, parm=None, seed_offset=0, n_agents=SYNTH1024_AGENTS):
    """BASELINE.json configs[4], as SURVEY 8(d) config 5 defines it: 1024 vehicles do not fit one 100x100 map, so the
    stress instance is ceil(1024/50) = 21 independent worlds - instances ex0..ex20 of the map100by100/agents50/obstacle
    set in one batch (inter-vehicle planes only inside a world, every world keeps its obstacles), truncated to 1024
    agents: the last world keeps its first 24 agents with all their planes.  Returns (worlds, infos)."""
    worlds, infos, left = [], [], int(n_agents)
    k = 0
    while left > 0:
        w, info = map100_world(k % MAP100_SET_SIZE, veh=veh, parm=parm, seed_offset=seed_offset + 1000 * (k // MAP100_SET_SIZE))
        if w.Na > left:
            w = w.subset(0, left)
            info = dict(info, Na=w.Na, n_planes=int(w.plane_off[-1]), truncated_to=left)
        worlds.append(w)
        infos.append(info)
        left -= w.Na
        k += 1
    return worlds, infos


WORKLOADS = ("map100", "map50", "synth1024", "room50", "agents100")
# The regimes the two BASELINE sets do not reach (SURVEY 8 sizes): `room50` = benchmark/room/agents50 (238 obstacles of radius
# 0.5 forming walls: five times the box work per point, obstacle staging in LDS), `agents100` = benchmark/map100by100/agents100/
# obstacle (twice the vehicles, so about twice the separating planes per agent: the plane state of many agents no longer fits LDS
# beside the rest; 2-tuple obstacles, radius from config.yaml's obsRadius).  Twelve instances each (ex0 .. ex11).
ROOM_AGENTS50 = "room_agents50_ex{}.yaml"            # (benchmark/room/agents50/map_100by100_agents50_ex{}.yaml, renamed: the
MAP100_AGENTS100 = "map_100by100_obst50_agents100_ex{}.yaml"   # name collides with the empty map100 family)
EXTRA_SET_SIZE = 12


def workload_jobs(name, n_instances=None, seed_offset=0, front="auto"):
    """(builder, k, seed_offset, front) jobs of a named bench workload, one per world, for a process pool."""
    if name == "map100":
        n = MAP100_SET_SIZE if n_instances is None else max(1, min(int(n_instances), MAP100_SET_SIZE))
        return [("map100", k, seed_offset, front) for k in range(n)]
    if name == "map50":
        n = MAP50_SET_SIZE if n_instances is None else max(1, min(int(n_instances), MAP50_SET_SIZE))
        return [("map50", k, seed_offset, front) for k in range(n)]
    if name == "synth1024":
        return [("synth1024", k, seed_offset, front) for k in range(21)]
    if name in ("room50", "agents100"):
        n = EXTRA_SET_SIZE if n_instances is None else max(1, min(int(n_instances), EXTRA_SET_SIZE))
        return [(name, k, seed_offset, front) for k in range(n)]
    raise ValueError("unknown workload %r (one of %s)" % (name, ", ".join(WORKLOADS)))


def build_job(job):
    """Build one world of a workload (see workload_jobs); returns (World, info)."""
    kind, k, seed_offset, front = job
    if kind == "map50":
        return map50_world(k, seed_offset=seed_offset, front=front)
    if kind == "room50":
        return build_world(ROOM_AGENTS50.format(k), seed=k + seed_offset, front=front)
    if kind == "agents100":
        return build_world(MAP100_AGENTS100.format(k), seed=k + seed_offset, front=front)
    w, info = map100_world(k, seed_offset=seed_offset, front=front)
    if kind == "synth1024" and k == 20:
        left = SYNTH1024_AGENTS - 20 * 50
        w = w.subset(0, left)
        info = dict(info, Na=w.Na, n_planes=int(w.plane_off[-1]), truncated_to=left)
    return w, info


def job_agents(job):
    """Number of agents of a workload job's world, known without building it (sharding plans need it up front)."""
    kind, k = job[0], job[1]
    if kind == "map50":
        return 25
    if kind == "agents100":
        return 100
    if kind == "synth1024" and k == 20:
        return SYNTH1024_AGENTS - 20 * 50
    return 50


def build_jobs_parallel(jobs, procs=8, start_method="spawn"):
    """[(World, info)] for a list of workload jobs, built by a process pool.  `spawn` by default: safe to call from a
    process that has already initialised the GPU (the workers never touch it)."""
    procs = max(1, min(int(procs), len(jobs)))
    if procs == 1:
        return [build_job(j) for j in jobs]
    from multiprocessing import get_context
    with get_context(start_method).Pool(procs) as pool:
        return pool.map(build_job, jobs)
