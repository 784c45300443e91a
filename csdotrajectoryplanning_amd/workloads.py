"""Named workloads: benchmark instance (data fixture) -> front-end stand-in -> bridge -> World.

Instance files under tests/golden/instances/ are data files of the reference's benchmark set (benchmark/map50by50,
benchmark/map100by100, benchmark/room); BASELINE.json's configs name them.  The coarse paths come from
synth.rollout_paths (GENERATOR_NAME) because the reference's PBS front end is out of scope.
"""
import os

from . import config, instance, synth
from .solver import interpolate_and_planes

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
INSTANCE_DIR = os.path.join(_ROOT, "tests", "golden", "instances")

MAP50_AGENTS25 = "map_50by50_obst25_agents25_ex0.yaml"
MAP100_AGENTS50 = "map_100by100_obst50_agents50_ex{}.yaml"


def build_world(instance_file, seed=0, veh=None, parm=None, preprocess=None):
    """Returns (World, info).  `preprocess` defaults to the shipped bridge (csdo_preprocess); tests pass the oracle's."""
    veh = veh or config.vehicle_from_config()
    parm = parm or config.qp_parm_from_config()
    path = instance_file if os.path.isabs(instance_file) else os.path.join(INSTANCE_DIR, instance_file)
    inst = instance.load_instance(path, obs_radius=veh.obs_radius)
    S, A, G = synth.rollout_paths(inst, veh, seed)
    st, ac, po = synth.pack_paths(S, A)
    if preprocess is None:
        world, pairs, legal = interpolate_and_planes(st, ac, po, G, veh, parm, inst.dimx, inst.dimy, inst.obstacles)
    else:
        world, pairs, legal = preprocess(st, ac, po, G, veh, parm, inst)
    info = dict(instance=os.path.basename(path), generator=synth.GENERATOR_NAME, seed=seed, Na=world.Na, Nt=world.Nt,
                n_pairs=int(len(pairs)), n_planes=int(world.plane_off[-1]), initial_inter_legal=int(legal),
                paths=(st, ac, po, G))
    return world, info


MAP100_SET_SIZE = 60   # benchmark/map100by100/agents50/obstacle holds ex0 .. ex59


def map100_world(k, veh=None, parm=None, seed_offset=0):
    """Instance ex{k} of the map100by100/agents50/obstacle set; the front-end stand-in is seeded with k + seed_offset."""
    return build_world(MAP100_AGENTS50.format(k), seed=k + seed_offset, veh=veh, parm=parm)
