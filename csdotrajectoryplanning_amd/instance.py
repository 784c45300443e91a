"""Benchmark instance files (hybrid_a_star/Instance.cc:25-63 of the reference).

Format: `agents: [{start: [x,y,yaw], name, goal: [x,y,yaw]}]`, `map: {dimensions: [X,Y], obstacles: [[x,y(,r)]..]|null}`.
Two-element obstacles take the default radius (Constants::obsRadius).
"""
from dataclasses import dataclass

import numpy as np
import yaml


@dataclass
class Instance:
    dimx: float
    dimy: float
    obstacles: np.ndarray  # [n_obs, 3] x, y, r in file order
    starts: np.ndarray     # [Na, 3]
    goals: np.ndarray      # [Na, 3]
    name: str = ""

    @property
    def num_agents(self):
        return int(self.starts.shape[0])


def load_instance(path, obs_radius=float(np.float32(0.8))) -> Instance:
    with open(path) as f:
        doc = yaml.safe_load(f)
    dims = doc["map"]["dimensions"]
    obs = []
    for node in (doc["map"].get("obstacles") or []):
        r = float(node[2]) if len(node) > 2 else obs_radius
        obs.append([float(node[0]), float(node[1]), r])
    starts = [[float(v) for v in a["start"]] for a in doc["agents"]]
    goals = [[float(v) for v in a["goal"]] for a in doc["agents"]]
    return Instance(float(int(dims[0])), float(int(dims[1])), np.array(obs, dtype=np.float64).reshape(-1, 3),
                    np.array(starts, dtype=np.float64), np.array(goals, dtype=np.float64), name=str(path))
