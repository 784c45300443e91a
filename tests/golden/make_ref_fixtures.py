"""Generates tests/golden/ref_collision_verdicts.npz and tests/golden/ref_result_header.json by IMPORTING the reference's
own Python tools (build container only: /root/reference does not exist on the GPU box, so only the vectors travel):

  scripts/collision_detection.py:20-96   collision_rect_and_rect / collision_circle_and_rect (with collision_geometry's
                                         Rectangle(pos='rear_axle_center') exactly as scripts/visualize.py:40-52 builds it)
  scripts/analysis_result.py:53-101      read_solution_status, the positional parser of the result YAML's header

  scripts/visualize.py:256-281           Animation.getState (the interpolation between states the per-frame collision prints
                                         of :219-247 look at) with check_collision / check_obs_collision of :40-52
  scripts/visualize_corridor.py:92-103   translate2np, the parser of the corridor dump (sqp/utils.cc:62-89)

The fixtures hold INPUTS (random poses / obstacle discs; a result file written by csdotrajectoryplanning_amd.results) and
the reference's OUTPUTS (collision verdicts; parsed header values).  tests/test_results.py checks results.validate and
results.write_solutions against them.   python tests/golden/make_ref_fixtures.py
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference/scripts")
os.environ.setdefault("MPLBACKEND", "Agg")

if not hasattr(np, "Inf"):                     # the reference predates numpy 2.0, which dropped the alias it uses
    np.Inf = np.inf
import collision_detection as ref_cd          # noqa: E402  (the reference's module)
from collision_geometry import Circle, Rectangle  # noqa: E402
import analysis_result as ref_ar              # noqa: E402

LF, LB, W = 2.0, 1.0, 2.0                     # scripts/visualize.py:23-25


def rect(p):
    return Rectangle(p[0], p[1], p[2], length=LF + LB, width=W, pos="rear_axle_center", lb=LB)


def main():
    rng = np.random.default_rng(20260203)
    n = 6000
    # rectangle / rectangle: second pose within a few metres of the first so that about half of the pairs overlap
    a = np.column_stack([rng.uniform(10, 90, n), rng.uniform(10, 90, n), rng.uniform(-2 * np.pi, 2 * np.pi, n)])
    off_r, off_th = rng.uniform(0.0, 5.0, n), rng.uniform(0, 2 * np.pi, n)
    b = np.column_stack([a[:, 0] + off_r * np.cos(off_th), a[:, 1] + off_r * np.sin(off_th),
                         rng.uniform(-2 * np.pi, 2 * np.pi, n)])
    rr = np.array([bool(ref_cd.collision_rect_and_rect(rect(p), rect(q))) for p, q in zip(a, b)])
    # circle / rectangle
    c = np.column_stack([rng.uniform(10, 90, n), rng.uniform(10, 90, n), rng.uniform(-2 * np.pi, 2 * np.pi, n)])
    o_r, o_th = rng.uniform(0.0, 4.5, n), rng.uniform(0, 2 * np.pi, n)
    obs = np.column_stack([c[:, 0] + o_r * np.cos(o_th), c[:, 1] + o_r * np.sin(o_th), rng.uniform(0.3, 1.5, n)])
    cr = np.array([bool(ref_cd.collision_circle_and_rect(Circle(o[0], o[1], o[2]), rect(p))) for p, o in zip(c, obs)])
    np.savez_compressed(os.path.join(HERE, "ref_collision_verdicts.npz"), rect_a=a, rect_b=b, rect_rect=rr,
                        rect_c=c, circle=obs, circle_rect=cr, LF=LF, LB=LB, W=W)
    print("rect/rect: %d of %d collide; circle/rect: %d of %d collide" % (rr.sum(), n, cr.sum(), n))

    # result YAML header: written by this repo's writer, parsed by the reference's positional parser
    from csdotrajectoryplanning_amd import results
    cases = []
    sol = np.zeros((2, 4, 6))
    sol[..., 0] = np.arange(4)[None, :] * 1.5 + np.array([[3.0], [20.0]])
    sol[..., 1] = 7.25
    sol[..., 2] = 0.1
    sol[..., 3] = 0.05
    sol[:, :-1, 4] = 0.8
    sol[:, :-1, 5] = -0.01
    for k, stats in enumerate([
        dict(runtime=1.234, runtime_search=0.5, runtime_preprocess=0.01, runtime_optimization=0.7,
             runtime_decentralized_optimization=0.05, search_status=2, solver_status=1),
        dict(runtime=20.0, runtime_search=19.0, runtime_preprocess=0.25, runtime_optimization=3.5,
             runtime_decentralized_optimization=0.75, search_status=1, solver_status=-2),
        dict(search_status=0, solver_status=-3),          # everything else missing: the reference prints -1
    ]):
        path = os.path.join(HERE, "_tmp_result_%d.yaml" % k)
        results.write_solutions(path, sol, stats)
        st = ref_ar.Status()
        st.clear()
        ref_ar.read_solution_status(path, st, with_solver_status=True)
        with open(path) as f:
            text = f.read()
        os.remove(path)
        cases.append(dict(stats=stats, yaml_text=text,
                          parsed=dict(cost=st.cost[-1], makespan=st.makespan[-1], flowtime=st.flowtime[-1],
                                      runtime=st.runtime[-1], runtime_search=st.runtime_search[-1],
                                      runtime_dqp=st.runtime_dqp[-1], search_success=int(st.search_success[-1]),
                                      success=bool(st.success[-1]))))
    with open(os.path.join(HERE, "ref_result_header.json"), "w") as f:
        json.dump(dict(generator="tests/golden/make_ref_fixtures.py", parser="scripts/analysis_result.py:53-101",
                       solutions=sol.tolist(), cases=cases), f, indent=1)
    print("header cases:", [c["parsed"] for c in cases])


def substep_and_corridor_fixtures():
    """ref_substep_frames.npz: random trajectories (inputs) and, for 1, 3 and 10 frames per move, the reference's interpolated
    poses and per-frame collision verdicts (outputs).  ref_corridor_parse.json: a corridor dump written by results.write_corridors
    (text) and what the reference's parser makes of it."""
    import yaml
    import matplotlib
    matplotlib.use = lambda *a, **k: None          # the script asks for Qt5Agg at import; nothing is drawn here
    import visualize as ref_vis                     # noqa: E402  (the reference's module)
    import visualize_corridor as ref_vc             # noqa: E402
    from csdotrajectoryplanning_amd import config, results
    ref_vis.LF, ref_vis.LB, ref_vis.carWidth = LF, LB, W
    rng = np.random.default_rng(20260303)
    Na, Nt = 6, 9
    # vehicles that pass close to each other and to two discs, with yaw crossing +-pi so that getState's unwrapping runs
    sol = np.zeros((Na, Nt, 3))
    for a in range(Na):
        p = np.array([rng.uniform(18, 32), rng.uniform(18, 32)])
        yaw = rng.uniform(-np.pi, np.pi)
        for t in range(Nt):
            sol[a, t] = [p[0], p[1], ((yaw + np.pi) % (2 * np.pi)) - np.pi]
            yaw += rng.uniform(-0.5, 0.5)
            p = p + rng.uniform(0.0, 2.2) * np.array([np.cos(yaw), np.sin(yaw)])
    sol[0, :, 2] = np.linspace(2.6, 3.9, Nt)
    sol[0, :, 2] = ((sol[0, :, 2] + np.pi) % (2 * np.pi)) - np.pi        # jumps from +pi to -pi between two states
    obstacles = np.array([[25.0, 25.0, 0.8], [21.5, 28.0, 1.2], [29.0, 22.0, 0.8]])
    sched = [[{"t": t, "x": float(sol[a, t, 0]), "y": float(sol[a, t, 1]), "yaw": float(sol[a, t, 2])} for t in range(Nt)]
             for a in range(Na)]
    out = dict(solutions=sol, obstacles=obstacles, LF=LF, LB=LB, W=W)
    for S in (1, 3, 10):
        nf = (Nt - 1) * S + 1
        frames = np.zeros((Na, nf, 3))
        veh_hits, obs_hits = [], []
        for f in range(nf):
            pos = [ref_vis.Animation.getState(None, f / S, sched[a]) for a in range(Na)]
            for a in range(Na):
                frames[a, f] = pos[a]
            for i in range(Na):
                for j in range(i + 1, Na):
                    if ref_vis.check_collision(pos[i], pos[j]):
                        veh_hits.append((f, i, j))
            for a in range(Na):
                for k, o in enumerate(obstacles):
                    if ref_vis.check_obs_collision(pos[a], list(o)):
                        obs_hits.append((f, a, k))
        out["frames_%d" % S] = frames
        out["vehicle_hits_%d" % S] = np.array(veh_hits, np.int64).reshape(-1, 3)
        out["obstacle_hits_%d" % S] = np.array(obs_hits, np.int64).reshape(-1, 3)
        print("frames per move %d: %d frames, %d vehicle and %d obstacle collision triples" % (S, nf, len(veh_hits), len(obs_hits)))
    np.savez_compressed(os.path.join(HERE, "ref_substep_frames.npz"), **out)

    veh = config.vehicle_from_config()
    x0 = np.zeros((2, 3, 6))
    x0[..., 0] = [[10.0, 11.5, 13.0], [40.25, 40.25, 41.0]]
    x0[..., 1] = [[7.0, 7.125, 7.5], [33.0, 34.0, 35.5]]
    x0[..., 2] = [[0.0, 0.3, 0.6], [-1.57, -1.2, 3.1]]
    cor = rng.uniform(0.0, 50.0, (2, 3, 8))
    cor[0, 0] = [1.25, 20.1000000001, 1.25, 9.99999999, 1.25, 123456.789, 0.0625, 48.75]   # (PyYAML reads 1e-05 as a string)
    path = os.path.join(HERE, "_tmp_corridors.yaml")
    results.write_corridors(path, cor, x0, veh)
    with open(path) as f:
        text = f.read()
    with open(path) as f:
        parsed = ref_vc.translate2np(yaml.load(f, Loader=yaml.FullLoader))
    os.remove(path)
    with open(os.path.join(HERE, "ref_corridor_parse.json"), "w") as f:
        json.dump(dict(generator="tests/golden/make_ref_fixtures.py", parser="scripts/visualize_corridor.py:92-103",
                       x0_bar=x0.tolist(), corridors=cor.tolist(), yaml_text=text, parsed=parsed.tolist()), f, indent=1)
    print("corridor dump parsed by the reference:", parsed.shape)


if __name__ == "__main__":
    main()
    substep_and_corridor_fixtures()
