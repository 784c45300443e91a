"""Generates tests/golden/ref_collision_verdicts.npz and tests/golden/ref_result_header.json by IMPORTING the reference's
own Python tools (build container only: /root/reference does not exist on the GPU box, so only the vectors travel):

  scripts/collision_detection.py:20-96   collision_rect_and_rect / collision_circle_and_rect (with collision_geometry's
                                         Rectangle(pos='rear_axle_center') exactly as scripts/visualize.py:40-52 builds it)
  scripts/analysis_result.py:53-101      read_solution_status, the positional parser of the result YAML's header

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


if __name__ == "__main__":
    main()
