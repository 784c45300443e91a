"""Full-chain parity, agent by agent, ON THE CPU (round 5).  The device program's lane-serial host build (tests/emu) returns the
bits of the HIP build over the whole SQP chain - one shared sin / cos / tan / atan2 (csrc/csdo_math.h); asserted on the GPU for
every agent of all five workloads by tests/test_gpu_sets.py::test_full_chain_hip_build_is_bit_identical_to_its_lane_serial_build -
so everything about "the product against the oracle" can be computed here, without GPU minutes:
  product (= HIP)      vs oracle        the parity statement (north_star: 1e-4 on trajectory states)
  product              vs oracle_xm     the oracle built with the product's trigonometry: what the FORMULATION alone does
  oracle_xm            vs oracle        the oracle's two trigonometries: what another LIBM alone does
  oracle_fma           vs oracle        the oracle built with fused multiply-adds: the reference algorithm's own sensitivity
and, round 6, against THE ARBITER - oracle_q: the oracle with OSQP's linear algebra (scaling, LDL', ADMM updates, residuals and the
tests on them) in IEEE binary128, everything the reference defines in double left in double, the solution rounded to double once per
QP (oracle/Makefile: libcsdo_oracle_q.so; oracle_qxm: the same with the product's trigonometry) - the exact-arithmetic iterate path
of the reference algorithm on the same double-precision QP data:
  product              vs oracle_q      how far the product is from the exact path
  oracle               vs oracle_q      how far a double-precision OSQP is from it: the yardstick for the line above
  oracle_fma, oracle_xm vs oracle_q     the same for the oracle's other builds
  product              vs oracle_qxm    the product against the exact path with ITS trigonometry: formulation + rounding of the solve alone
  product_refined      vs oracle_q      the product with csdo_qp_parm::solve_refinement = 1 (one refinement step on the KKT residual per solve)
  product_lagged       vs oracle_q      ... = 2 (the residual joins the next iteration's rhs: one solve per iteration)
The arbiter takes minutes per workload: whole-workload results are cached under oracle/_cache (scripts/arbiter_run.py).
Every outlier (|product - oracle| > 1e-4 or different counts) is then run alone with QpParm.max_iter = 1..10 on all of them,
and the cut at which it parts from the oracle is classified: a termination check that flips (ADMM counts differ at that cut),
a 0.1 m growth step of a safe box that flips (corridors differ by > 0.05 there), or plain amplification.
   python scripts/chain_parity.py --workload map100 --out profiles/r05_chain_map100.json --fixture tests/golden/chain_outliers_map100.json
   python scripts/chain_parity.py --workload map100 --hip        (on a GPU box: also runs HIP and insists on the emu build's bits)"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def with_max_iter(world, k):
    from csdotrajectoryplanning_amd.abi import QpParm
    from csdotrajectoryplanning_amd.problem import World
    p = QpParm.from_buffer_copy(bytes(world.parm))
    p.max_iter = float(k)
    return World(world.x0_bar, world.plane_off, world.planes, world.dimx, world.dimy, world.obstacles, world.veh, p)


def per_agent(got, ref):
    d = np.concatenate([np.abs(g.solutions - r.solutions).max(axis=(1, 2)) for g, r in zip(got, ref)])
    dc = np.concatenate([np.abs(g.corridors - r.corridors).max(axis=(1, 2)) for g, r in zip(got, ref)])
    same = np.concatenate([(g.sqp_iters == r.sqp_iters) & (g.admm_iters == r.admm_iters) &
                           (g.last_status == r.last_status) for g, r in zip(got, ref)])
    return d, dc, same


def stats(d, dc, same):
    return {"same_counts": int(same.sum()), "max_d": float(d.max()), "median_d": float(np.median(d)),
            "n_gt_1e-6": int((d > 1e-6).sum()), "n_gt_1e-4": int((d > 1e-4).sum()), "n_gt_1e-3": int((d > 1e-3).sum()),
            "n_gt_2e-2": int((d > 2e-2).sum()), "n_box_step_flipped": int((dc > 0.05).sum()),
            "n_gt_1e-4_without_box_flip": int(((d > 1e-4) & (dc < 0.05)).sum())}


def classify(row, pair):
    """The cut k (1-based) at which the pair parts (first d > 1e-5 that is more than 30 x the previous cut's), and what flipped there."""
    d, dc, same = row["by_k"][pair]["d"], row["by_k"][pair]["d_corridor"], row["by_k"][pair]["same_counts"]
    for k in range(10):
        prev = max(d[k - 1], 1e-12) if k else 1e-12
        if d[k] > 1e-5 and d[k] > 30.0 * prev:
            kind = "termination_check" if not same[k] else ("growth_step" if dc[k] > 0.05 else "amplification")
            # a flipped growth step shows in the corridors one cut EARLIER than in the states it then moves
            if kind == "amplification" and k and dc[k - 1] > 0.05:
                kind = "growth_step"
            return {"k": k + 1, "kind": kind, "d_before": prev, "d_at": d[k]}
    return {"k": None, "kind": "amplification", "d_before": None, "d_at": None}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", choices=("map100", "map50", "synth1024", "room50", "agents100"), default="map100")
    ap.add_argument("--threads", type=int, default=min(os.cpu_count() or 8, 32))
    ap.add_argument("--instances", type=int, default=None)
    ap.add_argument("--hip", action="store_true", help="also run the HIP library and insist on the lane-serial build's bits (needs a GPU)")
    ap.add_argument("--max-chains", type=int, default=60, help="outliers that are also run alone with max_iter = 1..10 (the worst ones)")
    ap.add_argument("--out", default=None)
    ap.add_argument("--no-cache", action="store_true", help="recompute the oracle builds' whole-workload results (oracle/_cache)")
    ap.add_argument("--fixture", default=None, help="write the outlier fixture (tests/golden/chain_outliers_<workload>.json)")
    args = ap.parse_args()
    import arbiter_run
    from csdotrajectoryplanning_amd import workloads
    from tests import emu_lib, oracle_lib
    worlds = [w for w, _ in workloads.build_jobs_parallel(workloads.workload_jobs(args.workload, args.instances), args.threads)]
    solvers = {"product": lambda ws: emu_lib.solve_batch(ws, 0, args.threads),
               "product_refined": lambda ws: emu_lib.solve_batch([arbiter_run.with_refinement(w) for w in ws], 0, args.threads),
               "product_lagged": lambda ws: emu_lib.solve_batch([arbiter_run.with_refinement(w, 2) for w in ws], 0, args.threads),
               "oracle": lambda ws: oracle_lib.solve_batch(ws, args.threads),
               "oracle_xm": lambda ws: oracle_lib.solve_batch_xm(ws, args.threads),
               "oracle_fma": lambda ws: oracle_lib.solve_batch_fma(ws, args.threads),
               "oracle_q": lambda ws: oracle_lib.solve_batch_variant(ws, "q", args.threads),
               "oracle_qxm": lambda ws: oracle_lib.solve_batch_variant(ws, "qxm", args.threads)}
    # whole-workload results: the arbiter's (and, when the set is complete, everyone's) from the cache of scripts/arbiter_run.py
    full = {}
    for k, f in solvers.items():
        path = arbiter_run.cache_path(args.workload, k)
        cached = not k.startswith("product") and args.instances is None and os.path.exists(path) and not args.no_cache
        if cached:
            full[k] = arbiter_run.unpack(np.load(path))
            assert [s_.solutions.shape[0] for s_ in full[k]] == [w.Na for w in worlds]
        else:
            full[k] = f(worlds)
            if args.instances is None and not k.startswith("product"):
                os.makedirs(arbiter_run.CACHE, exist_ok=True)
                np.savez_compressed(path, **arbiter_run.pack(full[k]))
    report = {"workload": args.workload, "agents": int(sum(w.Na for w in worlds)),
              "product": "lane-serial host build of the device program (tests/emu): the HIP build's bits", "pairs": {}}
    if args.hip:
        from csdotrajectoryplanning_amd.solver import DsqpHandle
        h = DsqpHandle(0)
        hip = h.solve_batch(worlds)
        h.close()
        n_bad = sum(int(not (np.array_equal(a.solutions, b.solutions) and np.array_equal(a.corridors, b.corridors)
                             and np.array_equal(a.admm_iters, b.admm_iters) and np.array_equal(a.sqp_iters, b.sqp_iters)
                             and np.array_equal(a.last_status, b.last_status))) for a, b in zip(hip, full["product"]))
        report["hip_equals_lane_serial_build"] = n_bad == 0
        assert n_bad == 0, "%d worlds differ between HIP and the lane-serial build" % n_bad
    pairs = [("product", "oracle"), ("product", "oracle_xm"), ("oracle_xm", "oracle"), ("oracle_fma", "oracle"),
             ("product", "oracle_q"), ("oracle", "oracle_q"), ("oracle_fma", "oracle_q"), ("oracle_xm", "oracle_q"),
             ("product", "oracle_qxm"), ("oracle_xm", "oracle_qxm"), ("product_refined", "oracle_q"), ("product_refined", "oracle"),
             ("product_lagged", "oracle_q")]
    per = {}
    for a, b in pairs:
        per[(a, b)] = per_agent(full[a], full[b])
        report["pairs"]["%s_vs_%s" % (a, b)] = stats(*per[(a, b)])
    d, dc, same = per[("product", "oracle")]
    first = np.cumsum([0] + [w.Na for w in worlds])

    def world_agent(g):
        wi = int(np.searchsorted(first, g, side="right") - 1)
        return wi, int(g - first[wi])
    out_idx = np.nonzero(~same | (d > 1e-4))[0]
    out_idx = out_idx[np.argsort(-d[out_idx], kind="stable")]
    outliers = [world_agent(g) for g in out_idx]
    # the worst outliers alone, chain cut after k = 1..10 QPs, on every solver
    chained = outliers[:args.max_chains]
    singles = [worlds[wi].subset(a, a + 1) for wi, a in chained]
    chain = {k: [] for k in solvers}
    for k in range(1, 11):
        ws = [with_max_iter(w, k) for w in singles]
        for name, f in solvers.items():
            chain[name].append(f(ws) if ws else [])
    rows = []
    for j, (wi, a) in enumerate(outliers):
        g = first[wi] + a
        row = {"world": wi, "agent": a, "Nt": int(worlds[wi].Nt), "d": float(d[g]), "d_corridor": float(dc[g]),
               "same_counts": bool(same[g]),
               "sqp": [int(full["product"][wi].sqp_iters[a]), int(full["oracle"][wi].sqp_iters[a])],
               "admm": [int(full["product"][wi].admm_iters[a]), int(full["oracle"][wi].admm_iters[a])],
               "status": [int(full["product"][wi].last_status[a]), int(full["oracle"][wi].last_status[a])],
               "d_oracle_fma": float(per[("oracle_fma", "oracle")][0][g]), "d_oracle_xm": float(per[("oracle_xm", "oracle")][0][g]),
               "d_product_oracle_xm": float(per[("product", "oracle_xm")][0][g]),
               "d_product_q": float(per[("product", "oracle_q")][0][g]), "d_oracle_q": float(per[("oracle", "oracle_q")][0][g]),
               "d_product_qxm": float(per[("product", "oracle_qxm")][0][g]),
               "d_product_refined_q": float(per[("product_refined", "oracle_q")][0][g])}
        if j < len(chained):
            row["by_k"] = {}
            for x, y in pairs:
                dk, ck, sk = [], [], []
                for k in range(10):
                    sx, sy = chain[x][k][j], chain[y][k][j]
                    dk.append(float(np.abs(sx.solutions - sy.solutions).max()))
                    ck.append(float(np.abs(sx.corridors - sy.corridors).max()))
                    sk.append(bool(sx.admm_iters[0] == sy.admm_iters[0] and sx.sqp_iters[0] == sy.sqp_iters[0]
                                   and sx.last_status[0] == sy.last_status[0]))
                row["by_k"]["%s_vs_%s" % (x, y)] = {"d": dk, "d_corridor": ck, "same_counts": sk}
            row["admm_by_k"] = {n: [int(chain[n][k][j].admm_iters[0]) for k in range(10)] for n in solvers}
            row["parts_at"] = classify(row, "product_vs_oracle")
        rows.append(row)
    report["outliers"] = rows
    kinds = [r["parts_at"]["kind"] for r in rows if "parts_at" in r]
    report["outliers_part_at"] = {k: kinds.count(k) for k in sorted(set(kinds))}
    # agents beyond 1e-4 per pair, and how the product's outliers overlap with the reference algorithm's own sensitive agents
    sets = {"%s_vs_%s" % (a, b): set(np.nonzero(~per[(a, b)][2] | (per[(a, b)][0] > 1e-4))[0].tolist()) for a, b in pairs}
    po = sets["product_vs_oracle"]
    report["outlier_counts"] = {k: len(v) for k, v in sets.items()}
    report["product_outliers_also_in"] = {k: len(po & v) for k, v in sets.items() if k != "product_vs_oracle"}
    report["product_outliers_in_none_of_the_oracles_own"] = [list(world_agent(g)) for g in sorted(po - sets["oracle_fma_vs_oracle"] - sets["oracle_xm_vs_oracle"])]
    # the reference algorithm's own rounding-sensitive agents: the oracle against itself built with fused multiply-adds
    df, dcf, samef = per[("oracle_fma", "oracle")]
    sens = [[*world_agent(g), float(df[g])] for g in np.nonzero(~samef | (df > 1e-6))[0]]
    report["oracle_sensitive"] = sens
    # ---- against the arbiter: who is closer to the exact-arithmetic path, agent by agent
    dpq, _, spq = per[("product", "oracle_q")]
    doq, _, soq = per[("oracle", "oracle_q")]
    q = lambda a, pp: float(np.quantile(a, pp))
    report["arbiter"] = {
        "what": "oracle_q = the oracle with OSQP's linear algebra in IEEE binary128 (double-precision QP data, x* rounded to double per QP)",
        "beyond_1e-4_or_other_counts": {"product": int((~spq | (dpq > 1e-4)).sum()), "oracle": int((~soq | (doq > 1e-4)).sum()),
                                        "product_refined": int((~per[("product_refined", "oracle_q")][2] | (per[("product_refined", "oracle_q")][0] > 1e-4)).sum()),
                                        "product_lagged": int((~per[("product_lagged", "oracle_q")][2] | (per[("product_lagged", "oracle_q")][0] > 1e-4)).sum()),
                                        "oracle_fma": report["pairs"]["oracle_fma_vs_oracle_q"]["n_gt_1e-4"],
                                        "oracle_xm": report["pairs"]["oracle_xm_vs_oracle_q"]["n_gt_1e-4"]},
        "quantiles_of_d": {"product": {"median": q(dpq, .5), "p90": q(dpq, .9), "p99": q(dpq, .99), "max": float(dpq.max())},
                           "product_refined": {"median": q(per[("product_refined", "oracle_q")][0], .5), "p90": q(per[("product_refined", "oracle_q")][0], .9),
                                               "p99": q(per[("product_refined", "oracle_q")][0], .99), "max": float(per[("product_refined", "oracle_q")][0].max())},
                           "product_lagged": {"median": q(per[("product_lagged", "oracle_q")][0], .5), "p90": q(per[("product_lagged", "oracle_q")][0], .9),
                                              "p99": q(per[("product_lagged", "oracle_q")][0], .99), "max": float(per[("product_lagged", "oracle_q")][0].max())},
                           "oracle": {"median": q(doq, .5), "p90": q(doq, .9), "p99": q(doq, .99), "max": float(doq.max())}},
        "agents_where_the_product_is_closer_to_q_than_the_oracle_is": int((dpq < doq).sum()),
        "beyond_1e-4_of_q_both": int(((dpq > 1e-4) & (doq > 1e-4)).sum()),
        "beyond_1e-4_of_q_product_only": [list(world_agent(g)) for g in np.nonzero((dpq > 1e-4) & ~(doq > 1e-4))[0]],
        "beyond_1e-4_of_q_oracle_only": [list(world_agent(g)) for g in np.nonzero(~(dpq > 1e-4) & (doq > 1e-4))[0]]}
    if args.fixture:
        fx = {"workload": args.workload, "agents": report["agents"], "arbiter": report["arbiter"],
              "_note": "scripts/chain_parity.py (CPU): `outliers` = agents whose full SQP chain - computed by the lane-serial host build of "
                       "the device program, whose bits the HIP build returns (asserted on the GPU) - differs from the oracle's by more than "
                       "1e-4 or in its counts; `oracle_sensitive` = agents on which the oracle differs from ITSELF built with "
                       "-ffp-contract=fast -mfma by more than 1e-6.  tests/test_chain_cpu.py recomputes the list; a GPU run only has to "
                       "return the lane-serial build's bits.",
              "outlier_counts": report["outlier_counts"],
              "outliers": [{k: r[k] for k in ("world", "agent", "Nt", "d", "d_corridor", "same_counts", "sqp", "admm", "status",
                                               "d_oracle_fma", "d_oracle_xm", "d_product_q", "d_oracle_q", "d_product_qxm", "d_product_refined_q") if k in r} | ({"parts_at": r["parts_at"]} if "parts_at" in r else {})
                           for r in report["outliers"]],
              "oracle_sensitive": sens}
        with open(args.fixture, "w") as f:
            f.write(json.dumps(fx, indent=1) + "\n")
    print(json.dumps({k: v for k, v in report.items() if k not in ("outliers", "oracle_sensitive")}, indent=1))
    print("outliers:", [(r["world"], r["agent"], "%.2e" % r["d"], r.get("parts_at", {}).get("kind"), r.get("parts_at", {}).get("k")) for r in report["outliers"]])
    if args.out:
        with open(args.out, "w") as f:
            f.write(json.dumps(report, indent=1) + "\n")


if __name__ == "__main__":
    main()
