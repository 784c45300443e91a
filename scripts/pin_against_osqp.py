#!/usr/bin/env python
"""Pins this repository's oracle against the REAL OSQP - for an environment that has it (this project's containers do not; the oracle is
"parity unpinned" until someone runs this):

    pip install osqp==0.6.3            # the version the reference builds against (README.md:16, CMakeLists.txt:9,82)
    python scripts/pin_against_osqp.py [--dir tests/golden/osqp_pin] [--tol 1e-6]

For every tests/golden/osqp_pin/qp_XX.npz (scripts/make_osqp_pin_kit.py: assembled agent QPs of the golden worlds as
sqp/dsqp_solver.cc:457-502 hands them to OSQP, with the oracle's x*, y*, iteration count, status and rho at every check) it runs OSQP with
the settings of sqp/dsqp_solver.cc:476-487 - defaults, max_iter = 400, verbose off, polish off, warm start x only - and
adaptive_rho_interval = 25 (upstream's default 0 picks 25 k iterations from wall-clock timing, which no restatement can follow; the
oracle and the GPU backend pin it and expose it as a parameter), then compares: status, iteration count, x (max abs), y, final rho.
Exit code 0: every QP agrees (iteration counts and statuses identical, |x - x_oracle| <= --tol); 1: some differ (a table says which);
2: OSQP is not importable.

Two caveats of the Python wrapper that the C program tests/cpp/pin_osqp.c does not have:
  * osqp's Python interface clips |bounds| beyond 1e30 to +-1e30 (OSQP_INFTY) on setup, and so does the C setup (scaling.c treats
    bounds beyond OSQP_INFTY * MIN_SCALING as infinite); the reference passes a true -inf lower bound on the inter-vehicle rows
    (sqp/dsqp_solver.cc:1121-1123), which makes OSQP's primal-infeasibility certificate sum NaN for agents that have such rows - the
    certificate then never fires.  The oracle reproduces that quirk.  With the wrapper's clipping the certificate CAN fire: a QP that
    is reported primal infeasible here and max-iter / solved-inaccurate by the oracle is that difference, not a restatement error
    (the kit's QPs are all feasible, so it should not occur on them);
  * wrapper versions from 0.6.2.post0 on copy the matrices; any of them with the 0.6.x C core is fine.

--self-check: no OSQP needed - runs the ORACLE on the dumps and insists on the stored results (what tests/test_osqp_pin_kit.py does)."""
import argparse
import glob
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def load(path):
    import scipy.sparse as sp
    z = np.load(path)
    n, m = int(z["n"]), int(z["m"])
    P = sp.csc_matrix((z["P_data"], z["P_indices"], z["P_indptr"]), shape=(n, n))
    A = sp.csc_matrix((z["A_data"], z["A_indices"], z["A_indptr"]), shape=(m, n))
    return z, P, A


def run_osqp(z, P, A):
    import osqp
    prob = osqp.OSQP()
    prob.setup(P, z["q"], A, z["l"], z["u"], max_iter=int(z["max_iter"]), verbose=False, polish=False, warm_start=True,
               adaptive_rho=True, adaptive_rho_interval=int(z["adaptive_rho_interval"]), check_termination=25,
               eps_abs=1e-3, eps_rel=1e-3, eps_prim_inf=1e-4, eps_dual_inf=1e-4, rho=0.1, sigma=1e-6, alpha=1.6, scaling=10,
               scaled_termination=False)
    prob.warm_start(x=z["x_warm"])
    r = prob.solve()
    return r.x, r.y, int(r.info.iter), int(r.info.status_val), float(r.info.rho_estimate), int(r.info.rho_updates)


def run_oracle(z, P, A):
    from tests import oracle_lib
    x, y, info, hist = oracle_lib.osqp_hist(P, z["q"], A, z["l"], z["u"], z["x_warm"], max_iter=int(z["max_iter"]),
                                             adaptive_rho_interval=int(z["adaptive_rho_interval"]))
    return x, y, info["iter"], info["status"], float(hist[-1, 0]) if len(hist) else float("nan"), info["rho_updates"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dir", default=os.path.join(ROOT, "tests", "golden", "osqp_pin"))
    ap.add_argument("--tol", type=float, default=1e-6)
    ap.add_argument("--self-check", action="store_true")
    ap.add_argument("--export", default=None, help="write qp_XX.bin (flat little-endian binaries for tests/cpp/pin_osqp.c) into this directory and exit")
    args = ap.parse_args()
    if args.export:
        os.makedirs(args.export, exist_ok=True)
        for f in sorted(glob.glob(os.path.join(args.dir, "qp_*.npz"))):
            z = np.load(f)
            with open(os.path.join(args.export, os.path.basename(f)[:-4] + ".bin"), "wb") as o:
                np.array([z["n"], z["m"], len(z["P_data"]), len(z["A_data"]), z["max_iter"], z["adaptive_rho_interval"], z["oracle_iter"],
                          z["oracle_status"]], dtype="<i4").tofile(o)
                for k, dt in (("P_indptr", "<i4"), ("P_indices", "<i4"), ("P_data", "<f8"), ("A_indptr", "<i4"), ("A_indices", "<i4"),
                              ("A_data", "<f8"), ("q", "<f8"), ("l", "<f8"), ("u", "<f8"), ("x_warm", "<f8"), ("oracle_x", "<f8")):
                    np.ascontiguousarray(z[k], dtype=dt).tofile(o)
        print("wrote", len(glob.glob(os.path.join(args.export, "qp_*.bin"))), "files to", args.export)
        return 0
    if not args.self_check:
        try:
            import osqp
            print("osqp", osqp.__version__)
        except Exception as e:   # noqa: BLE001
            print("OSQP is not importable here (%s): pip install osqp==0.6.3, or build tests/cpp/pin_osqp.c against osqp.h" % e)
            return 2
    files = sorted(glob.glob(os.path.join(args.dir, "qp_*.npz")))
    bad = 0
    print("%-8s %-34s %5s %5s  %6s %6s  %10s %10s  %s" % ("qp", "what", "iter", "ref", "status", "ref", "max|dx|", "max|dy|", "verdict"))
    for f in files:
        z, P, A = load(f)
        x, y, it, st, rho, nup = (run_oracle if args.self_check else run_osqp)(z, P, A)
        dx = float(np.abs(np.asarray(x) - z["oracle_x"]).max()) if x is not None and np.all(np.isfinite(x)) else float("inf")
        dy = float(np.abs(np.asarray(y) - z["oracle_y"]).max()) if y is not None and np.all(np.isfinite(y)) else float("inf")
        ok = it == int(z["oracle_iter"]) and st == int(z["oracle_status"]) and dx <= args.tol
        bad += not ok
        print("%-8s %-34s %5d %5d  %6d %6d  %10.2e %10.2e  %s" % (os.path.basename(f)[:-4], str(z["what"])[:34], it, int(z["oracle_iter"]), st,
                                                                 int(z["oracle_status"]), dx, dy, "ok" if ok else "DIFFERENT"))
    print("%d of %d QPs agree with the oracle's stored results%s" % (len(files) - bad, len(files),
                                                                     "" if args.self_check else " - the oracle is pinned on them" if not bad else ""))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
