"""The OSQP pin kit (tests/golden/osqp_pin, scripts/make_osqp_pin_kit.py, scripts/pin_against_osqp.py, tests/cpp/pin_osqp.c): twenty
assembled agent QPs with the oracle's results, for whoever has OSQP 0.6.3 - no environment of this project does, so the oracle stays
"parity unpinned" here.  What CAN be checked here: the dumps are what the oracle solves (same results again, to the bit), they span
the regimes the kit promises, and the independent numpy ADMM of tests/admm_numpy.py (SuperLU on the KKT matrix, written from the
published algorithm) walks the same iterate path on them."""
import glob
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KIT = os.path.join(ROOT, "tests", "golden", "osqp_pin")
sys.path.insert(0, os.path.join(ROOT, "scripts"))


def _files():
    return sorted(glob.glob(os.path.join(KIT, "qp_*.npz")))


def test_the_kit_spans_what_it_promises():
    zs = [np.load(f) for f in _files()]
    assert len(zs) == 20
    its = [int(z["oracle_iter"]) for z in zs]
    assert min(its) == 25 and max(its) == 400 and sum(1 for z in zs if int(z["oracle_status"]) == 2) >= 3       # cap-bound QPs
    assert sum(1 for z in zs if int(z["planes"]) == 0) >= 3 and max(int(z["planes"]) for z in zs) > 300           # with / without rows
    assert {int(z["Nt"]) for z in zs} == {91, 169} and max(int(z["sqp_iteration"]) for z in zs) == 10
    for z in zs:
        n, m = int(z["n"]), int(z["m"])
        assert n == 6 * int(z["Nt"]) - 2 and m == 13 * int(z["Nt"]) + 4 * int(z["planes"])
        assert np.isneginf(z["l"]).sum() == 4 * int(z["planes"]) and np.all(z["q"] == 0)                          # true -inf, dsqp_solver.cc:1121
        assert len(z["oracle_checks"]) == int(z["oracle_iter"]) // 25 and np.all(z["oracle_checks"][:, 0] > 0)


def test_the_oracle_returns_the_stored_results_to_the_bit():
    import pin_against_osqp as kit
    for f in _files():
        z, P, A = kit.load(f)
        x, y, it, st, rho, nup = kit.run_oracle(z, P, A)
        assert it == int(z["oracle_iter"]) and st == int(z["oracle_status"]) and nup == int(z["oracle_rho_updates"])
        assert np.array_equal(x, z["oracle_x"]) and np.array_equal(y, z["oracle_y"]), f


@pytest.mark.parametrize("k", [0, 3, 7, 14, 17])
def test_an_independent_admm_walks_the_same_path_on_the_dumps(k):
    import pin_against_osqp as kit
    from tests import admm_numpy
    z, P, A = kit.load(_files()[k])
    r = admm_numpy.solve(P, z["q"], A, z["l"], z["u"], z["x_warm"], max_iter=int(z["max_iter"]), interval=25)
    assert (r["iter"], r["status"]) == (int(z["oracle_iter"]), int(z["oracle_status"]))
    np.testing.assert_allclose(r["rho_hist"], z["oracle_checks"][:, 0], rtol=1e-6)
    assert np.abs(r["x"] - z["oracle_x"]).max() < 1e-6


def test_the_script_says_so_when_osqp_is_missing():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "pin_against_osqp.py")], capture_output=True, text=True)
    try:
        import osqp  # noqa: F401
        assert r.returncode in (0, 1)
    except Exception:
        assert r.returncode == 2 and "pip install osqp==0.6.3" in r.stdout
