"""Host side of the multi-GPU handle (csdo_dsqp_create_multi): the sharding rule behind the C ABI is the one the N-process path
uses (sharding.shard_bounds_weighted), item for item.  No GPU needed: csdo_dsqp_shard_bounds is host code."""
import ctypes as C

import numpy as np

from csdotrajectoryplanning_amd import abi, sharding
from csdotrajectoryplanning_amd._lib import lib


def _cuts(w, n_blocks):
    w = np.ascontiguousarray(w, dtype=np.float64)
    cuts = np.zeros(n_blocks + 1, np.int32)
    assert lib().csdo_dsqp_shard_bounds(abi.as_double_p(w) if len(w) else None, len(w), n_blocks, abi.as_int32_p(cuts)) == 0
    return cuts


def test_c_abi_rule_equals_the_python_rule():
    rng = np.random.default_rng(3)
    for trial in range(400):
        n = int(rng.integers(0, 400))
        nb = int(rng.integers(1, 12))
        kind = trial % 4
        w = (rng.uniform(0.1, 1.0, n) if kind == 0 else rng.lognormal(0, 1.5, n) if kind == 1 else
             np.where(rng.uniform(size=n) < 0.3, 0.0, rng.uniform(0, 5, n)) if kind == 2 else np.full(n, 2.5))
        want = sharding.shard_bounds_weighted(w, nb)
        got = _cuts(w, nb)
        assert [(int(got[r]), int(got[r + 1])) for r in range(nb)] == [(int(a), int(b)) for a, b in want], (trial, n, nb)


def test_blocks_are_contiguous_cover_everything_and_balance_the_work():
    rng = np.random.default_rng(4)
    w = rng.lognormal(0, 1.0, 3000)
    c = _cuts(w, 8)
    assert c[0] == 0 and c[-1] == 3000 and np.all(np.diff(c) > 0)
    loads = np.array([w[c[r]:c[r + 1]].sum() for r in range(8)])
    assert loads.max() / loads.mean() < 1.01
    # fewer items than blocks: every item is a block of its own, the other blocks are empty; all-zero weights: equal counts
    few = _cuts(np.ones(3), 5)
    assert few[0] == 0 and few[-1] == 3 and sorted(np.diff(few).tolist()) == [0, 0, 1, 1, 1]
    assert [(int(few[r]), int(few[r + 1])) for r in range(5)] == sharding.shard_bounds_weighted(np.ones(3), 5)
    assert _cuts(np.zeros(10), 4).tolist() == [0, 3, 6, 8, 10]
    assert _cuts(np.zeros(0), 3).tolist() == [0, 0, 0, 0]


def test_bad_arguments():
    cuts = np.zeros(4, np.int32)
    assert lib().csdo_dsqp_shard_bounds(None, 5, 3, abi.as_int32_p(cuts)) == abi.CSDO_EINVAL
    assert lib().csdo_dsqp_shard_bounds(abi.as_double_p(np.ones(5)), 5, 0, abi.as_int32_p(cuts)) == abi.CSDO_EINVAL
    assert lib().csdo_dsqp_multi_count(None) == abi.CSDO_EINVAL
    h = C.c_void_p()
    assert lib().csdo_dsqp_create_multi(C.byref(h), None, 2) == abi.CSDO_EINVAL
    devs = np.zeros(2, np.int32)
    assert lib().csdo_dsqp_create_multi(C.byref(h), abi.as_int32_p(devs), 0) == abi.CSDO_EINVAL
