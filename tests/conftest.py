import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # torch brings its own HIP runtime, which must be the one libcsdo_hip.so binds to: a process that loads the library first and
    # torch later ends up with two runtimes and torch finds no GPU (tests/test_gpu_dist.py uses torch.distributed on the device)
    try:
        import torch  # noqa: F401
    except Exception:
        pass


@pytest.fixture(scope="session")
def oracle():
    from tests import oracle_lib
    oracle_lib.lib()
    return oracle_lib


@pytest.fixture(scope="session")
def emu():
    from tests import emu_lib
    emu_lib.lib()
    return emu_lib


@pytest.fixture(scope="session")
def veh_parm():
    from csdotrajectoryplanning_amd import config
    return config.vehicle_from_config(), config.qp_parm_from_config()


@pytest.fixture(scope="session")
def world_map50(veh_parm):
    from csdotrajectoryplanning_amd import workloads
    veh, parm = veh_parm
    return workloads.build_world(workloads.MAP50_AGENTS25, seed=0, veh=veh, parm=parm)


@pytest.fixture(scope="session")
def world_map100(veh_parm):
    from csdotrajectoryplanning_amd import workloads
    veh, parm = veh_parm
    return workloads.map100_world(0, veh=veh, parm=parm, front="stand-in")   # the golden fixtures' world


@pytest.fixture(scope="session")
def gpu_handle():
    from csdotrajectoryplanning_amd.solver import DsqpHandle
    h = DsqpHandle(0)
    yield h
    h.close()
