import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# Tests take the tile of the committed tuning table when a shape is in it and the library's heuristic otherwise: timing two
# dozen candidates for every new shape (what the engine does on first eager use outside tests) multiplies the suite's run time.
os.environ.setdefault("GEO4D_AUTOTUNE", "0")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: long-running CPU test")


def cpu_threads(cap=32):
    """Thread count for CPU oracles: more OpenMP threads than ~32 make these small fp32 convs / einsums SLOWER on the many-core
    hosts of the GPU boxes (measured in round 1: oversubscription, not speed-up)."""
    import torch
    n = max(1, min(os.cpu_count() or 1, cap))
    torch.set_num_threads(n)
    return n


@pytest.fixture(scope="session")
def dev():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


@pytest.fixture(scope="session")
def _full_engine_session(dev):
    import bench
    return bench.build("bf16", dev)


@pytest.fixture
def full_engine(_full_engine_session):
    """ONE instance of the shipped 1.44 B-parameter configuration (+ both VAEs) shared by every full-size test: building it costs
    tens of seconds of host time. Every test gets it in the bf16 mode and it is put back into bf16 afterwards, so the MODE never
    depends on collection order; the WEIGHTS do (re-filling 1.44 B parameters per test costs more than the tests): a test that
    compares against a fixture loads the name-keyed seeded weights itself, the others only use engine-vs-engine properties."""
    import bench
    model, pvae = _full_engine_session
    bench.set_mode(model, pvae, "bf16")
    yield model, pvae
    bench.set_mode(model, pvae, "bf16")
