import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
# the oracle uses OpenMP; keep small test cases from oversubscribing big hosts
os.environ.setdefault("OMP_NUM_THREADS", str(min(8, os.cpu_count() or 1)))
os.environ.setdefault("OMP_WAIT_POLICY", "PASSIVE")

import __graft_entry__ as entry  # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden")
P = 0xFFFFFFFF00000001
GEN = 14293326489335486720      # plonky2 GoldilocksField::MULTIPLICATIVE_GROUP_GENERATOR (coset shift)
ROOT32 = 7277203076849721926     # POWER_OF_TWO_GENERATOR = GEN^((p-1)/2^32)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def built():
    """Build every native piece once per session (seconds when already built)."""
    entry.build()
    return True


@pytest.fixture(scope="session")
def pkg(built):
    return entry.load_package()


@pytest.fixture(scope="session")
def orc(built):
    return entry.load_oracle()
