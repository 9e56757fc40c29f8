import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as O
    O.lib()
    return O


@pytest.fixture(scope="session")
def capi():
    from locus_amd import capi as capi_mod
    capi_mod.lib()  # fails loudly if liblocus_hip.so is missing
    return capi_mod


@pytest.fixture(scope="session")
def ctx(capi):
    return capi.Context(0)  # raises LocusHipError(LH_EDEVICE) without a GPU: no CPU fallback


GOLDEN = os.path.join(ROOT, "tests", "golden")


@pytest.fixture(scope="session")
def garage(oracle):
    q = oracle.read_pcd_xyzi(os.path.join(GOLDEN, "query_82_garage.pcd"))
    r = oracle.read_pcd_xyzi(os.path.join(GOLDEN, "reference_82_garage.pcd"))
    return q, r
