import json
import os
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, ".."))
for p in (ROOT, HERE):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(HERE, "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_hashes():
    return json.load(open(os.path.join(GOLDEN, "golden_hashes.json")))


@pytest.fixture(scope="session")
def golden_params():
    return json.load(open(os.path.join(GOLDEN, "golden_params.json")))


@pytest.fixture(scope="session")
def oracle_lib():
    from oracle import oracle
    oracle.lib()
    return oracle
