import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def orc():
    from tests.oracle_lib import Oracle
    return Oracle()


@pytest.fixture(scope="session")
def whisk_kat():
    with open(os.path.join(ROOT, "tests", "golden", "whisk_kat.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def oracle_vectors():
    with open(os.path.join(ROOT, "tests", "golden", "oracle_vectors.json")) as f:
        return json.load(f)
