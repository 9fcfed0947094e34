import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def oracle():
    from oracle.binding import Oracle
    return Oracle()


@pytest.fixture(scope="session")
def golden_cases():
    import gzip
    import json
    p = os.path.join(ROOT, "tests", "golden", "extz2_kat.json.gz")
    with gzip.open(p, "rb") as f:
        return json.loads(f.read().decode())["cases"]
