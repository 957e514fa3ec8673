import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

GOLDEN = ROOT / "tests" / "golden"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: test needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def vectors():
    import json
    return json.loads((GOLDEN / "vectors.json").read_text())


@pytest.fixture(scope="session")
def data_txt():
    return (GOLDEN / "data.txt").read_bytes()


@pytest.fixture(scope="session")
def data_lzma():
    return (GOLDEN / "data.eos.l3.lzma").read_bytes()
