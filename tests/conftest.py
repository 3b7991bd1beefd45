import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


@pytest.hookimpl(tryfirst=True)
def pytest_cmdline_main(config):
    """The CPU suite (`-m "not gpu"`) spreads over four workers when pytest-xdist is there and no -n was given: its tests are
    independent processes' worth of work (checker differentials, the stand-in stacks, gloo worlds) and take ~9 minutes in a row,
    ~3 side by side.  Never for the GPU suite (one device), never when a worker count was asked for; ZLNG_TESTS_SERIAL=1 opts out."""
    if (getattr(config.option, "markexpr", "") == "not gpu" and hasattr(config.option, "numprocesses") and not config.option.numprocesses
            and not os.environ.get("ZLNG_TESTS_SERIAL") and not getattr(config.option, "collectonly", False)):
        config.option.numprocesses = min(4, os.cpu_count() or 1)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle_py import Oracle
    return Oracle()


@pytest.fixture(scope="session")
def manifest():
    import json
    with open(os.path.join(ROOT, "tests", "golden", "manifest.json")) as f:
        return json.load(f)
