import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _build_everything():
    import __graft_entry__ as g

    g.build()


@pytest.fixture(scope="session")
def rr():
    import rust_renderer_amd

    return rust_renderer_amd


@pytest.fixture(scope="session")
def oa():
    import oracle_api

    return oracle_api
