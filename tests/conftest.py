import os
import sys

import pytest

try:  # before anything loads libfractal_hip.so: the library then binds to the HIP runtime torch carries
    import torch  # noqa: F401  (in a process that uses both, torch must come first)
except ImportError:
    pass

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (HERE, ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    import oracle_lib

    oracle_lib.lib()
    oracle_lib.set_log2_mode(oracle_lib.LOG2_LIBM)
    return oracle_lib
