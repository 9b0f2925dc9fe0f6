import os
import sys

import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


# torch ships its own HIP runtime; when libhmsg.so (linked against /opt/rocm) is loaded first, torch later finds
# "No HIP GPUs".  Loading torch first makes both use one runtime, as bench.py does.
try:
    import torch  # noqa: F401
except Exception:  # pragma: no cover
    pass
