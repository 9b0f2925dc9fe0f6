import os
import sys

import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


@pytest.hookimpl(tryfirst=True)
def pytest_cmdline_main(config):
    """The CPU suite (-m "not gpu") spends its time in the kernel simulator, one full pipeline per test: spread it over a
    few worker processes when pytest-xdist is there and the caller did not choose (-n ...).  GPU runs stay serial (one
    device).  HMSG_TEST_WORKERS=0 switches this off."""
    try:
        import xdist  # noqa: F401
    except Exception:
        return None
    want = int(os.environ.get("HMSG_TEST_WORKERS", "6"))
    if (want > 1 and getattr(config.option, "numprocesses", None) is None and not getattr(config.option, "collectonly", False)
            and (getattr(config.option, "markexpr", "") or "").strip() == "not gpu"
            and len(getattr(config.option, "file_or_dir", None) or []) <= 1 and not os.environ.get("PYTEST_XDIST_WORKER")):
        args = config.option.file_or_dir or []
        if not args or os.path.isdir(args[0]):          # (whole-suite runs only: a single file is quicker in-process)
            config.option.numprocesses = min(want, os.cpu_count() or 1)
            config.option.dist = "load"
    return None


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


# torch ships its own HIP runtime; when libhmsg.so (linked against /opt/rocm) is loaded first, torch later finds
# "No HIP GPUs".  Loading torch first makes both use one runtime, as bench.py does.
try:
    import torch  # noqa: F401
except Exception:  # pragma: no cover
    pass
