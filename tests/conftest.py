import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


@pytest.hookimpl(tryfirst=True)
def pytest_cmdline_main(config):
    """The CPU suite (`-m "not gpu"`) is dominated by single-threaded emulator runs of whole models: spread it over a few worker
    processes with pytest-xdist when it is installed and no -n was given (BBDM_TESTS_SERIAL=1 keeps one process).  The GPU suite
    always runs in one process (one device, timing-sensitive tests)."""
    opt = config.option
    # (never inside an xdist worker: a worker that switched xdist on again would spawn workers of its own, recursively)
    if hasattr(config, "workerinput") or os.environ.get("PYTEST_XDIST_WORKER"):
        return None
    if getattr(opt, "numprocesses", None) or os.environ.get("BBDM_TESTS_SERIAL") or "not gpu" not in (opt.markexpr or ""):
        return None
    try:
        import xdist  # noqa: F401
    except ImportError:
        return None
    if not hasattr(opt, "numprocesses"):
        return None
    sys.path.insert(0, os.path.join(ROOT, "tools", "hipemu"))
    try:                                        # build the emulated kernel library ONCE, before the workers race for it
        import build as hipemu_build
        hipemu_build.build()
    except Exception as e:                      # pragma: no cover -- the tests that need it will report the failure
        print(f"conftest: hipemu pre-build failed: {e!r}", file=sys.stderr)
    opt.numprocesses = max(1, min(6, (os.cpu_count() or 2) - 2))
    opt.dist = "load"
    opt.tx = ["popen"] * opt.numprocesses
    return None


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "reference: needs /root/reference mounted (build container only)")


def pytest_collection_modifyitems(config, items):
    have_ref = os.path.isdir("/root/reference/model")
    skip_ref = pytest.mark.skip(reason="/root/reference not mounted")
    for item in items:
        if "reference" in item.keywords and not have_ref:
            item.add_marker(skip_ref)
