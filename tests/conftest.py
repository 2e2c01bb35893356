import os, sys
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run via gpurun); everything else runs on CPU")


def pytest_sessionstart(session):
    # a fresh checkout has no built libraries (they are git-ignored): build them once, like `__graft_entry__.build()` does.
    # (Building is not a fallback: without the HIP library the product path fails loudly, and so would these tests.)
    if not os.path.exists(os.path.join(ROOT, "frizbee_amd", "libfrizbee_hip.so")):
        import __graft_entry__
        __graft_entry__.build()
