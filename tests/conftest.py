import os
import sys

import pytest
# torch first: its wheel bundles a HIP runtime of its own; when libseqrush_amd.so (linked against /opt/rocm) is the first
# to initialise HIP in the process, a later torch.cuda init finds "No HIP GPUs".  Loaded in this order both share torch's.
import torch  # noqa: F401

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _build_native():
    import __graft_entry__ as ge
    ge.build()


def _have_gpu():
    try:
        from seqrush_amd import _lib
        return _lib.load().sr_device_count() > 0
    except Exception:
        return False


@pytest.fixture(scope="session")
def gpu():
    if not _have_gpu():
        pytest.fail("-m gpu tests need a HIP device: seqrush_amd has no CPU fallback")
    return True


def usable_cpus():
    """CPUs this process may really use: affinity mask capped by the cgroup quota"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(float(q) / float(per))))
    except Exception:
        pass
    return max(1, n)


def canon_gfa(text):
    """H/S/P lines in order + L lines as a sorted multiset (the reference writes L lines
    in randomized HashSet order, src/bidirected_ops.rs:11, 898-907)."""
    lines = [l for l in text.strip().split("\n") if l]
    return [l for l in lines if l[0] != "L"], sorted(l for l in lines if l[0] == "L")
