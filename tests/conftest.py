import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


def pytest_terminal_summary(terminalreporter, exitstatus, config):
    """Seconds per test file (setup + call + teardown), slowest first: the driver runs `pytest -m gpu` under a time limit,
    so what a file costs is printed with every run (review of round 5: keep the GPU suite below 400 s)."""
    per_file = {}
    for reports in terminalreporter.stats.values():
        for rep in reports:
            dur, node = getattr(rep, "duration", None), getattr(rep, "nodeid", "")
            if dur is not None and node:
                per_file[node.split("::")[0]] = per_file.get(node.split("::")[0], 0.0) + dur
    if per_file:
        terminalreporter.write_sep("-", "seconds per test file")
        for name, sec in sorted(per_file.items(), key=lambda kv: -kv[1]):
            terminalreporter.write_line(f"{sec:8.1f}s  {name}")
        terminalreporter.write_line(f"{sum(per_file.values()):8.1f}s  total (in-test time; process start-up and collection come on top)")
