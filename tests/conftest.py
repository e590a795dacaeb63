import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


class Golden:
    def __init__(self):
        with open(os.path.join(GOLDEN_DIR, "golden.json")) as f:
            self.meta = json.load(f)
        self.arrays = np.load(os.path.join(GOLDEN_DIR, "golden.npz"))
        self.cases = {c["name"]: c for c in self.meta["cases"]}

    def vector_cases(self):
        return [c for c in self.meta["cases"] if c["kind"] == "vector"]

    def cli_cases(self):
        return [c for c in self.meta["cases"] if c["kind"] == "cli"]

    def inputs(self, case):
        return {c: self.arrays["%s/in/%s" % (case["name"], c)] for c, _ in case["chroms"]}

    def outputs(self, case):
        return {c: self.arrays["%s/out/%s" % (case["name"], c)] for c, _ in case["chroms"]}


_golden = None


def golden():
    global _golden
    if _golden is None:
        _golden = Golden()
    return _golden


@pytest.fixture(scope="session")
def gold():
    return golden()


def bits_equal(a, b):
    a = np.ascontiguousarray(a, np.float64)
    b = np.ascontiguousarray(b, np.float64)
    return a.shape == b.shape and a.tobytes() == b.tobytes()


def first_diff(a, b):
    a = np.asarray(a).view(np.uint64)
    b = np.asarray(b).view(np.uint64)
    idx = np.flatnonzero(a != b)
    return None if idx.size == 0 else int(idx[0])


@pytest.fixture(autouse=True)
def _forget_cli_runs():
    """tests/cli_compare.py remembers what every run() of a test did; start each test with an empty list"""
    import cli_compare
    del cli_compare.RUNS[:]
    yield


@pytest.hookimpl(hookwrapper=True)
def pytest_runtest_makereport(item, call):
    """a failing CLI test leaves argv, environment, library id, stdin, stdout and stderr of its runs under
    gpurun_out/artifacts/ (the directory gpurun brings back from the GPU box)"""
    outcome = yield
    report = outcome.get_result()
    if report.when == "call" and report.failed:
        try:
            import cli_compare
            where = cli_compare.write_artifacts(item.nodeid)
            if where:
                report.sections.append(("artifacts", "kept under " + where))
        except Exception as e:                      # noqa: BLE001
            report.sections.append(("artifacts", "could not be written: %r" % (e,)))
