import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    with open(os.path.join(GOLDEN_DIR, "golden.json")) as fh:
        return json.load(fh)


@pytest.fixture(scope="session")
def oracle():
    from oracle import ffq_oracle
    ffq_oracle.lib()
    return ffq_oracle


@pytest.fixture(scope="session")
def pkg():
    import fastqandfurious_amd  # noqa: F401
    return fastqandfurious_amd


@pytest.fixture(scope="session")
def gpu_ctx(pkg):
    from fastqandfurious_amd import build, hip
    build.build()                                    # (rebuilds iff the in-tree library's build id is not the sources')
    # (FFQ_USE_PROBE_BUILD=1: a debugging session runs the suite on the instrumented build of the same sources)
    assert hip.build_id() == (build.source_id(probe=True) + "+probes" if os.environ.get("FFQ_USE_PROBE_BUILD") == "1" else build.source_id()), \
        "libffq_hip.so was not built from the sources in this tree"
    ctx = hip.default_context(0)
    return ctx


@pytest.fixture(autouse=True)
def _forget_input_history(request):
    """GPU tests assert which kernels ran (res.path): start each from a context without
    memory of the previous test's input."""
    if request.node.get_closest_marker("gpu") is not None:
        request.getfixturevalue("gpu_ctx").forget()
        # the tests choose fbufsize to exercise the refill / carry logic: reads are not coalesced
        # unless a test asks for it (tests/test_iter_decode.py does)
        from fastqandfurious_amd import _fastqandfurious as C
        default = type(C.entrypos).coalesce_bytes
        C.entrypos.coalesce_bytes = 0
        yield
        C.entrypos.coalesce_bytes = default
        return
    yield


def golden_file(name):
    with open(os.path.join(GOLDEN_DIR, "data", name), "rb") as fh:
        return fh.read()


def end_matches(run, end_state, end_offset):
    """Does an oracle/GPU end state agree with a captured reference run?

    run: {"rows", "error", "hang"} from make_golden.py.  end_offset is the
    buffer coordinate (sentinel included) where the failing search started;
    the reference prints globaloffset + offset = end_offset - 1."""
    err = run["error"]
    if run["hang"]:
        return end_state == 4
    if err is None:
        return end_state == 0
    if err == "Incomplete final quality string at byte":
        return end_state == 2
    if err.startswith("Incomplete entry at byte"):
        return end_state == 3 and int(err.rsplit(" ", 1)[1]) == end_offset - 1
    if err.startswith("Entry is invalid at byte"):
        return end_state == 4 and int(err.rsplit(" ", 1)[1]) == end_offset - 1
    return False


def rows_of(table):
    return [[int(x) for x in r] for r in np.asarray(table).reshape(-1, 6)]
