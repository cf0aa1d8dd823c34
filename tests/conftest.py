import os
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

# test hooks of the product library (fault injection, per-call re-reading of the tuning environment
# variables): must be in the environment before the library initialises, see csrc/qs_xfer.h
os.environ.setdefault("QS_HIP_TEST_HOOKS", "1")

import jpegqs_pkg  # noqa: E402

GOLDEN = Path(__file__).resolve().parent / "golden"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(autouse=True)
def _gpu_tests_never_take_the_cpu_back_end(request, monkeypatch):
    """libjpegqs.so runs its CPU back end when no HIP device is visible (csrc/qs_cpu.c).  A test marked `gpu` must
    never pass that way: for those, JPEGQS_BACKEND=hip forbids the route (the call fails instead) -- also in the
    CLIs and demo programs they start, which inherit the environment."""
    if request.node.get_closest_marker("gpu") is not None:
        monkeypatch.setenv("JPEGQS_BACKEND", "hip")
        monkeypatch.delenv("QS_HIP_FORCE_CPU", raising=False)
    yield


@pytest.fixture(scope="session")
def pkg():
    return jpegqs_pkg.load()


@pytest.fixture(scope="session")
def synth(pkg):
    return pkg.synth


@pytest.fixture(scope="session")
def oracle():
    """our plain-C restatement (oracle/libqs_oracle.so); built on demand with gcc"""
    from oracle.oracle import Oracle
    return Oracle()


@pytest.fixture(scope="session")
def reference():
    """the compiled unmodified reference (oracle/_ref); only where it was built"""
    from oracle import oracle as om
    if not om.have_ref("none"):
        if not om.build_ref():
            pytest.skip("oracle/_ref not built and /root/reference not mounted")
    return om.Reference("none")


@pytest.fixture(scope="session")
def hip(pkg):
    """the product library; never skipped -- a missing .so is a failure"""
    return pkg.HipQS()


@pytest.fixture(scope="session")
def gpu(hip):
    if hip.device_count() <= 0:
        pytest.fail("no HIP device visible although the test is marked gpu")
    return hip


@pytest.fixture(scope="session")
def big_plane(synth):
    """BASELINE configs[2] input: 8192x8192 luma, JPEG quality 50 (about 9 s to build, shared)"""
    return synth.synth_gray(8192, 8192, 50)
