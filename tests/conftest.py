import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def kat():
    from tests.kat import load
    return load()


DEV_ONLY_SWITCHES = ("RTC_AMD_JIT_SOURCE", "RTC_AMD_JIT_FLAGS", "RTC_AMD_JIT_PRINT", "RTC_AMD_TREE_WAVES", "RTC_AMD_REG_LEVELS",
                     "RTC_AMD_BLOCKS_Y", "RTC_AMD_BLOCK_S", "RTC_AMD_BLOCK_S_TOP", "RTC_AMD_BLOCK_ORDER", "RTC_AMD_FILL_WGS", "RTC_AMD_TILE_FILL_WGS",
                     "RTC_AMD_CLUSTER_MIN_RUN", "RTC_AMD_CLUSTER_LEAF", "RTC_AMD_CLUSTER_GMAX", "RTC_AMD_CLUSTER_STATS", "RTC_AMD_TRI_NAIVE")


@pytest.fixture
def dev_lib():
    """The development build of the library (librtc_amd_dev.so, -DRTC_DEV_SWITCHES) beside the one that ships: the few
    tests that pin tuning constants, substitute compiler flags or break a guarantee on purpose (RTC_AMD_TRI_NAIVE) run
    against it -- the shipped library does not even contain those switches' names.  Everything the package does inside
    the test goes to that library; the shipped one is back afterwards."""
    from ray_tracer_challenge_amd import _lib as L
    if L.lib().rtc_dev_switches() == 1:  # the whole suite is running against the development build (RTC_AMD_LIB)
        yield L.lib()
        return
    with L.use_library(L.DEV_LIB_PATH) as lib:
        assert lib.rtc_dev_switches() == 1
        yield lib
