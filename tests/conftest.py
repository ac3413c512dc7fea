import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
# an abort underneath a test (glibc heap check, an assertion in a runtime library) leaves its C-level call stack here (util.cpp: abort_trace)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
os.environ.setdefault("MCF_ABORT_TRACE_FILE", os.path.join(ROOT, "gpurun_out", "abort_trace.txt"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # build what is missing (both builds are quick and need no GPU); prebuilt .so files travel with the snapshot
    if not os.path.exists(os.path.join(ROOT, "oracle", "libns_oracle.so")):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")], stdout=subprocess.DEVNULL)
    if not os.path.exists(os.path.join(ROOT, "mincostflow_amd", "libmcf_hip.so")):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "mincostflow_amd", "csrc")], stdout=subprocess.DEVNULL)


@pytest.fixture(scope="session")
def have_gpu():
    import mincostflow_amd as M
    return M.device_count() > 0
