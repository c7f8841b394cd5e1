import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")
TRACKS = ("rounded_rectangle", "handling_track", "modena_2019", "berlin_2018")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
    return {k: z[k] for k in z.files}


@pytest.fixture(scope="session")
def golden():
    return {name: load_golden(name) for name in TRACKS}


@pytest.fixture(scope="session")
def emu_lib():
    """TEST-ONLY: the unchanged csrc/*.hip compiled against the SIMT interpreter in tests/emu (no GPU needed)."""
    import subprocess
    path = os.path.join(ROOT, "tests", "emu", "libmcq_emu.so")
    src = [os.path.join(ROOT, "global_racetrajectory_optimization_amd", "csrc", f)
           for f in ("mcq_kernels.hip", "mcq_kernels.h", "mcq_api.hip", "mcq_kkt.inc", "mcq_tri.inc", "mcq_gi.inc")]
    src.append(os.path.join(ROOT, "tests", "emu", "include", "hip", "hip_runtime.h"))
    def stale():
        return not os.path.exists(path) or any(os.path.getmtime(s) > os.path.getmtime(path) for s in src)

    if stale():
        # several pytest-xdist workers may get here at once: one builds (build_emu.sh publishes the library with a rename), the others wait
        import fcntl
        with open(path + ".lock", "w") as lock:
            fcntl.flock(lock, fcntl.LOCK_EX)
            if stale():
                subprocess.run([os.path.join(ROOT, "tests", "emu", "build_emu.sh")], check=True)
    return path


@pytest.fixture(scope="session")
def gpu_engine():
    from global_racetrajectory_optimization_amd import engine
    eng = engine.Engine(0)
    yield eng
    eng.close()


@pytest.fixture(scope="session")
def rccl_stub():
    """TEST-ONLY: the shared-memory stand-in for librccl.so.1 (tests/stub/rccl_stub.cpp, synchronous build) that lets the engine's collective
    path run with several ranks on the SIMT-interpreted library; tests hand its path to the engine through $MCQ_RCCL_LIB."""
    import fcntl
    import subprocess
    d = os.path.join(ROOT, "tests", "stub")
    path = os.path.join(d, "librccl_stub_sync.so")
    src = os.path.join(d, "rccl_stub.cpp")
    if not os.path.exists(path) or os.path.getmtime(src) > os.path.getmtime(path):
        with open(path + ".lock", "w") as lock:
            fcntl.flock(lock, fcntl.LOCK_EX)
            if not os.path.exists(path) or os.path.getmtime(src) > os.path.getmtime(path):
                subprocess.run([os.path.join(d, "build_stub.sh"), "sync-only"], check=True)
    return path
