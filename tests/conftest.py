import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
# The HIP runtime aborts SILENTLY on an asynchronous queue error at its default log level; level 1 (errors only; nothing is printed on a
# healthy run) makes it say which one.  Set before the first HIP call of the session; a value from the environment wins.
os.environ.setdefault("AMD_LOG_LEVEL", "1")
os.environ.setdefault("LIBC_FATAL_STDERR_", "1")      # glibc's own fatal messages (heap consistency checks) to stderr, not to the controlling terminal


def _install_abort_trace(config):
    """native call stack on SIGABRT (tests/cpp/abort_trace.c, built by `make`): installed after pytest's faulthandler, which it chains to.
    It writes where faulthandler writes -- the real stderr that plugin duplicated before the capture took fd 2 -- and appends the tail of the
    CAPTURED stderr of the dying test (the last words of glibc or the HIP runtime go there and are lost with the process otherwise)."""
    import ctypes
    so = os.path.join(ROOT, "tests", "cpp", "libabort_trace.so")
    if os.path.exists(so):
        try:
            fd = 2
            try:
                from _pytest.faulthandler import fault_handler_stderr_fd_key
                fd = config.stash.get(fault_handler_stderr_fd_key, 2)
            except Exception:       # noqa: BLE001 (another pytest: plain stderr)
                pass
            ctypes.CDLL(so).abort_trace_install_fd(int(fd))
        except (OSError, AttributeError):
            pass

# Qi60 / Pi60: the reference's 61-bit NTT-friendly test primes (ring/test_params.go:15-32), data only.
QI60 = [0x1fffffffffe00001, 0x1fffffffffc80001, 0x1fffffffffb40001, 0x1fffffffff500001,
        0x1fffffffff380001, 0x1fffffffff000001, 0x1ffffffffef00001, 0x1ffffffffee80001,
        0x1ffffffffeb40001, 0x1ffffffffe780001, 0x1ffffffffe600001, 0x1ffffffffe4c0001,
        0x1ffffffffdf40001, 0x1ffffffffdac0001, 0x1ffffffffda40001, 0x1ffffffffc680001,
        0x1ffffffffc000001, 0x1ffffffffb880001, 0x1ffffffffb7c0001, 0x1ffffffffb300001,
        0x1ffffffffb1c0001, 0x1ffffffffadc0001, 0x1ffffffffa400001, 0x1ffffffffa140001,
        0x1ffffffff9d80001, 0x1ffffffff9140001, 0x1ffffffff8ac0001, 0x1ffffffff8a80001,
        0x1ffffffff81c0001, 0x1ffffffff7800001, 0x1ffffffff7680001, 0x1ffffffff7080001]
PI60 = [0x1ffffffff6c80001, 0x1ffffffff6140001, 0x1ffffffff5f40001, 0x1ffffffff5700001,
        0x1ffffffff4bc0001, 0x1ffffffff4380001, 0x1ffffffff3240001, 0x1ffffffff2dc0001,
        0x1ffffffff1a40001, 0x1ffffffff11c0001, 0x1ffffffff0fc0001, 0x1ffffffff0d80001,
        0x1ffffffff0c80001, 0x1ffffffff08c0001, 0x1fffffffefd00001, 0x1fffffffef9c0001,
        0x1fffffffef600001, 0x1fffffffeef40001, 0x1fffffffeed40001, 0x1fffffffeed00001,
        0x1fffffffeebc0001, 0x1fffffffed540001, 0x1fffffffed440001, 0x1fffffffed2c0001,
        0x1fffffffed200001, 0x1fffffffec940001, 0x1fffffffec6c0001, 0x1fffffffebe80001,
        0x1fffffffebac0001, 0x1fffffffeba40001, 0x1fffffffeb4c0001, 0x1fffffffeb280001]


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionstart(session):
    _install_abort_trace(session.config)


def uniform_mod(rng, q, shape):
    """i.i.d. uniform residues in [0, q) (rejection-free: 128-bit product method is overkill for tests)"""
    hi = rng.integers(0, 1 << 62, size=shape, dtype=np.uint64)
    return (hi % np.uint64(q)).astype(np.uint64)


@pytest.fixture(scope="session")
def oracle():
    import oracle as orc
    orc.lib()
    return orc


@pytest.fixture(scope="session")
def rh():
    import matrix_fhe_lattigo_amd as m
    m.lib()
    return m
