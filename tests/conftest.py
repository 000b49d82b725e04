import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    _use_sanitizer_builds()


def _use_sanitizer_builds():
    """LCHD_ASAN=1 (README "Sanitizer run"; CPU tests only): the AddressSanitizer + UBSan build of the CPython helper takes the place of
    loco_hd_amd._fastpack BEFORE the package is imported (oracle/oracle.py picks its own sanitizer build from the same variable).  The
    interpreter itself is not instrumented, so the run needs LD_PRELOAD=$(gcc -print-file-name=libasan.so)."""
    import importlib.util
    import os
    import subprocess

    if not os.environ.get("LCHD_ASAN"):
        return
    csrc = ROOT / "loco_hd_amd" / "csrc"
    subprocess.check_call(["make", "-C", str(csrc), "asan"], stdout=subprocess.DEVNULL)
    so = next((csrc / "asan").glob("_fastpack*.so"))
    spec = importlib.util.spec_from_file_location("loco_hd_amd._fastpack", so)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    sys.modules["loco_hd_amd._fastpack"] = mod
    import loco_hd_amd.api as api

    assert api._fastpack is mod, "the sanitizer build of _fastpack did not replace the regular one"


def pytest_sessionstart(session):
    """Build the native pieces if this is a fresh checkout (hipcc cross-compiles gfx950 without a GPU; ~40 s)."""
    import subprocess

    lib = ROOT / "loco_hd_amd" / "libloco_hd_hip.so"
    csrc = ROOT / "loco_hd_amd" / "csrc"
    srcs = list(csrc.glob("*.hip")) + list(csrc.glob("*.h")) + list(csrc.glob("*.c")) + [ROOT / "include" / "loco_hd_hip.h"]
    ext = list((ROOT / "loco_hd_amd").glob("_fastpack*.so"))
    newest = max(p.stat().st_mtime for p in srcs)
    if not lib.exists() or not ext or min(lib.stat().st_mtime, ext[0].stat().st_mtime) < newest:
        subprocess.check_call(["make", "-C", str(ROOT / "loco_hd_amd" / "csrc"), "-j8"], stdout=subprocess.DEVNULL)


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure; never imported by the product package)."""
    from oracle import oracle as orc

    orc.build()
    return orc
