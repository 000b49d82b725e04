import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


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
