"""The boundary is a C ABI: the header must compile as plain C, and a plain-C client must link and run."""
import shutil
import subprocess
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent


def _build(tmp_path):
    exe = tmp_path / "cabi_smoke"
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", str(ROOT / "include"),
                           str(ROOT / "tests" / "cabi_smoke.c"), "-o", str(exe), "-L", str(ROOT / "loco_hd_amd"), "-lloco_hd_hip",
                           "-lm", f"-Wl,-rpath,{ROOT / 'loco_hd_amd'}"])
    return exe


@pytest.mark.skipif(shutil.which("gcc") is None, reason="gcc not available")
def test_header_is_plain_c_and_client_links(tmp_path):
    exe = _build(tmp_path)
    assert exe.exists()


@pytest.mark.gpu
def test_c_client_runs(tmp_path):
    exe = _build(tmp_path)
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    assert "cabi smoke ok" in out.stdout
