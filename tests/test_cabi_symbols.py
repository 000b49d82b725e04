"""The C-ABI library must load on a machine without a GPU and export every symbol include/loco_hd_hip.h declares."""
import ctypes
import re
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent


def declared_functions():
    text = (ROOT / "include" / "loco_hd_hip.h").read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(lchd_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_the_four_drivers():
    names = declared_functions()
    for fn in ("lchd_from_anchors", "lchd_from_dmxs", "lchd_from_coords", "lchd_from_primitives", "lchd_from_primitives_dev",
               "lchd_ctx_create", "lchd_wf_cdf", "lchd_sd_run", "lchd_last_error"):
        assert fn in names


def test_library_exports_every_declared_symbol():
    from loco_hd_amd import _native

    lib = ctypes.CDLL(str(_native.LIB_PATH))
    missing = [fn for fn in declared_functions() if not hasattr(lib, fn)]
    assert not missing, missing
    # the ctypes prototypes in loco_hd_amd/_native.py cover the same set
    assert sorted(_native._PROTOS) == declared_functions()


def test_no_oracle_in_product_path():
    """The product package must never import / link the CPU oracle."""
    for path in (ROOT / "loco_hd_amd").rglob("*"):
        if path.suffix in {".py", ".hip", ".h", ".cpp"} or path.name == "Makefile":
            text = path.read_text(errors="replace")
            assert "oracle" not in text.lower(), path
