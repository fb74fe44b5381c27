"""CPU: repository rules -- the product never imports the oracle, never reads /root/reference, and the oracle says it is
test infrastructure."""
import os
import re

from conftest import ROOT

PKG = os.path.join(ROOT, "experiment-yolo_amd")


def _py_files(top):
    for d, _, fs in os.walk(top):
        for f in fs:
            if f.endswith((".py", ".hip", ".h", ".cc")):
                yield os.path.join(d, f)


def test_product_does_not_touch_oracle_or_reference():
    for path in list(_py_files(PKG)) + [os.path.join(ROOT, "include", "dealyolo_hip.h")]:
        src = open(path, errors="ignore").read()
        assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f"{path} imports the oracle"
        assert "/root/reference" not in src or path.endswith(".h"), f"{path} reads the reference tree"


def test_tools_do_not_import_oracle():
    for path in _py_files(os.path.join(ROOT, "tools")):
        assert not re.search(r"^\s*(from|import)\s+oracle\b", open(path).read(), flags=re.M), f"{path} imports the oracle"


def test_gpu_side_entry_points_do_not_read_reference():
    for f in ("bench.py", "__graft_entry__.py"):
        assert "/root/reference" not in open(os.path.join(ROOT, f)).read()
    for path in _py_files(os.path.join(ROOT, "tests")):
        if os.sep + "golden" + os.sep in path and os.path.basename(path) in ("make_golden.py", "_refimport.py", "check_export.py"):
            continue  # fixture generator / export check: build container only
        assert "/root/reference" not in open(path).read() or os.path.basename(path) == "test_layout.py", path


def test_oracle_is_labelled_test_infrastructure():
    hdr = open(os.path.join(ROOT, "oracle", "__init__.py")).read()
    assert "TEST INFRASTRUCTURE" in hdr
    for f in os.listdir(os.path.join(ROOT, "oracle")):
        if f.endswith(".py") and f != "__init__.py":
            assert "test infrastructure" in open(os.path.join(ROOT, "oracle", f)).read().lower(), f
