"""The C-ABI library loads on a CPU-only box and exports every symbol include/trs_abi.h declares; argument
validation (no kernel launch needed) reports errors through return codes and trs_last_error_string()."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "trs_abi.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(trs_[a-z0-9_]+)\s*\(", src)))


@pytest.fixture(scope="module")
def lib():
    from torecsys_amd import build, _abi
    build.build()
    return _abi.load()


def test_exports_every_declared_symbol(lib):
    from torecsys_amd import _abi
    names = _declared()
    assert len(names) >= 25
    for n in names:
        assert hasattr(lib, n), f"{n} declared in trs_abi.h but not exported"
    assert sorted(_abi.SIGNATURES) == names, "ctypes SIGNATURES out of sync with the header"


def test_version_and_sizes(lib):
    assert lib.trs_version() == 1
    assert lib.trs_csr_workspace_bytes(1000, 100) >= 400
    assert lib.trs_scatter_workspace_bytes(100000, 10, 16, 1) >= 8


def test_argument_errors_without_gpu(lib):
    from torecsys_amd import _abi
    null = ctypes.c_void_p(0)
    rc = lib.trs_gather_rows(null, 10, 4, 0, null, 0, null, 2, 2, null, null, null)
    assert rc == -1 and "NULL" in _abi.last_error()
    one = ctypes.c_void_p(16)
    rc = lib.trs_gather_rows(one, 10, 4, 7, one, 0, null, 2, 2, one, null, null)
    assert rc == -2 and "dtype" in _abi.last_error()
    rc = lib.trs_fm_fwd(one, 4, 0, 8, 0, one, null, null)
    assert rc == -1
    with pytest.raises(RuntimeError, match="trs_pair_dot_fwd failed"):
        _abi.call("trs_pair_dot_fwd", null, 1, 2, 4, 0, null, null)


def test_no_oracle_import_in_product():
    pkg = os.path.join(ROOT, "torecsys_amd")
    for f in os.listdir(pkg):
        if f.endswith(".py"):
            src = open(os.path.join(pkg, f)).read()
            assert "oracle" not in src.replace("no CPU", ""), f"{f} mentions the oracle"
    src = open(os.path.join(ROOT, "bench.py")).read() if os.path.exists(os.path.join(ROOT, "bench.py")) else ""
    assert "/root/reference" not in src
