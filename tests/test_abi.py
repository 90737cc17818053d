"""The C-ABI shared library loads (no GPU needed) and exports every symbol that
include/rlpyt_hip.h declares; the ctypes binding covers exactly the same set."""
import os
import re
import subprocess

from conftest import ROOT


def header_functions():
    src = open(os.path.join(ROOT, "include", "rlpyt_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    names = re.findall(r"\b(rlpyt_[a-z0-9_]+)\s*\(", src)
    return sorted(set(names))


def test_library_exports_every_declared_symbol():
    from rlpyt_amd import _lib
    declared = header_functions()
    assert len(declared) >= 25
    out = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True,
                         text=True, check=True).stdout
    exported = {line.split()[-1] for line in out.splitlines() if " T " in line}
    missing = [n for n in declared if n not in exported]
    assert not missing, f"declared in rlpyt_hip.h but not exported: {missing}"


def test_binding_matches_header():
    from rlpyt_amd import _lib
    assert sorted(_lib.EXPORTED_SYMBOLS) == header_functions()
    assert _lib.lib.rlpyt_hip_abi_version() == _lib.ABI_VERSION


def test_no_cpu_fallback_without_gpu():
    """On a CPU-only torch the product path must raise, not fall back."""
    import pytest
    import torch
    from rlpyt_amd import _lib, ops
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(_lib.HipError):
        ops.valid_from_done(torch.zeros(4, 2, dtype=torch.bool))


def test_product_does_not_import_oracle():
    """Nothing under rlpyt_amd/ may reference the oracle (test infrastructure only)."""
    bad = []
    for dp, _dn, fns in os.walk(os.path.join(ROOT, "rlpyt_amd")):
        for fn in fns:
            if fn.endswith((".py", ".hip", ".cpp", ".h")):
                txt = open(os.path.join(dp, fn)).read()
                if re.search(r"^\s*(from|import)\s+oracle\b", txt, flags=re.M) or \
                        "/root/reference" in txt:
                    bad.append(os.path.join(dp, fn))
    assert not bad, bad
