"""The C-ABI library loads without a GPU and exports every symbol include/mi355_isdf.h declares;
the binding table in pyscf_isdf_amd/lib.py covers exactly that set; the product refuses to run
without a GPU (no CPU fallback)."""
import ctypes
import os
import re
import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, 'include', 'mi355_isdf.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(isdf_[a-zA-Z0-9_]+)\s*\(', text)))


def test_library_exports_every_declared_symbol():
    from pyscf_isdf_amd import lib
    l = lib.load()
    names = _declared()
    assert len(names) >= 20
    for n in names:
        assert hasattr(l, n), 'missing symbol %s' % n
    assert sorted(lib.SIGNATURES) == names
    assert l.isdf_abi_version() == lib.ABI_VERSION


def test_no_gpu_fails_loudly():
    import torch
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    from pyscf_isdf_amd import lib
    with pytest.raises(lib.IsdfError):
        lib.Handle(0)
    import cells
    from pyscf_isdf_amd.isdf import ISDF
    df = ISDF(cells.cell_he_c(), c_isdf=2)
    with pytest.raises(lib.IsdfError):
        df.build()


def test_product_never_imports_oracle():
    """Static check: no module under pyscf_isdf_amd/ mentions the oracle package."""
    pkg = os.path.join(ROOT, 'pyscf_isdf_amd')
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(('.py', '.hip', '.h')):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r'^\s*(from|import)\s+oracle\b', src, flags=re.M), f
