"""CPU tests of the C-ABI: the HIP library loads without a GPU, exports every symbol include/nm.h declares, and refuses
to run without a device (no CPU fallback)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols(header='nm.h'):
    txt = open(os.path.join(ROOT, 'include', header)).read()
    txt = re.sub(r'/\*.*?\*/', '', txt, flags=re.S)
    return sorted(set(re.findall(r'\b(nm_[a-z_0-9]+)\s*\(', txt)))


def test_header_symbols_are_exported_and_bound():
    from neuralmelting_amd import _lib
    L = C.CDLL(_lib.LIB_PATH)
    syms = declared_symbols()
    assert len(syms) >= 25
    for s in syms:
        assert hasattr(L, s), s
    assert sorted(_lib.SYMBOLS) == syms          # the Python binding covers exactly the header
    dsyms = declared_symbols('nm_distr.h')
    assert dsyms == sorted(_lib.DISTR_SYMBOLS)
    for s in dsyms:
        assert hasattr(L, s), s
    psyms = declared_symbols('nm_parse.h')
    assert psyms == sorted(_lib.PARSE_SYMBOLS)
    for s in psyms:
        assert hasattr(L, s), s


def test_config_struct_matches_header():
    from neuralmelting_amd import _lib
    # 12 x int32/uint32, 2 x double, 2 pointers, natural alignment
    assert C.sizeof(_lib.NMConfig) == 12 * 4 + 2 * 8 + 2 * 8 + 2 * 4


def test_product_never_imports_oracle():
    import subprocess
    out = subprocess.run(['grep', '-rIl', 'oracle', os.path.join(ROOT, 'neuralmelting_amd'), '--include=*.py', '--include=*.h',
                          '--include=*.hip'], capture_output=True, text=True).stdout.split()
    assert out == []


def test_no_gpu_means_loud_error():
    import torch
    if torch.cuda.is_available():
        pytest.skip('a GPU is present')
    import neuralmelting_amd as nm
    with pytest.raises(nm.NMError) as e:
        nm.Engine(256, np.float32([1, 8]), np.float32([0.25, 2.5]))
    assert e.value.code == -2 and 'no HIP device' in str(e.value)


def test_replica_members_are_all_inlined():
    """The block kernel keeps a replica's state in a `Replica` object that must stay in registers.  An out-of-line member function
    takes the object's address: it then lives in scratch memory, its pointer members lose their address space and every list load
    becomes a flat load (seen in round 3 when rebuild() grew past the inliner's threshold: the 8^3 kernel ran 14 % slower with
    identical results).  Device functions that were not inlined keep their symbol in the library's embedded code object."""
    from neuralmelting_amd import _lib
    blob = open(_lib.LIB_PATH, 'rb').read()
    assert b'_ZN2nm13gaussian_fill' in blob      # (the check sees device symbols: this one is out of line on purpose)
    assert b'_ZN2nm7Replica' not in blob
