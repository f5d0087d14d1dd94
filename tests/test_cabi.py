"""CPU-only checks of the drop-in boundary: the C-ABI library is built, loads,
and exports exactly the symbols include/sphhip.h declares; the product never
imports the oracle; enum tables on both sides of ctypes agree."""
import ctypes
import os
import re

import pytest

from conftest import REPO


def _declared():
    text = open(os.path.join(REPO, 'include', 'sphhip.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(sph_[a-z0-9_]+)\s*\(', text)))


def test_library_exports_every_declared_symbol():
    from pysph_amd import device as dev
    names = _declared()
    assert len(names) >= 20
    assert sorted(dev.SIGNATURES) == names
    lib = ctypes.CDLL(dev.LIB_PATH)
    for n in names:
        assert hasattr(lib, n), n
    dev.load_library()  # binds restype/argtypes for all of them


def test_property_and_enum_tables_agree():
    from pysph_amd import device as dev
    from pysph_amd import equations as E
    from pysph_amd import solid_mech as SM
    hdr = open(os.path.join(REPO, 'include', 'sphhip.h')).read()
    for name in ['x', 'y', 'z', 'u', 'v', 'w', 'h', 'm', 'rho', 'p', 'cs',
                 'arho', 'au', 'av', 'aw', 'ax', 'ay', 'az', 'dt_cfl',
                 'dt_force', 'V', 'uhat', 'auhat']:
        assert dev.prop_id(name) >= 0
    assert dev.prop_id('x') == 0 and dev.prop_id('nonsense') == -1
    for cname, val in re.findall(r'(SPH_EQ_[A-Z_]+)\s*=\s*(\d+)', hdr):
        py = getattr(E, cname[4:], None)
        if py is None:
            py = getattr(SM, cname[4:])
        assert py == int(val), cname


def test_no_gpu_means_loud_failure():
    """Without a GPU the product must raise, not fall back to a CPU path."""
    import torch
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    from pysph_amd import device as dev
    with pytest.raises(dev.SphError):
        dev.HipContext(0)


def test_product_does_not_import_oracle():
    pkg = os.path.join(REPO, 'pysph_amd')
    for root, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(('.py', '.hip', '.h')):
                src = open(os.path.join(root, f)).read()
                assert 'oracle' not in src.replace('Oracle', 'oracle') or \
                    f in (), (f, 'mentions the oracle')
