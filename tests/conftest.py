import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

GOLDEN = os.path.join(REPO, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line(
        'markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def pytest_sessionstart(session):
    """The shared objects are not in the git history: a checkout that was never
    built, OR whose libraries were built from other sources than the ones in
    the tree now (pysph_amd/csrc/srchash.py: sha256 of csrc/ + include/ stored by
    the Makefile in libsphhip.stamp), is (re)built once with the Makefile
    __graft_entry__.build() uses.  No fallback is involved: without hipcc a
    stale or missing library is an error, not a skipped check."""
    import shutil
    import subprocess
    sys.path.insert(0, os.path.join(REPO, 'pysph_amd', 'csrc'))
    import srchash
    lib = os.path.join(REPO, 'pysph_amd', 'libsphhip.so')
    hipcc = shutil.which('hipcc') or ('/opt/rocm/bin/hipcc' if os.path.exists('/opt/rocm/bin/hipcc') else None)
    if os.path.exists(lib) and srchash.is_current():
        return
    if not hipcc:
        raise RuntimeError('pysph_amd/libsphhip.so is missing or was built from different sources '
                           '(libsphhip.stamp != srchash) and there is no hipcc to rebuild it')
    subprocess.check_call(['make', '-C', os.path.join(REPO, 'pysph_amd', 'csrc'), '-j8'])
    assert srchash.is_current()


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name))


def arrays_from_golden(g, which='in'):
    """Rebuild pysph_amd ParticleArrays from a golden file's stored state."""
    from pysph_amd.particle_array import ParticleArray
    names = []
    for key in g.files:
        parts = key.split('/')
        if parts[0] == which and parts[1] not in names:
            names.append(parts[1])
    out = []
    for name in names:
        props = {}
        for key in g.files:
            parts = key.split('/')
            if parts[0] == which and parts[1] == name:
                props[parts[2]] = g[key].copy()
        consts = {}
        for key in g.files:
            parts = key.split('/')
            if parts[0] == 'const' and parts[1] == name:
                consts[parts[2]] = g[key].copy()
        pa = ParticleArray(name=name, constants=consts, **props)
        nreal = 'nreal/%s' % name
        if nreal in g.files:
            pa.set_num_real_particles(int(g[nreal]))
        out.append(pa)
    return out


@pytest.fixture(scope='session')
def oracle():
    from oracle import oracle as orc
    orc.lib()
    return orc
