"""sha256 over the sources libsphhip.so / libsphcomm.so are built from; the
Makefile stores it next to the libraries (libsphhip.stamp) and the test session
/ __graft_entry__.build() compare it, so that a stale shared object is never
tested or benchmarked silently."""
import glob
import hashlib
import os

HERE = os.path.dirname(os.path.abspath(__file__))


def source_hash():
    files = sorted(glob.glob(os.path.join(HERE, '*.hip')) + glob.glob(os.path.join(HERE, '*.h')) +
                   glob.glob(os.path.join(HERE, '..', '..', 'include', '*.h')) +
                   [os.path.join(HERE, 'Makefile')])
    h = hashlib.sha256()
    for f in files:
        h.update(os.path.basename(f).encode())
        h.update(open(f, 'rb').read())
    return h.hexdigest()


def stamp_path():
    return os.path.join(HERE, '..', 'libsphhip.stamp')


def is_current():
    try:
        return open(stamp_path()).read().strip() == source_hash()
    except OSError:
        return False


if __name__ == '__main__':
    print(source_hash())
