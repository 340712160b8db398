import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def pytest_sessionstart(session):
    """The native pieces are built in-tree by __graft_entry__.build(); a checkout without them (built artefacts are
    git-ignored) builds them once here.  This only makes sure the HIP library EXISTS - nothing falls back to the CPU."""
    lib = os.path.join(ROOT, 'ml4ca_amd', 'lib', 'libdpenv.so')
    orc = os.path.join(ROOT, 'oracle', 'libdpenv_oracle.so')
    if not (os.path.exists(lib) and os.path.exists(orc)):
        import __graft_entry__
        __graft_entry__.build()
