"""pytest configuration: markers, import paths, shared fixtures."""
import sys
from pathlib import Path

import numpy as np
import pytest

REPO = Path(__file__).resolve().parent.parent
PKG = REPO / "matrix-multiplication_amd"
for p in (str(REPO), str(PKG), str(Path(__file__).resolve().parent)):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _ensure_built():
    """The built libraries travel with the tree; build them if this is a fresh checkout."""
    import sysconfig
    ext = PKG / ("custom_mm" + sysconfig.get_config_var("EXT_SUFFIX"))
    if not (PKG / "libmi_spmm.so").exists() or not ext.exists():
        import __graft_entry__
        __graft_entry__.build()


@pytest.fixture(scope="session")
def built():
    _ensure_built()
    return PKG


@pytest.fixture(scope="session")
def golden():
    data = np.load(REPO / "tests" / "golden" / "golden_v1.npz")

    class Golden:
        names = [str(n) for n in data["__names__"]]

        def case(self, name):
            prefix = name + "/"
            return {k[len(prefix):]: data[k] for k in data.files if k.startswith(prefix)}

        def cases(self, family):
            return [n for n in self.names if n.startswith(family + "/")]

    return Golden()


@pytest.fixture(scope="session")
def oracle_mod():
    import oracle
    oracle.build()
    return oracle
