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


# ---- fixtures of the GPU parity modules (tests/test_gpu_*.py); module scope: each module imports the product afresh ----

@pytest.fixture(scope="module")
def dev():
    import torch
    assert torch.cuda.is_available(), "GPU tests need the MI355X; the HIP path has no fallback"
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def cmm(built):
    for k in ("custom_mm", "matmuls"):
        sys.modules.pop(k, None)
    import custom_mm
    assert custom_mm.__file__.endswith(".so") and "matrix-multiplication_amd" in custom_mm.__file__
    custom_mm.init_cublas()
    custom_mm.init_cusparse()
    return custom_mm


@pytest.fixture(scope="module")
def mm(cmm):
    import matmuls
    assert matmuls.custom_mm is cmm
    return matmuls


@pytest.fixture(scope="module")
def capi(built, cmm):
    import ctypes
    lib = ctypes.CDLL(str(built / "libmi_spmm.so"))
    vp, i64, i32 = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32
    lib.mi_spmm_csr_f32_variant.argtypes = [ctypes.c_int, vp, vp, vp, i64, i32, i32, i32, vp, i64, vp, i64, vp]
    return lib
