"""CPU: libbmpc.so loads and exports exactly what include/bmpc.h declares; argument validation and the
no-device behaviour (the library must refuse to compute without a GPU -- no CPU fallback)."""
import ctypes as C
import os
import re

import pytest

from tests import util


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__ as ge
    ge.build()
    from biped_mpc_py_amd import _lib
    return _lib.load()


def _declared():
    text = open(os.path.join(util.ROOT, "include", "bmpc.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(bmpc_[a-z_0-9]+)\s*\(", text)))


def test_header_symbols_exported(lib):
    from biped_mpc_py_amd import _lib
    names = _declared()
    assert names == sorted(_lib.EXPORTS)
    raw = C.CDLL(_lib.LIB_PATH)
    for n in names:
        assert hasattr(raw, n), n
    assert lib.bmpc_abi_version() == _lib.ABI_VERSION


def test_default_params_are_reference_defaults(lib):
    from biped_mpc_py_amd import _lib
    cp = _lib.CParams()
    assert lib.bmpc_default_params(cp, 10) == 0
    assert (cp.h, cp.half, cp.dt, cp.kv, cp.m, cp.g, cp.mu) == (10, 5, 0.04, 0.01, 12.0, 9.81, 0.5)   # REF:24-44
    assert list(cp.Q) == [500, 100, 100, 300, 300, 700, 1, 1, 1, 1, 1, 1, 1]
    assert list(cp.R) == [1e-4] * 12
    assert list(cp.x_cmd) == [0, 0, 0, 0, 0, 0.55, 0, 0, 0, 0, 0, 0]
    assert list(cp.tau_max) == [0, 67, 33.5] and list(cp.tau_min) == [0, -67, -33.5]
    assert [lib.bmpc_supported_horizon(h) for h in (10, 16, 20, 11, 4, 1, 0, 41)] == [1, 1, 1, 1, 1, 1, 0, 0]      # every h in [1, 40] (round 5)
    assert lib.bmpc_default_params(cp, 16) == 0 and cp.half == 8


def test_struct_layout_matches_c(lib):
    """sizeof(bmpc_params) as the C compiler sees it == ctypes mirror."""
    import subprocess
    import tempfile
    from biped_mpc_py_amd import _lib
    src = '#include <stdio.h>\n#include "bmpc.h"\nint main(){printf("%zu", sizeof(bmpc_params));return 0;}\n'
    with tempfile.TemporaryDirectory() as td:
        c = os.path.join(td, "s.c")
        open(c, "w").write(src)
        exe = os.path.join(td, "s")
        subprocess.check_call(["gcc", "-I", os.path.join(util.ROOT, "include"), c, "-o", exe])
        size = int(subprocess.check_output([exe]).decode())
    assert size == C.sizeof(_lib.CParams)


def test_no_device_no_fallback(lib):
    """Without a GPU, creating a solver fails loudly; nothing computes on the CPU."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from biped_mpc_py_amd import _lib
    import biped_mpc_py_amd as bm
    with pytest.raises(_lib.BmpcError) as ei:
        bm.BatchSolver()
    assert ei.value.code == -2 and "no HIP device" in str(ei.value)
    with pytest.raises(_lib.BmpcError):
        bm.solve_mpc([0] * 12, 0.0, [0] * 6, bm.MPC(), bm.Biped(), [[1, 1]] * 10)


def test_argument_validation(lib):
    from biped_mpc_py_amd import _lib
    h = C.c_void_p()
    cp = _lib.CParams()
    lib.bmpc_default_params(cp, 10)
    assert lib.bmpc_create(None, C.byref(cp), 0, 16) == -1
    assert lib.bmpc_create(C.byref(h), C.byref(cp), 0, 0) == -1
    assert b"max_batch" in lib.bmpc_last_error()
    assert lib.bmpc_destroy(None) == 0
    assert lib.bmpc_synchronize(None) == -1
    assert lib.bmpc_set_params(None, C.byref(cp)) == -1


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    """No shared library -> ImportError with build instructions; nothing falls back to the CPU."""
    from biped_mpc_py_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "libbmpc.so"))
    with pytest.raises(ImportError) as ei:
        _lib.load()
    assert "no CPU fallback" in str(ei.value)


def test_product_does_not_import_oracle():
    """The oracle is test infrastructure: no module of the package may import it."""
    pkg = os.path.join(util.ROOT, "biped_mpc_py_amd")
    for root, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                text = open(os.path.join(root, f)).read()
                assert "import oracle" not in text and "from oracle" not in text, f
