"""CPU-only checks: the C-ABI library loads and exports every symbol the header declares."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "flmarl.h")).read()
    return sorted(set(re.findall(r"^(?:int|long|const char\*)\s+(flm_\w+)\s*\(", text, flags=re.M)))


def test_header_declares_symbols():
    names = _declared()
    assert "flm_graph_build" in names and "flm_observations" in names and len(names) >= 7


def test_library_exports_every_declared_symbol():
    import __graft_entry__ as g
    g.build()
    lib = ctypes.CDLL(os.path.join(ROOT, "flatland_marl_amd", "libflmarl_hip.so"))
    for name in _declared():
        assert hasattr(lib, name), name
    assert lib.flm_abi_version() >= 1


def test_binding_covers_header():
    from flatland_marl_amd import _lib
    assert set(_declared()) <= set(_lib.exported_symbols())


def test_product_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import numpy as np
    import flatland_marl_amd as fm
    with pytest.raises(fm.FlmError):
        fm.ObservationBatch(np.zeros((4, 4), np.uint16), [[0, 0]], [0], [[1, 1]], [1.0], 10)


def test_product_does_not_import_oracle():
    pkg = os.path.join(ROOT, "flatland_marl_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                text = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in text.replace("no oracle", ""), os.path.join(dirpath, f)


def test_entry_points_reject_null_and_zero_arguments():
    """Every launching entry point checks its arguments on the host before it touches the device: with null
    pointers and zero sizes it returns FLM_E_ARG (no GPU needed, nothing is launched)."""
    from flatland_marl_amd import _lib
    lib = _lib.load()
    skip = {"flm_abi_version"} | set(_lib._LONG_RESULT)
    checked = 0
    for name, argtypes in _lib._SIGNATURES.items():
        if name in skip:
            continue
        args = [None if t is ctypes.c_void_p else t(0) for t in argtypes]
        assert getattr(lib, name)(*args) == -1, name
        checked += 1
    assert checked >= 25
    # size queries answer 0 or a positive count, never garbage, for degenerate sizes
    assert lib.flm_dqn_param_count(300, 151, 2) == 19673091
    assert lib.flm_dqn_param_count(300, 151, 5) == 19673091 + 3 * 513
