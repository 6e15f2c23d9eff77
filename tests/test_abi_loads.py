"""CPU-side check: the C-ABI library loads and exports every symbol include/apx.h declares (no compute calls)."""
import os
import re

import pytest

from apex_amd import _lib

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    txt = open(os.path.join(REPO, "include", "apx.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(apx_[a-z_0-9]+)\s*\(", txt)))


def test_header_and_binding_agree():
    assert _declared_symbols() == sorted(_lib.SIGNATURES.keys())


def test_library_exports_every_declared_symbol():
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    lib = _lib.load()
    for name in _declared_symbols():
        assert hasattr(lib, name), name
    assert lib.apx_version() >= 100
    assert lib.apx_mlp_param_count(50, 256, 10) == 81418      # actor 81 418 params
    assert lib.apx_mlp_param_count(50, 256, 1) == 79105       # critic 79 105 params
    assert lib.apx_ppo_workspace_bytes(64, 50, 256, 10) > 0


def test_compute_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from apex_amd import engine
    with pytest.raises(_lib.ApxError):
        engine.returns_scan(torch.zeros(2, 2), torch.zeros(2, 2, dtype=torch.uint8), torch.zeros(2, 2), torch.zeros(2), 0.99)
