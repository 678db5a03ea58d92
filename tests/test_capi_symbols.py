"""CPU-only: the product library builds for gfx950, loads, and exports every symbol include/lobsim.h declares."""
import re
from pathlib import Path

import pytest

from rl4mm_amd import _capi, _lib

ROOT = Path(__file__).resolve().parent.parent


def test_header_and_binding_agree():
    text = (ROOT / "include" / "lobsim.h").read_text()
    declared = set(re.findall(r"\b(lobsim_[a-z_]+)\s*\(", text))
    bound = {"lobsim_" + n for n in _capi.PROTOTYPES}
    assert declared == bound, (declared ^ bound)


def test_library_exports_every_symbol():
    b = _lib.binding()  # builds with hipcc if needed (cross-compiles without a GPU)
    assert set(b.fn) == set(_capi.PROTOTYPES)


def test_struct_sizes_match_header():
    import ctypes as C

    assert C.sizeof(_capi.FeatureDesc) == 56
    assert C.sizeof(_capi.RewardDesc) == 24
    assert C.sizeof(_capi.Cfg) == 14 * 4 + 3 * 8 + 2 * 8 + 2 * 24 + 16 * 56
    assert C.sizeof(_capi.EnvState) == 11 * 8 + 6 * 4 + 7 * 8
    assert C.sizeof(_capi.OrderView) == 24


def test_create_fails_loudly_without_gpu():
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    cfg = _capi.make_cfg(4, 10, [_capi.make_feature(_capi.F_INVENTORY, min_value=-1e6, max_value=1e6)])
    with pytest.raises(_capi.LobsimError):
        _lib.new_sim(cfg)
