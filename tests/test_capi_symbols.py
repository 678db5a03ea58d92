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
    assert C.sizeof(_capi.Cfg) == 16 * 4 + 3 * 8 + 2 * 8 + 2 * 24 + 16 * 56 + 2 * 4
    assert C.sizeof(_capi.EnvState) == 11 * 8 + 6 * 4 + 7 * 8
    assert C.sizeof(_capi.OrderView) == 24


def test_header_struct_sizes_compile_time(tmp_path):
    """A C translation unit against include/lobsim.h: the sizes/offsets the ctypes mirror assumes are _Static_asserted by
    the C compiler itself (a maintainer binding the header from another language gets the same layout)."""
    import ctypes as C
    import subprocess

    offs = "\n".join(
        f'_Static_assert(offsetof({cname}, {f}) == {getattr(cls, f).offset}, "{cname}.{f}");'
        for cname, cls in (("lobsim_feature", _capi.FeatureDesc), ("lobsim_reward", _capi.RewardDesc), ("lobsim_cfg", _capi.Cfg),
                           ("lobsim_env_state", _capi.EnvState), ("lobsim_order_view", _capi.OrderView), ("lobsim_limits", _capi.Limits))
        for f, _ in cls._fields_
    )  # fmt: skip
    src = tmp_path / "sizes.c"
    src.write_text(
        '#include <stddef.h>\n#include "lobsim.h"\n'
        f'_Static_assert(sizeof(lobsim_feature) == {C.sizeof(_capi.FeatureDesc)}, "lobsim_feature");\n'
        f'_Static_assert(sizeof(lobsim_reward) == {C.sizeof(_capi.RewardDesc)}, "lobsim_reward");\n'
        f'_Static_assert(sizeof(lobsim_cfg) == {C.sizeof(_capi.Cfg)}, "lobsim_cfg");\n'
        f'_Static_assert(sizeof(lobsim_env_state) == {C.sizeof(_capi.EnvState)}, "lobsim_env_state");\n'
        f'_Static_assert(sizeof(lobsim_order_view) == {C.sizeof(_capi.OrderView)}, "lobsim_order_view");\n'
        f'_Static_assert(sizeof(lobsim_limits) == {C.sizeof(_capi.Limits)}, "lobsim_limits");\n'
        f'_Static_assert(sizeof(lobsim_feature) == 56, "documented size");\n{offs}\nint main(void) {{ return 0; }}\n'
    )
    subprocess.check_call(["gcc", "-std=c11", "-Wall", "-Werror", "-I", str(ROOT / "include"), "-c", str(src), "-o", str(tmp_path / "sizes.o")])


def test_integration_md_struct_excerpt_matches_binding():
    """INTEGRATION.md section 4 shows a ctypes lobsim_feature: its field list must be the real one."""
    text = (ROOT / "INTEGRATION.md").read_text()
    block = text[text.index("class lobsim_feature(C.Structure)") :]
    block = block[: block.index("h = C.c_void_p()")]
    fields = re.findall(r'\("([a-z_]+)",\s*C\.(c_[a-z0-9]+)\)', block)
    import ctypes as C

    assert [(n, getattr(C, t)) for n, t in fields] == list(_capi.FeatureDesc._fields_)


def test_create_fails_loudly_without_gpu():
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    cfg = _capi.make_cfg(4, 10, [_capi.make_feature(_capi.F_INVENTORY, min_value=-1e6, max_value=1e6)])
    with pytest.raises(_capi.LobsimError):
        _lib.new_sim(cfg)
