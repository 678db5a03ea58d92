"""The memory-safety net under the kernels (`-m gpu`): the diagnostic build `liblobsim_bounds.so` (-DLOBSIM_BOUNDS -DLOBSIM_REDZONE,
rl4mm_amd/_lib.py:build_bounds_library) on drawn configurations, on the deep-queue ops, and on deliberately damaged state.

GPU AddressSanitizer cannot be had on the pool this was built on.  What stands in for it: every computed index into an env's arrays is
held against its extent (a violation sets LOBSIM_ST_INTERNAL with the source line of the check and redirects the access to element 0),
and 4 KiB sentinel zones sit around every per-env array and behind every env's part of the arrays whose stride is a run-time value.
The product library is the same source without the two macros; these tests show (a) that the diagnostic build agrees with the oracle
and reports nothing on the configurations the fuzzer draws, and (b) that it does report when the state IS damaged."""
import ctypes as C
import sys
from pathlib import Path

import numpy as np
import pytest

import oracle_lib as ol
from rl4mm_amd import _capi, _lib

sys.path.insert(0, str(Path(__file__).resolve().parent / "diag"))
import config_fuzz  # noqa: E402
import test_gpu_deep_queue as dq  # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def bounds():
    return _lib.bounds_binding()


@pytest.mark.parametrize("seed", [3, 11, 58, 131, 404, 977, 2024, 4711])
def test_drawn_configurations_on_the_bounds_build(bounds, seed):
    """Eight configurations of tests/diag/config_fuzz.py (quote range, step size, feature subsets and windows, normalisation, RollingSharpe,
    drifting days, several days per handle, multi-step launches, on-device policies, masks, the info calculator) on the diagnostic
    build: every env equal to the oracle at every step - LOBSIM_ST_INTERNAL would show as a status difference - and every red zone
    intact at the end."""
    err, desc = config_fuzz.run_case(seed, 48, new_gpu=lambda cfg: _capi.Sim(bounds, cfg))
    assert err is None, f"seed {seed}: {err}\n{desc}"
    assert desc["red_zones"] > 0


def test_deep_queue_ops_on_the_bounds_build(bounds):
    """3 500 orders on one price, removals from the middle, 600 more (tests/test_gpu_deep_queue.py): the moves between slab classes and the
    squeezes of a 64-page slab with every slot index checked."""
    gpu, cpu = dq._pair(lib=bounds)
    rng = np.random.Generator(np.random.PCG64(7))
    price = dq.MID - dq.TICK
    for what, ops in dq._ops_on_one_price(rng, price, 3500):
        for k in range(0, len(ops), 500):
            assert list(gpu.debug_process_orders(0, ops[k:k + 500])) == list(cpu.debug_process_orders(0, ops[k:k + 500]))
        dq._same(gpu, cpu, what)
        assert np.array_equal(dq._level(gpu, 0, price), dq._level(cpu, 0, price))
    zones, damaged = gpu.check_redzones()
    assert zones > 0 and damaged == 0
    gpu.close()


def _small(bounds, n=4):
    g, _ = ol.load_traj("synth_default_random")
    cfg = ol.cfg_from_golden(g, n_envs=n)
    sim = _capi.Sim(bounds, cfg)
    sim.add_stream(ol.golden_stream(g["stream_file"]))
    sim.set_portfolio(0, 10**9)
    sim.reset(np.full(n, g["starts_us"][0], dtype=np.int64), 0)
    return sim, cfg


@pytest.mark.parametrize("kind, what", [(0, "a slab id beyond the pool"), (1, "a ring position outside the ring")])
def test_damaged_state_is_reported_not_followed(bounds, kind, what):
    """The net catches what it is there for: env 1's state is damaged from the host (`lobsim_diag_corrupt`, diagnostic build only) - the
    best bid's queue pointer, a feature's ring position - and the next step reports LOBSIM_ST_INTERNAL with the line of the check that
    refused the access, on that env only; its neighbours step on, and no red zone was touched."""
    sim, cfg = _small(bounds)
    rng = np.random.Generator(np.random.PCG64(1))
    for _ in range(3):
        sim.step(rng.uniform(0, 10, size=(4, cfg.action_dim)))
    assert not sim.status().any()
    bounds.lib.lobsim_diag_corrupt.argtypes = [C.c_void_p, C.c_int32, C.c_int32]
    assert bounds.lib.lobsim_diag_corrupt(sim.h, 1, kind) == 0
    for _ in range(3):
        sim.step(rng.uniform(0, 10, size=(4, cfg.action_dim)))
    st = sim.status()
    assert st[1] & _capi.ST_INTERNAL, (what, st.tolist())
    line = (int(st[1]) >> 16) & 0x7fff
    src = (_lib.CSRC / "lobsim_kernels.hip").read_text().splitlines()
    assert 0 < line <= len(src) and ("BIDX" in src[line - 1] or "RIDX" in src[line - 1] or "BCHECK" in src[line - 1] or "SLAB_PTR" in src[line - 1]), (line, src[line - 1][:120])
    assert not (st[[0, 2, 3]] & _capi.ST_INTERNAL).any(), st.tolist()
    zones, damaged = sim.check_redzones()
    assert zones > 0 and damaged == 0
    sim.close()


def test_a_write_behind_an_envs_pages_is_seen(bounds):
    """One byte written behind env 2's queue pages (what an out-of-bounds slot store would do): the red-zone check names the zone."""
    sim, cfg = _small(bounds)
    assert sim.check_redzones()[1] == 0
    bounds.lib.lobsim_diag_corrupt.argtypes = [C.c_void_p, C.c_int32, C.c_int32]
    assert bounds.lib.lobsim_diag_corrupt(sim.h, 2, 2) == 0
    zones, damaged = sim.check_redzones()
    assert damaged == 1
    msg = bounds.fn["last_error"](sim.h).decode()
    assert "pool" in msg and "env 2" in msg and "offset 17" in msg, msg
    sim.close()


def test_the_product_library_has_no_zones():
    sim = _lib.new_sim(_capi.make_cfg(2, 10, [_capi.make_feature(_capi.F_SPREAD, 0, 100_000, 0, 5000)]))
    assert sim.check_redzones() == (0, 0)
    sim.close()
