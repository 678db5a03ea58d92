"""Test-side loader of the CPU oracle (oracle/liblobsim_oracle.so).  Only tests/ and bench.py's cpu_baseline
leg import this; the product package never does."""
import ctypes as C
import json
import subprocess
from pathlib import Path

import numpy as np

from rl4mm_amd import _capi
from rl4mm_amd.stream import PackedStream, pack_lobster

ROOT = Path(__file__).resolve().parent.parent
GOLD = ROOT / "tests" / "golden"
_binding = None


def oracle_binding() -> _capi.Binding:
    global _binding
    if _binding is None:
        so = ROOT / "oracle" / "liblobsim_oracle.so"
        subprocess.check_call(["make", "-C", str(ROOT / "oracle"), "-s"])  # (a no-op when the library is newer than its source and the header)
        b = _capi.Binding(so, prefix="oracle_", skip=_capi.ORACLE_SKIP)
        P64, PD = C.POINTER(C.c_int64), C.POINTER(C.c_double)
        b.lib.oracle_beta_volumes.restype = None
        b.lib.oracle_beta_volumes.argtypes = [C.POINTER(_capi.Cfg), PD, P64, P64]
        b.lib.oracle_peek_agent_orders.restype = C.c_int
        b.lib.oracle_peek_agent_orders.argtypes = [C.c_void_p, C.c_int32, PD, P64, C.c_int32]
        PI32, PU8 = C.POINTER(C.c_int32), C.POINTER(C.c_uint8)
        b.lib.oracle_rollout.restype = C.c_int
        b.lib.oracle_rollout.argtypes = [C.c_void_p, C.c_int32, PD, C.c_int32, PD, PU8, PD, PD, PU8, PD, PI32, PD]
        b.lib.oracle_set_threads.restype = None
        b.lib.oracle_set_threads.argtypes = [C.c_int]
        b.lib.oracle_set_threads(usable_cpus())  # (a GPU box shows 256 CPUs and grants 16: 256 OpenMP threads per small step call crawl)
        _binding = b
    return _binding


def usable_cpus() -> int:
    """CPUs this process can keep busy: the affinity mask capped by the cgroup's CPU quota."""
    import os

    n = len(os.sched_getaffinity(0))
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            n = max(1, min(n, int(float(q) / float(per) + 0.5)))
    except (OSError, ValueError):
        pass
    return n


def oracle_sim(cfg) -> _capi.Sim:
    return _capi.Sim(oracle_binding(), cfg)


def oracle_fills(sim, env=0, cap=4096) -> np.ndarray:
    return sim.fills(env, cap)


def oracle_high_water(sim) -> np.ndarray:
    """[n_envs, 4] = most orders at one price, most live agent orders, most buy levels, most sell levels an env has held so far."""
    out = np.zeros((sim.n_envs, 4), dtype=np.int32)
    sim.b.lib.oracle_get_high_water.restype = C.c_int
    sim.b.lib.oracle_get_high_water.argtypes = [C.c_void_p, C.c_int32, C.POINTER(C.c_int32)]
    for e in range(sim.n_envs):
        assert sim.b.lib.oracle_get_high_water(sim.h, e, out[e].ctypes.data_as(C.POINTER(C.c_int32))) == 0
    return out


def oracle_peek_orders(sim, action, env=0, cap=1024) -> np.ndarray:
    out = np.zeros((cap, 5), dtype=np.int64)
    a = np.ascontiguousarray(action, dtype=np.float64)
    n = sim.b.lib.oracle_peek_agent_orders(sim.h, env, a.ctypes.data_as(C.POINTER(C.c_double)), out.ctypes.data_as(C.POINTER(C.c_int64)), cap)
    return out[:n, :4]


REWARDS = {"PnL": (_capi.R_PNL, 0, 0.0), "AD": (_capi.R_INVENTORY_ADJUSTED, 1, 0.1), "SD": (_capi.R_INVENTORY_ADJUSTED, 0, 0.1),
           "RS": (_capi.R_ROLLING_SHARPE, 0, 0.0, 120, 60)}  # RollingSharpe() defaults, RewardFunctions.py:39


def reward_desc(spec):
    """Golden reward spec: a get_reward_function name, or ["RS", max_window, min_window]."""
    return REWARDS[spec] if isinstance(spec, str) else (_capi.R_ROLLING_SHARPE, 0, 0.0, int(spec[1]), int(spec[2]))


def cfg_from_golden(g: dict, n_envs: int = 1) -> _capi.Cfg:
    """Golden trajectory config (json written by oracle/gen_golden.py) -> lobsim_cfg.  Reward names map like
    rl4mm/gym/utils.py:29-39 (inventory_aversion default 0.1)."""
    feats = [_capi.make_feature(**f) for f in g["features"]]
    return _capi.make_cfg(
        n_envs, g["n_levels"], feats, step_us=g["step_us"], episode_us=g["episode_us"], warmup_us=g["warmup_us"],
        tick_size=g["tick_size"], outer_levels=g["outer_levels"], min_quote_level=g["min_quote_level"],
        max_quote_level=g["max_quote_level"], concentration=g["concentration"] if g["use_concentration"] else None,
        market_order_clearing=g["market_order_clearing"], market_order_fraction_of_inventory=g["market_order_fraction_of_inventory"],
        enter_spread=g["enter_spread"], inc_prev_action_in_obs=g["inc_prev_action_in_obs"],
        per_step_reward=reward_desc(g["per_step_reward"]), terminal_reward=reward_desc(g["terminal_reward"]),
    )  # fmt: skip


_streams = {}


def golden_stream(name: str) -> PackedStream:
    if name not in _streams:
        if name == "msft":  # every book row is a snapshot, like the reference's own tests (book_snapshot_freq=None)
            _streams[name] = pack_lobster(GOLD / "msft_message_50.csv", GOLD / "msft_orderbook_50.csv.gz", "MSFT", "2012-06-21", 50, None, 1000, 1000)
        else:
            _streams[name] = PackedStream.load(GOLD / f"{name}.npz")
    return _streams[name]


def load_traj(name: str):
    z = np.load(GOLD / f"traj_{name}.npz", allow_pickle=False)
    return json.loads(str(z["cfg"])), z


def snapshot_row_from_levels(levels, n_levels: int) -> np.ndarray:
    """[(side, price, vol)] -> LOBSTER row (sell_p, sell_v, buy_p, buy_v)*L with price 0 = absent level."""
    row = np.zeros(4 * n_levels, dtype=np.int64)
    nb = ns = 0
    for side, price, vol in levels:
        if side == 0:
            row[4 * nb + 2], row[4 * nb + 3] = price, vol
            nb += 1
        else:
            row[4 * ns], row[4 * ns + 1] = price, vol
            ns += 1
    return row


def oracle_rollout(sim, policy, params, T, actions=None, want_actions=True):
    """Host mirror of Sim.rollout_device: -> dict(obs, reward, done, ret, steps, actions)."""
    n, A, D = sim.n_envs, sim.action_dim, sim.obs_dim
    p8 = np.zeros(8, dtype=np.float64)
    p8[: len(params)] = params
    obs = np.zeros((n, D)); reward = np.zeros(n); done = np.zeros(n, dtype=np.uint8); ret = np.zeros(n)
    steps = np.zeros(n, dtype=np.int32); acts_out = np.zeros((max(T, 1), n, A)) if want_actions else None
    PD, PU8, PI32 = C.POINTER(C.c_double), C.POINTER(C.c_uint8), C.POINTER(C.c_int32)
    a = None if actions is None else np.ascontiguousarray(actions, dtype=np.float64)
    rc = sim.b.lib.oracle_rollout(sim.h, policy, p8.ctypes.data_as(PD), T, None if a is None else a.ctypes.data_as(PD), None,
                                  obs.ctypes.data_as(PD), reward.ctypes.data_as(PD), done.ctypes.data_as(PU8), ret.ctypes.data_as(PD),
                                  steps.ctypes.data_as(PI32), None if acts_out is None else acts_out.ctypes.data_as(PD))
    assert rc == 0
    return dict(obs=obs, reward=reward, done=done, ret=ret, steps=steps, actions=None if acts_out is None else acts_out[:T])


TERADACTYL_GOLDEN = {  # the agents oracle/gen_golden.py drove the reference with (policy parameters of lobsim_rollout_device)
    "msft_teradactyl_ad": [100, 7.0, 0.5, 9.0, 1.0, 0, 0, 100],
    "synth_teradactyl_ad_spread": [300, 10.0, 0.5, 10.0, 1.0, 3, 0, 300],
}


class OracleHandle:
    """The CPU oracle where rl4mm_amd.env expects a device handle (tests and oracle/fuzz_facade.py only: the product has no CPU path).
    Everything is forwarded to the oracle's Sim; `host_io` / `step_host_io` - the array path writes actions into the handle's pinned
    buffers and reads the step's outputs from them - work over plain numpy arrays."""

    def __init__(self, cfg):
        self._s = oracle_sim(cfg)
        n, A, D = self._s.n_envs, self._s.action_dim, self._s.obs_dim
        self._io = dict(actions=np.zeros((n, A)), active=np.ones(n, dtype=np.uint8), obs=np.zeros((n, D)), obs_f32=np.zeros((n, D), dtype=np.float32),
                        reward=np.zeros(n), done=np.zeros(n, dtype=np.uint8))  # fmt: skip

    def __getattr__(self, name):
        return getattr(self._s, name)

    def host_io(self):
        return self._io

    def step_host_io(self, use_active=False, want_obs=1):
        o, r, d = self._s.step(self._io["actions"], self._io["active"] if use_active else None)
        self._io["obs"][:], self._io["obs_f32"][:], self._io["reward"][:], self._io["done"][:] = o, o, r, d

    def limits(self):
        big = 1 << 20
        return dict(level_slabs=big, ladder_levels=big, far_levels=big, queue_slots=big, agent_orders=big, fill_log_rows=big, bytes_per_env=0, queue_pages=0)

    def drop_stream(self, sid):
        pass
