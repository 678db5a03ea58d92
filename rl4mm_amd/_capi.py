"""ctypes view of include/lobsim.h: struct layouts, function prototypes and a thin numpy-facing wrapper.

The binding is parameterised by (shared library, symbol prefix) so the parity tests can drive the CPU oracle
(``oracle_*`` symbols of oracle/liblobsim_oracle.so, loaded only from tests/ and bench.py's cpu_baseline) through
exactly the same calls as the product library (``lobsim_*`` symbols of rl4mm_amd/csrc/liblobsim.so).  Nothing in
this package ever opens the oracle.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Sequence

import numpy as np

MAX_FEATURES = 16
MAX_ACTION_DIM = 5
ABI_VERSION = 6  # LOBSIM_ABI_VERSION of the include/lobsim.h this file mirrors
FWD_UPDATE_FEATURES = 1  # enum lobsim_forward_flags

# status bits (include/lobsim.h: enum lobsim_status)
ST_EMPTY_BOOK, ST_BAD_MESSAGE, ST_LEVEL_OVERFLOW, ST_QUEUE_OVERFLOW = 1, 2, 4, 8
ST_AGENT_OVERFLOW, ST_STREAM_END, ST_ASSERT, ST_NOT_RESET = 16, 32, 64, 128
ST_INTERNAL = 256  # (the diagnostic build only: bits 16 - 30 = the line of lobsim_kernels.hip whose bounds check failed)

F_SPREAD, F_PRICE_MOVE, F_VOLATILITY, F_INVENTORY, F_EPISODE_PROPORTION, F_TIME_OF_DAY = 1, 2, 3, 4, 5, 6
F_TRADE_DIRECTION_IMBALANCE, F_TRADE_VOLUME_IMBALANCE, F_BOOK_IMBALANCE, F_PRICE_RANGE, F_PRICE, F_AMIHUD_LAMBDA = 7, 8, 9, 10, 11, 12
R_PNL, R_INVENTORY_ADJUSTED, R_ROLLING_SHARPE = 1, 2, 3
POLICY_GIVEN, POLICY_FIXED, POLICY_TERADACTYL = 0, 1, 2
INFO_COLS = ["asset_price", "inventory", "cash", "aum", "market_spread", "agent_spread", "agent_weighted_spread_value", "midprice_offset",
             "weighted_midprice_offset", "market_order_count", "market_order_total_volume"]  # enum lobsim_info_col
TRACE_COLS = ["reward", "aum", "inventory", "asset_price", "market_spread", "weighted_midprice_offset", "a0", "a1", "a2", "a3", "a4"]  # enum lobsim_trace_col
INFO_DIM, TRACE_DIM = len(INFO_COLS), len(TRACE_COLS)


class FeatureDesc(C.Structure):
    _fields_ = [
        ("kind", C.c_int32),
        ("lookback", C.c_int32),
        ("update_us", C.c_int64),
        ("min_value", C.c_double),
        ("max_value", C.c_double),
        ("dparam", C.c_double),
        ("iparam", C.c_int64),
        ("norm_len", C.c_int64),
    ]


class RewardDesc(C.Structure):
    _fields_ = [("kind", C.c_int32), ("asymmetrically_dampened", C.c_int32), ("inventory_aversion", C.c_double),
                ("max_window", C.c_int32), ("min_window", C.c_int32)]


class Cfg(C.Structure):
    _fields_ = [
        ("n_envs", C.c_int32),
        ("n_levels", C.c_int32),
        ("tick_size", C.c_int32),
        ("outer_levels", C.c_int32),
        ("min_quote_level", C.c_int32),
        ("max_quote_level", C.c_int32),
        ("action_dim", C.c_int32),
        ("use_concentration", C.c_int32),
        ("market_order_clearing", C.c_int32),
        ("enter_spread", C.c_int32),
        ("inc_prev_action_in_obs", C.c_int32),
        ("n_features", C.c_int32),
        ("device_id", C.c_int32),
        ("ladder_levels", C.c_int32),
        ("max_levels", C.c_int32),
        ("agent_orders", C.c_int32),
        ("step_us", C.c_int64),
        ("episode_us", C.c_int64),
        ("warmup_us", C.c_int64),
        ("concentration", C.c_double),
        ("market_order_fraction_of_inventory", C.c_double),
        ("per_step_reward", RewardDesc),
        ("terminal_reward", RewardDesc),
        ("features", FeatureDesc * MAX_FEATURES),
        ("queue_pages", C.c_int32),
        ("reserved1", C.c_int32),
    ]


class EnvState(C.Structure):
    _fields_ = [
        ("now_us", C.c_int64),
        ("terminal_us", C.c_int64),
        ("inventory", C.c_int64),
        ("cash", C.c_int64),
        ("price", C.c_double),
        ("best_buy", C.c_int64),
        ("best_sell", C.c_int64),
        ("min_buy_price", C.c_int64),
        ("max_sell_price", C.c_int64),
        ("initial_buy_range", C.c_int64),
        ("initial_sell_range", C.c_int64),
        ("n_buy_levels", C.c_int32),
        ("n_sell_levels", C.c_int32),
        ("n_agent_orders", C.c_int32),
        ("status", C.c_int32),
        ("stream_id", C.c_int32),
        ("n_resyncs", C.c_int32),
        ("agent_fill_count", C.c_int64),
        ("agent_buy_volume", C.c_int64),
        ("agent_sell_volume", C.c_int64),
        ("ext_buy_count", C.c_int64),
        ("ext_sell_count", C.c_int64),
        ("ext_buy_volume", C.c_int64),
        ("ext_sell_volume", C.c_int64),
    ]

    def as_dict(self):
        return {name: getattr(self, name) for name, _ in self._fields_}


class OrderView(C.Structure):
    _fields_ = [("id", C.c_int64), ("volume", C.c_int64), ("kind", C.c_int32), ("reserved", C.c_int32)]


class Limits(C.Structure):
    _fields_ = [("level_slabs", C.c_int32), ("ladder_levels", C.c_int32), ("far_levels", C.c_int32), ("queue_slots", C.c_int32),
                ("agent_orders", C.c_int32), ("fill_log_rows", C.c_int32), ("bytes_per_env", C.c_int64), ("queue_pages", C.c_int32), ("reserved", C.c_int32)]

    def as_dict(self):
        return {name: getattr(self, name) for name, _ in self._fields_}


class HostIO(C.Structure):
    _fields_ = [("actions", C.c_void_p), ("active", C.c_void_p), ("obs", C.c_void_p), ("obs_f32", C.c_void_p), ("reward", C.c_void_p), ("done", C.c_void_p)]


ORDER_VIEW_DTYPE = np.dtype([("id", "<i8"), ("volume", "<i8"), ("kind", "<i4"), ("reserved", "<i4")])

_P64 = C.POINTER(C.c_int64)
_P32 = C.POINTER(C.c_int32)
_PD = C.POINTER(C.c_double)
_PF = C.POINTER(C.c_float)
_PU8 = C.POINTER(C.c_uint8)

# name -> (restype, argtypes); every symbol include/lobsim.h declares
PROTOTYPES = {
    "abi_version": (C.c_int, []),
    "create": (C.c_int, [C.POINTER(Cfg), C.POINTER(C.c_void_p)]),
    "destroy": (None, [C.c_void_p]),
    "last_error": (C.c_char_p, [C.c_void_p]),
    "set_stream": (C.c_int, [C.c_void_p, C.c_void_p]),
    "add_stream": (C.c_int, [C.c_void_p, _P64, C.c_int64, _P64, C.c_int64]),
    "drop_stream": (C.c_int, [C.c_void_p, C.c_int32]),
    "set_portfolio": (C.c_int, [C.c_void_p, _P32, C.c_int32, _P64, _P64]),
    "reset": (C.c_int, [C.c_void_p, _P32, C.c_int32, _P32, _P64, _PD]),
    "step": (C.c_int, [C.c_void_p, _PD, _PU8, _PD, _PD, _PU8]),
    "step_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "sync": (C.c_int, [C.c_void_p]),
    "rollout_device": (C.c_int, [C.c_void_p, C.c_int32, _PD, C.c_int32] + [C.c_void_p] * 8),
    "set_info_buffer": (C.c_int, [C.c_void_p, C.c_void_p]),
    "set_trace_buffer": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32]),
    "obs_dim": (C.c_int, [C.c_void_p]),
    "get_env_state": (C.c_int, [C.c_void_p, C.c_int32, C.POINTER(EnvState)]),
    "get_portfolios": (C.c_int, [C.c_void_p, _P64, _P64, _PD]),
    "get_best_prices": (C.c_int, [C.c_void_p, _P64, _P64]),
    "get_book": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, _P64, _P64, _P32, C.c_int32, _P32]),
    "get_level_orders": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int64, C.POINTER(OrderView), C.c_int32, _P32]),
    "get_limits": (C.c_int, [C.c_void_p, C.POINTER(Limits)]),
    "get_status": (C.c_int, [C.c_void_p, _P32]),
    "host_io_buffers": (C.c_int, [C.c_void_p, C.POINTER(HostIO)]),
    "step_host_io": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32]),
    "step_device_skip": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "reset_device": (C.c_int, [C.c_void_p, _P32, C.c_int32, _P32, _P64, C.c_void_p, C.c_void_p, C.c_void_p]),
    "reset_episode": (C.c_int, [C.c_void_p, _P32, C.c_int32, _P32, _P64]),
    "forward": (C.c_int, [C.c_void_p, _P32, C.c_int32, _P64, _P64, _P32, C.c_int32, C.c_int32, _PD]),
    "get_fills": (C.c_int, [C.c_void_p, C.c_int32, _P64, C.c_int32, _P32]),
    "features_reset": (C.c_int, [C.c_void_p, _P32, C.c_int32]),
    "debug_set_book": (C.c_int, [C.c_void_p, C.c_int32, _P64, C.c_int64]),
    "debug_process_orders": (C.c_int, [C.c_void_p, C.c_int32, _P64, C.c_int32, _P64]),
    "debug_check_redzones": (C.c_int, [C.c_void_p, _P32, _P32]),
}
# the oracle has no device entry points
ORACLE_SKIP = {"drop_stream", "set_stream", "step_device", "sync", "rollout_device", "set_info_buffer", "set_trace_buffer", "host_io_buffers", "step_host_io",
               "step_device_skip", "reset_device", "debug_check_redzones"}


def _ptr(a: Optional[np.ndarray], typ):
    return None if a is None else a.ctypes.data_as(typ)


class LobsimError(RuntimeError):
    pass


class Binding:
    """Function table of one shared library."""

    def __init__(self, path: str, prefix: str = "lobsim_", skip: Sequence[str] = (), check_abi: bool = True):
        self.path = str(path)
        self.prefix = prefix
        self.lib = C.CDLL(self.path)
        self.fn = {}
        for name, (res, args) in PROTOTYPES.items():
            if name in skip:
                continue
            f = getattr(self.lib, prefix + name)  # AttributeError if the library lacks a declared symbol
            f.restype = res
            f.argtypes = args
            self.fn[name] = f
        got = self.fn["abi_version"]()
        if got != ABI_VERSION and check_abi:  # (a prebuilt library from before a struct changed would read lobsim_cfg at the wrong offsets)
            raise LobsimError(f"{self.path}: {prefix}abi_version() = {got}, this binding mirrors include/lobsim.h at version {ABI_VERSION}: rebuild the library")


def make_feature(kind, lookback=0, update_us=100_000, min_value=0.0, max_value=0.0, dparam=0.0, iparam=0, norm_len=0) -> FeatureDesc:
    return FeatureDesc(int(kind), int(lookback), int(update_us), float(min_value), float(max_value), float(dparam), int(iparam), int(norm_len))


def make_cfg(
    n_envs: int,
    n_levels: int,
    features: Sequence[FeatureDesc],
    step_us: int = 100_000,
    episode_us: int = 1_800_000_000,
    warmup_us: Optional[int] = None,
    tick_size: int = 100,
    outer_levels: int = 20,
    min_quote_level: int = 0,
    max_quote_level: int = 10,
    concentration: Optional[float] = None,
    market_order_clearing: bool = False,
    market_order_fraction_of_inventory: float = 0.0,
    enter_spread: bool = False,
    inc_prev_action_in_obs: bool = False,
    per_step_reward=(R_PNL, 0, 0.0),
    terminal_reward=(R_PNL, 0, 0.0),
    device_id: int = 0,
    ladder_levels: int = 0,
    max_levels: int = 0,
    agent_orders: int = 0,
    queue_pages: int = 0,
) -> Cfg:
    cfg = Cfg()
    cfg.n_envs, cfg.n_levels, cfg.tick_size, cfg.outer_levels = n_envs, n_levels, tick_size, outer_levels
    cfg.min_quote_level, cfg.max_quote_level = min_quote_level, max_quote_level
    cfg.use_concentration = int(concentration is not None)
    cfg.concentration = float(concentration or 0.0)
    cfg.action_dim = (2 if concentration is not None else 4) + int(bool(market_order_clearing))
    cfg.market_order_clearing = int(bool(market_order_clearing))
    cfg.market_order_fraction_of_inventory = float(market_order_fraction_of_inventory)
    cfg.enter_spread = int(bool(enter_spread))
    cfg.inc_prev_action_in_obs = int(bool(inc_prev_action_in_obs))
    assert len(features) <= MAX_FEATURES
    cfg.n_features = len(features)
    for i, f in enumerate(features):
        cfg.features[i] = f
    cfg.step_us, cfg.episode_us = int(step_us), int(episode_us)
    if warmup_us is None:
        warmup_us = max([f.lookback * f.update_us for f in features] + [0])
    cfg.warmup_us = int(warmup_us)
    def _rw(t):
        t = tuple(t)
        return RewardDesc(*(t + (120, 60)[len(t) - 3 :])) if len(t) < 5 else RewardDesc(*t)

    cfg.per_step_reward = _rw(per_step_reward)
    cfg.terminal_reward = _rw(terminal_reward)
    cfg.device_id = device_id
    cfg.ladder_levels = ladder_levels
    cfg.max_levels = max_levels
    cfg.agent_orders = agent_orders
    cfg.queue_pages = queue_pages
    return cfg


class Sim:
    """One simulator handle behind a :class:`Binding` (numpy in, numpy out)."""

    def __init__(self, binding: Binding, cfg: Cfg):
        self.b = binding
        self.cfg = cfg
        self.h = C.c_void_p()
        rc = binding.fn["create"](C.byref(cfg), C.byref(self.h))
        if rc != 0 or not self.h:
            raise LobsimError(f"{binding.prefix}create failed with code {rc}")
        self.n_envs = cfg.n_envs
        self.action_dim = cfg.action_dim
        self.obs_dim = binding.fn["obs_dim"](self.h)
        self._streams = []

    def _check(self, rc, what):
        if rc != 0:
            msg = self.b.fn["last_error"](self.h)
            raise LobsimError(f"{self.b.prefix}{what} failed ({rc}): {msg.decode() if msg else ''}")

    def close(self):
        if self.h:
            self.b.fn["destroy"](self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- data -----------------------------------------------------------------------------------------------
    def add_stream(self, stream) -> int:
        assert stream.n_levels == self.cfg.n_levels, "stream depth must equal cfg.n_levels"
        msgs = np.ascontiguousarray(stream.msgs, dtype=np.int64)
        snaps = np.ascontiguousarray(stream.snaps, dtype=np.int64)
        sid = self.b.fn["add_stream"](self.h, _ptr(msgs, _P64), msgs.shape[0], _ptr(snaps, _P64), snaps.shape[0])
        if sid < 0:
            self._check(sid, "add_stream")
        if sid < len(self._streams):
            self._streams[sid] = stream  # (a slot lobsim_drop_stream gave back)
        else:
            self._streams.append(stream)
        return sid

    def drop_stream(self, sid: int):
        """lobsim_drop_stream: frees the device copy; the id is handed out again by a later add_stream"""
        self._check(self.b.fn["drop_stream"](self.h, int(sid)), "drop_stream")
        self._streams[sid] = None

    def set_portfolio(self, inventory, cash, env_ids=None):
        ids = np.arange(self.n_envs, dtype=np.int32) if env_ids is None else np.ascontiguousarray(env_ids, dtype=np.int32)
        inv = np.ascontiguousarray(np.broadcast_to(np.asarray(inventory, dtype=np.int64), ids.shape))
        csh = np.ascontiguousarray(np.broadcast_to(np.asarray(cash, dtype=np.int64), ids.shape))
        self._check(self.b.fn["set_portfolio"](self.h, _ptr(ids, _P32), len(ids), _ptr(inv, _P64), _ptr(csh, _P64)), "set_portfolio")

    # ---- env API ---------------------------------------------------------------------------------------------
    def reset(self, episode_start_us, stream_ids=0, env_ids=None) -> np.ndarray:
        ids = np.arange(self.n_envs, dtype=np.int32) if env_ids is None else np.ascontiguousarray(env_ids, dtype=np.int32)
        sids = np.ascontiguousarray(np.broadcast_to(np.asarray(stream_ids, dtype=np.int32), ids.shape))
        starts = np.ascontiguousarray(np.broadcast_to(np.asarray(episode_start_us, dtype=np.int64), ids.shape))
        obs = np.zeros((len(ids), self.obs_dim), dtype=np.float64)
        self._check(self.b.fn["reset"](self.h, _ptr(ids, _P32), len(ids), _ptr(sids, _P32), _ptr(starts, _P64), _ptr(obs, _PD)), "reset")
        return obs

    def step(self, actions, active=None):
        actions = np.ascontiguousarray(actions, dtype=np.float64).reshape(self.n_envs, self.action_dim)
        act = None if active is None else np.ascontiguousarray(active, dtype=np.uint8)
        # (the HIP library writes all three in full; the oracle leaves the rows of envs it does not step alone, so with a mask they
        # start as zeros)
        alloc = np.empty if act is None else np.zeros
        obs = alloc((self.n_envs, self.obs_dim), dtype=np.float64)
        reward = alloc(self.n_envs, dtype=np.float64)
        done = alloc(self.n_envs, dtype=np.uint8)
        self._check(self.b.fn["step"](self.h, _ptr(actions, _PD), _ptr(act, _PU8), _ptr(obs, _PD), _ptr(reward, _PD), _ptr(done, _PU8)), "step")
        return obs, reward, done

    def step_device(self, d_actions: int, d_active: int, d_obs: int, d_obs_f32: int, d_reward: int, d_done: int, d_return_acc: int = 0):
        self._check(self.b.fn["step_device"](self.h, d_actions, d_active, d_obs, d_obs_f32, d_reward, d_done, d_return_acc), "step_device")

    def step_device_skip(self, d_actions: int, d_skip: int, d_obs: int, d_obs_f32: int, d_reward: int, d_done: int, d_return_acc: int = 0):
        self._check(self.b.fn["step_device_skip"](self.h, d_actions, d_skip, d_obs, d_obs_f32, d_reward, d_done, d_return_acc), "step_device_skip")

    def reset_device(self, episode_start_us, stream_ids, env_ids, d_obs: int, d_obs_f32: int, d_done: int):
        ids = np.arange(self.n_envs, dtype=np.int32) if env_ids is None else np.ascontiguousarray(env_ids, dtype=np.int32)
        sids = np.ascontiguousarray(np.broadcast_to(np.asarray(stream_ids, dtype=np.int32), ids.shape))
        starts = np.ascontiguousarray(np.broadcast_to(np.asarray(episode_start_us, dtype=np.int64), ids.shape))
        self._check(self.b.fn["reset_device"](self.h, _ptr(ids, _P32), len(ids), _ptr(sids, _P32), _ptr(starts, _P64), d_obs, d_obs_f32, d_done), "reset_device")

    def host_io(self) -> dict:
        """numpy views of the handle's pinned step buffers (lobsim_host_io_buffers): fill ``actions``, call step_host_io, read the rest in place."""
        io = HostIO()
        self._check(self.b.fn["host_io_buffers"](self.h, C.byref(io)), "host_io_buffers")
        N, A, D = self.n_envs, self.action_dim, self.obs_dim

        def view(addr, ctype, shape):
            n = int(np.prod(shape))
            return np.ctypeslib.as_array((ctype * n).from_address(addr)).reshape(shape)

        return dict(actions=view(io.actions, C.c_double, (N, A)), active=view(io.active, C.c_uint8, (N,)), obs=view(io.obs, C.c_double, (N, D)),
                    obs_f32=view(io.obs_f32, C.c_float, (N, D)), reward=view(io.reward, C.c_double, (N,)), done=view(io.done, C.c_uint8, (N,)))

    def step_host_io(self, use_active: bool = False, want_obs: int = 1):
        self._check(self.b.fn["step_host_io"](self.h, int(bool(use_active)), int(want_obs)), "step_host_io")

    def limits(self) -> dict:
        lim = Limits()
        self._check(self.b.fn["get_limits"](self.h, C.byref(lim)), "get_limits")
        return lim.as_dict()

    def check_redzones(self):
        """(zones looked at, zones damaged) - 0 / 0 on the product library, which has none (lobsim_debug_check_redzones)."""
        n, bad = C.c_int32(0), C.c_int32(0)
        self._check(self.b.fn["debug_check_redzones"](self.h, C.byref(n), C.byref(bad)), "debug_check_redzones")
        return int(n.value), int(bad.value)

    def status(self) -> np.ndarray:
        out = np.zeros(self.n_envs, dtype=np.int32)
        self._check(self.b.fn["get_status"](self.h, _ptr(out, _P32)), "get_status")
        return out

    # ---- simulator level (OrderbookSimulator.reset_episode / forward_step) ---------------------------------------
    def reset_episode(self, start_us, stream_ids=0, env_ids=None):
        ids = np.arange(self.n_envs, dtype=np.int32) if env_ids is None else np.ascontiguousarray(env_ids, dtype=np.int32)
        sids = np.ascontiguousarray(np.broadcast_to(np.asarray(stream_ids, dtype=np.int32), ids.shape))
        starts = np.ascontiguousarray(np.broadcast_to(np.asarray(start_us, dtype=np.int64), ids.shape))
        self._check(self.b.fn["reset_episode"](self.h, _ptr(ids, _P32), len(ids), _ptr(sids, _P32), _ptr(starts, _P64)), "reset_episode")

    def forward(self, until_us, orders=None, env_ids=None, update_features: bool = False):
        """orders: None, or one [k_i, 6] array per env (rows {type, side, price, volume (-1 none), id, is_external}).  Returns the
        features' values [n, obs_dim] when update_features, else None."""
        ids = np.arange(self.n_envs, dtype=np.int32) if env_ids is None else np.ascontiguousarray(env_ids, dtype=np.int32)
        until = np.ascontiguousarray(np.broadcast_to(np.asarray(until_us, dtype=np.int64), ids.shape))
        rows = n_ord = None
        max_orders = 0
        if orders is not None:
            per_env = [np.ascontiguousarray(o, dtype=np.int64).reshape(-1, 6) for o in orders]
            assert len(per_env) == len(ids)
            max_orders = max(1, max(len(o) for o in per_env))
            rows = np.zeros((len(ids), max_orders, 6), dtype=np.int64)
            n_ord = np.array([len(o) for o in per_env], dtype=np.int32)
            for i, o in enumerate(per_env):
                rows[i, : len(o)] = o
        obs = np.zeros((len(ids), self.obs_dim), dtype=np.float64) if update_features else None
        self._check(self.b.fn["forward"](self.h, _ptr(ids, _P32), len(ids), _ptr(until, _P64), _ptr(rows, _P64), _ptr(n_ord, _P32), max_orders,
                                         FWD_UPDATE_FEATURES if update_features else 0, _ptr(obs, _PD)), "forward")
        return obs

    def fills(self, env: int = 0, cap: int = 8192) -> np.ndarray:
        """FilledOrders of the env's last forward: int64 rows {list (0 internal / 1 external), side, price, volume}."""
        out = np.zeros((cap, 4), dtype=np.int64)
        n = C.c_int32(0)
        self._check(self.b.fn["get_fills"](self.h, env, _ptr(out, _P64), cap, C.byref(n)), "get_fills")
        return out[: min(n.value, cap)]

    def features_reset(self, env_ids=None):
        ids = np.arange(self.n_envs, dtype=np.int32) if env_ids is None else np.ascontiguousarray(env_ids, dtype=np.int32)
        self._check(self.b.fn["features_reset"](self.h, _ptr(ids, _P32), len(ids)), "features_reset")

    def rollout_device(self, policy: int, params, T: int, d_actions: int, d_active: int, d_obs: int, d_obs_f32: int, d_reward: int, d_done: int,
                       d_return_acc: int = 0, d_steps_taken: int = 0):
        p8 = np.zeros(8, dtype=np.float64)
        p8[: len(params)] = params
        self._check(self.b.fn["rollout_device"](self.h, policy, _ptr(p8, _PD), T, d_actions, d_active, d_obs, d_obs_f32, d_reward, d_done, d_return_acc, d_steps_taken), "rollout_device")

    def sync(self):
        self._check(self.b.fn["sync"](self.h), "sync")

    def set_info_buffer(self, d_info: int):
        self._check(self.b.fn["set_info_buffer"](self.h, d_info), "set_info_buffer")

    def set_trace_buffer(self, d_trace: int, t_capacity: int):
        self._check(self.b.fn["set_trace_buffer"](self.h, d_trace, int(t_capacity)), "set_trace_buffer")

    def set_stream(self, hip_stream: int):
        self._check(self.b.fn["set_stream"](self.h, hip_stream), "set_stream")

    # ---- inspection ------------------------------------------------------------------------------------------
    def env_state(self, env: int) -> dict:
        st = EnvState()
        self._check(self.b.fn["get_env_state"](self.h, env, C.byref(st)), "get_env_state")
        return st.as_dict()

    def portfolios(self):
        inv = np.zeros(self.n_envs, dtype=np.int64)
        cash = np.zeros(self.n_envs, dtype=np.int64)
        price = np.zeros(self.n_envs, dtype=np.float64)
        self._check(self.b.fn["get_portfolios"](self.h, _ptr(inv, _P64), _ptr(cash, _P64), _ptr(price, _PD)), "get_portfolios")
        return inv, cash, price

    def best_prices(self):
        bb = np.zeros(self.n_envs, dtype=np.int64)
        bs = np.zeros(self.n_envs, dtype=np.int64)
        self._check(self.b.fn["get_best_prices"](self.h, _ptr(bb, _P64), _ptr(bs, _P64)), "get_best_prices")
        return bb, bs

    def book(self, env: int, side: int, max_levels: int = 256):
        prices = np.zeros(max_levels, dtype=np.int64)
        vols = np.zeros(max_levels, dtype=np.int64)
        cnt = np.zeros(max_levels, dtype=np.int32)
        n = C.c_int32(0)
        self._check(self.b.fn["get_book"](self.h, env, side, _ptr(prices, _P64), _ptr(vols, _P64), _ptr(cnt, _P32), max_levels, C.byref(n)), "get_book")
        return prices[: n.value], vols[: n.value], cnt[: n.value]

    def level_orders(self, env: int, side: int, price: int, max_orders: int = 1024) -> np.ndarray:
        out = np.zeros(max_orders, dtype=ORDER_VIEW_DTYPE)
        n = C.c_int32(0)
        self._check(
            self.b.fn["get_level_orders"](self.h, env, side, int(price), out.ctypes.data_as(C.POINTER(OrderView)), max_orders, C.byref(n)),
            "get_level_orders",
        )
        return out[: n.value]

    def queues(self, env: int) -> np.ndarray:
        """Whole central book as int64 rows (side, price, kind, id, volume), best level first, FIFO inside - the
        format of oracle/gen_golden.py:dump_queues."""
        rows = []
        for side in (0, 1):
            prices, _, _ = self.book(env, side, 1024)
            for p in prices:
                for o in self.level_orders(env, side, int(p)):
                    rows.append((side, int(p), int(o["kind"]), int(o["id"]), int(o["volume"])))
        return np.array(rows, dtype=np.int64).reshape(-1, 5)

    def top(self, env: int, k: int) -> np.ndarray:
        out = np.zeros((2, k, 3), dtype=np.int64)
        for side in (0, 1):
            p, v, c = self.book(env, side, k)
            out[side, : len(p), 0], out[side, : len(p), 1], out[side, : len(p), 2] = p, v, c
        return out

    # ---- engine-level ----------------------------------------------------------------------------------------
    def debug_set_book(self, env: int, snapshot_row: Optional[np.ndarray], now_us: int = 0):
        row = None if snapshot_row is None else np.ascontiguousarray(snapshot_row, dtype=np.int64)
        if row is not None:
            assert row.size == 4 * self.cfg.n_levels
        self._check(self.b.fn["debug_set_book"](self.h, env, _ptr(row, _P64), int(now_us)), "debug_set_book")

    def debug_process_orders(self, env: int, orders) -> np.ndarray:
        orders = np.ascontiguousarray(orders, dtype=np.int64).reshape(-1, 6)
        fills = np.zeros(8, dtype=np.int64)
        self._check(self.b.fn["debug_process_orders"](self.h, env, _ptr(orders, _P64), orders.shape[0], _ptr(fills, _P64)), "debug_process_orders")
        return fills
