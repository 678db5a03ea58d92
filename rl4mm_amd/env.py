"""Host-side mirror of the reference's Gym interface for the LOB step, backed by the HIP library.

Same names, constructor arguments and return shapes as the reference so it drops into the existing RLlib loop:

* ``HistoricalOrderbookEnvironment``      single env, ``reset() -> obs``, ``step(a) -> (obs, reward, done, info)``
  (reference rl4mm/gym/HistoricalOrderbookEnvironment.py:51-178)
* ``HistoricalOrderbookVectorEnv``        the same for N envs per launch, RLlib ``VectorEnv`` shaped
  (``vector_reset / reset_at / vector_step``); this is what a rollout worker should hold
* feature / reward descriptor classes     same class names and kwargs as rl4mm/features/Features.py and
  rl4mm/rewards/RewardFunctions.py, but they only *describe* the computation: the arithmetic runs in the kernel
* ``env_creator(env_config)``             rl4mm/gym/utils.py:42-87 with the same env_config keys
* ``StreamCatalog``                       replaces ``HistoricalDatabase`` as the data source: packed (ticker, day)
  streams (rl4mm_amd/stream.py)

`gym` and `ray` are optional.  When they import, ``HistoricalOrderbookEnvironment`` IS a ``gym.Env`` and
``HistoricalOrderbookVectorEnv`` IS a ``ray.rllib.env.VectorEnv`` (RLlib's env check accepts nothing else), spaces are
``gym.spaces.Box`` and ``env_creator`` sets ``env.spec = EnvSpec(id, max_episode_steps)`` like the reference
(rl4mm/gym/utils.py:83-86).  Without them the bases are plain classes and spaces fall back to a minimal Box with the
attributes RLlib and the reference's agents read (low, high, shape, dtype, sample, seed).
"""
from __future__ import annotations

from dataclasses import dataclass
from datetime import datetime, timedelta
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import pandas as pd

from . import _capi, _lib
from .stream import PackedStream

try:  # pragma: no cover - gym is not installed in the build image
    from gym.spaces import Box  # type: ignore
except Exception:  # noqa: BLE001

    class Box:  # minimal stand-in
        def __init__(self, low, high, shape=None, dtype=np.float32):
            if shape is None:
                shape = np.shape(low)
            self.low = np.broadcast_to(np.asarray(low, dtype=dtype), shape).copy()
            self.high = np.broadcast_to(np.asarray(high, dtype=dtype), shape).copy()
            self.shape = tuple(shape)
            self.dtype = np.dtype(dtype)
            self._rng = np.random.default_rng()

        def seed(self, seed=None):
            self._rng = np.random.default_rng(seed)
            return [seed]

        def sample(self):
            return self._rng.uniform(self.low, self.high).astype(self.dtype)

        def contains(self, x):
            return bool(np.all(x >= self.low) and np.all(x <= self.high))


try:  # the bases RLlib checks for (ray.rllib.env.base_env convert_to_base_env / check_env accept gym.Env and VectorEnv only)
    import gym as _gym  # type: ignore

    _GymEnv = _gym.Env
except Exception:  # noqa: BLE001
    _gym = None
    _GymEnv = object
try:
    from ray.rllib.env.vector_env import VectorEnv as _RayVectorEnv  # type: ignore
except Exception:  # noqa: BLE001
    _RayVectorEnv = object


class _EnvSpec:
    """gym.envs.registration.EnvSpec's two fields the reference sets (rl4mm/gym/utils.py:83-86); used when gym is absent."""

    def __init__(self, id, max_episode_steps=None):  # noqa: A002 - gym's own argument name
        self.id, self.max_episode_steps = id, max_episode_steps

    def __repr__(self):
        return f"EnvSpec({self.id})"


def make_env_spec(episode_length: timedelta, step_size: timedelta):
    steps = episode_length / step_size  # a float, like the reference's `episode_length / timedelta(seconds=step_size)`
    if _gym is not None:
        try:
            from gym.envs.registration import EnvSpec  # type: ignore

            return EnvSpec(id="HistoricalOrderbookEnvironment-v0", max_episode_steps=steps)
        except Exception:  # noqa: BLE001
            pass
    return _EnvSpec(id="HistoricalOrderbookEnvironment-v0", max_episode_steps=steps)


US = timedelta(microseconds=1)


def _us(td: timedelta) -> int:
    return int(td / US)


# ---------------------------------------------------------------------------------------------------------------
# feature / reward descriptors (rl4mm/features/Features.py, rl4mm/rewards/RewardFunctions.py)
# ---------------------------------------------------------------------------------------------------------------
@dataclass
class Portfolio:  # Features.py:24-27
    inventory: int
    cash: float


class Feature:
    kind = 0

    def __init__(self, name, min_value, max_value, update_frequency, lookback_periods, normalisation_on=False, max_norm_len=10000):
        assert update_frequency <= timedelta(minutes=1), "HFT update frequency must be less than 1 minute."  # Features.py:53
        self.name, self.min_value, self.max_value = name, min_value, max_value
        self.update_frequency, self.lookback_periods = update_frequency, lookback_periods
        self.normalisation_on, self.max_norm_len = normalisation_on, max_norm_len  # Features.py:56-61

    @property
    def window_size(self) -> timedelta:
        return self.lookback_periods * self.update_frequency

    def desc(self) -> _capi.FeatureDesc:
        return _capi.make_feature(self.kind, self.lookback_periods, _us(self.update_frequency), self.min_value, self.max_value,
                                  norm_len=self.max_norm_len if self.normalisation_on else 0)

    def desc_for(self, horizon_us: int) -> _capi.FeatureDesc:
        """desc() with the normalisation history sized to what one reset-to-terminal stretch (warm-up + episode) can
        append: the reference clears the deque at every reset, so a longer max_norm_len is never reached and the
        device would only reserve memory for it."""
        d = self.desc()
        if d.norm_len:
            d.norm_len = min(d.norm_len, horizon_us // _us(self.update_frequency) + 4)
        return d


class Spread(Feature):  # Features.py:113
    kind = _capi.F_SPREAD

    def __init__(self, name="Spread", min_value=0, max_value=50 * 100, update_frequency=timedelta(seconds=0.1), normalisation_on=False, max_norm_len=10000):
        super().__init__(name, min_value, max_value, update_frequency, 0, normalisation_on, max_norm_len)


class BookImbalance(Feature):  # :132
    kind = _capi.F_BOOK_IMBALANCE

    def __init__(self, update_frequency=timedelta(seconds=0.1), normalisation_on=False, max_norm_len=10000):
        super().__init__("BookImbalance", -1, 1, update_frequency, 0, normalisation_on, max_norm_len)


class PriceMove(Feature):  # :151
    kind = _capi.F_PRICE_MOVE

    def __init__(self, name="MidpriceMove", min_value=-100 * 100, max_value=100 * 100, update_frequency=timedelta(seconds=1), lookback_periods=10, normalisation_on=False, max_norm_len=100_000):
        super().__init__(name, min_value, max_value, update_frequency, lookback_periods, normalisation_on, max_norm_len)


class PriceRange(Feature):  # :178
    kind = _capi.F_PRICE_RANGE

    def __init__(self, name="PriceRange", min_value=0, max_value=100 * 100, update_frequency=timedelta(seconds=1), lookback_periods=10, normalisation_on=False, max_norm_len=100_000):
        super().__init__(name, min_value, max_value, update_frequency, lookback_periods, normalisation_on, max_norm_len)


class Volatility(Feature):  # :203
    kind = _capi.F_VOLATILITY

    def __init__(self, name="Volatility", min_value=0, max_value=1.0, update_frequency=timedelta(seconds=1), lookback_periods=10, normalisation_on=False, max_norm_len=100_000):
        super().__init__(name, min_value, max_value, update_frequency, lookback_periods, normalisation_on, max_norm_len)


class Price(Feature):  # :327
    kind = _capi.F_PRICE

    def __init__(self, name="Price", min_value=0, max_value=10_000 * 10_000, update_frequency=timedelta(seconds=1), normalisation_on=False, max_norm_len=100_000):
        super().__init__(name, min_value, max_value, update_frequency, 0, normalisation_on, max_norm_len)


class AmihudLambda(Feature):  # :245
    kind = _capi.F_AMIHUD_LAMBDA

    def __init__(self, name="AmihudLambda", min_value=0, max_value=1.0, update_frequency=timedelta(seconds=0.1), lookback_periods=10, slowing_factor=10, normalisation_on=False, max_norm_len=100_000):
        super().__init__(name, min_value, max_value, update_frequency, (lookback_periods + 1) * slowing_factor, normalisation_on, max_norm_len)
        self.true_lookback_periods, self.slowing_factor = lookback_periods, slowing_factor

    def desc(self):
        d = super().desc()
        d.iparam = int(self.slowing_factor)
        return d


class TradeDirectionImbalance(Feature):  # :351
    kind = _capi.F_TRADE_DIRECTION_IMBALANCE

    def __init__(self, name="TradeImbalance", update_frequency=timedelta(seconds=0.1), lookback_periods=600, track_internal=False, normalisation_on=False, max_norm_len=100_000):
        super().__init__(name, -1.0, 1.0, update_frequency, lookback_periods, normalisation_on, max_norm_len)
        self.track_internal = track_internal  # Features.py:379-381, 438-440: the agent's own fills count as well

    def desc(self):
        d = super().desc()
        d.iparam = int(bool(self.track_internal))
        return d


class TradeVolumeImbalance(TradeDirectionImbalance):  # :410
    kind = _capi.F_TRADE_VOLUME_IMBALANCE

    def __init__(self, name="TradeVolumeImbalance", **kw):
        super().__init__(name, **kw)


class Inventory(Feature):  # :474
    kind = _capi.F_INVENTORY

    def __init__(self, name="Inventory", min_value=-1000000, max_value=1000000, update_frequency=timedelta(seconds=0.1), normalisation_on=False, max_norm_len=100_000):
        super().__init__(name, min_value, max_value, update_frequency, 0, normalisation_on, max_norm_len)


class EpisodeProportion(Feature):  # :493
    kind = _capi.F_EPISODE_PROPORTION

    def __init__(self, name="EpisodeProportion", update_frequency=timedelta(seconds=0.1), episode_length=timedelta(minutes=60), normalisation_on=False, max_norm_len=100_000):
        super().__init__(name, 0.0, 1.0, update_frequency, 0, normalisation_on, max_norm_len)
        self.episode_length = episode_length
        self.step_size = update_frequency / episode_length  # :505

    def desc(self):
        d = super().desc()
        d.dparam = float(self.step_size)
        return d


class TimeOfDay(Feature):  # :515
    kind = _capi.F_TIME_OF_DAY

    def __init__(self, name="TimeOfDay", update_frequency=timedelta(minutes=1), normalisation_on=False, max_norm_len=100_000, n_buckets=10):
        super().__init__(name, 0, n_buckets - 1, update_frequency, 0, normalisation_on, max_norm_len)
        self.n_buckets = n_buckets
        self.bucket_size = (timedelta(hours=15, minutes=30) - timedelta(hours=10)) / n_buckets  # :526-528

    def desc(self):
        d = super().desc()
        d.iparam = _us(self.bucket_size)
        return d


class RewardFunction:
    def desc(self) -> Tuple[int, int, float]:
        raise NotImplementedError


class PnL(RewardFunction):  # RewardFunctions.py:97
    def desc(self):
        return (_capi.R_PNL, 0, 0.0)


class InventoryAdjustedPnL(RewardFunction):  # :107
    def __init__(self, inventory_aversion: float, asymmetrically_dampened: bool = False):
        self.inventory_aversion, self.asymmetrically_dampened = inventory_aversion, asymmetrically_dampened

    def desc(self):
        return (_capi.R_INVENTORY_ADJUSTED, int(self.asymmetrically_dampened), float(self.inventory_aversion))


class RollingSharpe(RewardFunction):  # RewardFunctions.py:38
    def __init__(self, max_window_size: int = 120, min_window_size: int = 60):
        assert max_window_size >= min_window_size, "Error with window sizes"
        assert max_window_size <= 128, "the device path holds at most 128 AUM values per window"
        self.max_window_size, self.min_window_size = max_window_size, min_window_size

    def desc(self):
        return (_capi.R_ROLLING_SHARPE, 0, 0.0, self.max_window_size, self.min_window_size)


def get_reward_function(reward_function: str, inventory_aversion: float = 0.1) -> RewardFunction:  # rl4mm/gym/utils.py:29-39
    if reward_function == "AD":
        return InventoryAdjustedPnL(inventory_aversion=inventory_aversion, asymmetrically_dampened=True)
    if reward_function == "SD":
        return InventoryAdjustedPnL(inventory_aversion=inventory_aversion, asymmetrically_dampened=False)
    if reward_function == "PnL":
        return PnL()
    if reward_function == "RS":
        return RollingSharpe()
    raise NotImplementedError("You must specify one of 'AS', 'SD', 'PnL', or 'RS'")


def get_default_features(step_size: timedelta, episode_length: timedelta, normalisation_on: bool = False) -> List[Feature]:
    """HistoricalOrderbookEnvironment.get_default_features (HistoricalOrderbookEnvironment.py:395-440)."""
    assert step_size <= timedelta(seconds=0.1), "Default features require a minimum step size of 0.1 seconds."
    n = normalisation_on
    return [
        Spread(update_frequency=step_size, normalisation_on=n),
        PriceMove(name="price_move_0.1_s", update_frequency=timedelta(seconds=0.1), lookback_periods=1, normalisation_on=n),
        PriceMove(name="price_move_10_s", update_frequency=timedelta(seconds=1), lookback_periods=10, normalisation_on=n),
        Volatility(name="volatility_1_min", update_frequency=timedelta(seconds=0.1), lookback_periods=600, normalisation_on=n),
        Volatility(name="volatility_5_min", update_frequency=timedelta(seconds=1), lookback_periods=300, normalisation_on=n),
        Inventory(update_frequency=step_size, normalisation_on=n),
        EpisodeProportion(update_frequency=step_size, episode_length=episode_length, normalisation_on=n),
        TimeOfDay(n_buckets=10, normalisation_on=n),
        TradeDirectionImbalance(update_frequency=timedelta(seconds=0.1), lookback_periods=600, normalisation_on=n),
        TradeVolumeImbalance(update_frequency=timedelta(seconds=0.1), lookback_periods=600, normalisation_on=n),
    ]


def beta_volumes(actions: np.ndarray, quote_levels: int = 10, concentration: Optional[float] = None, active_volume: int = 100):
    """BetaOrderDistributor.convert_action (rl4mm/gym/action_interpretation/OrderDistributors.py:23-56) for a batch of
    actions [N, A] -> (buy [N, Q], sell [N, Q]) integer volumes; same closed form as the kernel (x^(a-1) (1-x)^(b-1),
    the 1/B(a,b) of the pdf cancels in the normalisation), checked against the scipy table in tests/golden."""
    a = np.asarray(actions, dtype=np.float64) + 0.000001
    if concentration is not None:
        ab, bb, as_, bs = a[:, 0], concentration - a[:, 0] + 0.000001, a[:, 1], concentration - a[:, 1] + 0.000001
    else:
        ab, bb, as_, bs = a[:, 0], a[:, 1], a[:, 2], a[:, 3]
    x = 1 / quote_levels * np.array([i + 0.5 for i in range(quote_levels)])
    out = []
    for al, be in ((ab, bb), (as_, bs)):
        with np.errstate(all="ignore"):
            pdf = np.power(x[None, :], al[:, None] - 1.0) * np.power(1.0 - x[None, :], be[:, None] - 1.0)
            tot = np.array([row.sum() for row in pdf])  # per-row np.sum = the reference's summation order
            q = pdf / tot[:, None] * active_volume
        # where the reference's distribution is NaN (scipy's pdf of a shape parameter <= 0: concentration mode, an action at or above
        # the concentration; a pdf that sums to nothing) its volumes are np.round(nan).astype(int), which no caller can use - the step
        # raises unless it clears the inventory, and the info calculator's argmax over their signs is 0: the kernels and this function
        # say "no volume at any level" there
        nan = ~((tot > 0) & np.isfinite(tot)) | ((be <= 0) if concentration is not None else np.zeros(len(tot), dtype=bool))
        out.append(np.where(nan[:, None] | ~np.isfinite(q), 0, np.round(np.where(np.isfinite(q), q, 0.0))).astype(np.int64))
    return out[0], out[1]


class SimpleInfoCalculator:
    """rl4mm/gym/order_tracking/InfoCalculators.py:17-89, batched: the `info` dict of every env from bulk device reads
    (price, inventory, cash, best prices) and the actions.  Host side by design: it is evaluation tooling, not the
    rollout hot path (the reference default is info_calculator=None)."""

    def __init__(self, market_order_fraction_of_inventory: float = 0.0, enter_spread: bool = False, order_distributor=None, concentration: float = None,
                 quote_levels: int = 10, active_volume: int = 100):
        self.market_order_fraction_of_inventory = market_order_fraction_of_inventory
        self.enter_spread, self.concentration = enter_spread, concentration
        self.quote_levels, self.active_volume = quote_levels, active_volume  # BetaOrderDistributor defaults (:29)
        self.market_order_count = None
        self.market_order_total_volume = None
        self._device_rows = None

    def attach(self, sim):
        """Have the step kernel compute the info arrays in its epilogue (lobsim_set_info_buffer): calculate_arrays then
        costs one device-to-host copy instead of bulk getters plus a host-side Beta evaluation."""
        import torch

        # The kernel works the info out from the ENV's configuration (its quote levels, enter_spread, concentration, clearing
        # fraction); the reference's calculator carries its own copies of those (InfoCalculators.py:17-29).  Only when they
        # say the same do the device rows equal what this object would compute on the host: otherwise stay on the host path.
        c = sim.cfg
        same = (self.quote_levels == c.max_quote_level - c.min_quote_level and c.min_quote_level == 0 and self.active_volume == 100
                and bool(self.enter_spread) == bool(c.enter_spread)
                and (self.concentration is None) == (not c.use_concentration) and (self.concentration is None or float(self.concentration) == c.concentration)
                and (not c.market_order_clearing or float(self.market_order_fraction_of_inventory) == c.market_order_fraction_of_inventory))
        if not same:
            self._device_rows = None
            return self
        self._device_rows = torch.zeros((sim.n_envs, _capi.INFO_DIM), dtype=torch.float64, device=torch.device("cuda", sim.cfg.device_id))
        torch.cuda.synchronize()
        sim.set_info_buffer(self._device_rows.data_ptr())
        return self

    def calculate_arrays(self, sim, actions: np.ndarray) -> Dict[str, np.ndarray]:
        if self._device_rows is not None:
            sim.sync()
            rows = self._device_rows.cpu().numpy()
            out = {name: rows[:, i].copy() for i, name in enumerate(_capi.INFO_COLS)}
            for k in ("inventory", "cash", "market_order_count"):
                out[k] = out[k].astype(np.int64)
            out["market_spread"] = out["market_spread"].astype(np.int64) if np.all(np.isfinite(out["market_spread"])) else out["market_spread"]
            return out
        actions = np.asarray(actions, dtype=np.float64)
        n = actions.shape[0]
        if self.market_order_count is None:
            self.market_order_count = np.zeros(n, dtype=np.int64)
            self.market_order_total_volume = np.zeros(n, dtype=np.float64)
        inv, cash, price = sim.portfolios()
        bb, bs = sim.best_prices()
        spread = (bs - bb).astype(np.int64)
        buy, sell = beta_volumes(actions, self.quote_levels, self.concentration, self.active_volume)
        levels = np.arange(self.quote_levels)
        best_buy, best_sell = np.sign(buy).argmax(axis=1), np.sign(sell).argmax(axis=1)  # :66-67
        midprice_offset = (best_sell - best_buy) / 2
        agent_spread = (best_buy + best_sell).astype(np.float64)
        buy_com, sell_com = buy @ levels / self.active_volume, sell @ levels / self.active_volume
        weighted_offset = (sell_com - buy_com) / 2
        weighted_spread = buy_com + sell_com
        if not self.enter_spread:
            agent_spread = agent_spread + spread
            weighted_spread = weighted_spread + spread
        A = actions.shape[1]
        if A in (3, 5):  # :52-56, with the post-step inventory like the reference
            # (an env whose step raised never reaches the calculator in the reference: HistoricalOrderbookEnvironment.py:163-178)
            hit = (np.abs(inv) > actions[:, -1]) & (np.asarray(sim.status()) == 0)
            self.market_order_count = self.market_order_count + hit
            self.market_order_total_volume = self.market_order_total_volume + np.where(hit, np.round(np.abs(inv) * self.market_order_fraction_of_inventory), 0.0)
        return dict(
            asset_price=price, inventory=inv, cash=cash, aum=cash + price * inv, market_spread=spread, agent_spread=agent_spread,
            agent_weighted_spread_value=weighted_spread, midprice_offset=midprice_offset, weighted_midprice_offset=weighted_offset,
            market_order_count=self.market_order_count.copy(), market_order_total_volume=self.market_order_total_volume.copy(),
        )

    def calculate_dicts(self, sim, actions: np.ndarray) -> List[dict]:
        """One dict per env with exactly the reference's keys (including its literal agent_weighted_spread=[1])."""
        arr = self.calculate_arrays(sim, actions)
        actions = np.asarray(actions, dtype=np.float64)
        A = actions.shape[1]
        out = []
        for i in range(actions.shape[0]):
            d = dict(
                asset_price=float(arr["asset_price"][i]), inventory=int(arr["inventory"][i]), cash=int(arr["cash"][i]), aum=float(arr["aum"][i]),
                market_spread=int(arr["market_spread"][i]), agent_spread=arr["agent_spread"][i], agent_weighted_spread=[1],
                midprice_offset=arr["midprice_offset"][i], weighted_midprice_offset=arr["weighted_midprice_offset"][i],
            )
            if A in (2, 3):
                d["bid_action"], d["ask_action"] = (actions[i][[0]],), (actions[i][[1]],)
            else:
                d["bid_action"], d["ask_action"] = (actions[i][[0, 1]],), (actions[i][[2, 3]],)
            if A in (3, 5):
                d["market_order_action"] = (actions[i][[-1]],)
            d["market_order_count"] = int(arr["market_order_count"][i])
            d["market_order_total_volume"] = arr["market_order_total_volume"][i]
            out.append(d)
        return out


# ---------------------------------------------------------------------------------------------------------------
# data source
# ---------------------------------------------------------------------------------------------------------------
class StreamCatalog:
    """(ticker, 'YYYY-MM-DD') -> PackedStream.  Stands where ``HistoricalDatabase`` stood."""

    def __init__(self):
        self._streams: Dict[Tuple[str, str], PackedStream] = {}

    def add(self, stream: PackedStream) -> "StreamCatalog":
        self._streams[(stream.ticker, stream.trading_date)] = stream
        return self

    def get(self, ticker: str, date: str) -> PackedStream:
        try:
            return self._streams[(ticker, date)]
        except KeyError:
            raise KeyError(f"no packed stream for {ticker} on {date}; loaded: {sorted(self._streams)}") from None

    def items(self):
        return self._streams.items()

    def __len__(self):
        return len(self._streams)


class _StreamResidency:
    """Which (ticker, day) streams are on the device.  The reference reads a day from its database when an episode starts on it and
    holds none of them (rl4mm/database/HistoricalDatabase.py:46-62,103-135); a two-week ``min_date..max_date`` range at ~150 MB per
    ticker-day does not have to sit in HBM either: a stream is uploaded when a start draw first selects its day, and when the device
    copies exceed ``budget_bytes`` the least recently drawn days that no env is replaying are dropped (``lobsim_drop_stream``; an env
    keeps its stream until it is reset onto another one).  budget 0 = no limit (days are still uploaded on first use)."""

    def __init__(self, sim, catalog: StreamCatalog, budget_bytes: int = 0):
        self.sim, self.catalog, self.budget = sim, catalog, int(budget_bytes)
        self.sid: Dict[Tuple[str, str], int] = {}     # resident streams
        self.bytes: Dict[Tuple[str, str], int] = {}
        self.last_use: Dict[Tuple[str, str], int] = {}
        self.clock = 0
        self.uploads = self.drops = 0

    def has(self, key) -> bool:
        try:
            self.catalog.get(*key)
            return True
        except KeyError:
            return False

    def ensure(self, keys) -> Dict[Tuple[str, str], int]:
        """stream ids of `keys`, uploading the ones that are not on the device (the budget is enforced afterwards: trim)"""
        self.clock += 1
        for key in keys:
            if key not in self.sid:
                stream = self.catalog.get(*key)
                self.sid[key] = self.sim.add_stream(stream)
                self.bytes[key] = int(stream.msgs.nbytes + stream.snaps.nbytes)
                self.uploads += 1
            self.last_use[key] = self.clock
        return self.sid

    def trim(self, in_use) -> None:
        """drop least-recently-used streams that no env replays until the device copies fit the budget"""
        if self.budget <= 0:
            return
        total = sum(self.bytes.values())
        for key in sorted(self.sid, key=lambda k: self.last_use[k]):
            if total <= self.budget:
                break
            if key in in_use:
                continue
            try:  # (the bookkeeping follows the device: a stream the library refuses to drop - an env's last reset still names it - stays tracked)
                self.sim.drop_stream(self.sid[key])
            except _capi.LobsimError:
                continue
            del self.sid[key]
            total -= self.bytes.pop(key)
            del self.last_use[key]
            self.drops += 1

    @property
    def resident_bytes(self) -> int:
        return sum(self.bytes.values())


def default_device_id(env_config=None) -> int:
    """The GPU a rollout worker's envs go to when the caller names none: RLlib hands ``env_creator`` an EnvContext whose
    ``worker_index`` is 0 for the local worker and 1 .. num_workers for the remote ones (rl4mm/helpers/main_helper.py:262-263 sets
    ``num_workers``): worker w steps its envs on device (w - 1) mod the number of visible GPUs; a process started by torchrun uses its
    LOCAL_RANK; otherwise device 0."""
    import os

    n_dev = 0
    try:
        import torch

        n_dev = torch.cuda.device_count()
    except Exception:  # noqa: BLE001
        pass
    n_dev = max(1, n_dev)
    w = getattr(env_config, "worker_index", None)
    if w is None and isinstance(env_config, dict):
        w = env_config.get("worker_index")
    if w:
        return (int(w) - 1) % n_dev
    if os.environ.get("LOCAL_RANK"):
        return int(os.environ["LOCAL_RANK"]) % n_dev
    return 0


# ---------------------------------------------------------------------------------------------------------------
# environments
# ---------------------------------------------------------------------------------------------------------------
class HistoricalOrderbookVectorEnv(_RayVectorEnv):
    """N independent copies of the reference env stepped by one kernel launch: a ``ray.rllib.env.VectorEnv`` when ray imports.

    ``vector_reset / reset_at / vector_step`` have RLlib's list-of-per-env-values shape.  A caller that wants the batch
    as it comes off the device uses the ARRAY path: ``vector_reset_arrays() -> obs[N, D]``,
    ``vector_step_arrays(actions[N, A]) -> (obs[N, D], reward[N], done[N] bool, infos or None)`` - no per-env Python objects
    (building 4 x 4096 list entries costs 35 x the kernel); the kernel reads the actions from and writes its outputs to pinned
    host memory itself (``lobsim_step_host_io``).  A learner that lives on the GPU - the reference's consumer is a torch PPO
    learner (rl4mm/helpers/main_helper.py:260-261) - uses the TENSOR path: ``vector_reset_tensors() -> obs``,
    ``vector_step_tensors(actions) -> (obs, reward, done)`` with torch tensors on the env's device, launched on torch's
    current stream, nothing crossing PCIe and no host synchronisation per step.

    Errors.  Where the reference raises inside ``step`` (``EmptyOrderbookError``, an assert: rl4mm/orderbook/Exchange.py:32-37,
    59-60,180-186) or where a capacity of this implementation is exceeded, the env stops: it reports ``done=True`` once, with
    ``infos[i]["error"]`` (list path) naming the condition, and ``env.status`` (int32[N], bits of ``enum lobsim_status``) says
    which envs stopped and why on every path.  ``env.limits`` holds the capacities of the handle (``lobsim_get_limits``).

    ``reset_at``.  RLlib resets finished envs one ``reset_at(i)`` call at a time, and every env of a batch ends its episode on
    the same step.  A reset replays the features' warm-up (3000 steps for the default features) - tens of milliseconds of
    latency for one env, but the same tens of milliseconds for 4096 of them in one launch.  So the first ``reset_at`` after a
    step that reported dones resets ALL finished envs in one launch, drawing their start times in index order (the order in
    which RLlib's loop calls ``reset_at``, so ``np.random`` ends where the one-at-a-time calls leave it), and the calls
    for the other finished envs return their cached rows."""

    def __init__(
        self,
        num_envs: int,
        catalog: StreamCatalog,
        features: List[Feature] = None,
        max_distribution_param: float = 10.0,
        ticker="MSFT",
        step_size: timedelta = timedelta(seconds=0.1),
        episode_length: timedelta = timedelta(minutes=30),
        initial_portfolio: Portfolio = None,
        min_quote_level: int = 0,
        max_quote_level: int = 10,
        min_date: datetime = datetime(2019, 1, 2),
        max_date: datetime = datetime(2019, 1, 2),
        min_start_timedelta: timedelta = timedelta(hours=10),
        max_end_timedelta: timedelta = timedelta(hours=15, minutes=30),
        market_order_clearing: bool = False,
        inc_prev_action_in_obs: bool = False,
        max_inventory: int = 100000,
        per_step_reward_function: RewardFunction = None,
        terminal_reward_function: RewardFunction = None,
        info_calculator=None,
        concentration: Optional[float] = None,
        market_order_fraction_of_inventory: float = 0.0,
        enter_spread: bool = False,
        n_levels: int = 50,
        outer_levels: int = 20,
        device_id: int = 0,
        ladder_levels: int = 0,
        max_levels: int = 0,
        obs_dtype=np.float64,
        agent_orders: int = 0,
        stream_budget_mb: float = 0.0,
        queue_pages: int = 0,
    ):
        # action space, HistoricalOrderbookEnvironment.py:85-94
        if concentration is not None:
            assert concentration >= max_distribution_param, "Concentration is less than max_distribution_param."
            self.action_space = Box(low=0.0, high=concentration, shape=(2,), dtype=np.float64)
        else:
            self.action_space = Box(low=0.0, high=max_distribution_param, shape=(4,), dtype=np.float64)
        if market_order_clearing:
            self.action_space = Box(low=np.append(self.action_space.low, [0.0]), high=np.append(self.action_space.high, [max_inventory]), dtype=np.float64)
        assert episode_length % step_size == timedelta(0), "Episode length must be a multiple of step size!"  # :100
        assert min_start_timedelta + episode_length <= max_end_timedelta, "Episode is too long"  # :372
        if (market_order_clearing and market_order_fraction_of_inventory <= 0.0) or (not market_order_clearing and market_order_fraction_of_inventory > 0.0):
            raise Exception("market_order_fraction_of_inventory must be positive if and only if market order clearing is on")  # :376-384
        self.info_calculator = info_calculator  # None (the reference default, main_helper.py:233) -> info == {}
        self.num_envs = num_envs
        self.catalog = catalog
        # `ticker` may be a list: env i then replays tickers[i mod T] - a batch of the reference's per-ticker envs
        # (HistoricalOrderbookEnvironment.py:58,96,136-140) over several tickers, BASELINE configs[4]'s "mixed tickers".  Every env draws
        # its own trading day and start time (global np.random, in env order, like one reference env after the other)
        self.tickers = [ticker] if isinstance(ticker, str) else [str(t) for t in ticker]
        assert len(self.tickers) >= 1 and len(set(self.tickers)) == len(self.tickers), "ticker: one name or a list of distinct names"
        self.ticker, self.step_size, self.episode_length = self.tickers[0] if len(self.tickers) == 1 else tuple(self.tickers), step_size, episode_length
        self._env_ticker = np.arange(num_envs, dtype=np.int64) % len(self.tickers)  # ticker index of every env
        self.min_date, self.max_date = min_date, max_date
        self.min_start_timedelta, self.max_end_timedelta = min_start_timedelta, max_end_timedelta
        self.initial_portfolio = initial_portfolio or Portfolio(inventory=0, cash=1000)  # :104
        self.features = features or get_default_features(step_size, episode_length)
        self.inc_prev_action_in_obs = inc_prev_action_in_obs
        self.per_step_reward_function = per_step_reward_function or InventoryAdjustedPnL(inventory_aversion=10 ** (-4))  # :72
        self.terminal_reward_function = terminal_reward_function or InventoryAdjustedPnL(inventory_aversion=0.1)  # :73
        low = np.array([np.float32(f.min_value) for f in self.features])
        high = np.array([np.float32(f.max_value) for f in self.features])
        if inc_prev_action_in_obs:
            low = np.concatenate((low, np.float32(self.action_space.low)))
            high = np.concatenate((high, np.float32(self.action_space.high)))
        self.observation_space = Box(low=low, high=high, dtype=np.float32)  # :125-135
        self.max_feature_window_size = max(f.window_size for f in self.features)
        cfg = _capi.make_cfg(
            num_envs, n_levels, [f.desc_for(_us(self.max_feature_window_size) + _us(episode_length)) for f in self.features], step_us=_us(step_size), episode_us=_us(episode_length),
            warmup_us=_us(self.max_feature_window_size), outer_levels=outer_levels, min_quote_level=min_quote_level, max_quote_level=max_quote_level,
            concentration=concentration, market_order_clearing=market_order_clearing, market_order_fraction_of_inventory=market_order_fraction_of_inventory,
            enter_spread=enter_spread, inc_prev_action_in_obs=inc_prev_action_in_obs, per_step_reward=self.per_step_reward_function.desc(),
            terminal_reward=self.terminal_reward_function.desc(), device_id=device_id,
            ladder_levels=ladder_levels,  # price levels per side kept ON CHIP (0 = 64 for books of <= 56 visible levels, else 128): a performance knob
            max_levels=max_levels,  # an env's budget of price levels (0 = 512: 2 x 512 levels over both sides, ~2.7 MiB of device memory per env): the
                                    # reference keeps stale levels that left LOBSTER's visible range for the rest of the episode - one per tick of
                                    # price drift; beyond the budget an env stops with LOBSIM_ST_LEVEL_OVERFLOW (env.limits tells what the handle has)
            agent_orders=agent_orders,  # the env's table of live agent orders: 0 = 256, which cannot overflow (include/lobsim.h); 128 = smaller, a little faster
            queue_pages=queue_pages,  # KiB of order-queue memory per env (0 = two pages per level of the budget: 2304 at the default; up to 4352 for thick AND drifting books)
        )  # fmt: skip
        # observations as the reference returns them are float64 (np.array of Python floats, HistoricalOrderbookEnvironment.py:189-194); its
        # observation_space is float32 (:131-135), which is what RLlib casts to anyway: obs_dtype=np.float32 has the kernel write that
        self.obs_dtype = np.dtype(obs_dtype)
        assert self.obs_dtype in (np.dtype(np.float64), np.dtype(np.float32)), "obs_dtype is float64 (the reference's arrays) or float32 (its observation_space)"
        if _RayVectorEnv is not object:
            super().__init__(self.observation_space, self.action_space, num_envs)
        self.sim = _lib.new_sim(cfg)  # raises if the HIP library or the GPU is missing: there is no CPU path
        # streams go to the device when a start draw first selects their day (and leave it again under stream_budget_mb: _StreamResidency)
        self._residency = _StreamResidency(self.sim, catalog, int(stream_budget_mb * 2**20))
        self._env_key: List[Optional[Tuple[str, str]]] = [None] * num_envs  # the (ticker, day) every env replays
        self.last_draw = None  # (env ids, [(ticker, day)], start_us) of the latest reset: what a test needs to mirror it on another simulator
        cash = self.initial_portfolio.cash
        assert float(cash) == int(cash), "cash must be integer valued (it is kept exact as int64)"
        self.sim.set_portfolio(int(self.initial_portfolio.inventory), int(cash))
        self._dones = np.zeros(num_envs, dtype=bool)
        self._any_done = False
        self._day_table = None
        self.limits = self.sim.limits()  # capacities of this handle (lobsim_get_limits): level_slabs, queue_slots, agent_orders, ...
        self.status = np.zeros(num_envs, dtype=np.int32)  # enum lobsim_status bits of the envs that stopped on an error (refreshed when a step reports done)
        self._io = None          # pinned step buffers (array path), mapped on first use
        self._out_pool = []      # output arrays of vector_step_arrays(copy=True), reused once the caller has dropped them
        self._reset_cache = {}   # env index -> first observation of a reset made ahead of its reset_at call
        self._t = None           # tensor path state

    def __repr__(self):
        lim = self.limits
        return (f"{type(self).__name__}(num_envs={self.num_envs}, ticker={self.ticker!r}, step_size={self.step_size}, obs_dtype={self.obs_dtype.name}, "
                f"limits: {lim['level_slabs']} price levels per env (both sides; {lim['ladder_levels']} per side on chip), {lim['queue_slots']} orders per level, "
                f"{lim['agent_orders']} agent orders, {lim['bytes_per_env'] / 2**20:.1f} MiB per env; streams on the device: {len(self._residency.sid)}, {self._residency.resident_bytes / 2**20:.0f} MiB)")

    # ---- start-time draw: same np.random calls, in the same order, as the reference (:196, :333-351) -------------
    def _random_offset_timestamp(self) -> timedelta:
        max_offset_steps = int((self.max_end_timedelta - self.episode_length - self.min_start_timedelta) / self.step_size)
        try:
            random_offset_steps = np.random.randint(low=0, high=max_offset_steps)
        except ValueError:
            random_offset_steps = 0
        ts = self.min_start_timedelta + random_offset_steps * self.step_size
        return ts - timedelta(microseconds=ts.microseconds)

    def _get_random_trading_day(self) -> datetime:
        from .trading_calendar import get_next_trading_dt

        trading_dates = pd.bdate_range(self.min_date, self.max_date)
        d = get_next_trading_dt(pd.to_datetime(np.random.choice(trading_dates)).to_pydatetime())  # rl4mm/utils/utils.py:54-57 (NASDAQ sessions)
        return datetime.combine(d.date(), datetime.min.time())

    def _get_random_start_time(self) -> datetime:
        return self._get_random_trading_day() + self._random_offset_timestamp()

    def _draw_slow(self, n: int, env_ids=None):
        """One env at a time through the reference's own call sequence (the definition the batched draw is tested against).
        -> (day index into pd.bdate_range(min_date, max_date), start_us), like _draw_days."""
        days = pd.bdate_range(self.min_date, self.max_date)
        pos = {pd.Timestamp(d): i for i, d in enumerate(days)}
        day, starts = np.zeros(n, dtype=np.int64), np.zeros(n, dtype=np.int64)
        for i in range(n):
            # _get_random_start_time (:196,333-351), the business day kept as an index: the NASDAQ session it maps to is looked up per ticker
            d = pd.to_datetime(np.random.choice(days))
            off = self._random_offset_timestamp()
            day[i] = pos[pd.Timestamp(d)]
            starts[i] = _us(off)
        return day, starts

    def _day_lookup(self):
        """Per candidate business day (pd.bdate_range(min_date, max_date), what np.random.choice draws from): the date string of the
        NASDAQ session it maps to (rl4mm/utils/utils.py:54-57)."""
        if self._day_table is None:
            from .trading_calendar import get_next_trading_dt

            self._day_table = [get_next_trading_dt(pd.to_datetime(d).to_pydatetime()).strftime("%Y-%m-%d") for d in pd.bdate_range(self.min_date, self.max_date)]
        return self._day_table

    def _draw_days(self, n: int):
        """Start day (index into the business days of [min_date, max_date]) and start second of n episodes with the global
        ``np.random`` state advanced exactly as n calls of the reference's ``_get_random_start_time`` advance it
        (HistoricalOrderbookEnvironment.py:196,333-351): per env one ``np.random.choice`` over the business days, then one
        ``np.random.randint`` over the start offsets.  Legacy ``RandomState.randint(0, m)`` draws 32-bit words, masks them to the next
        power of two and rejects values >= m (no word at all when m == 1): the words of all n envs are pulled in one call and the
        accept / reject walk is replayed here, then the generator is wound to the word the last env stopped at.  ~4 ms for 4096 envs
        instead of 0.5 s."""
        n_days = len(self._day_lookup())
        max_offset_steps = int((self.max_end_timedelta - self.episode_length - self.min_start_timedelta) / self.step_size)
        if n_days == 0:
            raise ValueError("'a' cannot be empty unless no samples are taken")  # np.random.choice of an empty range
        if n == 0:
            return np.zeros(0, dtype=np.int64), np.zeros(0, dtype=np.int64)
        if max(n_days, max_offset_steps) > 0xFFFFFFFF:
            return self._draw_slow(n)

        def mask_of(m):  # smallest 2^k - 1 >= m - 1
            r, k = m - 1, 0
            while (1 << k) - 1 < r:
                k += 1
            return (1 << k) - 1

        need_day, need_off = n_days > 1, max_offset_steps > 1  # (m <= 1: the value is 0 and no word is drawn; m <= 0 raises -> 0)
        rs = np.random.mtrand._rand  # the global legacy generator the reference uses
        state0 = rs.get_state()
        day = np.zeros(n, dtype=np.int64)
        off = np.zeros(n, dtype=np.int64)
        if need_day or need_off:
            md, mo = mask_of(n_days) if need_day else 0, mask_of(max_offset_steps) if need_off else 0
            # expected words per env, with head-room; more are pulled if the walk runs out
            per_env = (need_day * (md + 1) / max(n_days, 1)) + (need_off * (mo + 1) / max(max_offset_steps, 1))
            k_words = int(n * per_env * 1.25) + 64
            while True:
                rs.set_state(state0)
                words = np.frombuffer(rs.bytes(4 * k_words), dtype="<u4")
                vd = (words & md).tolist() if need_day else None
                vo = (words & mo).tolist() if need_off else None
                p, ok = 0, True
                dl, ol = [0] * n, [0] * n
                try:
                    for i in range(n):
                        if need_day:
                            while vd[p] >= n_days:
                                p += 1
                            dl[i] = vd[p]
                            p += 1
                        if need_off:
                            while vo[p] >= max_offset_steps:
                                p += 1
                            ol[i] = vo[p]
                            p += 1
                except IndexError:
                    ok = False
                if ok:
                    break
                k_words *= 2
            rs.set_state(state0)
            if p:
                rs.bytes(4 * p)  # exactly the words the n envs consumed
            day[:] = dl
            off[:] = ol
        start_us = _us(self.min_start_timedelta) + off * _us(self.step_size)
        return day, (start_us // 1_000_000) * 1_000_000  # "start episode on sec" (:341)

    def _draw(self, n: int, env_ids=None):
        """(stream ids, start_us) of n new episodes - of envs `env_ids` (default 0 .. n-1): each env's day is looked up under ITS
        ticker, the streams of the days drawn are uploaded if they are not on the device yet, and the envs' bookkeeping is moved to
        their new (ticker, day).  Call _after_reset() once the device has reset them."""
        ids = np.arange(n) if env_ids is None else np.asarray(env_ids, dtype=np.int64)
        day, start_us = self._draw_days(n)
        dates = self._day_lookup()
        keys = [(self.tickers[int(self._env_ticker[e])], dates[int(d)]) for e, d in zip(ids, day)]
        missing = sorted({k for k in keys if not self._residency.has(k)})
        if missing:
            raise KeyError(f"no packed stream loaded for {missing[0]}")
        sid = self._residency.ensure(sorted(set(keys)))
        self.last_draw = (ids.copy(), keys, start_us.copy())
        return np.array([sid[k] for k in keys], dtype=np.int32), start_us

    def _after_reset(self):
        """the device has moved the envs onto their new streams - only now does the bookkeeping follow (a reset that raised leaves every
        env on the stream the library still has it on) - and days nobody replays any more may leave the device (stream_budget_mb)"""
        if self.last_draw is not None:
            for e, k in zip(self.last_draw[0], self.last_draw[1]):
                self._env_key[int(e)] = k
        self._residency.trim({k for k in self._env_key if k is not None})

    # ---- errors --------------------------------------------------------------------------------------------------
    @staticmethod
    def error_text(status: int) -> str:
        """What a status word means, in the reference's words where it raises on the same condition."""
        parts = []
        if status & _capi.ST_EMPTY_BOOK:
            parts.append("EmptyOrderbookError: Trying take liquidity from empty side of the book.")  # Exchange.py:186
        if status & _capi.ST_ASSERT:
            # Exchange.py:60; also RollingSharpe's non-positive AUM (RewardFunctions.py) and the KeyError of a step whose Beta volumes are
            # NaN (an action at or above the concentration, shape parameters whose pdf underflows: HistoricalOrderbookEnvironment.py:238-240)
            parts.append("AssertionError: Order volume must be positive (or: the action's Beta distribution is NaN; or: RollingSharpe met a non-positive AUM).")
        if status & _capi.ST_BAD_MESSAGE:
            parts.append("AssertionError: cross_trade / trading_halt message in the stream")  # HistoricalOrderGenerator.py:54-56
        if status & _capi.ST_LEVEL_OVERFLOW:
            parts.append("LOBSIM_ST_LEVEL_OVERFLOW: more price levels than the env's budget (max_levels; the reference never drops a level that left "
                         "LOBSTER's visible range: one stale level per tick of price drift) - pass a larger max_levels (up to 2048) or shorten the episode")
        if status & _capi.ST_QUEUE_OVERFLOW:
            parts.append("LOBSIM_ST_QUEUE_OVERFLOW: more live orders at one price than a level's queue holds (lobsim_limits.queue_slots: 4096 - or 64 / 256 / 1024 with the env's queue_pages used up)")
        if status & _capi.ST_AGENT_OVERFLOW:
            parts.append("LOBSIM_ST_AGENT_OVERFLOW: more live agent orders than the env's table holds (agent_orders=128 was asked for - the default of 256 cannot overflow for up to 18 quote levels; entries freed in a step are reusable from the next one)")
        if status & _capi.ST_STREAM_END:
            parts.append("LOBSIM_ST_STREAM_END: no snapshot at the episode's start, or stepped past the end of the loaded stream")
        if status & _capi.ST_NOT_RESET:
            parts.append("LOBSIM_ST_NOT_RESET: step() before reset()")
        return "; ".join(parts) or f"status {status}"

    def _note_dones(self, done: np.ndarray):
        """Bookkeeping after a step that reported dones: which of them are errors (the reference would have raised)."""
        self._dones |= done
        self._any_done = True
        self.status = self.sim.status()

    # ---- array path ------------------------------------------------------------------------------------------------
    def vector_reset_arrays(self) -> np.ndarray:
        sids, starts = self._draw(self.num_envs)
        obs = self.sim.reset(starts, sids)
        self._after_reset()
        self._dones[:] = False
        self._any_done = False
        self._reset_cache = {}
        self.status = self.sim.status()
        return obs.astype(self.obs_dtype, copy=False)

    def vector_step_arrays(self, actions, want_infos: bool = True, copy: bool = True):
        """-> (obs obs_dtype[N, D], reward f64[N], done bool[N], infos).  infos is None when nothing is to be said (no info
        calculator - the reference's default - and no env stopped on an error) or when not asked for; otherwise one dict per env.
        copy=False returns views of the handle's pinned buffers, valid until the next step."""
        if self._io is None:
            self._io = self.sim.host_io()
        io = self._io
        self._reset_cache = {}  # (observations of resets made ahead of their reset_at call: stale once the envs step)
        np.copyto(io["actions"], np.asarray(actions, dtype=np.float64).reshape(io["actions"].shape))
        use_active = self._any_done  # (nothing to mask until an env has finished)
        if use_active:
            np.logical_not(self._dones, out=io["active"].view(np.bool_))
        f32 = self.obs_dtype == np.float32
        self.sim.step_host_io(use_active, 2 if f32 else 1)
        obs, reward, done = io["obs_f32" if f32 else "obs"], io["reward"], io["done"].view(np.bool_)
        if copy:
            o, r, d = self._out_buffers(obs.dtype)
            np.copyto(o, obs), np.copyto(r, reward), np.copyto(d, done)
            obs, reward, done = o, r, d
        if use_active:  # an env that finished earlier and was not reset did not step: it keeps saying done, with no reward
            reward = np.where(self._dones, 0.0, reward)
            new_done = done & ~self._dones
        else:
            new_done = done
        if new_done.any():
            self._note_dones(new_done)
        infos = None
        if want_infos:
            if self.info_calculator is not None:
                infos = self.info_calculator.calculate_dicts(self.sim, actions)
            bad = np.nonzero(new_done & (self.status != 0))[0] if new_done.any() else ()
            if len(bad):
                infos = infos if infos is not None else [{} for _ in range(self.num_envs)]
                for i in bad:
                    infos[i] = dict(infos[i], error=self.error_text(int(self.status[i])), status=int(self.status[i]))
        return obs, reward, done, infos

    def _out_buffers(self, obs_dtype):
        """Arrays for one step's outputs.  Fresh memory for 4096 envs' worth of outputs costs more than the copy itself (the pages are
        faulted in on first touch), so the memory of a set the caller has let go of - the usual case, a loop that consumes a step's
        outputs before the next step - is handed out again.  "Let go of" is observed, not inferred: every hand-out is a NEW array
        object over the pooled buffer (np.frombuffer: its base is the buffer, not another ndarray, so numpy does not collapse the base
        of a view the caller takes - a view keeps the handed-out array alive), and the pool keeps only weak references to what it
        handed out; while any of them is alive - the array itself, a slice, a reshape - the buffer is left alone and a new one is made,
        so holding on to results stays safe on every Python (no reference-count arithmetic)."""
        import weakref

        N, D = self.num_envs, self.sim.obs_dim
        item = np.dtype(obs_dtype).itemsize
        for entry in self._out_pool:
            if entry["dtype"] == obs_dtype and all(r() is None for r in entry["out"]):
                break
        else:
            entry = dict(dtype=np.dtype(obs_dtype), mem=(bytearray(N * D * item), bytearray(N * 8), bytearray(N)), out=())
            if len(self._out_pool) >= 8:
                self._out_pool.pop(0)  # (all in use: the oldest buffers now belong to whoever still holds arrays over them)
            self._out_pool.append(entry)
        bufs = (np.frombuffer(entry["mem"][0], dtype=obs_dtype), np.frombuffer(entry["mem"][1], dtype=np.float64), np.frombuffer(entry["mem"][2], dtype=np.bool_))
        bufs[0].shape = (N, D)  # (in place: .reshape() would hand out a view of an ndarray, whose views numpy re-bases on that hidden array)
        entry["out"] = tuple(weakref.ref(b) for b in bufs)
        return bufs

    # ---- tensor path: torch tensors on the env's device, torch's current stream, no PCIe, no host sync per step ----------
    def _tensors(self):
        if self._t is None:
            import torch

            dev = torch.device("cuda", self.sim.cfg.device_id)
            D, N = self.sim.obs_dim, self.num_envs
            f32 = self.obs_dtype == np.float32
            self._t = dict(torch=torch, dev=dev, obs=torch.zeros((N, D), dtype=torch.float32 if f32 else torch.float64, device=dev),
                           reward=torch.zeros(N, dtype=torch.float64, device=dev), done=torch.zeros(N, dtype=torch.uint8, device=dev), bound=None, side=None)  # fmt: skip
        return self._t

    def _bind_stream(self, t):
        """Launch on torch's current stream.  The null stream cannot be handed to the library (0 means "your own stream"): work on
        it is ordered against a side stream with two stream waits."""
        torch = t["torch"]
        cur = torch.cuda.current_stream(t["dev"])
        target = cur
        if cur.cuda_stream == 0:
            if t["side"] is None:
                t["side"] = torch.cuda.Stream(device=t["dev"])
            target = t["side"]
            target.wait_stream(cur)
        if t["bound"] != target.cuda_stream:
            self.sim.set_stream(target.cuda_stream)  # (synchronises the previous stream: a one-off)
            t["bound"] = target.cuda_stream
        return cur, target

    def vector_reset_tensors(self):
        """env.reset() of every env -> obs tensor [N, D] (obs_dtype) on the device."""
        t = self._tensors()
        self._bind_stream(t)  # (the reset below synchronises the stream it runs on)
        sids, starts = self._draw(self.num_envs)
        f32 = self.obs_dtype == np.float32
        self.sim.reset_device(starts, sids, None, 0 if f32 else t["obs"].data_ptr(), t["obs"].data_ptr() if f32 else 0, t["done"].data_ptr())
        self._after_reset()
        self._dones[:] = False
        self._any_done = False
        self._reset_cache = {}
        return t["obs"]

    def vector_step_tensors(self, actions):
        """actions: float64 CUDA tensor [N, A] on the env's device -> (obs [N, D], reward f64 [N], done uint8 [N]): the env's
        own tensors, overwritten by the next step.  Asynchronous.  An env that reports done is left alone by later steps (its
        rows keep their last values, done stays 1) until reset_done_tensors() / vector_reset_tensors()."""
        t = self._tensors()
        torch = t["torch"]
        assert actions.is_cuda and actions.dtype == torch.float64 and actions.is_contiguous() and tuple(actions.shape) == (self.num_envs, self.sim.action_dim)
        cur, target = self._bind_stream(t)
        f32 = self.obs_dtype == np.float32
        p = t["obs"].data_ptr()
        self.sim.step_device_skip(actions.data_ptr(), t["done"].data_ptr(), 0 if f32 else p, p if f32 else 0, t["reward"].data_ptr(), t["done"].data_ptr())
        if target is not cur:
            cur.wait_stream(target)
        return t["obs"], t["reward"], t["done"]

    def reset_done_tensors(self):
        """Reset every env whose done flag is set (one launch; synchronises: the flags are read on the host).  Returns the indices
        reset; ``status`` then holds the error bits those envs had stopped with (0 = a normal episode end)."""
        t = self._tensors()
        self._bind_stream(t)  # (the reset below synchronises the stream it runs on)
        self.sim.sync()
        ids = np.nonzero(t["done"].cpu().numpy())[0].astype(np.int32)
        if len(ids) == 0:
            return ids
        self.status = self.sim.status()
        sids, starts = self._draw(len(ids), ids)
        f32 = self.obs_dtype == np.float32
        self.sim.reset_device(starts, sids, ids, 0 if f32 else t["obs"].data_ptr(), t["obs"].data_ptr() if f32 else 0, t["done"].data_ptr())
        self._after_reset()
        return ids

    # ---- VectorEnv API ------------------------------------------------------------------------------------------
    def vector_reset(self) -> List[np.ndarray]:
        return list(self.vector_reset_arrays())

    def reset_at(self, index: Optional[int] = None) -> np.ndarray:
        index = 0 if index is None else int(index)
        row = self._reset_cache.pop(index, None)
        if row is not None:
            return row
        if self._dones[index]:
            # the first reset_at after a step that reported dones: every finished env in ONE launch, their start times drawn in
            # index order (= the order of RLlib's reset_at calls); the others' observations wait in the cache
            ids = np.nonzero(self._dones)[0].astype(np.int32)
            obs = self.reset_many(ids)
            self._reset_cache = {int(i): obs[k] for k, i in enumerate(ids)}
            return self._reset_cache.pop(index)
        return self.reset_many([index])[0]

    def reset_many(self, indices: Sequence[int]) -> np.ndarray:
        ids = np.asarray(indices, dtype=np.int32)
        sids, starts = self._draw(len(ids), ids)
        obs = self.sim.reset(starts, sids, ids)
        self._after_reset()
        self._dones[ids] = False
        self._any_done = bool(self._dones.any())
        # (a reset can fail like the reference's - its warm-up replays the day with the features live: an empty side, a feature that
        # raises, no data at the drawn time - and says so here; the env then reports done at its next step)
        self.status[ids] = self.sim.status()[ids]
        return obs.astype(self.obs_dtype, copy=False)

    def vector_step(self, actions):
        obs, reward, done, infos = self.vector_step_arrays(actions)
        if infos is None:
            infos = [{} for _ in range(self.num_envs)]
        return list(obs), reward.tolist(), done.tolist(), infos

    def get_sub_environments(self):
        return []

    def restart_at(self, index: Optional[int] = None) -> None:
        """RLlib (ray >= 2.0) calls this for a sub-environment that raised, before it resets it: the env's device state is rebuilt by
        the reset that follows, so there is nothing to tear down here (the reference env is a CPU object that RLlib would re-create)."""
        return None

    def try_render_at(self, index: Optional[int] = None):
        return None  # the reference's render() is a no-op too (HistoricalOrderbookEnvironment.py:353-358)

    def portfolios(self):
        """(inventory int64[N], cash int64[N], price f64[N]) straight from device state."""
        return self.sim.portfolios()

    def close(self):
        self.sim.close()


class HistoricalOrderbookEnvironment(_GymEnv):
    """Single-env facade with the reference's gym.Env surface (reset / step / seed / render / spaces); a ``gym.Env`` when gym
    imports (rl4mm/gym/HistoricalOrderbookEnvironment.py:51)."""

    metadata = {"render.modes": ["human"]}

    def __init__(self, catalog: StreamCatalog = None, **kwargs):
        if catalog is None:
            raise ValueError("pass catalog=StreamCatalog(...): packed streams replace the reference's HistoricalDatabase")
        kwargs.pop("simulator", None)
        kwargs.pop("order_distributor", None)
        kwargs.pop("preload_orders", None)
        self._venv = HistoricalOrderbookVectorEnv(1, catalog, **kwargs)
        self.action_space = self._venv.action_space
        self.observation_space = self._venv.observation_space
        self.features = self._venv.features
        self.spec = None

    def reset(self):
        obs = self._venv.reset_at(0)
        if self._venv.status[0]:  # the reference's reset() raises what its warm-up raised (HistoricalOrderbookEnvironment.py:147-161)
            self._raise(int(self._venv.status[0]))
        return obs

    def step(self, action: np.ndarray):
        obs, reward, done, info = self._venv.vector_step(np.asarray(action, dtype=np.float64)[None, :])
        if done[0] and self._venv.status[0]:
            self._raise(int(self._venv.status[0]))
        return obs[0], float(reward[0]), bool(done[0]), info[0]

    @staticmethod
    def _raise(status: int):  # the exceptions the reference raises on the same conditions
        if status & _capi.ST_EMPTY_BOOK:
            raise EmptyOrderbookError("Trying take liquidity from empty side of the book.")
        if status & _capi.ST_ASSERT:  # (the reference's exception type differs by cause - AssertionError, KeyError, Exception -; the status word has one bit for them)
            raise AssertionError("Order volume must be positive (or: the action's Beta distribution is NaN; or: RollingSharpe met a non-positive AUM).")
        raise RuntimeError(f"env stopped with status {status} (include/lobsim.h: enum lobsim_status): " + HistoricalOrderbookVectorEnv.error_text(status))

    def seed(self, seed=42):
        self.np_random = np.random.RandomState(seed)
        return [seed]

    def render(self, mode="human"):
        pass

    def close(self):
        self._venv.close()

    @staticmethod
    def get_default_features(step_size: timedelta, episode_length: timedelta, normalisation_on: bool = False):
        return get_default_features(step_size, episode_length, normalisation_on)


class EmptyOrderbookError(Exception):  # rl4mm/orderbook/Exchange.py:32-33
    pass


def env_creator(env_config: dict, catalog: StreamCatalog = None, num_envs: Optional[int] = None):
    """rl4mm/gym/utils.py:42-87 with the same env_config keys.  ``num_envs`` (or env_config["num_envs"]) > 1 returns the batched
    VectorEnv instead of the single-env facade.

    Keys beyond the reference's (all optional): ``catalog`` (the packed streams), ``num_envs``, ``ticker`` as a LIST (env i replays
    tickers[i mod T]), and what places and sizes the envs on the GPU - ``device_id`` (default: by the rollout worker's index,
    ``default_device_id``: with ``num_workers=8`` on an 8-GPU node every worker's envs land on their own device instead of all on
    device 0), ``obs_dtype`` ("float32": the kernel writes the observation_space's dtype, 10 - 15 % faster through the array path),
    ``max_levels``, ``ladder_levels``, ``outer_levels``, ``agent_orders``, ``stream_budget_mb``, ``queue_pages``."""
    catalog = catalog if catalog is not None else env_config.get("catalog")
    episode_length = timedelta(minutes=env_config["episode_length"])
    step_size = timedelta(seconds=env_config["step_size"])
    if env_config["features"] == "agent_state":
        features = [Inventory(max_value=env_config["max_inventory"], normalisation_on=env_config["normalisation_on"])]
    elif env_config["features"] == "full_state":
        features = get_default_features(step_size, episode_length, env_config["normalisation_on"])
    else:
        raise ValueError(env_config["features"])
    kw = dict(
        ticker=env_config["ticker"], episode_length=episode_length, features=features, max_inventory=env_config["max_inventory"],
        min_date=datetime.strptime(env_config["min_date"], "%Y-%m-%d"), max_date=datetime.strptime(env_config["max_date"], "%Y-%m-%d"),
        step_size=step_size, initial_portfolio=Portfolio(inventory=env_config["initial_inventory"], cash=env_config["initial_cash"]),
        per_step_reward_function=get_reward_function(env_config["per_step_reward_function"]),
        terminal_reward_function=get_reward_function(env_config["terminal_reward_function"]),
        market_order_clearing=env_config["market_order_clearing"], market_order_fraction_of_inventory=env_config["market_order_fraction_of_inventory"],
        concentration=env_config["concentration"], min_quote_level=env_config["min_quote_level"], max_quote_level=env_config["max_quote_level"],
        enter_spread=env_config["enter_spread"], info_calculator=env_config["info_calculator"], n_levels=env_config["n_levels"],  # (a required key, as in rl4mm/gym/utils.py:81)
    )  # fmt: skip
    dev = env_config.get("device_id")
    kw["device_id"] = default_device_id(env_config) if dev is None else int(dev)
    for key in ("max_levels", "ladder_levels", "outer_levels", "agent_orders", "stream_budget_mb", "queue_pages"):
        if env_config.get(key) is not None:
            kw[key] = env_config[key]
    if env_config.get("obs_dtype") is not None:
        kw["obs_dtype"] = np.dtype(env_config["obs_dtype"])
    n = num_envs or env_config.get("num_envs", 1)
    env = HistoricalOrderbookVectorEnv(n, catalog, **kw) if n > 1 else HistoricalOrderbookEnvironment(catalog=catalog, **kw)
    env.spec = make_env_spec(episode_length, step_size)  # rl4mm/gym/utils.py:83-86
    return env


def generate_trajectory(agent, env):
    """rl4mm/gym/utils.py:100-117."""
    observations, rewards, actions, infos = [], [], [], []
    obs = env.reset()
    observations.append(obs)
    while True:
        action = agent.get_action(obs)
        obs, reward, done, info = env.step(action)
        observations.append(obs)
        actions.append(action)
        rewards.append(reward)
        infos.append(info)
        if done:
            break
    return {"observations": observations, "actions": actions, "rewards": rewards, "infos": infos}
