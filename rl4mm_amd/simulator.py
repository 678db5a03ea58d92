"""Host-side mirror of the reference's simulator surface, backed by the HIP library's simulator-level entry points.

* order records            ``Order / LimitOrder / MarketOrder / Cancellation / Deletion / FilledOrders / Orderbook``
                           (reference rl4mm/orderbook/models.py:17-107: same names, fields and properties)
* ``OrderbookSimulator``   ``reset_episode(start_date) -> Orderbook``, ``forward_step(until, internal_orders) -> FilledOrders``,
                           ``now_is``, ``min_buy_price / max_sell_price / initial_*_price_range``, ``exchange.central_orderbook`` /
                           ``exchange.internal_orderbook`` (rl4mm/simulation/OrderbookSimulator.py:23-135)

The book lives on the GPU (one env of a ``lobsim_t`` handle): ``reset_episode`` is ``lobsim_reset_episode``, ``forward_step`` is one
``lobsim_forward`` launch - the internal orders, then the window's messages, then the outer-level resync - and the
``FilledOrders`` it returns are read back from the kernel's fill log (``lobsim_get_fills``).  The ``Orderbook`` objects are
snapshots decoded from device state when asked for.  There is no CPU path: without the HIP library and a GPU the constructor raises.

What differs from the reference, by construction:
* the data source is a ``StreamCatalog`` / ``PackedStream`` (rl4mm_amd/stream.py) where the reference takes a ``HistoricalDatabase``;
  ``order_generators`` is accepted and ignored (the packed stream IS the historical order generator's output);
* the agent's orders are identified by ``internal_id`` as in the reference: ``forward_step`` gives every LimitOrder it submits the next
  id (ascending in submission order, like OrderIDConvertor.py:12-19), the library files the order under it (``lobsim_order_view.reserved``)
  and finds the target of a Cancellation / Deletion by it on the device (round 6; until then the host turned ids into FIFO ranks and had to
  refuse the calls in which it could not know them).  The outer-level resync re-adds the agent's orders of a level under FRESH ids, as the
  reference does (OrderbookSimulator.py:114-129 re-submits them): read them from ``exchange.internal_orderbook`` like the reference's env does;
* ``FilledOrders.internal`` entries are ``LimitOrder`` records (direction, price, volume) for resting fills and aggressing fills
  alike - the reference makes the latter ``MarketOrder`` copies with a price; every consumer reads direction / price / volume.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from datetime import datetime, timedelta
from typing import Dict, List, Optional, Tuple

import numpy as np

from . import _capi, _lib
from .stream import PackedStream

try:  # (the reference's container; plain dicts kept sorted otherwise)
    from sortedcontainers import SortedDict
except Exception:  # noqa: BLE001  # pragma: no cover
    SortedDict = None

US = timedelta(microseconds=1)
SIDE = {"buy": 0, "sell": 1}
SIDE_NAME = ("buy", "sell")


# ---------------------------------------------------------------------------------------------------------------------
# order records, rl4mm/orderbook/models.py:17-61
# ---------------------------------------------------------------------------------------------------------------------
@dataclass
class Order:
    timestamp: datetime
    direction: str  # "buy" | "sell"
    ticker: str
    internal_id: Optional[int]
    external_id: Optional[int]
    is_external: bool

    def __lt__(self, other):
        return self.timestamp < other.timestamp


@dataclass
class MarketOrder(Order):
    volume: int
    price: Optional[int] = None


@dataclass
class LimitOrder(Order):
    price: int
    volume: int


@dataclass
class Deletion(Order):
    price: int
    volume: Optional[int]


@dataclass
class Cancellation(Deletion):
    volume: int


@dataclass
class FilledOrders:  # models.py:104-107
    internal: List[Order] = field(default_factory=list)
    external: List[Order] = field(default_factory=list)


def _sorted(d: dict):
    return SortedDict(d) if SortedDict is not None else dict(sorted(d.items()))


class Orderbook:
    """What callers read of the reference's ``Orderbook`` (rl4mm/orderbook/models.py:64-101): the two sides as price -> FIFO list
    of ``LimitOrder`` (ascending prices, a ``SortedDict`` when sortedcontainers is installed) and the touch statistics derived
    from them.  A snapshot decoded from device state, not the live book."""

    def __init__(self, buy: dict, sell: dict, ticker: str, tick_size: int = 100):
        self.buy, self.sell, self.ticker, self.tick_size = buy, sell, ticker, tick_size

    def _touch(self, side: str):
        """(best price, resting volume there) of one side; (0, 0) / (inf, 0) when the side is empty, like the reference's defaults."""
        book = getattr(self, side)
        if not book:
            return (0, 0) if side == "buy" else (np.inf, 0)
        price = max(book) if side == "buy" else min(book)
        return price, sum(o.volume for o in book[price])

    best_buy_price = property(lambda self: self._touch("buy")[0])
    best_sell_price = property(lambda self: self._touch("sell")[0])
    best_buy_volume = property(lambda self: self._touch("buy")[1])
    best_sell_volume = property(lambda self: self._touch("sell")[1])

    @property
    def spread(self):
        return self.best_sell_price - self.best_buy_price

    @property
    def midprice(self):
        return (self.best_sell_price + self.best_buy_price) / 2

    @property
    def imbalance(self):  # (bid volume - ask volume) / (bid volume + ask volume) at the touch, models.py:88-90
        (_, bv), (_, av) = self._touch("buy"), self._touch("sell")
        return (bv - av) / (bv + av)

    @property
    def microprice(self):  # the touch prices weighted by the OPPOSITE side's share of the volume, models.py:92-97
        imb = self.imbalance
        return (1 + imb) / 2 * self.best_sell_price + (1 - imb) / 2 * self.best_buy_price

    def __repr__(self):
        return f"Orderbook({self.ticker}: {len(self.buy)} buy levels, {len(self.sell)} sell levels, touch {self.best_buy_price} / {self.best_sell_price})"


def convert_to_lobster_format(orderbook: Orderbook, n_levels: int = 10) -> dict:
    """A book as one LOBSTER row dict - ``{side}_price_{i}`` / ``{side}_volume_{i}`` for the best n_levels levels of each side, floats -
    what the reference's tests compare with the database's snapshot rows (rl4mm/extras/orderbook_comparison.py:6-19)."""
    row = {}
    for side, best_first in (("buy", True), ("sell", False)):
        levels = getattr(orderbook, side)
        for i, price in enumerate(sorted(levels, reverse=best_first)[:n_levels]):
            row[f"{side}_price_{i}"] = float(price)
            row[f"{side}_volume_{i}"] = float(sum(o.volume for o in levels[price]))
    return row


class _Exchange:
    """The attributes of ``Exchange`` (rl4mm/orderbook/Exchange.py:40-57,149-169) that callers of the simulator read; every access
    decodes the current device state."""

    def __init__(self, owner: "OrderbookSimulator"):
        self._o = owner
        self.ticker = owner.ticker
        self.tick_size = 100

    @property
    def central_orderbook(self) -> Orderbook:
        return self._o._read_book(internal=False)

    @property
    def internal_orderbook(self) -> Orderbook:
        return self._o._read_book(internal=True)

    @property
    def best_buy_price(self):
        return self._o._state()["best_buy"]

    @property
    def best_sell_price(self):
        b = self._o._state()["best_sell"]
        return np.inf if b == np.iinfo(np.int64).max else b

    @property
    def orderbook_price_range(self) -> Tuple[int, int]:
        book = self.central_orderbook
        return (next(p for p in book.buy if p > 0), next(p for p in reversed(book.sell) if p < 9999999999))


class EmptyOrderbookError(Exception):  # rl4mm/orderbook/Exchange.py:32-33
    pass


class OrderbookSimulator:
    def __init__(
        self,
        ticker: str = "MSFT",
        exchange=None,
        order_generators=None,
        n_levels: int = 50,
        database=None,
        preload_orders: bool = True,
        episode_length: timedelta = timedelta(minutes=30),
        warm_up: timedelta = timedelta(seconds=0),
        outer_levels: int = 20,
        device_id: int = 0,
        ladder_levels: int = 0,
        max_levels: int = 0,
        _new_sim=None,  # (tests of this class's host logic bind another library with the same C-ABI)
    ) -> None:
        if database is None:
            raise ValueError("pass database=StreamCatalog(...) or a PackedStream: packed streams replace the reference's HistoricalDatabase")
        self.ticker = ticker
        self.n_levels = n_levels
        self.database = database
        self.preload_orders = preload_orders
        self.episode_length, self.warm_up, self.outer_levels = episode_length, warm_up, outer_levels
        self.now_is: datetime = datetime(2000, 1, 1)  # OrderbookSimulator.py:41
        cfg = _capi.make_cfg(1, n_levels, [_capi.make_feature(_capi.F_INVENTORY, min_value=-1e6, max_value=1e6)], outer_levels=outer_levels,
                             device_id=device_id, ladder_levels=ladder_levels, max_levels=max_levels)  # fmt: skip
        self.sim = (_new_sim or _lib.new_sim)(cfg)  # raises without the HIP library / a GPU
        self.exchange = _Exchange(self)
        self._stream_ids: Dict[str, int] = {}
        self._day0: Optional[datetime] = None
        self._next_internal_id = 0
        self._n_resyncs = 0

    # ---- data ---------------------------------------------------------------------------------------------------
    def _stream_for(self, day: datetime) -> int:
        key = day.strftime("%Y-%m-%d")
        if key not in self._stream_ids:
            db = self.database
            stream = db if isinstance(db, PackedStream) else db.get(self.ticker, key)
            if stream.trading_date != key or stream.ticker != self.ticker:
                raise KeyError(f"no packed stream for {self.ticker} on {key}")
            self._stream_ids[key] = self.sim.add_stream(stream)
        return self._stream_ids[key]

    def _us(self, t: datetime) -> int:
        return int((t - self._day0) / US)

    # ---- OrderbookSimulator.py:55-68 ------------------------------------------------------------------------------
    def reset_episode(self, start_date: datetime, start_book: Optional[Orderbook] = None) -> Orderbook:
        assert start_date.microsecond == 0, "Episodes must be started on the second."
        self._day0 = datetime(start_date.year, start_date.month, start_date.day)
        sid = self._stream_for(self._day0)
        self.sim.reset_episode(self._us(start_date), sid)
        st = self._state()
        assert not (st["status"] & _capi.ST_STREAM_END), f"There is no data before the episode start time: {start_date}"
        if start_book is not None:
            self._install_book(start_book)
            st = self._state()
        self._raise_on_status(st["status"])
        self.now_is = start_date
        self._next_internal_id = 0
        self._n_resyncs = 0
        return self.exchange.central_orderbook

    def _install_book(self, book: Orderbook):
        """``self.exchange.central_orderbook = start_book`` (:58) + ``_reset_initial_price_ranges`` (:185-188): a level whose first order is a
        snapshot order (``internal_id == -1``: what get_historical_start_book builds, :156-175) goes to the device as that; every other order
        of the book follows in queue order as an external limit order under an id no later message can name (the reference's id map knows
        nothing of a start book's orders either: OrderIDConvertor.py:7-37) - or as the agent's own when ``is_external`` is false."""
        L = self.n_levels
        row = np.zeros(4 * L, dtype=np.int64)  # sell_p, sell_v, buy_p, buy_v per level; price 0 = absent
        extra = []
        fake_id = -(1 << 40)
        for side, levels, col in ((1, book.sell, 0), (0, book.buy, 2)):
            prices = sorted(levels, reverse=(side == 0))
            if len(prices) > L:
                raise ValueError(f"start_book holds {len(prices)} {SIDE_NAME[side]} levels, the simulator was made for n_levels={L}")
            for i, p in enumerate(prices):
                q = list(levels[p])
                if q and q[0].internal_id == -1:
                    row[4 * i + col], row[4 * i + col + 1] = int(p), int(q[0].volume)
                    q = q[1:]
                for o in q:
                    fake_id -= 1
                    extra.append((1, side, int(p), int(o.volume), fake_id if o.is_external else 0, 1 if o.is_external else 0))
        self.sim.debug_set_book(0, row, -1)
        if extra:
            self.sim.debug_process_orders(0, np.array(extra, dtype=np.int64))
            self.sim.debug_set_book(0, None, -2)

    # ---- OrderbookSimulator.py:70-88 ------------------------------------------------------------------------------
    def forward_step(self, until: datetime, internal_orders: Optional[List[Order]] = None) -> FilledOrders:
        assert until > self.now_is, f"The current time is {self.now_is.time()}, but we are trying to step forward in time until {until.time()}!"
        rows = self._rows_of(internal_orders or [])
        self.sim.forward(self._us(until), orders=[rows] if len(rows) else None)
        st = self._state()
        self._raise_on_status(st["status"])
        self.now_is = until
        if st["n_resyncs"] != self._n_resyncs:  # the resync re-added the agent's orders of the levels it replaced under fresh ids, taken from the
            self._n_resyncs = st["n_resyncs"]   # device's counter: the ids this class hands out next must lie beyond them
            tags = [int(o["reserved"]) for side in (0, 1) for p in self.sim.book(0, side, 4096)[0] for o in self.sim.level_orders(0, side, int(p), 4096) if o["kind"] == 1]
            self._next_internal_id = max([self._next_internal_id] + tags)
        filled = FilledOrders()
        for lst, side, price, vol in self.sim.fills(0):
            o = LimitOrder(timestamp=until, direction=SIDE_NAME[side], ticker=self.ticker, internal_id=None, external_id=None, is_external=bool(lst),
                           price=int(price), volume=int(vol))  # fmt: skip
            (filled.external if lst else filled.internal).append(o)
        return filled

    # ---- re-synchronisation bookkeeping, OrderbookSimulator.py:47-53 -----------------------------------------------
    @staticmethod
    def _inf(v):
        return np.inf if v == np.iinfo(np.int64).max else v

    @property
    def min_buy_price(self):
        return self._inf(self._state()["min_buy_price"])

    @property
    def max_sell_price(self):
        return self._state()["max_sell_price"]

    @property
    def initial_buy_price_range(self):
        return self._inf(self._state()["initial_buy_range"])

    @property
    def initial_sell_price_range(self):
        return self._inf(self._state()["initial_sell_range"])

    def close(self):
        self.sim.close()

    # ---- internals ----------------------------------------------------------------------------------------------
    def _state(self) -> dict:
        return self.sim.env_state(0)

    @staticmethod
    def _raise_on_status(status: int):
        if status & _capi.ST_EMPTY_BOOK:
            raise EmptyOrderbookError("Trying take liquidity from empty side of the book.")
        if status & _capi.ST_ASSERT:
            raise AssertionError("Order volume must be positive.")
        if status:
            raise RuntimeError(f"simulator stopped with status {status} (include/lobsim.h: enum lobsim_status)")

    def _rows_of(self, orders: List[Order]) -> np.ndarray:
        """Internal orders -> rows of lobsim_forward {type, side, price, volume, id, who} (include/lobsim.h).  A new limit order gets the next
        internal id and is filed under it on the device; a Cancellation / Deletion carries the internal id it names (who = 2: the device
        looks the order up) - also one that names an order of this very call, one the call's own orders may have swept, or one that never
        rested: the list is processed in order on the device, as Exchange.process_order does (Exchange.py:58-69)."""
        rows = []
        for o in orders:
            assert not o.is_external, "internal_orders are the agent's own orders (historical messages come from the packed stream)"
            side = SIDE[o.direction]
            if isinstance(o, LimitOrder):
                self._next_internal_id += 1
                o.internal_id = self._next_internal_id
                rows.append((_capi_type("limit"), side, int(o.price), int(o.volume), o.internal_id, 0))
            elif isinstance(o, MarketOrder):
                rows.append((_capi_type("market"), side, 0, int(o.volume), 0, 0))
            elif isinstance(o, Deletion):  # Cancellation is a Deletion
                typ = _capi_type("cancellation" if isinstance(o, Cancellation) else "deletion")
                iid = o.internal_id if (o.internal_id is not None and 0 < o.internal_id < (1 << 31) - 1) else (1 << 31) - 1  # (None / -1: names no order)
                rows.append((typ, side, int(o.price), -1 if o.volume is None else int(o.volume), int(iid), 2))
            else:
                raise TypeError(f"unsupported internal order {type(o).__name__}")
        return np.array(rows, dtype=np.int64).reshape(-1, 6)

    def _read_book(self, internal: bool) -> Orderbook:
        day0 = self._day0 or datetime(2000, 1, 1)
        sides = ({}, {})
        for side in (0, 1):
            for price in self.sim.book(0, side, 4096)[0]:
                price = int(price)
                for ov in self.sim.level_orders(0, side, price, 4096):
                    kind, vol = int(ov["kind"]), int(ov["volume"])
                    if internal and kind != 1:
                        continue
                    if kind == 1:  # (the order's tag on the device IS its internal id)
                        o = LimitOrder(timestamp=self.now_is, direction=SIDE_NAME[side], ticker=self.ticker, internal_id=int(ov["reserved"]), external_id=None,
                                       is_external=False, price=price, volume=vol)  # fmt: skip
                    else:  # tracked external order (kind 0) or snapshot aggregate (kind 2: internal_id == -1, external)
                        o = LimitOrder(timestamp=day0, direction=SIDE_NAME[side], ticker=self.ticker, internal_id=-1 if kind == 2 else None,
                                       external_id=None if kind == 2 else int(ov["id"]), is_external=True, price=price, volume=vol)  # fmt: skip
                    sides[side].setdefault(price, []).append(o)
        return Orderbook(buy=_sorted(sides[0]), sell=_sorted(sides[1]), ticker=self.ticker, tick_size=100)


def _capi_type(name: str) -> int:
    return {"limit": 1, "cancellation": 2, "deletion": 3, "market": 4}[name]


def create_order(order_type: str, order_dict: dict) -> Order:
    """rl4mm/orderbook/create_order.py:4-42."""
    d = dict(order_dict)
    if order_type == "limit":
        return LimitOrder(**{k: d[k] for k in ("timestamp", "direction", "ticker", "internal_id", "external_id", "is_external", "price", "volume")})
    if order_type == "market":
        return MarketOrder(**{k: d[k] for k in ("timestamp", "direction", "ticker", "internal_id", "external_id", "is_external", "volume")})
    if order_type == "cancellation":
        return Cancellation(**{k: d[k] for k in ("timestamp", "direction", "ticker", "internal_id", "external_id", "is_external", "price", "volume")})
    if order_type == "deletion":
        return Deletion(**{k: d[k] for k in ("timestamp", "direction", "ticker", "internal_id", "external_id", "is_external", "price", "volume")})
    raise NotImplementedError(order_type)
