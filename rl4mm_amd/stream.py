"""LOBSTER CSV pair -> packed int64 message stream + snapshot table (host side, numpy/pandas).

This replaces the reference's database layer *as a data format*: the stream a simulator replays is
whatever ``HistoricalDatabase.get_messages`` / ``get_last_snapshot`` would have returned, so the packer
restates the semantics that layer imposes:

* Q1  timestamps: LOBSTER float seconds -> ``pd.to_timedelta(unit="s")``
      (reference rl4mm/database/database_population_helpers.py:126-129) -> SQL DateTime, i.e. floor to
      microseconds.
* Q2  replay order: ``ORDER BY timestamp, id`` where ``id`` is the *string*
      ``"{freq}_L{nnn}_NASDAQ_{ticker}_{date}_{rowindex}"`` (rl4mm/database/HistoricalDatabase.py:111,
      database_population_helpers.py:163-180), so microsecond ties are broken lexicographically by the
      decimal row index ("10" < "9").
* Q3  type / direction mapping (database_population_helpers.py:132-136,151-160); hidden executions are
      dropped (rl4mm/simulation/HistoricalOrderGenerator.py:57).
* Q4  snapshot rows: for each target time on the ``book_snapshot_freq`` grid the book row of the last
      message at or before it, stamped with *that message's* time
      (database_population_helpers.py:45-62,139-148); ``book_snapshot_freq=None`` keeps every row.

Packed layout (all little-endian int64, C-contiguous):

``msgs[n, 4]``   = ``[t_us, external_id, price, volume | type << 48 | side << 52]``
                   type 1 limit, 2 cancellation, 3 deletion, 4 market, 6 cross_trade, 7 trading_halt;
                   side 0 buy / 1 sell, already flipped for executions.
``snaps[m, 2 + 4 L]`` = ``[t_us, source_row, (sell_price_i, sell_vol_i, buy_price_i, buy_vol_i) for i < L]``
                   (column order of rl4mm/orderbook/helpers.py:52-55), sorted by the same (t_us, id) key.
"""
from __future__ import annotations

import gzip
import io
from dataclasses import dataclass
from pathlib import Path
from typing import Optional, Union

import numpy as np
import pandas as pd

MSG_WORDS = 4
TYPE_SHIFT = 48
SIDE_SHIFT = 52
VOL_MASK = (1 << 48) - 1

TYPE_LIMIT, TYPE_CANCEL, TYPE_DELETE, TYPE_MARKET, TYPE_HIDDEN, TYPE_CROSS, TYPE_HALT = 1, 2, 3, 4, 5, 6, 7
SIDE_BUY, SIDE_SELL = 0, 1

DUMMY_SELL_PRICE = 9999999999  # LOBSTER filler for missing levels (rl4mm/orderbook/Exchange.py:163-169)
DUMMY_BUY_PRICE = -9999999999


@dataclass
class PackedStream:
    """One (ticker, trading day) worth of replayable data."""

    ticker: str
    trading_date: str  # "YYYY-MM-DD"
    n_levels: int
    msgs: np.ndarray  # int64 [n, 4]
    snaps: np.ndarray  # int64 [m, 2 + 4 * n_levels]

    def __post_init__(self):
        self.msgs = np.ascontiguousarray(self.msgs, dtype=np.int64).reshape(-1, MSG_WORDS)
        self.snaps = np.ascontiguousarray(self.snaps, dtype=np.int64).reshape(-1, 2 + 4 * self.n_levels)

    @property
    def n_msgs(self) -> int:
        return int(self.msgs.shape[0])

    @property
    def n_snaps(self) -> int:
        return int(self.snaps.shape[0])

    # --- the two queries the reference's simulator makes of its database -------------------------------
    def first_msg_after(self, t_us: int) -> int:
        """Index of the first message with ``timestamp > t_us`` (HistoricalDatabase.get_messages:109)."""
        return int(np.searchsorted(self.msgs[:, 0], t_us, side="right"))

    def last_snapshot_at(self, t_us: int) -> int:
        """Index of the last snapshot with ``timestamp <= t_us`` (get_last_snapshot:46-62); -1 if none."""
        return int(np.searchsorted(self.snaps[:, 0], t_us, side="right")) - 1

    def save(self, path: Union[str, Path]) -> None:
        np.savez_compressed(
            path,
            ticker=np.array(self.ticker),
            trading_date=np.array(self.trading_date),
            n_levels=np.array(self.n_levels),
            msgs=self.msgs,
            snaps=self.snaps,
        )

    @staticmethod
    def load(path: Union[str, Path]) -> "PackedStream":
        z = np.load(path, allow_pickle=False)
        return PackedStream(str(z["ticker"]), str(z["trading_date"]), int(z["n_levels"]), z["msgs"], z["snaps"])


def unpack_meta(meta: np.ndarray):
    """-> (volume, type, side) arrays from the 4th message word."""
    meta = np.asarray(meta, dtype=np.int64)
    return meta & VOL_MASK, (meta >> TYPE_SHIFT) & 0xF, (meta >> SIDE_SHIFT) & 0x1


def _open_text(path: Union[str, Path]):
    path = str(path)
    if path.endswith(".gz"):
        return io.TextIOWrapper(gzip.open(path, "rb"))
    return open(path, "r")


_POW10 = 10 ** np.arange(19, dtype=np.int64)


def _id_suffix_order(t_us: np.ndarray, id_index: np.ndarray) -> np.ndarray:
    """Permutation sorting rows by (t_us, str(id_index)) - the (timestamp, id) order of Q2.

    The string comparison is done on integers: decimal strings compare like their digits left-aligned to a common width
    ("10" -> 10, "9" -> 90), and a string that is a proper prefix of another ("1", "10") sorts first, i.e. by length."""
    ids = np.asarray(id_index, dtype=np.int64)
    assert ids.size == 0 or (ids.min() >= 0 and ids.max() < 10**18)
    ndig = np.maximum(np.searchsorted(_POW10, ids, side="right"), 1)  # digits of str(id); str(0) has one
    width = int(ndig.max()) if ids.size else 1
    left_aligned = ids * _POW10[width - ndig]
    return np.lexsort((ndig, left_aligned, t_us))


def _snapshot_rows(ts_ns: np.ndarray, freq: Optional[str]) -> np.ndarray:
    """Row positions (within this batch) whose book rows become snapshots (Q4).

    For ``freq=None`` every row.  Otherwise one row per grid time ``T`` in
    ``[ceil(first), ceil(last)]``: the *last* row carrying the greatest timestamp ``<= T``; grid times
    before the first message select nothing; repeated selections collapse.
    """
    n = len(ts_ns)
    if freq is None:
        # messages.timestamp itself is the interval series; get_book_snapshots keeps rows whose *index*
        # is in it, i.e. all rows.
        return np.arange(n)
    ts = pd.DatetimeIndex(ts_ns.astype("datetime64[ns]"))
    if freq.endswith("S") and not freq.endswith("MS"):  # pandas >= 2.2 spells the reference's "S" as "s"
        freq = freq[:-1] + "s"
    targets = pd.date_range(ts[0].ceil(freq), ts[-1].ceil(freq), freq=freq).values.astype("datetime64[ns]")
    targets = targets.astype(np.int64)
    # last row of each distinct timestamp
    is_last = np.ones(n, dtype=bool)
    is_last[:-1] = ts_ns[1:] != ts_ns[:-1]
    uniq_rows = np.nonzero(is_last)[0]
    uniq_ts = ts_ns[uniq_rows]
    pos = np.searchsorted(uniq_ts, targets, side="right") - 1
    pos = pos[pos >= 0]
    rows = uniq_rows[pos]
    # pandas .unique() keeps first-appearance order; rows are non-decreasing so this is a plain unique
    return np.unique(rows)


def pack_lobster(
    message_path: Union[str, Path],
    orderbook_path: Union[str, Path],
    ticker: str,
    trading_date: str,
    n_levels: int,
    book_snapshot_freq: Optional[str] = "S",
    max_rows: Optional[int] = None,
    batch_size: int = 1_000_000,
) -> PackedStream:
    """Pack one LOBSTER message/orderbook CSV pair.

    ``max_rows`` / ``batch_size`` have the meaning of ``populate_database``'s arguments
    (rl4mm/database/populate_database.py:38-95): the reference ingests in batches and computes snapshot
    rows and the id offset per batch, which is observable at batch boundaries, so it is kept.
    """
    with _open_text(message_path) as f:
        m = pd.read_csv(
            f,
            header=None,
            names=["time", "type", "external_id", "volume", "price", "direction"],
            usecols=[0, 1, 2, 3, 4, 5],
            nrows=max_rows,
        )
    with _open_text(orderbook_path) as f:
        books = pd.read_csv(f, header=None, nrows=len(m)).values.astype(np.int64)
    assert books.shape[1] >= 4 * n_levels, "orderbook file has fewer levels than requested"
    books = books[:, : 4 * n_levels]
    return pack_arrays(
        m.time.values,
        m.type.values,
        m.external_id.values,
        m.volume.values,
        m.price.values,
        m.direction.values,
        books,
        ticker,
        trading_date,
        n_levels,
        book_snapshot_freq,
        batch_size,
    )


def pack_arrays(
    time_s,
    mtype,
    external_id,
    volume,
    price,
    direction,
    books,
    ticker: str,
    trading_date: str,
    n_levels: int,
    book_snapshot_freq: Optional[str] = "S",
    batch_size: int = 1_000_000,
) -> PackedStream:
    """Same as :func:`pack_lobster` from in-memory LOBSTER columns (``books[i]`` = book after row i)."""
    time_s = np.asarray(time_s, dtype=np.float64)
    n = len(time_s)
    # Q1: float seconds -> pandas ns (same third-party conversion the reference calls) -> floor to us.
    ts_ns = pd.to_timedelta(time_s, unit="s").values.astype("timedelta64[ns]").astype(np.int64)
    t_us = ts_ns // 1000
    mtype = np.asarray(mtype, dtype=np.int64)
    direction = np.asarray(direction, dtype=np.int64)
    # Q3: limit/cancel/delete keep LOBSTER's side (-1 sell, +1 buy); executions name the aggressor.
    is_mkt = mtype == TYPE_MARKET
    side = np.where(is_mkt, np.where(direction == 1, SIDE_SELL, SIDE_BUY), np.where(direction == -1, SIDE_SELL, SIDE_BUY))

    # id suffix = row label + first label of the batch (get_message_dict_from_series:171 adds start_index
    # to a label that is already global) - only its decimal string matters, as the tie-break.
    row = np.arange(n, dtype=np.int64)
    batch_start = (row // batch_size) * batch_size
    id_index = row + batch_start

    order = _id_suffix_order(t_us, id_index)
    keep = order[mtype[order] != TYPE_HIDDEN]
    vol = np.asarray(volume, dtype=np.int64)
    assert np.all(vol[keep] >= 0) and np.all(vol[keep] <= VOL_MASK)
    msgs = np.empty((len(keep), MSG_WORDS), dtype=np.int64)
    msgs[:, 0] = t_us[keep]
    msgs[:, 1] = np.asarray(external_id, dtype=np.int64)[keep]
    msgs[:, 2] = np.asarray(price, dtype=np.int64)[keep]
    msgs[:, 3] = vol[keep] | (mtype[keep] << TYPE_SHIFT) | (side[keep].astype(np.int64) << SIDE_SHIFT)

    # Q4: snapshot rows, per ingestion batch.
    snap_rows = []
    for b0 in range(0, n, batch_size):
        b1 = min(n, b0 + batch_size)
        snap_rows.append(b0 + _snapshot_rows(ts_ns[b0:b1], book_snapshot_freq))
    snap_rows = np.concatenate(snap_rows) if snap_rows else np.zeros(0, dtype=np.int64)
    # book ids: row label within the (re-indexed) books frame + batch start
    snap_ids = []
    for b0 in range(0, n, batch_size):
        b1 = min(n, b0 + batch_size)
        k = int(np.sum((snap_rows >= b0) & (snap_rows < b1)))
        snap_ids.append(np.arange(k, dtype=np.int64) + b0)
    snap_ids = np.concatenate(snap_ids) if snap_ids else np.zeros(0, dtype=np.int64)
    s_t = t_us[snap_rows]
    s_order = _id_suffix_order(s_t, snap_ids)
    books = np.asarray(books, dtype=np.int64)
    snaps = np.empty((len(snap_rows), 2 + 4 * n_levels), dtype=np.int64)
    snaps[:, 0] = s_t[s_order]
    snaps[:, 1] = snap_rows[s_order]
    snaps[:, 2:] = books[snap_rows[s_order], : 4 * n_levels]
    return PackedStream(ticker, trading_date, n_levels, msgs, snaps)
