"""Constructed LOBSTER-format streams whose price DRIFTS (test infrastructure).

The synthetic generator of the bench (rl4mm_amd/csrc/synth_lob.c) is mean-reverting: an env never holds more than ~40
price levels per side.  The reference's book is an unbounded SortedDict that keeps every level that left LOBSTER's
visible range (rl4mm/orderbook/models.py:64-69, Exchange.py:71-83,238-247), so on a day that trends the side the price
moves away from piles up one stale level per tick of excursion.  This module simulates a small "true" book whose best
prices follow a prescribed path and emits what LOBSTER would have recorded for it: messages only for events inside the
visible `n_levels` per side, the book row after every message.  The columns go through the real packer
(rl4mm_amd.stream.pack_arrays), so ordering / snapshot quirks are the product's own.

Per tick of upward movement: the orders resting at the best ask are executed (one execution message per resting order)
and a new limit buy opens the price they left; the bid that drops out of the visible range just stays in the simulated
book for good.  Between moves: limit orders, deletions, partial cancellations and small executions at random visible
levels.  Orders that were in the book before the first row are unknown to a simulator (it sees them as snapshot
aggregates): removing one exercises the reference's unknown-id rule (Exchange.py:128-139).
"""
from __future__ import annotations

from collections import deque

import numpy as np
from sortedcontainers import SortedDict

from rl4mm_amd.stream import PackedStream, pack_arrays

TICK = 100


def triangle_path(amplitude: int, seconds_per_tick: float):
    """0 -> +A -> -A -> +A ... in ticks, one tick every `seconds_per_tick` s."""

    def f(t_s: float) -> int:
        k = int(t_s / seconds_per_tick)
        period = 4 * amplitude
        k %= period
        if k <= amplitude:
            return k
        if k <= 3 * amplitude:
            return 2 * amplitude - k
        return k - 4 * amplitude

    return f


def ramp_path(seconds_per_tick: float, sign: int = 1):
    return lambda t_s: sign * int(t_s / seconds_per_tick)


def drift_lobster(seed: int, path, n_levels: int = 10, start_s: float = 34_200.0, duration_s: float = 2_400.0, mid: int = 2_700_000,
                  noise_per_s: float = 12.0, depth_pad: int = 40, max_excursion: int = 2500, target_depth: int = 0):
    """-> dict of LOBSTER columns (time_s, type, id, size, price, direction) + books[n, 4 * n_levels].
    target_depth > 0: queues as deep as a thick large-cap book's - every level starts with about that many resting orders and the
    noise keeps the visible levels near it (an order leaves with a probability proportional to its queue's length)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    next_id = [10_000_000]

    def new_id():
        next_id[0] += 1
        return next_id[0]

    # true book: side 0 buy / 1 sell; price -> deque of [id, vol]; best bid = mid - TICK, best ask = mid
    book = (SortedDict(lambda p: -p), SortedDict())  # best price first on both sides
    n_deep = n_levels + depth_pad + max_excursion
    for i in range(n_deep):
        for side, p in ((0, mid - TICK * (i + 1)), (1, mid + TICK * i)):
            n0 = int(rng.integers(1, 4)) if not target_depth or i >= n_levels + 5 else int(target_depth * rng.uniform(0.7, 1.1))
            book[side][p] = deque([[new_id(), int(100 * rng.integers(1, 40))] for _ in range(n0)])
    cols = {k: [] for k in ("time_s", "type", "id", "size", "price", "direction")}
    rows = []

    def best(side):
        return book[side].keys()[0]

    def visible(side):
        return list(book[side].keys()[:n_levels])

    def book_row():
        r = np.zeros(4 * n_levels, dtype=np.int64)
        vs, vb = visible(1), visible(0)
        for i in range(n_levels):
            if i < len(vs):
                r[4 * i], r[4 * i + 1] = vs[i], sum(o[1] for o in book[1][vs[i]])
            else:
                r[4 * i], r[4 * i + 1] = 9999999999, 0
            if i < len(vb):
                r[4 * i + 2], r[4 * i + 3] = vb[i], sum(o[1] for o in book[0][vb[i]])
            else:
                r[4 * i + 2], r[4 * i + 3] = -9999999999, 0
        return r

    def emit(t, typ, oid, size, price, side):
        cols["time_s"].append(t); cols["type"].append(typ); cols["id"].append(oid); cols["size"].append(size)
        cols["price"].append(price); cols["direction"].append(1 if side == 0 else -1)
        rows.append(book_row())

    def move(t, up: bool):
        """One tick: the best level of the side the price moves INTO is executed away, a new order takes its price on the
        other side."""
        eat = 1 if up else 0
        p = best(eat)
        q = book[eat][p]
        while q:
            oid, vol = q.popleft()
            if not q:
                del book[eat][p]
            emit(t, 4, oid, vol, p, eat)  # execution of a resting order: LOBSTER's direction = the resting side
            t += 1e-5
        oid, vol = new_id(), int(100 * rng.integers(1, 30))
        book[eat ^ 1].setdefault(p, deque()).append([oid, vol])
        emit(t, 1, oid, vol, p, eat ^ 1)

    level = 0
    t = start_s + 0.001
    end = start_s + duration_s
    book[0][best(0) - TICK * 3].append([new_id(), 100])  # (a first row, so that the first snapshot exists)
    emit(t, 1, next_id[0], 100, best(0) - TICK * 3, 0)
    while True:
        t += float(rng.exponential(1.0 / noise_per_s))
        if t >= end:
            break
        want = path(t - start_s)
        while level != want:
            up = want > level
            move(t, up)
            level += 1 if up else -1
            t += 2e-4
        side = int(rng.integers(0, 2))
        vis = visible(side)
        i = min(int(rng.geometric(0.3)) - 1, len(vis) - 1)
        p = vis[i]
        q = book[side][p]
        u = rng.random()
        if target_depth:  # mean-reverting queue length: arrivals at a constant rate, departures in proportion to the queue
            u = 0.2 if rng.random() < target_depth / (target_depth + len(q)) else (0.6 if u < 0.8 else u)
        if u < 0.45:
            oid, vol = new_id(), int(100 * rng.integers(1, 20))
            q.append([oid, vol])
            emit(t, 1, oid, vol, p, side)
        elif u < 0.78:
            if len(q) > 1 or (i >= 2 and rng.random() < 0.15):  # (a level rarely empties: most ticks of drift leave a stale level behind)
                j = int(rng.integers(0, len(q)))
                oid, vol = q[j]
                del q[j]
                if not q:
                    del book[side][p]
                emit(t, 3, oid, vol, p, side)
        elif u < 0.88:
            j = int(rng.integers(0, len(q)))
            if q[j][1] > 100:
                q[j][1] -= 100
                emit(t, 2, q[j][0], 100, p, side)
        else:
            p0 = vis[0]
            head = book[side][p0][0]
            if head[1] > 100:
                ex = int(100 * rng.integers(1, head[1] // 100))
                head[1] -= ex
                emit(t, 4, head[0], ex, p0, side)
    out = {k: np.array(v, dtype=np.float64 if k == "time_s" else np.int64) for k, v in cols.items()}
    out["books"] = np.array(rows, dtype=np.int64).reshape(-1, 4 * n_levels)
    return out


def drift_stream(seed: int, path, n_levels: int = 10, start_s: float = 34_200.0, duration_s: float = 2_400.0, **kw) -> PackedStream:
    g = drift_lobster(seed, path, n_levels, start_s, duration_s, **kw)
    return pack_arrays(g["time_s"], g["type"], g["id"], g["size"], g["price"], g["direction"], g["books"], "DRIFT", "2018-02-21", n_levels, "S")
