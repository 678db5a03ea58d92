"""tests/golden/sim_internal_orders.npz replayed through rl4mm_amd.simulator.OrderbookSimulator - on the oracle (tests/test_sim_golden.py,
CPU suite) and on the HIP library (tests/test_gpu_simulator.py).  The fixture was recorded from the reference's OrderbookSimulator
(oracle/gen_golden_simulator.py): internal-order lists of every shape - cancellations and deletions of own orders anywhere in a call,
behind own orders that cross the book or the agent's quotes, at the price of a marketable own order - and episodes that start from a
caller's `start_book`; after every call the FilledOrders, the central book order by order, the internal book with the identity of every
own order, the price ranges; the reference's exceptions end an episode and must be raised here on the same call."""
from datetime import datetime, timedelta
from pathlib import Path

import numpy as np

from rl4mm_amd import simulator as S
from rl4mm_amd.stream import PackedStream

GOLD = Path(__file__).resolve().parent / "golden" / "sim_internal_orders.npz"
T_LIMIT, T_CANCEL, T_DELETE, T_MARKET = 1, 2, 3, 4
BIG = 2**62


def n_cases() -> int:
    with np.load(GOLD) as g:
        return int(g["n_cases"])


def _book_rows(book):
    rows = []
    for side, d in ((0, book.buy), (1, book.sell)):
        prices = list(d.keys())
        for p in (reversed(prices) if side == 0 else prices):
            for o in d[p]:
                rows.append((side, int(p), int(o.volume), int(bool(o.is_external))))
    return rows


def _bound(v):
    return int(v) if np.isfinite(v) else (BIG if v > 0 else -BIG)


def replay_case(k: int, new_sim) -> dict:
    with np.load(GOLD) as g:
        c = {name[len(f"c{k}_"):]: g[name] for name in g.files if name.startswith(f"c{k}_")}
    L, outer, start_us, length_s = (int(x) for x in c["meta"])
    tick, date = str(c["ticker"]), str(c["date"])
    ps = PackedStream(tick, date, L, c["msgs"], c["snaps"])
    day0 = datetime.strptime(date, "%Y-%m-%d")
    sim = S.OrderbookSimulator(ticker=tick, n_levels=L, database=ps, episode_length=timedelta(seconds=length_s), outer_levels=outer, _new_sim=new_sim)
    try:
        start = day0 + timedelta(microseconds=start_us)
        book = sim.reset_episode(start)
        if len(c["book_edit"]):  # reset_episode(start_date, start_book): the historical book, edited as the recorder edited the reference's
            for side, price, vol, ext_id in c["book_edit"]:
                d = book.buy if side == 0 else book.sell
                if vol < 0:
                    del d[int(price)]
                else:
                    d[int(price)] = list(d[int(price)]) + [S.LimitOrder(timestamp=start, direction="buy" if side == 0 else "sell", ticker=tick, internal_id=7, external_id=int(ext_id),
                                                                       is_external=True, price=int(price), volume=int(vol))]  # fmt: skip
            sim.reset_episode(start, start_book=book)
        sub_of, id_of, n_sub = {}, {}, 0  # the class's internal id <-> the fixture's submission number
        now = start
        stats = dict(steps=0, orders=0, fills=0, raised=None)
        for step, until_us in enumerate(c["until_us"]):
            until = day0 + timedelta(microseconds=int(until_us))
            rows = c["orders"][c["orders"][:, 0] == step]
            raised = bool(len(rows)) and rows[-1, 1] == -1
            orders, limits = [], []
            for _, typ, side, price, vol, tgt, sub in rows[: len(rows) - int(raised)]:
                base = dict(timestamp=now, direction="buy" if side == 0 else "sell", ticker=tick, external_id=None, is_external=False)
                if typ == T_LIMIT:
                    orders.append(S.LimitOrder(internal_id=None, price=int(price), volume=int(vol), **base))
                    limits.append((orders[-1], int(sub)))
                elif typ == T_MARKET:
                    orders.append(S.MarketOrder(internal_id=None, volume=int(vol), **base))
                else:
                    cls = S.Deletion if typ == T_DELETE else S.Cancellation
                    orders.append(cls(internal_id=id_of[int(tgt)], price=int(price), volume=None if vol < 0 else int(vol), **base))
            stats["orders"] += len(orders)
            try:
                filled = sim.forward_step(until, orders)
            except Exception as ex:  # noqa: BLE001
                assert raised, f"case {k} step {step}: {type(ex).__name__} where the reference went on: {ex}"
                stats["raised"] = type(ex).__name__
                break
            assert not raised, f"case {k} step {step}: the reference raised on this call"
            for o, sub in limits:
                sub_of[o.internal_id], id_of[sub] = sub, o.internal_id
                n_sub = max(n_sub, sub + 1)
            want = c["fills"][c["fills"][:, 0] == step]
            for which, name in ((0, "internal"), (1, "external")):
                got = [(0 if o.direction == "buy" else 1, int(o.price), int(o.volume)) for o in getattr(filled, name)]
                assert got == [tuple(int(x) for x in r[2:]) for r in want[want[:, 1] == which]], f"case {k} step {step}: {name} fills differ"
                stats["fills"] += len(got)
            got = _book_rows(sim.exchange.central_orderbook)
            assert got == [tuple(int(x) for x in r[1:]) for r in c["central"][c["central"][:, 0] == step]], f"case {k} step {step}: central book differs"
            got = []
            ib = sim.exchange.internal_orderbook
            for side, d in ((0, ib.buy), (1, ib.sell)):
                prices = list(d.keys())
                for p in (reversed(prices) if side == 0 else prices):
                    for o in d[p]:
                        if o.internal_id not in sub_of:  # an own order the outer-level resync took out and put back: a new order under a new id (the recorder numbers it the same way)
                            sub_of[o.internal_id], id_of[n_sub], n_sub = n_sub, o.internal_id, n_sub + 1
                        got.append((side, int(p), int(o.volume), sub_of[o.internal_id]))
            assert got == [tuple(int(x) for x in r[1:]) for r in c["internal"][c["internal"][:, 0] == step]], f"case {k} step {step}: internal book differs"
            assert (_bound(sim.min_buy_price), _bound(sim.max_sell_price)) == tuple(int(x) for x in c["ranges"][step]), f"case {k} step {step}: price ranges differ"
            now = until
            stats["steps"] += 1
        else:
            assert not (len(c["orders"]) and c["orders"][-1, 1] == -1 and c["orders"][-1, 0] >= len(c["until_us"])), "fixture ends with a raise that was not replayed"
        return stats
    finally:
        sim.close()
