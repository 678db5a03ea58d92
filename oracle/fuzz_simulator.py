#!/usr/bin/env python3
"""Build container only: the simulator surface on its own - `OrderbookSimulator.reset_episode` / `forward_step(until, internal_orders)
-> FilledOrders` - the upstream reference against `rl4mm_amd.simulator.OrderbookSimulator` (order records, internal ids, the host-side
map from an own order's id to its FIFO rank, the fill log, the decoded books) running on the CPU oracle.

Run:  PYTHONPATH=oracle/refshim:/root/reference PYTHONDONTWRITEBYTECODE=1 python oracle/fuzz_simulator.py [first_seed=0] [end_seed=200]

Each case: a day of oracle/fuzz_reference.py, a start second, 25 - 60 forward steps of random length (50 ms .. 2 s, some ending on a
whole second: the outer-level resync), each with 0 - 4 internal orders - cancellations and deletions of the agent's own live orders
(whole and partial, by the id the simulator gave them; first in the list: rl4mm_amd.simulator refuses a cancellation BEHIND a
self-crossing own order of the same call), limit orders around and through the touch, market orders of a few shares.  After every step:
the FilledOrders lists (direction, price, volume, in order), the central book order by order (price, volume, whose), the internal book
(price, volume, internal id mapped to the submission it came from), the price-range bookkeeping; an exception on one side must be one
on the other.  The reference's own simulator tests (G2, G8) run through the same class in tests/; this draws the cases nobody wrote."""
import sys
import time
import traceback
from datetime import datetime, timedelta
from pathlib import Path

import numpy as np

REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO / "oracle"))
sys.path.insert(0, str(REPO / "tests"))
_args = sys.argv[1:]
sys.argv = sys.argv[:1]
import fuzz_reference as fr  # noqa: E402
import gen_golden as gg  # noqa: E402
import oracle_lib as ol  # noqa: E402
from rl4mm_amd import simulator as S  # noqa: E402

from rl4mm.orderbook import models as RM  # noqa: E402


def book_rows(book, id_of=None):
    """[(side, price, volume, is_external)] best price first within a side, FIFO within a price."""
    rows = []
    for side, d in ((0, book.buy), (1, book.sell)):
        prices = list(d.keys())
        for p in (reversed(prices) if side == 0 else prices):
            for o in d[p]:
                rows.append((side, int(p), int(o.volume), bool(o.is_external)))
    return rows


def run(seed):
    r = np.random.Generator(np.random.PCG64(77_000 + seed))
    r2 = np.random.Generator(np.random.PCG64(177_000 + seed))  # (dimensions added later draw from a generator of their own)
    kind = str(r.choice(["syn10", "syn10", "syn5", "drift", "thin", "thinx", "syn50"]))
    db, ps, tick, L, t0, t1 = fr.day(kind)
    outer = int(r.choice([5, 20, 40]))
    ref = gg.OrderbookSimulator(ticker=tick, order_generators=[gg.HistoricalOrderGenerator(tick, db, True)], n_levels=L, database=db, preload_orders=True,
                                episode_length=timedelta(seconds=t1 - t0), warm_up=timedelta(0), outer_levels=outer)  # fmt: skip
    mine = S.OrderbookSimulator(ticker=tick, n_levels=L, database=ps, episode_length=timedelta(seconds=t1 - t0), outer_levels=outer, _new_sim=ol.oracle_sim)
    start_s = int(r.integers(t0 + 1, max(t0 + 2, t1 - 25)))
    start = fr.DATE + timedelta(seconds=start_s)
    rb, mb = ref.reset_episode(start_date=start), mine.reset_episode(start)
    assert book_rows(rb, {}) == book_rows(mb, {}), "start books differ"
    now = start
    seen = []  # own orders met so far [(reference's internal id, the mirror's, direction, price, volume)]: targets of later cancellations, dead ones included
    n_steps, n_orders, n_fills = int(r.integers(25, 60)), 0, 0
    for step in range(n_steps):
        dt = float(r.choice([0.05, 0.1, 0.1, 0.25, 0.5, 1.0, 2.0]))
        until = now + timedelta(seconds=dt)
        if until > fr.DATE + timedelta(seconds=t1 - 2):
            break
        if r.random() < 0.3:  # end on a whole second now and then (update_outer_levels looks only then)
            until = until.replace(microsecond=0) + timedelta(seconds=1)
        ro, mo = [], []
        bb, bs = ref.exchange.best_buy_price, ref.exchange.best_sell_price
        ts = now
        for _ in range(int(r.choice([0, 0, 1, 1, 2, 3, 4]))):
            what = r.random()
            if what < 0.35 and seen:  # cancel / delete one of the agent's own orders (now and then one that is gone already)
                id_r, id_m, direction, price, v0 = seen[int(r.integers(0, len(seen)))]
                whole = r.random() < 0.5
                vol = None if whole and r.random() < 0.5 else int(v0 if whole else max(1, int(r.integers(1, v0 + 1))))
                cls_r, cls_m = (RM.Deletion, S.Deletion) if vol is None else (RM.Cancellation, S.Cancellation)
                kw = dict(timestamp=ts, direction=direction, ticker=tick, external_id=None, is_external=False, price=price, volume=vol)
                # anywhere in the call's list - also BEHIND own limit / market orders of the same call that cross the book or the agent's own
                # quotes (round 6: the device finds the target by its internal id; until then the mirror refused such calls)
                at = 0 if r2.random() < 0.4 else int(r2.integers(0, len(ro) + 1))
                ro.insert(at, cls_r(internal_id=id_r, **kw)); mo.insert(at, cls_m(internal_id=id_m, **kw))
            elif what < 0.85:
                side = "buy" if r.random() < 0.5 else "sell"
                ref_p = bb if side == "buy" else bs
                if not np.isfinite(ref_p) or ref_p <= 0: continue
                off = int(r.choice([-3, -2, -1, 0, 0, 1, 2, 4, 8]))  # > 0: away from the touch; < 0: towards and through it
                price = int(ref_p - off * 100) if side == "buy" else int(ref_p + off * 100)
                if price <= 0: continue
                vol = int(r.choice([1, 5, 10, 50, 200]))
                kw = dict(timestamp=ts, direction=side, ticker=tick, internal_id=None, external_id=None, is_external=False, price=price, volume=vol)
                o_r, o_m = RM.LimitOrder(**kw), S.LimitOrder(**kw)
                ro.append(o_r); mo.append(o_m)
            else:
                side = "buy" if r.random() < 0.5 else "sell"
                kw = dict(timestamp=ts, direction=side, ticker=tick, internal_id=None, external_id=None, is_external=False, volume=int(r.choice([1, 10, 100, 1000])))
                ro.append(RM.MarketOrder(**kw)); mo.append(S.MarketOrder(**kw))
        n_orders += len(ro)
        err_r = err_m = None
        try:
            fm = mine.forward_step(until, mo)
        except NotImplementedError:  # (the documented limit of the host-side rank map: the same orders in two calls instead)
            return dict(day=kind, skipped="cancellation behind a self-crossing own order", steps=step)
        except Exception as ex:  # noqa: BLE001
            err_m = type(ex).__name__
        import contextlib, io
        with contextlib.redirect_stdout(io.StringIO()):
            try:
                fr_ = ref.forward_step(until=until, internal_orders=ro)
            except Exception as ex:  # noqa: BLE001
                err_r = type(ex).__name__
        assert (err_r is None) == (err_m is None), f"step {step}: reference {err_r}, mirror {err_m}; orders {ro}"
        if err_r is not None:
            return dict(day=kind, steps=step, orders=n_orders, fills=n_fills, raised=(err_r, err_m))
        for name in ("internal", "external"):
            a = [(o.direction, int(o.price), int(o.volume)) for o in getattr(fr_, name)]
            b = [(o.direction, int(o.price), int(o.volume)) for o in getattr(fm, name)]
            assert a == b, f"step {step}: {name} fills differ\n{a}\n{b}\norders {ro}"
            n_fills += len(a)
        a, b = book_rows(ref.exchange.central_orderbook), book_rows(mine.exchange.central_orderbook)
        assert a == b, f"step {step}: central books differ at {next(i for i in range(min(len(a), len(b)) + 1) if i >= min(len(a), len(b)) or a[i] != b[i])}: {a[:4]} .. vs {b[:4]} ..\norders {ro}"
        ib_r, ib_m = ref.exchange.internal_orderbook, mine.exchange.internal_orderbook
        a, b = book_rows(ib_r), book_rows(ib_m)
        assert a == b, f"step {step}: internal books differ\n{a}\n{b}\norders {ro}"
        assert (ref.min_buy_price, ref.max_sell_price) == (mine.min_buy_price, mine.max_sell_price), f"step {step}: price ranges {ref.min_buy_price, ref.max_sell_price} vs {mine.min_buy_price, mine.max_sell_price}"
        # the own orders alive now, paired by their place in the (equal) internal books: side, price, FIFO rank
        alive = []
        for side_name in ("buy", "sell"):
            dr, dm = getattr(ib_r, side_name), getattr(ib_m, side_name)
            for p in dr.keys():
                for o_r, o_m in zip(dr[p], dm[p]):
                    alive.append((o_r.internal_id, o_m.internal_id, side_name, int(p), int(o_r.volume)))
        dead = [e for e in seen if e[0] not in {x[0] for x in alive}]
        seen = alive + (dead[-3:] if r.random() < 0.3 else [])
        now = until
    mine.close()
    return dict(day=kind, steps=n_steps, orders=n_orders, fills=n_fills)


if __name__ == "__main__":
    first = int(_args[0]) if len(_args) > 0 else 0
    end = int(_args[1]) if len(_args) > 1 else 200
    bad, t_all = 0, time.time()
    for seed in range(first, end):
        t = time.time()
        try:
            info = run(seed)
            print(f"seed {seed:4d} OK   {time.time() - t:5.1f}s {info}", flush=True)
        except AssertionError as ex:
            bad += 1
            print(f"seed {seed:4d} DIFF {time.time() - t:5.1f}s\n    -> {str(ex)[:1800]}", flush=True)
        except Exception:
            bad += 1
            print(f"seed {seed:4d} ERROR {time.time() - t:5.1f}s\n{traceback.format_exc()[-2500:]}", flush=True)
    print(f"{end - first} cases, {bad} with a difference or an error, {time.time() - t_all:.0f} s")
    sys.exit(1 if bad else 0)
